// 3x3 stride-1 convolution with 32 input channels on the split-bf16 matrix-core path, weights RESIDENT in LDS, optional
// ToRGB partial fused into the epilogue -- the generator's last StyledConv (32 -> 32 at 1024^2, model.py:537-549, 655-657) and the
// ToRGB behind it (model.py:422-448).
//
// (Round-3 measurements, 8 images at 1024^2, ms: generic kernel 0.97; this kernel as ONE 512-thread block per CU with dword stores
// from the accumulators 0.86; stores transposed through LDS 0.70; ablations of that version -- no MFMA loop 0.51, no halo loads
// 0.58, no stores 0.61, no epilogue 0.54: no single phase is the bound, the phases of the one resident block add up.  256-thread
// blocks, two per CU, on 8x16 tiles were SLOWER (0.75: 40 % halo overhead, twice the barriers); what is left is wave
// specialisation (loader / MFMA / store waves) -- not done.)
// That layer is HBM-bound (8 images: 1.07 GB in + 1.07 GB out, 0.36 ms at 6 TB/s; its 155 GFLOP are 0.15 ms of MFMA time), but
// ran at 0.97 ms on the generic kernel: with K = 288 a 256-pixel tile is only 54 MFMAs per wave, and the generic
// pipeline re-stages the weights every 3 taps (3 barriers per tile) and re-reads each A fragment per 32 columns.  Here:
//   * the 9 x 32 x 32 weights (split hi/lo bf16, 41 KB) are staged ONCE per block and stay in LDS;
//   * one stage = one 16x16-pixel tile: its 18x18 halo (32 channels, split once while staged) sits in a single LDS buffer and
//     is fetched into registers TWO tiles ahead (two register sets, even / odd tiles), so an HBM round trip has two whole tiles
//     to complete; 2 barriers per tile (3 with the ToRGB partial);
//   * the epilogue (demodulation, noise, bias, activation) leaves through an LDS staging tile (16-byte stores, one 128-byte
//     line per pixel: dword stores straight from the accumulators are store-issue bound) from which, when asked,
//     rgb_partial[b, c, p] = sum_co y[b, p, co] * ws[b, c, co] (the ToRGB 1x1 modulated conv) comes out of the same pass: the
//     134 MB activation per image is not read again; e4s_torgb_finish_f32 adds bias + FIR-upsampled skip.
// Arithmetic as conv_bf16x3.hip: three v_mfma_f32_32x32x16_bf16 per product on hi/lo-split fp32 operands, fp32 accumulate.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32, ROWB = 144, LO = 64;
constexpr int TW = 16, TH = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;      // 16 x 16-pixel tiles, 324 halo pixels
constexpr int BM = TH * TW, BN = 32, NTHR = 512;                                // 8 waves x (32 pixels x 32 channels)
constexpr int ITEMS = HALO * 4, AJ = (ITEMS + NTHR - 1) / NTHR;                 // 1296 items, 3 per thread
constexpr int BPIECES = 9 * BN * 8, BJ = (BPIECES + NTHR - 1) / NTHR;           // 2304 16-byte pieces, 5 per thread (once)
constexpr int A_BYTES = HALO * ROWB, B_BYTES = 9 * BN * ROWB;
constexpr int YLD = 36;                                                         // floats per pixel row of the output staging tile
constexpr int SMEM = B_BYTES + A_BYTES + BM * 8 + BM * YLD * 4 + 3 * BN * 4;
static_assert(SMEM <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void split_store(unsigned char* dst, const f32x8 v) {
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + LO) = l;
}

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

struct TileId { int tb, tyb, txb, nt; };

// XF: 1 = v * in_scale[b][c] while the halo is staged; RGB: 1 = also emit the ToRGB partial (needs Cout == 32)
template <int XF, int RGB>
__global__ __launch_bounds__(NTHR) void conv_c32_kernel(const e4s_conv_params p, const float* __restrict__ rgb_ws,
                                                        float* __restrict__ rgb_partial, const int ntn, const int tx_n,
                                                        const int per_img, const int ntiles, const int abl) {
    // abl (profiling builds only, -DE4S_ABLATIONS + env E4S_C32_ABL; results WRONG): 1 no MFMA loop, 2 no halo loads, 3 no output
    // stores, 4 no halo staging (split + LDS write), 5 no epilogue at all
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sB = smem;                                   // [9][32][ROWB]  resident weights of this block's n-tile
    unsigned char* sA = smem + B_BYTES;                         // [HALO][ROWB]
    int* s_out = reinterpret_cast<int*>(sA + A_BYTES);          // [BM] output pixel index or -1
    float* s_nz = reinterpret_cast<float*>(s_out + BM);         // [BM]
    float* sY = s_nz + BM;                                      // [BM][YLD]  output staging tile
    float* sWS = sY + BM * YLD;                                 // [3][32]    (RGB)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const size_t img_stride = (size_t)p.Hi * p.Wi * KC;
    const int HWo = p.Ho * p.Wo;

    auto decode = [&](int t) -> TileId {
        TileId id;
        const int mt = t % (p.B * per_img);                     // n-tile major: a block keeps its resident weights as long as it can
        id.nt = t / (p.B * per_img);
        id.tb = mt / per_img;
        const int rem = mt - id.tb * per_img;
        id.tyb = rem / tx_n;
        id.txb = rem - id.tyb * tx_n;
        return id;
    };
    auto item_src = [&](const TileId& id, int item, bool& ok) -> size_t {
        const int h = item >> 2, q = item & 3;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = id.tyb * TH + hy - 1, ix = id.txb * TW + hx - 1;
        ok = item < ITEMS && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        return ok ? ((size_t)iy * p.Wi + ix) * KC + q * 8 : (size_t)(q * 8);
    };
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    struct AReg {
        f32x8 a[AJ], sc;                // sc: the style scale of this thread's channel group (item & 3 == tid & 3 for every j)
        bool ok[AJ];
    };
    auto fetch_a = [&](AReg& R, const TileId& id, bool real) {
        const float* xb = p.x + (real ? (size_t)id.tb * img_stride : 0);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int item = tid + NTHR * j;
            const size_t off = item_src(id, item, R.ok[j]);
            if (abl != 2) R.a[j] = load8(xb + off);
        }
        if (XF) R.sc = load8(p.in_scale + (real ? (size_t)id.tb * KC : 0) + (tid & 3) * 8);
    };
    auto store_a = [&](const AReg& R) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int item = tid + NTHR * j;
            if (item < ITEMS) {
                f32x8 v = R.a[j];
                if (XF) v = v * R.sc;
                if (!R.ok[j]) v = zero8;
                split_store(sA + (item >> 2) * ROWB + (item & 3) * 16, v);
            }
        }
    };
    auto load_weights = [&](int nt) {                           // rows (tap, co) of the split [9][Cout][32] image, 128 bytes each
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
        f32x4 r[BJ];
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j, row = (i < BPIECES ? i : 0) >> 3, pc = i & 7;
            const int tap = row / BN, co = row - tap * BN;
            r[j] = *reinterpret_cast<const f32x4*>(wb + ((size_t)tap * p.Cout + nt * BN + co) * 128 + pc * 16);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j;
            if (i < BPIECES) *reinterpret_cast<f32x4*>(sB + (i >> 3) * ROWB + (i & 7) * 16) = r[j];
        }
    };

    if (first >= ntiles) return;
    // fragment rows: wave w owns pixels 32 w .. 32 w + 31 of the tile (two image rows)
    const int m_row = wave * 32 + li;
    const int arow = ((m_row / TW) * HALO_W + (m_row % TW)) * ROWB + kh * 16;
    const int brow = li * ROWB + kh * 16;

    int t_cur = first;
    TileId cur = decode(t_cur);
    int res_nt = cur.nt;
    load_weights(res_nt);
    AReg RE, RO;
    fetch_a(RE, cur, true);
    {
        const int t1 = t_cur + G;
        fetch_a(RO, decode(t1 < ntiles ? t1 : t_cur), t1 < ntiles);
    }

    // one tile: `R` holds its halo (fetched two tiles ago); after storing it the set is re-used for the tile two ahead
    auto process = [&](AReg& R) {
        __syncthreads();                                        // every reader of sA / sY of the previous tile is done
        if (cur.nt != res_nt) { res_nt = cur.nt; load_weights(res_nt); }
        if (abl != 4) store_a(R);
        {
            const int t2 = t_cur + 2 * G;
            fetch_a(R, decode(t2 < ntiles ? t2 : t_cur), t2 < ntiles);
        }
        if (tid < BM) {
            const int ay = cur.tyb * TH + tid / TW, ax = cur.txb * TW + tid % TW;
            const bool valid = ay < p.Ho && ax < p.Wo;
            s_out[tid] = valid ? (cur.tb * p.Ho + ay) * p.Wo + ax : -1;
            float nz = 0.f;
            if (valid && p.noise) nz = p.noise_w[0] * p.noise[(int64_t)cur.tb * p.noise_bstride + (int64_t)ay * p.Wo + ax];
            s_nz[tid] = nz;
        }
        if (RGB && tid < 3 * BN) sWS[tid] = rgb_ws[(size_t)cur.tb * 3 * BN + tid];
        __syncthreads();
        // ---- 9 taps x 2 k-halves x 3 MFMAs on the wave's 32 x 32 block ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < (abl == 1 ? 0 : 9); ++tap) {
            const unsigned char* At = sA + ((tap / 3) * HALO_W + (tap % 3)) * ROWB + arow;
            const unsigned char* Bt = sB + tap * (BN * ROWB) + brow;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(At + kk * 32);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(At + kk * 32 + LO);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bt + kk * 32);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bt + kk * 32 + LO);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
        // ---- epilogue ----
        if (abl == 5) { asm volatile("" :: "v"(acc[0]), "v"(acc[5])); return; }
        const int co = cur.nt * BN + li;
        const float osc = p.out_scale ? p.out_scale[(size_t)cur.tb * p.Cout + co] : 1.f;
        const float bsv = p.bias ? p.bias[co] : 0.f;
        const float gain = (p.act == 1) ? p.gain : 1.f;
        const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = acc[r] * osc + s_nz[row] + bsv;
            if (p.act) v = (v > 0.f ? v : v * p.alpha) * gain;
            sY[row * YLD + li] = v;
        }
        __syncthreads();
        // the tile leaves through LDS: 16 dword stores per lane straight from the accumulators (128 store instructions per
        // tile) are store-ISSUE bound on this chip (~70 cycles each: 9-10k cycles per tile against 3.5k of MFMA time);
        // transposed, a thread stores 4 x 16 bytes and 8 lanes cover one pixel's 128-byte line
        {
            const int c4 = tid & 7;
#pragma unroll
            for (int ps = 0; ps < BM / (NTHR / 8); ++ps) {
                const int px = ps * (NTHR / 8) + (tid >> 3);
                const int off = s_out[px];
                if (off >= 0 && abl != 3)
                    *reinterpret_cast<f32x4*>(p.y + (size_t)off * ycs + cur.nt * BN + c4 * 4) =
                        *reinterpret_cast<const f32x4*>(sY + px * YLD + c4 * 4);
            }
        }
        if (RGB) {
            // thread = (pixel, half of the 32 channels): 3 dot products of 16, combined across the lane pair
            const int px = tid >> 1, hf = tid & 1;
            const float* yp = sY + px * YLD + hf * 16;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(yp + c);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWS + hf * 16 + c);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWS + BN + hf * 16 + c);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(sWS + 2 * BN + hf * 16 + c);
                a0 += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
                a1 += v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
                a2 += v[0] * w2[0] + v[1] * w2[1] + v[2] * w2[2] + v[3] * w2[3];
            }
            a0 += __shfl_xor(a0, 1, 64);
            a1 += __shfl_xor(a1, 1, 64);
            a2 += __shfl_xor(a2, 1, 64);
            const int off = s_out[px];
            if (hf == 0 && off >= 0) {
                const int rem = off - cur.tb * HWo;
                float* o = rgb_partial + (size_t)cur.tb * 3 * HWo + rem;
                o[0] = a0;
                o[HWo] = a1;
                o[2 * (size_t)HWo] = a2;
            }
        }
    };

    for (;;) {
        process(RE);
        t_cur += G;
        if (t_cur >= ntiles) break;
        cur = decode(t_cur);
        process(RO);
        t_cur += G;
        if (t_cur >= ntiles) break;
        cur = decode(t_cur);
    }
}

// out = partial + bias[c] + upfirdn2d(skip, k4, up=2, pad=(2,1))   (ToRGB's tail, model.py:441-446); NCHW [B,3,H,W].
// Thread = 4 consecutive pixels of one row (16-byte loads / stores); grid (W/4 blocks of 64, H, B*3).
__global__ __launch_bounds__(64) void torgb_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                          const float* __restrict__ skip, const float* __restrict__ k4,
                                                          float* __restrict__ out, int H, int W) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x0 >= W) return;
    const int yy = blockIdx.y, plane = blockIdx.z;               // plane = b * 3 + ch
    const size_t row = ((size_t)plane * H + yy) * W + x0;
    const bool vec = (W & 3) == 0;                               // otherwise rows are not 16-byte aligned: element-wise tail path
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec) v = *reinterpret_cast<const f32x4*>(partial + row);
    else
        for (int e = 0; e < 4; ++e)
            if (x0 + e < W) v[e] = partial[row + e];
    v += bias[plane % 3];
    if (skip) {
        const int Hs = H >> 1, Ws = W >> 1;
        const float* sp = skip + (size_t)plane * Hs * Ws;
#pragma unroll
        for (int jy = 0; jy < 4; ++jy) {
            const int qy = yy + jy - 2;
            if (qy < 0 || (qy & 1) || (qy >> 1) >= Hs) continue;
            const float* sr = sp + (size_t)(qy >> 1) * Ws;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int qx = x0 + e + jx - 2;
                    if (qx < 0 || (qx & 1) || (qx >> 1) >= Ws) continue;
                    v[e] += sr[qx >> 1] * k4[15 - (jy * 4 + jx)];
                }
            }
        }
    }
    if (vec) *reinterpret_cast<f32x4*>(out + row) = v;
    else
        for (int e = 0; e < 4; ++e)
            if (x0 + e < W) out[row + e] = v[e];
}

int num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (!cus[dev & 63]) {
        hipDeviceProp_t prop;
        cus[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                            ? prop.multiProcessorCount : 256;
    }
    return cus[dev & 63];
}

template <int XF, int RGB>
int launch(const e4s_conv_params& p, const float* rgb_ws, float* rgb_partial, hipStream_t st) {
    auto kern = conv_c32_kernel<XF, RGB>;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, smem_set)) return e;
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wo + TW - 1) / TW, per_img = ((p.Ho + TH - 1) / TH) * tx_n;
    const int64_t ntiles = (int64_t)p.B * per_img * ntn;
    if (ntiles >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());               // persistent, one block per CU
    int abl = 0;
#ifdef E4S_ABLATIONS
    if (const char* e = getenv("E4S_C32_ABL")) abl = atoi(e);
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHR), SMEM, st, p, rgb_ws, rgb_partial, ntn, tx_n, per_img, (int)ntiles,
                       abl);
    E4S_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// p as e4s_conv_bf16x3_f32 (natural-order 3x3, stride 1, Cin == 32, Cout % 32 == 0, one style per sample or none, per-pixel
// noise, bias, act 0/1); w = the split image of the tap-packed weights [9][Cout][32].  rgb_ws [B][3][32] + rgb_partial
// [B][3][H][W] (both or neither; Cout == 32): also emit rgb_partial = ToRGB's 1x1 modulated conv of the layer's OUTPUT.
extern "C" int e4s_conv_c32_bf16x3_f32(const e4s_conv_params* pp, const float* rgb_ws, float* rgb_partial, void* stream) {
    const e4s_conv_params& p = *pp;
    if (p.Cin != KC || p.Cout % BN || p.ntaps != 9 || p.ncls != 1 || p.istride != 1 || p.ostride != 1 || p.tiles || p.labels ||
        p.in_stats || p.noise_per_channel || p.act == 2 || p.Ho != p.Hi || p.Wo != p.Wi || p.B <= 0 || p.stats_ws)
        return (int)hipErrorInvalidValue;
    if ((rgb_ws == nullptr) != (rgb_partial == nullptr) || (rgb_ws && p.Cout != BN)) return (int)hipErrorInvalidValue;
    if ((int64_t)p.B * p.Ho * p.Wo >= (1ll << 31)) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    if (p.in_scale) return rgb_ws ? launch<1, 1>(p, rgb_ws, rgb_partial, st) : launch<1, 0>(p, rgb_ws, rgb_partial, st);
    return rgb_ws ? launch<0, 1>(p, rgb_ws, rgb_partial, st) : launch<0, 0>(p, rgb_ws, rgb_partial, st);
}

extern "C" int e4s_torgb_finish_f32(const float* partial, const float* bias, const float* skip, const float* k4, float* out, int B,
                                    int H, int W, void* stream) {
    if (!partial || !bias || !out || (skip && (!k4 || ((H | W) & 1)))) return (int)hipErrorInvalidValue;
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    hipLaunchKernelGGL(torgb_finish_kernel, dim3((unsigned)(((W + 3) / 4 + 63) / 64), (unsigned)H, (unsigned)(B * 3)), dim3(64), 0,
                       as_stream(stream), partial, bias, skip, k4, out, H, W);
    E4S_CHECK_LAUNCH();
    return 0;
}
