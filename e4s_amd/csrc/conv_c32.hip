// 3x3 stride-1 convolution with 32 input channels on the split-bf16 matrix-core path, weights RESIDENT in LDS, optional
// ToRGB partial fused into the epilogue -- the generator's last StyledConv (32 -> 32 at 1024^2, model.py:537-549, 655-657) and the
// ToRGB behind it (model.py:422-448).
//
// That layer is HBM-bound (8 images: 1.07 GB in + 1.07 GB out, 0.36 ms at 6 TB/s; its 155 GFLOP are 0.15 ms of MFMA time), but
// ran at 0.97 ms on the generic kernel: with K = 288 a 256-pixel tile is only 54 MFMAs per wave, and the generic
// pipeline re-stages the weights every 3 taps (3 barriers per tile) and re-reads each A fragment per 32 columns.  Here:
//   * the 9 x 32 x 32 weights (split hi/lo bf16, 36 KB) are staged ONCE per block and stay in LDS;
//   * one stage = one 16x16-pixel tile: its 18x18 halo (32 channels, split once while staged) sits in a single LDS buffer and
//     is fetched into registers a tile ahead;
//   * the epilogue (demodulation, noise, bias, activation) leaves through an LDS staging tile (16-byte stores, one 128-byte
//     line per pixel: dword stores straight from the accumulators are store-issue bound) from which, when asked,
//     rgb_partial[b, c, p] = sum_co y[b, p, co] * ws[b, c, co] (the ToRGB 1x1 modulated conv) comes out of the same pass: the
//     134 MB activation per image is not read again; e4s_torgb_finish_f32 adds bias + FIR-upsampled skip.
// Round 3 ran this as ONE 512-thread block per CU (127 KB of LDS): 0.70 ms, and its ablations -- no MFMA loop 0.51, no halo loads 0.58,
// no stores 0.61, no epilogue 0.54 -- said no single phase is the bound, the phases of the one resident block add up (two 256-thread
// blocks on 8x16 tiles were slower then: 40 % halo overhead).  Round 5, after the same cut paid on the exact up-conv
// (upconv_bf16x3.hip): TWO 256-thread blocks per CU on the SAME 16x16 tile -- 128-byte LDS rows with the 16-byte granule XORed with
// (row >> 1) & 7 instead of 144-byte padded rows (A 41 KB, B 36 KB), the output staging tile aliased over the halo buffer (one more
// barrier per tile), four waves of 64 pixels (two accumulators; 6 fragment reads per 6 MFMAs instead of 4 per 3): 79 KB per block.
// Arithmetic as conv_bf16x3.hip: three v_mfma_f32_32x32x16_bf16 per product on hi/lo-split fp32 operands, fp32 accumulate.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32, ROWB = 128, LO = 64;                                     // [32 hi | 32 lo] bf16 per row, swizzled (swz below)
constexpr int TW = 16, TH = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;      // 16 x 16-pixel tiles, 324 halo pixels
constexpr int BM = TH * TW, BN = 32, NTHR = 256;                                // 4 waves x (64 pixels x 32 channels)
constexpr int ITEMS = HALO * 4, AJ = (ITEMS + NTHR - 1) / NTHR;                 // 1296 items, 6 per thread
constexpr int BPIECES = 9 * BN * 8, BJ = BPIECES / NTHR;                        // 2304 16-byte pieces, 9 per thread (once)
constexpr int A_BYTES = HALO * ROWB, B_BYTES = 9 * BN * ROWB;                   // 41 472 + 36 864
constexpr int YLD = 36;                                                         // floats per pixel row of the output staging tile
constexpr int SMEM = B_BYTES + A_BYTES + BM * 8 + 3 * BN * 4;                   // 80 768: two blocks per CU
static_assert(BPIECES % NTHR == 0 && BM == NTHR, "thread layout");
static_assert(BM * YLD * 4 <= A_BYTES, "the output staging tile aliases the halo buffer");
static_assert(2 * SMEM <= 160 * 1024, "two blocks per CU");

// byte offset of 16-byte granule g (0..3: 8 hi channels each, 4..7: the lo halves) of row r; the lo half of a granule is the same
// offset ^ 64, the second k-step ^ 32.  A ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...)
// that must hit 16 different 16-byte slots of the 256-byte bank line.  Weights: a group reads 16 rows of one tap at one logical granule;
// rows of one parity share a 128-byte half and have eight different (r >> 1) & 7 -- conflict free.  Halo: a group's pixels are columns
// x0 .. x0 + 15 of TWO image rows ({0-3, 12-15} of one, {4-11} of the next), so the swizzle keys on the halo COLUMN hx = r % 18 (the row
// pitch is even: a row's parity is its column's): (hx & 1, (hx >> 1) & 7) takes 16 different values -- conflict free (keyed on r itself,
// like the 144-byte padded rows of round 3, every group was 2-way: the 30.9 % of conflict cycles in r04f's counters).
__device__ __forceinline__ int swz(int r, int g) { return r * ROWB + ((g ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int swz_halo(int h, int g) { return h * ROWB + ((g ^ (((h % HALO_W) >> 1) & 7)) << 4); }

__device__ __forceinline__ void split_store(unsigned char* base, int off, const f32x8 v) {
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    *reinterpret_cast<bf16x8*>(base + off) = h;
    *reinterpret_cast<bf16x8*>(base + (off ^ LO)) = l;
}

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

struct TileId { int tb, tyb, txb, nt; };

// XF: 1 = v * in_scale[b][c] while the halo is staged; RGB: 1 = also emit the ToRGB partial (needs Cout == 32)
template <int XF, int RGB>
__global__ __launch_bounds__(NTHR, 2) void conv_c32_kernel(const e4s_conv_params p, const float* __restrict__ rgb_ws,
                                                           float* __restrict__ rgb_partial, const int ntn, const int tx_n,
                                                           const int per_img, const int ntiles, const int abl_arg) {
    // abl (profiling builds only, -DE4S_ABLATIONS + env E4S_C32_ABL; results WRONG): 1 no MFMA loop, 2 no halo loads, 3 no output
    // stores, 4 no halo staging (split + LDS write), 5 no epilogue at all
#ifdef E4S_ABLATIONS
    const int abl = abl_arg;
#else
    constexpr int abl = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sB = smem;                                   // [9][32][ROWB]  resident weights of this block's n-tile
    unsigned char* sA = smem + B_BYTES;                         // [HALO][ROWB]
    float* sY = reinterpret_cast<float*>(sA);                   // [BM][YLD]  output staging tile (aliases the halo: written after the MFMA loop)
    int* s_out = reinterpret_cast<int*>(sA + A_BYTES);          // [BM] output pixel index or -1
    float* s_nz = reinterpret_cast<float*>(s_out + BM);         // [BM]
    float* sWS = s_nz + BM;                                     // [3][32]    (RGB)

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
                split_store(sA, swz_halo(item >> 2, item & 3), v);
            }
        }
    };
    auto load_weights = [&](int nt) {                           // rows (tap, co) of the split [9][Cout][32] image, 128 bytes each
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
        f32x4 r[BJ];
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j, row = i >> 3, pc = i & 7;
            const int tap = row / BN, co = row - tap * BN;
            r[j] = *reinterpret_cast<const f32x4*>(wb + ((size_t)tap * p.Cout + nt * BN + co) * 128 + pc * 16);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j;
            *reinterpret_cast<f32x4*>(sB + swz(i >> 3, i & 7)) = r[j];
        }
    };

    if (first >= ntiles) return;
    // fragment rows: wave w owns pixels 64 w .. 64 w + 63 of the tile (four image rows) as two 32-row MFMA tiles; ro[tm][tap] = byte
    // offset of (halo row of the pixel shifted by the tap, k-half kh) -- the second k-step is ^ 32, the lo halves ^ 64
    int ro[2][9];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m_row = wave * 64 + tm * 32 + li;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            ro[tm][tap] = swz_halo((m_row / TW + tap / 3) * HALO_W + (m_row % TW) + tap % 3, kh);
    }
    const int brow = swz(li, kh);                               // + tap * BN * ROWB (a multiple of 16 rows: the swizzle term is the row's own)

    int t_cur = first;
    TileId cur = decode(t_cur);
    int res_nt = cur.nt;
    load_weights(res_nt);
    AReg R;
    fetch_a(R, cur, true);

    // one tile: `R` holds its halo (fetched one tile ago: with two blocks per CU a block's tile lasts ~8 us, several HBM round trips);
    // after storing it the set is re-used for the next tile
    auto process = [&]() {
        __syncthreads();                                        // every reader of sY / s_out of the previous tile is done
        if (cur.nt != res_nt) { res_nt = cur.nt; load_weights(res_nt); }
        if (abl != 4) store_a(R);
        {
            const int t1 = t_cur + G;
            fetch_a(R, decode(t1 < ntiles ? t1 : t_cur), t1 < ntiles);
        }
        {
            const int ay = cur.tyb * TH + tid / TW, ax = cur.txb * TW + tid % TW;
            const bool valid = ay < p.Ho && ax < p.Wo;
            s_out[tid] = valid ? (cur.tb * p.Ho + ay) * p.Wo + ax : -1;
            float nz = 0.f;
            if (valid && p.noise) nz = p.noise_w[0] * p.noise[(int64_t)cur.tb * p.noise_bstride + (int64_t)ay * p.Wo + ax];
            s_nz[tid] = nz;
        }
        if (RGB && tid < 3 * BN) sWS[tid] = rgb_ws[(size_t)cur.tb * 3 * BN + tid];
        __syncthreads();
        // ---- 9 taps x 2 k-halves x 2 row tiles x 3 MFMAs on the wave's 64 x 32 block ----
        f32x16 acc[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < (abl == 1 ? 0 : 9); ++tap) {
            const unsigned char* Bt = sB + tap * (BN * ROWB);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bt + (brow ^ (kk * 32)));
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bt + (brow ^ (kk * 32) ^ LO));
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    ah[tm] = *reinterpret_cast<const bf16x8*>(sA + (ro[tm][tap] ^ (kk * 32)));
                    al[tm] = *reinterpret_cast<const bf16x8*>(sA + (ro[tm][tap] ^ (kk * 32) ^ LO));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], bh, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bl, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bl, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bh, acc[1], 0, 0, 0);
            }
        }
        // ---- epilogue ----
        if (abl == 5) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][5])); return; }
        const int co = cur.nt * BN + li;
        const float osc = p.out_scale ? p.out_scale[(size_t)cur.tb * p.Cout + co] : 1.f;
        const float bsv = p.bias ? p.bias[co] : 0.f;
        const float gain = (p.act == 1) ? p.gain : 1.f;
        const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
        __syncthreads();                                        // every wave is through with the halo: sY may overwrite it
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[tm][r] * osc + s_nz[row] + bsv;
                if (p.act) v = (v > 0.f ? v : v * p.alpha) * gain;
                sY[row * YLD + li] = v;
            }
        __syncthreads();
        // the tile leaves through LDS: 16 dword stores per lane straight from the accumulators (128 store instructions per
        // tile) are store-ISSUE bound on this chip (~70 cycles each: 9-10k cycles per tile against 3.5k of MFMA time);
        // transposed, a thread stores 8 x 16 bytes and 8 lanes cover one pixel's 128-byte line
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
            // thread = (pixel, half of the 32 channels): 3 dot products of 16, combined across the lane pair; two passes of 128 pixels
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int px = pass * (NTHR / 2) + (tid >> 1), hf = tid & 1;
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
        }
    };

    for (;;) {
        process();
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
    const int slots = 2 * num_cus();                                               // persistent, two co-resident blocks per CU
    const int grid = (int)(ntiles < slots ? ntiles : slots);
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
