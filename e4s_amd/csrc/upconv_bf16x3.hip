// Exact up-sampling StyledConv on the bf16 matrix cores at fp32-class accuracy: conv_transpose2d(stride 2, 3x3) followed by the
// 4x4 FIR blur (src/models/stylegan2/model.py:287-300, 206-213) at 1x the MACs of the layer (9 Cin Cout per INPUT pixel).
// The polyphase form on e4s_conv_bf16x3_f32 (ncls = 4: four 3x3 phase kernels) spends 36; the two unmasked up-convs of a face
// swap (128 -> 64 into 512^2, 64 -> 32 into 1024^2) ran 4-7x above their HBM floor on it.
//
// Two kernels, loosely fused through the Infinity Cache (the intermediate of ONE launch group is sized to stay on-die):
//
//  (1) upconv_sub_kernel -- the transposed conv as a SUB-PIXEL GEMM.  With I[q] = sum_{2u + k = q} x[u] W[k] (no flip:
//      conv_transpose2d), the four parity classes of q are four small convolutions over the INPUT grid,
//          I[2a + dy, 2b + dx] = sum over shifts (sy, sx) in {0,-1}^2 with (sy == 0 or dy == 0) and (sx == 0 or dx == 0) of
//                                x[a + sy, b + sx] . W[dy - 2 sy, dx - 2 sx]
//      i.e. class (0,0) has 4 taps, (0,1) and (1,0) two, (1,1) one: 9 (shift, class) blocks, exactly the layer's MACs, and the
//      scatter-add of the transposed conv happens in the MFMA accumulators.  GEMM: M = anchors a on the (H+1) x (W+1) grid
//      (16x16 tiles, the halo kernel's LDS layout: one split hi/lo bf16 row per halo pixel, taps = shifted views), N = 4 Cout
//      (column = class * Cout + co), K = 4 shifts x Cin with the 7 zero (shift, class) blocks skipped per wave.  Arithmetic as
//      conv_bf16x3.hip: three v_mfma_f32_32x32x16_bf16 per product on hi/lo-split fp32 operands, fp32 accumulate.  Persistent
//      blocks, the stage pipeline runs through tile boundaries.  The epilogue stores the raw sums pixel-shuffled into
//      I [G][2H+2][2W+2][Cout] (row / column 2H+1 receive exact zeros: x is zero beyond the image).
//  (2) upconv_fir_kernel -- out = act(d[b,co] * sum_j kflip[j] I[o + j - 1] + noise_w * noise[b,o] + bias[co]) * gain:
//      the blur with pad (1,1) + NoiseInjection + FusedLeakyReLU (model.py:396-404) in one streaming pass over I.
//
// The entry point walks the batch in groups whose I fits ~144 MB, so kernel (2) reads what kernel (1) just wrote from the
// 256 MB Infinity Cache and HBM sees the layer's algorithmic bytes (x once, y once).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32, ROWB = 144, LO = 64;
constexpr int TW = 16, TH = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;     // 324 halo pixels (row/column 17 unused)
constexpr int BM = 256, BN = 128, NTHR = 512;
constexpr int WM = 4, WN = 2, TM = BM / (WM * 32), TN = BN / (WN * 32);        // waves 64 x 64
constexpr int NSH = 4;                                                        // shifts = pipeline stages per chunk
constexpr int ITEMS = HALO * 4, PIECE = ITEMS / NSH;                           // 1296 halo items, 324 per stage
constexpr int BITEMS = BN * 8, BJ = BITEMS / NTHR;                             // 1024 16-byte weight pieces per stage
constexpr int A_BYTES = HALO * ROWB, B_BYTES = BN * ROWB;
constexpr int SMEM = 2 * A_BYTES + 2 * B_BYTES + 3 * BM * 4;
static_assert(PIECE * NSH == ITEMS && PIECE <= NTHR && BJ * NTHR == BITEMS, "staging split");
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

struct SubParams {
    const float* x;            // NHWC [G, H, W, Cin] (this group's samples)
    const float* w;            // split image of [4 shifts][4 Cout][Cin]
    const float* in_scale;     // [G, Cin] or null
    float* I;                  // [G, 2H+2, 2W+2, Cout]
    int G, H, W, Cin, Cout;
};

struct TileId { int tb, tyb, txb, n0; };

// XF: 0 none, 1 v * in_scale[b][c] while the halo is staged (one style per sample: unmasked StyledConv, model.py:655-657)
template <int XF>
__global__ __launch_bounds__(NTHR) void upconv_sub_kernel(const SubParams p, const int ntn, const int tx_n, const int per_img,
                                                          const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][HALO][ROWB]
    unsigned char* sB = smem + 2 * A_BYTES;            // [2][BN][ROWB]
    int* s_out = reinterpret_cast<int*>(sB + 2 * B_BYTES);            // [3][BM]: I pixel index of anchor (2ay, 2ax), or -1

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tap masks below branch on it
    const int li = lane & 31, kh = lane >> 5;
    // waves w and w + 4 share a SIMD: give them different column halves so every SIMD carries the same MFMA count although
    // the parity classes have 4 / 2 / 2 / 1 taps
    const int wm = wave & 3, wn = wave >> 2;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int nchunk = p.Cin / KC;
    const int Ha = p.H + 1, Wa = p.W + 1;              // anchor grid
    const int IH = 2 * p.H + 2, IW = 2 * p.W + 2;
    const int NT = 4 * p.Cout;

    auto decode = [&](int t) -> TileId {
        TileId id;
        const int mt = t / ntn, nt = t - mt * ntn;
        id.n0 = nt * BN;
        id.tb = mt / per_img;
        const int rem = mt - id.tb * per_img;
        id.tyb = rem / tx_n;
        id.txb = rem - id.tyb * tx_n;
        return id;
    };
    auto fill_meta = [&](int buf, const TileId& id) {
        if (tid < BM) {
            const int ay = id.tyb * TH + tid / TW, ax = id.txb * TW + tid % TW;
            s_out[buf * BM + tid] = (ay < Ha && ax < Wa) ? (id.tb * IH + 2 * ay) * IW + 2 * ax : -1;
        }
    };
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w);
    const size_t wrow = (size_t)p.Cin * 4;
    const size_t img_stride = (size_t)p.H * p.W * p.Cin;

    auto item_src = [&](const TileId& id, int item, bool& ok) -> size_t {
        const int h = item >> 2, q = item & 3;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = id.tyb * TH + hy - 1, ix = id.txb * TW + hx - 1;
        ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        return ok ? ((size_t)iy * p.W + ix) * p.Cin + q * 8 : (size_t)(q * 8);
    };
    auto item_dst = [&](int item) -> int { return (item >> 2) * ROWB + (item & 3) * 16; };
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    int b_dst[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int i = tid + NTHR * j;
        b_dst[j] = (i >> 3) * ROWB + (i & 7) * 16;
    }
    auto b_src = [&](int j, int n0, int sh, int chunk) -> size_t {
        const int i = tid + NTHR * j;
        return ((size_t)sh * NT + n0 + (i >> 3)) * wrow + (size_t)chunk * 128 + (i & 7) * 16;
    };

    if (first >= ntiles) return;
    TileId cur = decode(first);
    int t_next = first + G;
    bool has_next = t_next < ntiles;
    TileId nxt = decode(has_next ? t_next : first);

    // ---- prologue ----
    {
        const float* xb = p.x + (size_t)cur.tb * img_stride;
        for (int item = tid; item < ITEMS; item += NTHR) {
            bool ok;
            const size_t off = item_src(cur, item, ok);
            f32x8 v = load8(xb + off);
            if (XF) v = v * load8(p.in_scale + (size_t)cur.tb * p.Cin + (item & 3) * 8);
            if (!ok) v = zero8;
            split_store(sA + item_dst(item), v);
        }
        f32x4 pb[BJ];
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wbytes + b_src(j, cur.n0, 0, 0));
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + b_dst[j]) = pb[j];
        fill_meta(0, cur);
    }
    __syncthreads();

    int arow[TM], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = (wm * TM + tm) * 32 + li;
        arow[tm] = ((m / TW) * HALO_W + (m % TW)) * ROWB + kh * 16;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = ((wn * TN + tn) * 32 + li) * ROWB + kh * 16;

    f32x16 acc[TM][TN];
    const bool piece_thr = tid < PIECE;
    unsigned sg = 0, cg = 0;
    int mbuf = 0;
    for (;;) {
        const int mnext = (mbuf + 1) % 3;
        if (has_next) fill_meta(mnext, nxt);
        // parity class of each of this wave's column blocks in this tile, and in which shift stages it has a tap:
        // shift index s = 2 * (sy == -1) + (sx == -1); class (dy, dx) is active iff (sy == 0 or dy == 0) and (sx == 0 or dx == 0)
        unsigned amask[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int cls = (cur.n0 + (wn * TN + tn) * 32) / p.Cout;
            const int dy = cls >> 1, dx = cls & 1;
            amask[tn] = 1u | (dx == 0 ? 2u : 0u) | (dy == 0 ? 4u : 0u) | ((dx == 0 && dy == 0) ? 8u : 0u);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

        for (int chunk = 0; chunk < nchunk; ++chunk) {
            const bool last_chunk = (chunk + 1 == nchunk);
            const TileId& own = last_chunk ? nxt : cur;
            const bool have_nc = !last_chunk || has_next;
            const int c_n = last_chunk ? 0 : chunk + 1;
            const float* xb_n = p.x + (size_t)own.tb * img_stride + c_n * KC;
            for (int ts = 0; ts < NSH; ++ts) {
                const unsigned char* Ab = sA + (cg & 1) * A_BYTES;
                const unsigned char* Bb = sB + (sg & 1) * B_BYTES;
                const bool last_ts = (ts + 1 == NSH);
                const bool more = !last_ts || have_nc;
                // view of the halo for shift (sy, sx): anchor (ay, ax) reads halo pixel (ay + sy + 1, ax + sx + 1)
                const int vy = 1 - (ts >> 1), vx = 1 - (ts & 1);
                const unsigned char* At = Ab + (vy * HALO_W + vx) * ROWB;
                // -- global -> VGPR: weights of the next stage, one piece of the next chunk's halo --
                f32x4 pb[BJ];
                f32x8 pa, px;
                bool pok;
                int pdst;
                {
                    const int n0_w = last_ts ? own.n0 : cur.n0;
                    const int sh_w = last_ts ? 0 : ts + 1;
                    const int ch_w = last_ts ? c_n : chunk;
#pragma unroll
                    for (int j = 0; j < BJ; ++j)
                        pb[j] = *reinterpret_cast<const f32x4*>(wbytes + b_src(j, more ? n0_w : cur.n0, more ? sh_w : 0, more ? ch_w : 0));
                    const int item = ts * PIECE + (piece_thr ? tid : 0);
                    const size_t off = item_src(own, item, pok);
                    pa = load8((have_nc ? xb_n : p.x) + off);
                    if (XF) px = load8(p.in_scale + (size_t)(have_nc ? own.tb : 0) * p.Cin + c_n * KC + (item & 3) * 8);
                    pdst = item_dst(item);
                }
                // -- MFMAs: 2 k-halves x TM row blocks x the column blocks that have a tap in this shift --
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 bh[TN], bl[TN];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        bh[tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32);
                        bl[tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32 + LO);
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(At + arow[tm] + kk * 32);
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(At + arow[tm] + kk * 32 + LO);
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            if ((amask[tn] >> ts) & 1u) {
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[tn], acc[tm][tn], 0, 0, 0);
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[tn], acc[tm][tn], 0, 0, 0);
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[tn], acc[tm][tn], 0, 0, 0);
                            }
                        }
                    }
                }
                // -- VGPR -> LDS --
                if (more) {
                    unsigned char* db = sB + ((sg + 1) & 1) * B_BYTES;
#pragma unroll
                    for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(db + b_dst[j]) = pb[j];
                }
                if (have_nc && piece_thr) {
                    f32x8 v = pa;
                    if (XF) v = v * px;
                    if (!pok) v = zero8;
                    split_store(sA + ((cg + 1) & 1) * A_BYTES + pdst, v);
                }
                __syncthreads();
                ++sg;
            }
            ++cg;
        }

        // ---- epilogue: raw sums, pixel-shuffled into I ----
        {
            const int* so = s_out + mbuf * BM;
            int coff[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = cur.n0 + (wn * TN + tn) * 32 + li;
                const int cls = n / p.Cout, co = n - cls * p.Cout;
                coff[tn] = ((cls >> 1) * IW + (cls & 1)) * p.Cout + co;
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int off = so[row];
                    if (off < 0) continue;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) p.I[(size_t)off * p.Cout + coff[tn]] = acc[tm][tn][r];
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        mbuf = mnext;
        t_next += G;
        has_next = t_next < ntiles;
        if (has_next) nxt = decode(t_next);
    }
}

// ---- (2) blur + demodulation + noise + bias + activation ------------------------------------------------------------------
struct FirParams {
    const float* I;            // [G, IH, IW, C]
    float* y;                  // [G, 2H, 2W, ycs] (first C channels written)
    const float* out_scale;    // [G, C] or null
    const float* noise;        // [*, 2H, 2W] or null; noise_bstride = 0 for a shared map
    const float* noise_w;
    const float* bias;         // [C] or null
    int64_t noise_bstride;
    int G, Ho, Wo, C, ycs, act;
    float alpha, gain;
    const float* k4;           // device, [4][4]: the taps are used flipped, kf[ay*4+ax] = k4[3-ay][3-ax] (upfirdn2d_kernel.cu:77)
};

// thread = (output column ox, 4 channels) x a strip of 4 output rows: 7 x 4 float4 reads for 4 outputs
__global__ __launch_bounds__(256) void upconv_fir_kernel(const FirParams p) {
    const int c4n = p.C >> 2;
    const int ppb = 256 / c4n;                               // output columns per block
    const int c4 = threadIdx.x % c4n, pxl = threadIdx.x / c4n;
    const int ox = blockIdx.x * ppb + pxl;
    const int oy0 = blockIdx.y * 4;
    const int b = blockIdx.z;
    if (ox >= p.Wo) return;
    const int IW = p.Wo + 2, IH = p.Ho + 2;
    const float* Ib = p.I + (size_t)b * IH * IW * p.C + c4 * 4;
    float kf[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) kf[a] = p.k4[15 - a];       // uniform: scalar loads
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ry = 0; ry < 7; ++ry) {
        const int qy = oy0 + ry - 1;
        if (qy < 0 || qy >= IH) continue;
        f32x4 v[4];
#pragma unroll
        for (int ax = 0; ax < 4; ++ax) {
            const int qx = ox + ax - 1;
            v[ax] = qx >= 0 ? *reinterpret_cast<const f32x4*>(Ib + ((size_t)qy * IW + qx) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ay = ry - r;                           // I row qy = (oy0 + r) + ay - 1
            if (ay < 0 || ay > 3) continue;
#pragma unroll
            for (int ax = 0; ax < 4; ++ax) acc[r] += v[ax] * kf[ay * 4 + ax];
        }
    }
    f32x4 d = {1.f, 1.f, 1.f, 1.f}, bs = {0.f, 0.f, 0.f, 0.f};
    if (p.out_scale) d = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)b * p.C + c4 * 4);
    if (p.bias) bs = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
    const float nw = p.noise ? p.noise_w[0] : 0.f;
    const float gain = (p.act == 1) ? p.gain : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.Ho) break;
        float nz = 0.f;
        if (p.noise) nz = nw * p.noise[b * p.noise_bstride + (int64_t)oy * p.Wo + ox];
        f32x4 v = acc[r] * d + bs + nz;
        if (p.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * p.alpha) * gain;
        }
        *reinterpret_cast<f32x4*>(p.y + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.ycs + c4 * 4) = v;
    }
}

// w [Cout,Cin,3,3] -> [4 shifts][4 Cout][Cin]: row (s, cls * Cout + co) = W[co][.][dy - 2 sy][dx - 2 sx] or zeros
__global__ void subpixel_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)16 * Cout * Cin;
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int co = (int)(r % Cout);
    const int cls = (int)((r / Cout) % 4), s = (int)(r / (4 * Cout));
    const int dy = cls >> 1, dx = cls & 1, sy = -(s >> 1), sx = -(s & 1);
    const bool act = (sy == 0 || dy == 0) && (sx == 0 || dx == 0);
    const int ky = dy - 2 * sy, kx = dx - 2 * sx;
    out[i] = act ? w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx] : 0.f;
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

// samples per launch group: the fp32 intermediate of a group should sit in the 256 MB Infinity Cache between the two kernels
inline int group_size(const e4s_conv_params& p) {
    const int64_t per = (int64_t)(2 * p.Hi + 2) * (2 * p.Wi + 2) * p.Cout * 4;
    int g = (int)((144ll << 20) / (per > 0 ? per : 1));
    if (g < 1) g = 1;
    if (g > p.B) g = p.B;
    return g;
}

}  // namespace

extern "C" int e4s_subpixel_weights_f32(const float* w, float* out, int Cout, int Cin, void* stream) {
    const int64_t n = (int64_t)16 * Cout * Cin;
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(subpixel_weights_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w, out, Cout, Cin);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_upconv_bf16x3_ws_floats(const e4s_conv_params* pp) {
    const e4s_conv_params& p = *pp;
    return (int64_t)group_size(p) * (2 * p.Hi + 2) * (2 * p.Wi + 2) * p.Cout;
}

// p: x NHWC [B,H,W,Cin], w = split image of e4s_subpixel_weights_f32's output, y NHWC [B,2H,2W,Cout (or y_cstride)],
// in_scale [B,Cin] | null, out_scale [B,Cout] | null, noise / noise_w / bias / act / alpha / gain as e4s_conv_bf16x3_f32;
// k4: DEVICE pointer to the 4x4 blur kernel (model.py:206-213, already x4); ws: e4s_upconv_bf16x3_ws_floats(p) floats.
extern "C" int e4s_upconv_bf16x3_f32(const e4s_conv_params* pp, const float* k4, float* ws, void* stream) {
    const e4s_conv_params& p = *pp;
    // Cout % 32: a 32-column MFMA block must not straddle two parity classes (column = class * Cout + co)
    if (p.Cin % KC || p.Cout % 32 || !k4 || !ws || p.labels || p.tiles || p.in_stats || p.noise_per_channel || p.act == 2 ||
        p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi || p.B <= 0 || (p.y_cstride && p.y_cstride % 4))
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    auto kern = p.in_scale ? upconv_sub_kernel<1> : upconv_sub_kernel<0>;
    static std::atomic<uint64_t> smem_set0{0}, smem_set1{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, p.in_scale ? smem_set1 : smem_set0)) return e;
    const int gs = group_size(p);
    const int ntn = (4 * p.Cout) / BN;
    const int tx_n = (p.Wi + 1 + TW - 1) / TW, per_img = ((p.Hi + 1 + TH - 1) / TH) * tx_n;
    const size_t ximg = (size_t)p.Hi * p.Wi * p.Cin, yimg = (size_t)p.Ho * p.Wo * (p.y_cstride ? p.y_cstride : p.Cout);
    for (int g0 = 0; g0 < p.B; g0 += gs) {
        const int G = (p.B - g0 < gs) ? p.B - g0 : gs;
        SubParams sp;
        sp.x = p.x + g0 * ximg;
        sp.w = p.w;
        sp.in_scale = p.in_scale ? p.in_scale + (size_t)g0 * p.Cin : nullptr;
        sp.I = ws;
        sp.G = G; sp.H = p.Hi; sp.W = p.Wi; sp.Cin = p.Cin; sp.Cout = p.Cout;
        const int64_t ntiles = (int64_t)G * per_img * ntn;
        const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHR), SMEM, st, sp, ntn, tx_n, per_img, (int)ntiles);
        E4S_CHECK_LAUNCH();
        FirParams fp;
        fp.I = ws;
        fp.y = p.y + g0 * yimg;
        fp.out_scale = p.out_scale ? p.out_scale + (size_t)g0 * p.Cout : nullptr;
        fp.noise = p.noise ? p.noise + (size_t)g0 * p.noise_bstride : nullptr;
        fp.noise_w = p.noise_w;
        fp.bias = p.bias;
        fp.noise_bstride = p.noise_bstride;
        fp.G = G; fp.Ho = p.Ho; fp.Wo = p.Wo; fp.C = p.Cout; fp.ycs = p.y_cstride ? p.y_cstride : p.Cout;
        fp.act = p.act; fp.alpha = p.alpha; fp.gain = p.gain;
        fp.k4 = k4;
        const int ppb = 256 / (p.Cout / 4);
        hipLaunchKernelGGL(upconv_fir_kernel, dim3((unsigned)((p.Wo + ppb - 1) / ppb), (unsigned)((p.Ho + 3) / 4), (unsigned)G),
                           dim3(256), 0, st, fp);
        E4S_CHECK_LAUNCH();
    }
    return 0;
}
