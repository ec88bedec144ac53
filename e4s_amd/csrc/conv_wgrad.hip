// Weight gradient of the 3x3 / 1x1 convolutions on fp32 MFMA (SURVEY.md 8(f) N1 "wgrad"; config 5):
//   dW[tap][co][ci] = sum over anchors a of  G[a][co] * X[a (+) tap][ci]
//   G[a][co] = gz[o(a)][co] * d[g(a)][co],   X[a (+) tap][ci] = x[a*istride + tap - 1][ci] * s[g(a)][ci]
// with o(a) = a*ostride + phase the output pixel of anchor a (ostride 2 = one phase of the polyphase up-conv) and g(a) its
// sample / region group -- the same operands the forward kernels contract over (Cin, taps), contracted over the PIXELS
// instead.  GEMM view: M = Cout, N = Cin, K = anchors, once per tap, exact fp32 (v_mfma_f32_32x32x2_f32: both operands
// are one float per lane -- lane = channel, k = pixel -- so the NHWC tiles feed the matrix cores without any transpose).
// Block = 4 waves = one 64 (co) x 64 (ci) tile of EVERY tap: 9 accumulators of 32x32 per wave live in registers across
// the whole pixel loop (144 VGPRs), the x tile is staged once per anchor tile with its halo and read through 9 shifted
// views, the gz tile once.  Split-K over anchor tiles; the partial [tap][co][ci] slabs are added in a fixed order
// (e4s_reduce_parts_f32): the result is bit-reproducible.  ~78 KB of LDS: two blocks per CU overlap staging and MFMAs.
#include "common.h"

namespace {

constexpr int NTHR = 256, BC = 64, MAXR = 16;

template <int IS>
struct WgTile {                       // anchor tile: 8x16 at stride 1, 4x8 at stride 2 (input halo <= 180 pixels)
    static constexpr int TH = IS == 1 ? 8 : 4, TW = IS == 1 ? 16 : 8, NA = TH * TW;
    static constexpr int HH = TH * IS + 2, HWD = TW * IS + 2, NH = HH * HWD;
};

__device__ __forceinline__ int label_of(const e4s_conv_wgrad_params& p, int b, int oy, int ox) {
    const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Ho)), p.Hm - 1);
    const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wo)), p.Wm - 1);
    return p.labels[((size_t)b * p.Hm + sy) * p.Wm + sx];
}

template <int IS, int NTAPS>
__global__ __launch_bounds__(NTHR, 2) void conv_wgrad_kernel(const e4s_conv_wgrad_params p, const int nct, const int nnt,
                                                             const int nsplit, const int tx_n, const int per_img) {
    using T = WgTile<IS>;
    constexpr int TW = T::TW, NA = T::NA, HWD = T::HWD, NH = T::NH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sG = reinterpret_cast<float*>(smem_raw);            // [NA][64]
    float* sX = sG + NA * BC;                                  // [NH][64]
    float* sS = sX + NH * BC;                                  // [MAXR][64]  s[g][ci0..]
    int* sgrp = reinterpret_cast<int*>(sS + MAXR * BC);        // [NA]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int nt = bid % nnt, ct = bid / nnt;
    const int co0 = ct * BC, ci0 = nt * BC;
    const int R = p.labels ? p.R : 1;
    const int ntiles = p.B * per_img;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = split; tile < ntiles; tile += nsplit) {
        const int tb = tile / per_img;
        const int rem = tile - tb * per_img;
        const int tyb = rem / tx_n, txb = rem - tyb * tx_n;
        __syncthreads();                                        // previous tile fully consumed
        // ---- groups of the tile's anchors; style slice ----
        if (tid < NA) {
            const int ay = tyb * T::TH + tid / TW, ax = txb * TW + tid % TW;
            int g = -1;
            if (ay < p.Ha && ax < p.Wa) g = p.labels ? label_of(p, tb, ay * p.ostride + p.py, ax * p.ostride + p.px) : 0;
            sgrp[tid] = g;
        }
        if (p.s) {
            for (int t = tid; t < R * BC; t += NTHR) {
                const int r = t / BC, c = t - r * BC;
                sS[t] = (ci0 + c < p.Cin) ? p.s[((size_t)tb * R + r) * p.Cin + ci0 + c] : 0.f;
            }
        }
        // ---- X halo tile: [NH][64 ci] ----
        for (int t = tid; t < NH * (BC / 4); t += NTHR) {
            const int h = t / (BC / 4), c4 = (t - h * (BC / 4)) * 4;
            const int hy = h / HWD, hx = h - hy * HWD;
            const int iy = tyb * T::TH * IS + hy - (NTAPS == 9 ? 1 : 0) + p.tap_shift, ix = txb * TW * IS + hx - (NTAPS == 9 ? 1 : 0) + p.tap_shift;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && ci0 + c4 < p.Cin)
                v = *reinterpret_cast<const f32x4*>(p.x + (((size_t)tb * p.Hi + iy) * p.Wi + ix) * p.Cin + ci0 + c4);
            *reinterpret_cast<f32x4*>(sX + h * BC + c4) = v;
        }
        __syncthreads();                                        // sgrp visible for the G staging below
        // ---- G tile: [NA][64 co] = gz * d[g] ----
        for (int t = tid; t < NA * (BC / 4); t += NTHR) {
            const int a = t / (BC / 4), c4 = (t - a * (BC / 4)) * 4;
            const int g = sgrp[a];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g >= 0 && co0 + c4 < p.Cout) {
                const int ay = tyb * T::TH + a / TW, ax = txb * TW + a % TW;
                const int oy = ay * p.ostride + p.py, ox = ax * p.ostride + p.px;
                v = *reinterpret_cast<const f32x4*>(p.gz + (((size_t)tb * p.Ho + oy) * p.Wo + ox) * p.Cout + co0 + c4);
                if (p.d) v *= *reinterpret_cast<const f32x4*>(p.d + ((size_t)tb * R + g) * p.Cout + co0 + c4);
            }
            *reinterpret_cast<f32x4*>(sG + a * BC + c4) = v;
        }
        __syncthreads();
        // ---- contraction over the tile's anchors, two per MFMA ----
#pragma unroll 2
        for (int kp = 0; kp < NA / 2; ++kp) {
            const int a = 2 * kp + kh;
            const float av = sG[a * BC + wm * 32 + li];
            float sv = 1.f;
            if (p.s) {
                const int g = sgrp[a];
                sv = sS[(g < 0 ? 0 : g) * BC + wn * 32 + li];
            }
            const int ay = a / TW, ax = a - ay * TW;
            const float* xb = sX + ((ay * IS) * HWD + ax * IS) * BC + wn * 32 + li;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int ty = NTAPS == 9 ? t / 3 : 0, tx = NTAPS == 9 ? t % 3 : 0;
                const float bv = xb[(ty * HWD + tx) * BC] * sv;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- partial slab of this split: ws[split][tap][co][ci] ----
    const size_t slab = (size_t)NTAPS * p.Cout * p.Cin;
    float* out = p.ws + (size_t)split * slab;
    const int ci = ci0 + wn * 32 + li;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < p.Cout && ci < p.Cin) out[((size_t)t * p.Cout + co) * p.Cin + ci] = acc[t][r];
        }
}

template <int IS>
constexpr int wg_smem() { return (WgTile<IS>::NA * BC + WgTile<IS>::NH * BC + MAXR * BC) * 4 + WgTile<IS>::NA * 4; }

int wg_nsplit(const e4s_conv_wgrad_params& p) {
    const int th = p.istride == 1 ? 8 : 4, tw = p.istride == 1 ? 16 : 8;
    const int ntiles = p.B * ((p.Ha + th - 1) / th) * ((p.Wa + tw - 1) / tw);
    const int ctiles = ((p.Cout + BC - 1) / BC) * ((p.Cin + BC - 1) / BC);
    int ns = 1024 / (ctiles > 0 ? ctiles : 1);
    if (ns > ntiles) ns = ntiles;
    if (ns > 2048) ns = 2048;
    return ns < 1 ? 1 : ns;
}

template <int IS, int NTAPS>
int launch_wg(const e4s_conv_wgrad_params& p, hipStream_t st) {
    using T = WgTile<IS>;
    auto kern = conv_wgrad_kernel<IS, NTAPS>;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), wg_smem<IS>(), smem_set)) return e;
    const int nct = (p.Cout + BC - 1) / BC, nnt = (p.Cin + BC - 1) / BC;
    const int tx_n = (p.Wa + T::TW - 1) / T::TW, per_img = ((p.Ha + T::TH - 1) / T::TH) * tx_n;
    const int ns = wg_nsplit(p);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nct * nnt * ns)), dim3(NTHR), wg_smem<IS>(), st, p, nct, nnt, ns, tx_n, per_img);
    E4S_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int64_t e4s_conv_wgrad_ws_floats(const e4s_conv_wgrad_params* p) {
    return e4s_reduce_parts_ws_floats(wg_nsplit(*p), (int64_t)p->ntaps * p->Cout * p->Cin);
}

extern "C" int e4s_conv_wgrad_f32(const e4s_conv_wgrad_params* pp, void* stream) {
    const e4s_conv_wgrad_params& p = *pp;
    if (p.Cin % 32 || p.Cout % 32 || (p.ntaps != 9 && p.ntaps != 1) || (p.istride != 1 && p.istride != 2) ||
        (p.ostride != 1 && p.ostride != 2) || !p.ws || (p.labels && (p.R < 1 || p.R > MAXR)) || p.B < 1)
        return (int)hipErrorInvalidValue;
    if ((p.Ha - 1) * p.ostride + p.py >= p.Ho || (p.Wa - 1) * p.ostride + p.px >= p.Wo) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int rc;
    if (p.istride == 1) rc = p.ntaps == 9 ? launch_wg<1, 9>(p, st) : launch_wg<1, 1>(p, st);
    else rc = p.ntaps == 9 ? launch_wg<2, 9>(p, st) : launch_wg<2, 1>(p, st);
    if (rc) return rc;
    return e4s_reduce_parts_f32(p.ws, p.dw, wg_nsplit(p), (int64_t)p.ntaps * p.Cout * p.Cin, 1.f, stream);
}
