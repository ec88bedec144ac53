// Exact up-sampling StyledConv: conv_transpose2d(stride 2, 3x3) followed by the 4x4 FIR blur
// (src/models/stylegan2/model.py:287-300, 206-213), tile-fused so that the (2H+1)^2 intermediate lives only in LDS.
//
// The polyphase form used by e4s_conv_mfma_f32 (ncls = 4) spends 36*Cin*Cout MACs per input pixel; the transposed
// conv itself needs 9.  This kernel does the minimal products on the matrix cores and the blur on the VALU:
//   P_k[u, co] = sum_ci x[u, ci] * s[ci] * W[co, ci, k]                 9 taps k, fp32 MFMA, rows u = halo pixels
//   I[q]       = sum_{2u + k = q} P_k[u]                               scatter-add into an LDS tile
//   out[p]     = act( d[co] * sum_j kflip[j] * I[p + j - 1] + noise + bias )
// Block = 6 x 14 input pixels (+1 halo on each side = 8 x 16 = 128 GEMM rows, one 32-row MFMA block per wave) x 32
// output channels -> a 12 x 28 output tile; 3.4*Cin MACs per output element instead of 9*Cin.  The nine tap
// accumulators (9 x 16 VGPRs) stay in registers across the whole K loop; the A tile (halo x 32 channels) is staged
// once per channel chunk and reused by the 9 taps, B (32 x 32 weights of one tap) is staged per step.
// Region-select: the style of a tile is uniform per pass; tiles whose output pixels belong to several regions run one
// pass per region present (style folded into the B tile while staging) and each pass writes only its own pixels.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int KC = 32, LDA = 36, NTHR = 256;
constexpr int TAH = 6, TAW = 14;                 // input pixels (anchors) per tile
constexpr int HW_ = TAW + 2;                     // halo width 16; halo rows = (TAH + 2) * 16 = 128
constexpr int ROWS = (TAH + 2) * HW_;
constexpr int BN = 32;
constexpr int OH = 2 * TAH, OW = 2 * TAW;        // 12 x 28 outputs
constexpr int IQH = 2 * (TAH + 2) + 1, IQW = 2 * HW_ + 1;   // 17 x 33: every position a halo pixel scatters to
                                                            // (the 15 x 31 interior feeds the tile's outputs)
constexpr int TG = 3;                            // taps staged (and MFMA'd) per barrier: 3 x 16 MFMAs per wave
constexpr int STAGE_WORDS = 2 * ROWS * LDA + 2 * TG * BN * LDA;
constexpr int ITILE_WORDS = IQH * IQW * BN;
constexpr int UNION_WORDS = STAGE_WORDS > ITILE_WORDS ? STAGE_WORDS : ITILE_WORDS;

struct UpSmem {
    float kf[16];
    int regmask;
    int pad_[3];
    unsigned char lab[OH * OW];                  // region of each output pixel of the tile (336 B)
    float nz[OH * OW];                           // noise_w * noise of each output pixel (per-pixel noise maps)
    float buf[UNION_WORDS];                      // K loop: A[2][128][36], B[2][3][32][36];  epilogue: I[17][33][32]
};

// ABL (ablation; only builds with -DE4S_ABLATIONS can select != 0): 0 full; 1 skip the K loop; 2 skip zero-fill + scatter; 3 skip the blur/store phase
template <int ABL>
__global__ __launch_bounds__(NTHR, 2) void upconv_kernel(const e4s_conv_params p, const float* __restrict__ k4,
                                                         const int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    UpSmem& sm = *reinterpret_cast<UpSmem*>(smem_raw);
    float* sA = sm.buf;
    float* sB = sm.buf + 2 * ROWS * LDA;
    float* sI = sm.buf;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const int tx_n = (p.Wi + TAW - 1) / TAW, per_img = ((p.Hi + TAH - 1) / TAH) * tx_n;
    const int tb = mt / per_img;
    const int rem0 = mt - tb * per_img;
    const int ay0 = (rem0 / tx_n) * TAH, ax0 = (rem0 % tx_n) * TAW;
    const int oy0 = 2 * ay0, ox0 = 2 * ax0;
    const int R = p.labels ? p.groups_per_batch : 1;

    // ---- tile metadata: flipped blur taps, labels of the output pixels, set of regions present -----------------
    if (tid < 16) sm.kf[tid] = k4[15 - tid];
    if (tid == 0) sm.regmask = 0;
    __syncthreads();
    for (int t = tid; t < OH * OW; t += NTHR) {
        const int oy = oy0 + t / OW, ox = ox0 + t % OW;
        int r = 255;
        if (oy < p.Ho && ox < p.Wo) {
            r = 0;
            if (p.labels) {
                const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Ho)), p.Hm - 1);
                const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wo)), p.Wm - 1);
                r = p.labels[((size_t)tb * p.Hm + sy) * p.Wm + sx];
            }
            atomicOr(&sm.regmask, 1 << r);
        }
        sm.lab[t] = (unsigned char)r;
        float nzv = 0.f;
        if (r != 255 && p.noise && !p.noise_per_channel)
            nzv = p.noise_w[0] * p.noise[(int64_t)tb * p.noise_bstride + (int64_t)oy * p.Wo + ox];
        sm.nz[t] = nzv;
    }
    __syncthreads();
    const int regmask = sm.regmask;

    // ---- staging roles ---------------------------------------------------------------------------------------
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;              // A: rows r0 + 32 j (j < 4); B: row r0
    unsigned a_off[4];                                   // element offsets (launcher checks they fit 32 bits)
    unsigned a_ok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + 32 * j;
        const int uy = ay0 + row / HW_ - 1, ux = ax0 + row % HW_ - 1;
        const bool ok = (unsigned)uy < (unsigned)p.Hi && (unsigned)ux < (unsigned)p.Wi;
        a_off[j] = ok ? (unsigned)(((tb * p.Hi + uy) * p.Wi + ux) * p.Cin) : 0u;
        a_ok |= (ok ? 1u : 0u) << j;
    }
    const int nchunk = p.Cin / KC;
    const int arow = (wave * 32 + li) * LDA, brow = li * LDA;

    for (int reg = 0; reg < R; ++reg) {
        if (!((regmask >> reg) & 1)) continue;
        const int g = tb * R + reg;
        const float* sscale = p.in_scale ? p.in_scale + (size_t)g * p.Cin : nullptr;

        f32x16 acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

        f32x4 pa[4], pb[TG], psc = {1.f, 1.f, 1.f, 1.f};
        auto fetch_a = [&](int c0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[j] = *reinterpret_cast<const f32x4*>(p.x + a_off[j] + c0 + c4);
        };
        auto store_a = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(sA + buf * (ROWS * LDA) + (r0 + 32 * j) * LDA + c4) =
                    ((a_ok >> j) & 1u) ? pa[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // B of one tap group (3 taps x 32 couts x 32 channels); the style scale is applied when the registers are
        // written to LDS, so that nothing waits on a load right after issuing it
        auto fetch_b = [&](int tg, int c0) {
#pragma unroll
            for (int j = 0; j < TG; ++j)
                pb[j] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(tg * TG + j) * p.Cout + n0 + r0) * p.Cin + c0 + c4);
            if (sscale) psc = *reinterpret_cast<const f32x4*>(sscale + c0 + c4);
        };
        auto store_b = [&](int buf) {
#pragma unroll
            for (int j = 0; j < TG; ++j)
                *reinterpret_cast<f32x4*>(sB + (buf * TG + j) * (BN * LDA) + r0 * LDA + c4) = pb[j] * psc;
        };

        fetch_a(0);
        fetch_b(0, 0);
        store_a(0);
        store_b(0);
        __syncthreads();

        int s = 0;
        for (int c = 0; c < (ABL == 1 ? 0 : nchunk); ++c) {
            const bool more_c = (c + 1 < nchunk);
            if (more_c) fetch_a((c + 1) * KC);
            const float* Ab = sA + (c & 1) * (ROWS * LDA) + arow;
#pragma unroll
            for (int tg = 0; tg < 9 / TG; ++tg, ++s) {
                const bool last = (tg == 9 / TG - 1);
                const bool more = !last || more_c;
                if (more) fetch_b(last ? 0 : tg + 1, (last ? c + 1 : c) * KC);
#pragma unroll
                for (int j = 0; j < TG; ++j) {
                    const float* Bb = sB + ((s & 1) * TG + j) * (BN * LDA) + brow;
#pragma unroll
                    for (int kk = 0; kk < KC / 8; ++kk) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(Ab + kk * 8 + kh * 4);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(Bb + kk * 8 + kh * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[tg * TG + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[tg * TG + j], 0, 0, 0);
                    }
                }
                if (last && more_c) store_a((c + 1) & 1);
                if (more) store_b((s + 1) & 1);
                __syncthreads();
            }
        }

        // ---- transposed-conv scatter: I[2u + k] += P_k[u] (the staging buffers are free now) -----------------
        if (ABL != 2)
            for (int t = tid; t < ITILE_WORDS / 4; t += NTHR) reinterpret_cast<f32x4*>(sI)[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        int ibase[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            ibase[r] = ((2 * (row / HW_)) * IQW + 2 * (row % HW_)) * BN + li;
        }
        __syncthreads();
        // q = 2u + k: within one tap every halo pixel hits a different q, and taps of different (ky, kx) parity never
        // meet -> four barrier-separated phases; inside a phase all reads are issued before the adds and the writes
        // (LDS float atomics measured ~100x slower than this on gfx950)
        auto scatter = [&](const f32x16& a, int k) {
            const int off = ((k / 3) * IQW + (k % 3)) * BN;
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = sI[ibase[r] + off];
#pragma unroll
            for (int r = 0; r < 16; ++r) sI[ibase[r] + off] = old[r] + a[r];
        };
        if (ABL != 2) {
            scatter(acc[0], 0); scatter(acc[1], 1); scatter(acc[3], 3); scatter(acc[4], 4);
            __syncthreads();
            scatter(acc[2], 2); scatter(acc[5], 5); scatter(acc[7], 7);
            __syncthreads();
            scatter(acc[6], 6);
            __syncthreads();
            scatter(acc[8], 8);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[k][r]), "v"(ibase[r]));
        }
        __syncthreads();

        // ---- blur + demodulation + noise + bias + activation; a pass writes only the pixels of its own region ------
        {
            const int co = tid & 31, grp = tid >> 5;
            const int col = n0 + co;
            const float dsc = p.out_scale ? p.out_scale[(size_t)g * p.Cout + col] : 1.f;
            const float bsv = p.bias ? p.bias[col] : 0.f;
            const float nw = p.noise ? p.noise_w[0] : 0.f;
            const float slp = (p.act == 2) ? p.slope[col] : p.alpha;
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const bool nz_pc = p.noise && p.noise_per_channel;
            float kf[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) kf[j] = sm.kf[j];
            // sliding 4x4 window down each output column: 4 new LDS reads per output instead of 16
            for (int px = grp; px < (ABL == 3 ? 0 : OW); px += NTHR / 32) {
                const float* ip = sI + (IQW + px + 1) * BN + co;          // I row 1, column px + 1
                float w0[4], w1[4], w2[4], w3[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    w0[jx] = ip[jx * BN];
                    w1[jx] = ip[(IQW + jx) * BN];
                    w2[jx] = ip[(2 * IQW + jx) * BN];
                }
#pragma unroll
                for (int py = 0; py < OH; ++py) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) w3[jx] = ip[((py + 3) * IQW + jx) * BN];
                    float v = 0.f;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx)
                        v += kf[jx] * w0[jx] + kf[4 + jx] * w1[jx] + kf[8 + jx] * w2[jx] + kf[12 + jx] * w3[jx];
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) { w0[jx] = w1[jx]; w1[jx] = w2[jx]; w2[jx] = w3[jx]; }
                    const int t = py * OW + px;
                    const int lab = sm.lab[t];
                    if (lab == 255 || (p.labels && lab != reg)) continue;
                    const int oy = oy0 + py, ox = ox0 + px;
                    const int64_t opix = ((int64_t)tb * p.Ho + oy) * p.Wo + ox;
                    v = v * dsc + bsv + sm.nz[t];
                    if (nz_pc) v += nw * p.noise[((int64_t)tb * p.noise_bstride + (int64_t)oy * p.Wo + ox) * p.Cout + col];
                    if (p.act) v = (v > 0.f ? v : v * slp) * gain;
                    p.y[opix * (p.y_cstride ? p.y_cstride : p.Cout) + col] = v;
                }
            }
        }
        __syncthreads();          // the I tile is overwritten by the next pass's staging
    }
}

}  // namespace

extern "C" int e4s_upconv_mfma_f32(const e4s_conv_params* pp, const float* k4, void* stream) {
    const e4s_conv_params& p = *pp;
    if (p.Cin % KC || p.Cout % BN || !k4) return (int)hipErrorInvalidValue;
    if (p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi) return (int)hipErrorInvalidValue;
    if ((int64_t)p.B * p.Hi * p.Wi * p.Cin >= (1ll << 31)) return (int)hipErrorInvalidValue;
    if (p.labels && (p.groups_per_batch < 1 || p.groups_per_batch > 16)) return (int)hipErrorInvalidValue;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(upconv_kernel<0>), (int)sizeof(UpSmem), smem_set)) return e;
    const int ntn = p.Cout / BN;
    const int64_t mtiles = (int64_t)p.B * ((p.Hi + TAH - 1) / TAH) * ((p.Wi + TAW - 1) / TAW);
    if (mtiles <= 0) return 0;
#ifdef E4S_ABLATIONS      // profiling builds only (E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build): tools/upconv_ablate.py
    static const int abl = getenv("E4S_UPCONV_ABL") ? atoi(getenv("E4S_UPCONV_ABL")) : 0;
    if (abl) {          // wrong results by construction
        auto set = [](const void* f) { hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(UpSmem)); };
        dim3 gr((unsigned)(mtiles * ntn)), bl(NTHR);
        if (abl == 1) { set((const void*)upconv_kernel<1>); hipLaunchKernelGGL(upconv_kernel<1>, gr, bl, sizeof(UpSmem), as_stream(stream), p, k4, ntn); }
        if (abl == 2) { set((const void*)upconv_kernel<2>); hipLaunchKernelGGL(upconv_kernel<2>, gr, bl, sizeof(UpSmem), as_stream(stream), p, k4, ntn); }
        if (abl == 3) { set((const void*)upconv_kernel<3>); hipLaunchKernelGGL(upconv_kernel<3>, gr, bl, sizeof(UpSmem), as_stream(stream), p, k4, ntn); }
        E4S_CHECK_LAUNCH();
        return 0;
    }
#endif
    hipLaunchKernelGGL(upconv_kernel<0>, dim3((unsigned)(mtiles * ntn)), dim3(NTHR), sizeof(UpSmem), as_stream(stream), p, k4, ntn);
    E4S_CHECK_LAUNCH();
    return 0;
}

// diagnostic: resident blocks per CU of the up-conv kernel according to the runtime
extern "C" int e4s_upconv_blocks_per_cu(void) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(upconv_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)sizeof(UpSmem));
    int n = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(upconv_kernel<0>), NTHR, sizeof(UpSmem));
    return n;
}
