// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, 157.3 TFLOP/s chip peak).
//
// One kernel serves every contraction on the E4S hot path:
//   * generator StyledConv, masked   : GEMM rows are OUTPUT PIXELS GATHERED BY REGION (plan mode):
//     every pixel is computed once, with the style of its own region, instead of the
//     reference's 12 full passes + mask multiply (model.py:386-400)
//   * generator up-conv              : conv_transpose2d(stride 2) (*) 4x4 blur (model.py:287-300)
//     folded into 4 phase-specific 3x3 kernels over the input grid (ncls = 4)
//   * generator StyledConv, unmasked : natural order, halo-tiled loader (SPATIAL)
//   * encoder Conv2d 3x3 (stride 1/2) and the 1x1 stride-2 shortcut (helpers.py:122-144)
//
// GEMM view: M = rows (pixels), N = Cout, K = taps x Cin.  A = activations (NHWC, so one tap of
// one pixel is a contiguous Cin vector), B = weights [cls][tap][Cout][Cin].
// Block = 256 threads = 4 waves; block tile BM x BN, K step 32; each wave owns TM x TN
// 32x32 MFMA blocks (16 accumulator VGPRs each).  Both operands are staged global -> VGPR ->
// LDS (double buffered, one barrier per K step; next step's global loads are in flight during
// the MFMAs).  LDS rows are 36 floats (144 B) so ds_read_b128 fragment reads are conflict free.
// Style modulation s[g][ci] is applied to the B tile while it is staged (input-scaling form,
// model.py:245-274); demodulation, noise, bias and leaky-ReLU live in the epilogue.
#include "common.h"

namespace {

constexpr int KC = 32;    // K step (input channels per stage)
constexpr int LDA = 36;   // padded LDS row, floats
constexpr int NTHR = 256;

template <int BM, int BN, bool SPATIAL>
struct SmemLayout {
    static constexpr int TH = 8, TW = BM / 8, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;
    static constexpr int META_WORDS = 4 * BM;
    static constexpr int A_WORDS = SPATIAL ? HALO * LDA : 2 * BM * LDA;
    static constexpr int B_WORDS = 2 * BN * LDA;
    static constexpr int BYTES = (META_WORDS + A_WORDS + B_WORDS) * 4;
};

template <int BM, int BN, int WM, int WN, bool SPATIAL>
__global__ __launch_bounds__(NTHR, 2) void conv_mfma_kernel(const e4s_conv_params p, const int ntn,
                                                            const int tiles_per_cls) {
    using L = SmemLayout<BM, BN, SPATIAL>;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int TH = L::TH, TW = L::TW, HALO_W = L::HALO_W, HALO = L::HALO;
    constexpr int AR = BM / 32, BR = BN / 32, HR = (HALO + 31) / 32;
    constexpr int PA = SPATIAL ? HR : AR;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_out = reinterpret_cast<int*>(smem_raw);
    float* s_nz = reinterpret_cast<float*>(s_out + BM);
    int* s_base = reinterpret_cast<int*>(s_nz + BM);
    int* s_yx = s_base + BM;
    float* sA = reinterpret_cast<float*>(s_yx + BM);
    float* sB = sA + L::A_WORDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- which tile ------------------------------------------------------------------
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const bool plan = (p.tiles != nullptr);
    int row_start = 0, g = 0, cls = 0, tt = mt, tb = 0, tyb = 0, txb = 0;
    if (plan) {
        if (mt >= p.meta[0]) return;
        row_start = p.tiles[mt * 4 + 0];
        g = p.tiles[mt * 4 + 1];
        cls = p.tiles[mt * 4 + 2];
    } else {
        cls = mt / tiles_per_cls;
        tt = mt - cls * tiles_per_cls;
        if (SPATIAL) {
            const int tx_n = p.Wa / TW, per_img = (p.Ha / TH) * tx_n;
            tb = tt / per_img;
            const int rem = tt - tb * per_img;
            tyb = rem / tx_n;
            txb = rem - tyb * tx_n;
            g = tb;
        } else {
            g = (tt * BM) / (p.Ha * p.Wa);
        }
    }
    const int py = (p.ncls == 4) ? (cls >> 1) : 0, px = (p.ncls == 4) ? (cls & 1) : 0;

    // ---- per-row metadata ------------------------------------------------------------
    if (tid < BM) {
        int b, ay, ax;
        bool valid = true;
        if (SPATIAL) {
            b = tb;
            ay = tyb * TH + tid / TW;
            ax = txb * TW + tid % TW;
        } else {
            const int anchor = plan ? p.rows[row_start + tid] : tt * BM + tid;
            valid = anchor >= 0;
            const int a = valid ? anchor : 0;
            const int hw = p.Ha * p.Wa;
            b = a / hw;
            const int rem = a - b * hw;
            ay = rem / p.Wa;
            ax = rem - ay * p.Wa;
        }
        if (valid) {
            const int oy = ay * p.ostride + py, ox = ax * p.ostride + px;
            s_out[tid] = (b * p.Ho + oy) * p.Wo + ox;
            const int64_t npix = (int64_t)b * p.noise_bstride + (int64_t)oy * p.Wo + ox;
            if (!p.noise) s_nz[tid] = 0.f;
            else if (p.noise_per_channel) s_nz[tid] = __int_as_float((int)npix);
            else s_nz[tid] = p.noise_w[0] * p.noise[npix];
            const int by = ay * p.istride, bx = ax * p.istride;
            s_base[tid] = (b * p.Hi + by) * p.Wi + bx;
            s_yx[tid] = (by << 16) | bx;
        } else {
            s_out[tid] = -1;
            s_nz[tid] = 0.f;
            s_base[tid] = 0;
            s_yx[tid] = 0x7fff7fff;
        }
    }
    __syncthreads();

    // ---- staging roles: thread -> (row r0 + 32 j, 16-byte chunk c of the 32-float K step) ----
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;
    int a_base[AR], a_yx[AR];
    if (!SPATIAL) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            a_base[j] = s_base[r0 + 32 * j];
            a_yx[j] = s_yx[r0 + 32 * j];
        }
    }
    const float* sscale = p.in_scale ? p.in_scale + (size_t)g * p.Cin : nullptr;

    f32x4 pa[PA], pb[BR];
    const int ntaps = p.ntaps, nchunk = p.Cin / KC, nstage = nchunk * ntaps;

    auto fetch_b = [&](int tap, int c0) {
        const float* wp = p.w + ((size_t)(cls * ntaps + tap) * p.Cout + n0) * p.Cin + c0 + c4;
#pragma unroll
        for (int j = 0; j < BR; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(r0 + 32 * j) * p.Cin);
        if (sscale) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(sscale + c0 + c4);
#pragma unroll
            for (int j = 0; j < BR; ++j) pb[j] *= sv;
        }
    };
    auto fetch_a_gather = [&](int tap, int c0) {
        const int oy = (ntaps == 9) ? tap / 3 - 1 : 0, ox = (ntaps == 9) ? tap % 3 - 1 : 0;
        const int doff = oy * p.Wi + ox;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int iy = (a_yx[j] >> 16) + oy, ix = (a_yx[j] & 0xffff) + ox;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(a_base[j] + doff) * p.Cin + c0 + c4);
            pa[j] = v;
        }
    };
    auto fetch_a_halo = [&](int c0) {
#pragma unroll
        for (int j = 0; j < HR; ++j) {
            const int h = r0 + 32 * j;
            const int hy = h / HALO_W, hx = h - hy * HALO_W;
            const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (h < HALO && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(tb * p.Hi + iy) * p.Wi + ix) * p.Cin + c0 + c4);
            pa[j] = v;
        }
    };
    auto store_b = [&](int buf) {
        float* d = sB + buf * (BN * LDA) + r0 * LDA + c4;
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(d + 32 * j * LDA) = pb[j];
    };
    auto store_a = [&](int buf) {
        if (SPATIAL) {
#pragma unroll
            for (int j = 0; j < HR; ++j) {
                const int h = r0 + 32 * j;
                if (h < HALO) *reinterpret_cast<f32x4*>(sA + h * LDA + c4) = pa[j];
            }
        } else {
            float* d = sA + buf * (BM * LDA) + r0 * LDA + c4;
#pragma unroll
            for (int j = 0; j < AR; ++j) *reinterpret_cast<f32x4*>(d + 32 * j * LDA) = pa[j];
        }
    };

    // ---- fragment addressing ---------------------------------------------------------
    int arow[TM], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = (wm * TM + tm) * 32 + li;
        arow[tm] = SPATIAL ? ((m / TW) * HALO_W + (m % TW)) * LDA : m * LDA;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = ((wn * TN + tn) * 32 + li) * LDA;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- prologue: stage 0 -----------------------------------------------------------
    fetch_b(0, 0);
    if (SPATIAL) fetch_a_halo(0); else fetch_a_gather(0, 0);
    store_b(0);
    store_a(0);
    __syncthreads();

    int tap = 0, c0 = 0;
    for (int s = 0; s < nstage; ++s) {
        int ntap = tap + 1, nc0 = c0;
        if (ntap == ntaps) { ntap = 0; nc0 += KC; }
        const bool more = (s + 1 < nstage);
        const bool new_chunk = (ntap == 0);
        if (more) {
            fetch_b(ntap, nc0);
            if (SPATIAL) { if (new_chunk) fetch_a_halo(nc0); } else fetch_a_gather(ntap, nc0);
        }
        // ---- MFMAs on stage s ----
        {
            const int buf = s & 1;
            const float* Ab = SPATIAL ? sA + ((tap / 3) * HALO_W + (tap % 3)) * LDA : sA + buf * (BM * LDA);
            const float* Bb = sB + buf * (BN * LDA);
#pragma unroll
            for (int kk = 0; kk < KC / 8; ++kk) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(Ab + arow[tm] + kk * 8 + kh * 4);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(Bb + brow[tn] + kk * 8 + kh * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
            }
        }
        if (more) {
            if (SPATIAL && new_chunk) __syncthreads();   // single A halo buffer: everyone done reading it
            store_b((s + 1) & 1);
            if (!SPATIAL || new_chunk) store_a((s + 1) & 1);
        }
        __syncthreads();
        tap = ntap;
        c0 = nc0;
    }

    // ---- epilogue: demod * acc + noise + bias, activation, scatter to NHWC ------------
    float osc[TN], bsv[TN], slp[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + (wn * TN + tn) * 32 + li;
        osc[tn] = p.out_scale ? p.out_scale[(size_t)g * p.Cout + col] : 1.f;
        bsv[tn] = p.bias ? p.bias[col] : 0.f;
        slp[tn] = (p.act == 2) ? p.slope[col] : p.alpha;
    }
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0;
    const bool nz_pc = p.noise && p.noise_per_channel;
    const float nzw = nz_pc ? p.noise_w[0] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int off = s_out[row];
            if (off < 0) continue;
            const float nz = s_nz[row];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float nzv = nz_pc ? nzw * p.noise[(size_t)__float_as_int(nz) * p.Cout + n0 + (wn * TN + tn) * 32 + li] : nz;
                float v = acc[tm][tn][r] * osc[tn] + nzv + bsv[tn];
                if (do_act) v = (v > 0.f ? v : v * slp[tn]) * gain;
                p.y[(size_t)off * p.Cout + n0 + (wn * TN + tn) * 32 + li] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool SPATIAL>
int launch(const e4s_conv_params& p, hipStream_t st) {
    using L = SmemLayout<BM, BN, SPATIAL>;
    auto kern = conv_mfma_kernel<BM, BN, WM, WN, SPATIAL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntn = p.Cout / BN;
    int mtiles, tiles_per_cls = 0;
    if (p.tiles) {
        mtiles = p.tiles_cap;
    } else {
        const int64_t anchors = (int64_t)p.B * p.Ha * p.Wa;
        if (((int64_t)p.Ha * p.Wa) % BM) return (int)hipErrorInvalidValue;   // a tile never straddles two samples
        tiles_per_cls = (int)(anchors / BM);
        mtiles = tiles_per_cls * p.ncls;
    }
    if (mtiles <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(mtiles * ntn), dim3(NTHR), L::BYTES, st, p, ntn, tiles_per_cls);
    E4S_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int e4s_conv_mfma_f32(const e4s_conv_params* pp, int spatial, void* stream) {
    const e4s_conv_params& p = *pp;
    hipStream_t st = as_stream(stream);
    if (p.Cin % KC || p.Cout % 32 || (p.ntaps != 9 && p.ntaps != 1) || (p.ncls != 1 && p.ncls != 4))
        return (int)hipErrorInvalidValue;
    if (p.Hi >= 32767 || p.Wi >= 32767) return (int)hipErrorInvalidValue;
    if (spatial) {
        if (p.tiles || p.istride != 1 || p.ntaps != 9 || p.Ha % 8 || p.Wa % 16) return (int)hipErrorInvalidValue;
        if (p.Cout % 128 == 0) return launch<128, 128, 2, 2, true>(p, st);
        if (p.Cout % 64 == 0) return launch<128, 64, 2, 2, true>(p, st);
        return launch<128, 32, 4, 1, true>(p, st);
    }
    if (p.Cout % 128 == 0) return launch<128, 128, 2, 2, false>(p, st);
    if (p.Cout % 64 == 0) return launch<128, 64, 2, 2, false>(p, st);
    return launch<128, 32, 4, 1, false>(p, st);
}
