// Backward of the region-select modulated convolution on fp32 MFMA.
//
// Forward (conv_mfma.hip, spatial mode):  out_pre[p,co] = d[r(p),co] * sum_{tap,ci} W[co,tap,ci] s[r(p),ci] x[p+tap-1, ci]
// Given gz = dL/d(out_pre) (NHWC) this kernel produces, in ONE pass over the same MACs as the forward,
//   dx[q,ci]  = sum_tap' s[r(p),ci] * T[q,tap',ci],   T[q,tap',ci] = sum_co Wt[tap',ci,co] * d[r(p),co] * gz[p,co],
//               p = the output pixel that input pixel q feeds through tap' (p = q + tap' - 1 with the flipped taps;
//               for the polyphase up-conv p = 2*(q + tap' - 1) + phase)
//   ds[b,r,ci] += sum_{q,tap' : r(p) = r} x[q,ci] * T[q,tap',ci]          (grad of the loss w.r.t. the modulation s)
// The style scale depends on the region of the *tap-shifted* pixel p and on the output column ci, so it cannot be
// folded into either GEMM operand: each (phase, tap') group accumulates into a temporary MFMA accumulator T which
// is then scaled by s[r(p)][ci] into the final accumulator -- the same T, multiplied by x, feeds the ds reduction
// (LDS atomics per tile, one global atomic per (region, ci) per tile).  The demodulation d[r(p)][co] is applied to the
// A fragment on its way into the MFMA, exactly like s in the forward kernel.
// GEMM view: rows = input pixels q (8x16 tiles), N = Cin (64 per block), K = (phase, tap', Cout); per stage one
// (tap', 32-channel) slice of gz (128 rows) and of Wt (64 rows) is staged global->VGPR->LDS, double buffered.
#include "common.h"

namespace {

constexpr int KC = 32, LDA = 36, NTHR = 256;
constexpr int BM = 128, TH = 8, TW = 16, MAXR = 16;
// BN = 64: 4 waves as 2 (M) x 2 (N), wave tile 64 x 32 (TM = 2);  BN = 32 (Cin = 32 layers): 4 x 1, wave tile 32 x 32

template <int BN>
struct BwdSmem {
    int out_off[BM];                 // dx pixel index of each row, -1 if outside
    int yx[BM];                      // (qy << 16) | qx
    unsigned char grp[36][BM];       // region of the tap-shifted pixel per (phase*9+tap', row); 255 = outside
    float sS[MAXR * BN];             // s[r][n0 + n]
    float dS[128 / BN][MAXR * BN];   // ds partial sums of this tile, one slab per wave row (wm = 0 .. 4/(BN/32)-1): no atomics
    int regmask;                     // regions touched by any (row, tap) of the tile
    int pad_[3];
    float sD[2][MAXR * LDA];         // d[r][k0 .. k0+32) of the current stage
    float A[2][BM * LDA];
    float B[2][BN * LDA];
};

template <int BN>
__global__ __launch_bounds__(NTHR, 2) void conv_bwd_kernel(const e4s_conv_bwd_params p, const int ntn, const int gsplit,
                                                           float* __restrict__ dx_ws) {
    constexpr int WN = BN / 32, WM = 4 / WN, TM = BM / (WM * 32);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    BwdSmem<BN>& sm = *reinterpret_cast<BwdSmem<BN>*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // gsplit > 1 (low resolutions: a handful of tiles, a serial loop of up to 36 tap groups x Cy/32 stages): the tap groups
    // are split over gsplit blocks per tile; each writes its partial dx / ds, added in group order afterwards
    const int logical0 = xcd_remap(blockIdx.x, gridDim.x);
    const int gs = logical0 % gsplit;
    const int logical = logical0 / gsplit;
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const int tx_n = (p.Wx + TW - 1) / TW, per_img = ((p.Hx + TH - 1) / TH) * tx_n;
    const int tb = mt / per_img;
    const int rem0 = mt - tb * per_img;
    const int tyb = rem0 / tx_n, txb = rem0 - tyb * tx_n;
    const int os = (p.ncls == 4) ? 2 : 1;
    const int R = p.labels ? p.R : 1;
    const int ngroups = p.ncls * 9;

    // ---- per-row metadata -------------------------------------------------------------
    if (tid < BM) {
        const int qy = tyb * TH + tid / TW, qx = txb * TW + tid % TW;
        const bool valid = qy < p.Hx && qx < p.Wx;
        sm.out_off[tid] = valid ? (tb * p.Hx + qy) * p.Wx + qx : -1;
        sm.yx[tid] = (qy << 16) | qx;
    }
    if (tid == 0) sm.regmask = 0;
    __syncthreads();
    for (int t = tid; t < ngroups * BM; t += NTHR) {
        const int grp = t / BM, row = t - grp * BM;
        const int ph = grp / 9, tp = grp - ph * 9;
        const int qy = tyb * TH + row / TW + tp / 3 - 1, qx = txb * TW + row % TW + tp % 3 - 1;
        unsigned char r = 255;
        if ((unsigned)qy < (unsigned)p.Hx && (unsigned)qx < (unsigned)p.Wx) {
            r = 0;
            if (p.labels) {
                const int oy = qy * os + (ph >> 1), ox = qx * os + (ph & 1);
                const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Hy)), p.Hm - 1);
                const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wy)), p.Wm - 1);
                r = p.labels[((size_t)tb * p.Hm + sy) * p.Wm + sx];
            }
        }
        sm.grp[grp][row] = r;
        if (r != 255) atomicOr(&sm.regmask, 1 << r);
    }
    for (int t = tid; t < R * BN; t += NTHR) {
        const int r = t / BN, n = t - r * BN;
        sm.sS[t] = p.s ? p.s[((size_t)tb * R + r) * p.Cx + n0 + n] : 1.f;
#pragma unroll
        for (int j = 0; j < 128 / BN; ++j) sm.dS[j][t] = 0.f;
    }
    __syncthreads();

    const int c4 = (tid & 7) * 4, r0 = tid >> 3;
    int a_yx[BM / 32];
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) a_yx[j] = sm.yx[r0 + 32 * j];
    const float* dtab = p.d ? p.d + (size_t)tb * R * p.Cy : nullptr;

    f32x4 pa[BM / 32], pb[BN / 32], pd = {1.f, 1.f, 1.f, 1.f};
    const int nchunk = p.Cy / KC;

    auto fetch = [&](int grp, int c0) {
        const int ph = grp / 9, tp = grp - ph * 9;
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;
        const float* wp = p.wt + ((size_t)grp * p.Cx + n0) * p.Cy + c0 + c4;
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(r0 + 32 * j) * p.Cy);
#pragma unroll
        for (int j = 0; j < BM / 32; ++j) {
            const int qy = (a_yx[j] >> 16) + dy, qx = (a_yx[j] & 0xffff) + dx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)qy < (unsigned)p.Hx && (unsigned)qx < (unsigned)p.Wx) {
                const int oy = qy * os + (ph >> 1), ox = qx * os + (ph & 1);
                v = *reinterpret_cast<const f32x4*>(p.gz + ((size_t)(tb * p.Hy + oy) * p.Wy + ox) * p.Cy + c0 + c4);
            }
            pa[j] = v;
        }
        if (dtab && tid < R * 8) pd = *reinterpret_cast<const f32x4*>(dtab + (size_t)(tid >> 3) * p.Cy + c0 + c4);
    };
    auto store = [&](int buf) {
        float* da = sm.A[buf] + r0 * LDA + c4;
#pragma unroll
        for (int j = 0; j < BM / 32; ++j) *reinterpret_cast<f32x4*>(da + 32 * j * LDA) = pa[j];
        float* db = sm.B[buf] + r0 * LDA + c4;
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) *reinterpret_cast<f32x4*>(db + 32 * j * LDA) = pb[j];
        if (dtab && tid < R * 8) *reinterpret_cast<f32x4*>(sm.sD[buf] + (tid >> 3) * LDA + c4) = pd;
    };

    int arow[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) arow[tm] = (wm * TM + tm) * 32 + li;
    const int brow = (wn * 32 + li) * LDA;

    // forward input of this tile in accumulator layout (rows x column li), for ds = sum x * T
    f32x16 xv[TM];
    const int regmask = sm.regmask;
    if (p.ds) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = sm.out_off[(wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
                xv[tm][r] = off >= 0 ? p.x[(size_t)off * p.Cx + n0 + wn * 32 + li] : 0.f;
            }
    }
    f32x16 acc[TM], tmp[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

    const int gper = (ngroups + gsplit - 1) / gsplit;
    const int g_lo = gs * gper, g_hi = min(g_lo + gper, ngroups);
    if (g_lo < g_hi) fetch(g_lo, 0);
    store(0);
    __syncthreads();

    int s = 0;
    for (int grp = g_lo; grp < g_hi; ++grp) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmp[tm][r] = 0.f;
        int srow[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int g = sm.grp[grp][arow[tm]];
            srow[tm] = (g == 255 ? 0 : g) * LDA;
        }
        for (int ch = 0; ch < nchunk; ++ch, ++s) {
            int ngrp = grp, nch = ch + 1;
            if (nch == nchunk) { nch = 0; ++ngrp; }
            const bool more = ngrp < g_hi;
            if (more) fetch(ngrp, nch * KC);
            {
                const int buf = s & 1;
                const float* Ab = sm.A[buf];
                const float* Bb = sm.B[buf];
                const float* Db = sm.sD[buf];
#pragma unroll
                for (int kk = 0; kk < KC / 8; ++kk) {
                    f32x4 a[TM];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        a[tm] = *reinterpret_cast<const f32x4*>(Ab + arow[tm] * LDA + kk * 8 + kh * 4);
                        if (dtab) a[tm] *= *reinterpret_cast<const f32x4*>(Db + srow[tm] + kk * 8 + kh * 4);
                    }
                    const f32x4 b = *reinterpret_cast<const f32x4*>(Bb + brow + kk * 8 + kh * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            tmp[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[e], tmp[tm], 0, 0, 0);
                }
            }
            if (more) store((s + 1) & 1);
            __syncthreads();
        }
        // ---- tap-group epilogue: final += s[r(p)][ci] * T ;  ds[r(p)][ci] += x[q,ci] * T ----------------------
        // ds: per region present, an in-register masked sum over this lane's 16*TM rows, one shuffle across the two
        // row halves of the wave, then a plain add into the wave-row's own LDS slab (LDS float atomics are ~100x slower)
        const int ncol = wn * 32 + li;
        int gq[TM][16];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int g = sm.grp[grp][row];
                gq[tm][r] = g;
                if (g != 255) acc[tm][r] += sm.sS[g * BN + ncol] * tmp[tm][r];    // outside the image T is exactly 0
            }
        }
        if (p.ds) {
            for (int rr = 0; rr < R; ++rr) {
                if (!((regmask >> rr) & 1)) continue;
                float part = 0.f;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part += (gq[tm][r] == rr) ? xv[tm][r] * tmp[tm][r] : 0.f;
                part += __shfl_xor(part, 32, 64);
                if (kh == 0) sm.dS[wm][rr * BN + ncol] += part;
            }
        }
    }

    // ---- store dx, flush ds -----------------------------------------------------------------------
    // 4x4 quad transposes turn the accumulators' one-column-per-lane layout into 16-byte NHWC stores (common.h)
    {
        float* dxo = gsplit > 1 ? dx_ws + (size_t)gs * ((size_t)p.B * p.Hx * p.Wx * p.Cx) : p.dx;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v0 = acc[tm][4 * g4], v1 = acc[tm][4 * g4 + 1], v2 = acc[tm][4 * g4 + 2], v3 = acc[tm][4 * g4 + 3];
                quad_transpose4(v0, v1, v2, v3, li);
                const int off = sm.out_off[(wm * TM + tm) * 32 + (li & 3) + 8 * g4 + 4 * kh];
                if (off >= 0)
                    *reinterpret_cast<f32x4*>(dxo + (size_t)off * p.Cx + n0 + wn * 32 + (li & ~3)) = f32x4{v0, v1, v2, v3};
            }
        }
    }
    if (p.ds) {
        __syncthreads();
        for (int t = tid; t < R * BN; t += NTHR) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 128 / BN; ++j) v += sm.dS[j][t];
            // one slot per (tile of the sample, region, channel); e4s_conv_bwd_mfma_f32 adds the tiles in order afterwards
            p.ds_ws[((size_t)rem0 * gsplit + gs) * ((size_t)p.B * R * p.Cx) + ((size_t)tb * R + t / BN) * p.Cx + n0 + (t % BN)] = v;
        }
    }
}

// [ncls][9][Cout][Cin] (forward layout) -> [ncls][9][Cin][Cout] with the taps flipped (tap' = 8 - tap)
__global__ void pack_bwd_kernel(const float* __restrict__ w, float* __restrict__ wt, int ncls, int cout, int cin) {
    const int64_t n = (int64_t)ncls * 9 * cout * cin;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % cout);
    int64_t r = i / cout;
    const int ci = (int)(r % cin); r /= cin;
    const int tp = (int)(r % 9);
    const int cls = (int)(r / 9);
    wt[i] = w[(((int64_t)cls * 9 + (8 - tp)) * cout + co) * cin + ci];
}

// the same through a 32 x 32 LDS tile (Cout, Cin multiples of 32): both sides coalesced -- 64 of these per config-5 G step (13.8 -> ~5 us each)
__global__ __launch_bounds__(256) void pack_bwd_tiled_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int ct = blockIdx.z, cls = ct / 9, tp = ct - cls * 9;
    const float* src = w + ((size_t)cls * 9 + (8 - tp)) * cout * cin;
    float* dst = wt + (size_t)ct * cin * cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[ty + 8 * j][tx] = src[(size_t)(co0 + ty + 8 * j) * cin + ci0 + tx];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(size_t)(ci0 + ty + 8 * j) * cout + co0 + tx] = t[tx][ty + 8 * j];
}

template <int BN>
int launch_bwd(const e4s_conv_bwd_params& p, int mtiles, int gsplit, float* dx_ws, hipStream_t st) {
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(conv_bwd_kernel<BN>), (int)sizeof(BwdSmem<BN>), smem_set)) return e;
    const int ntn = p.Cx / BN;
    hipLaunchKernelGGL(conv_bwd_kernel<BN>, dim3(mtiles * ntn * gsplit), dim3(NTHR), sizeof(BwdSmem<BN>), st, p, ntn, gsplit,
                       dx_ws);
    E4S_CHECK_LAUNCH();
    return 0;
}

}  // namespace

namespace {
// tap-group split of the low-resolution layers: enough blocks to cover the chip, each with >= 1 group
int bwd_gsplit(const e4s_conv_bwd_params& p) {
    const int per_img = ((p.Hx + TH - 1) / TH) * ((p.Wx + TW - 1) / TW);
    const int blocks = p.B * per_img * (p.Cx / (p.Cx % 64 == 0 ? 64 : 32));
    const int ngroups = p.ncls * 9;
    if (blocks >= 256) return 1;
    int gs = 512 / blocks;
    if (gs > ngroups) gs = ngroups;
    return gs < 1 ? 1 : gs;
}
}  // namespace

extern "C" int e4s_conv_bwd_mfma_f32(const e4s_conv_bwd_params* pp, void* stream) {
    const e4s_conv_bwd_params& p = *pp;
    if (p.Cx % 32 || p.Cy % KC || (p.ncls != 1 && p.ncls != 4)) return (int)hipErrorInvalidValue;
    if (p.labels && (p.R < 1 || p.R > MAXR)) return (int)hipErrorInvalidValue;
    if (p.Hy != p.Hx * (p.ncls == 4 ? 2 : 1) || p.Wy != p.Wx * (p.ncls == 4 ? 2 : 1)) return (int)hipErrorInvalidValue;
    if (p.ds && !p.x) return (int)hipErrorInvalidValue;
    const int per_img = ((p.Hx + TH - 1) / TH) * ((p.Wx + TW - 1) / TW);
    const int mtiles = p.B * per_img;
    if (mtiles <= 0) return 0;
    const int gsplit = bwd_gsplit(p);
    if ((p.ds || gsplit > 1) && !p.ds_ws) return (int)hipErrorInvalidValue;
    const int R = p.labels ? p.R : 1;
    const int64_t nds = (int64_t)p.B * R * p.Cx, ndx = (int64_t)p.B * p.Hx * p.Wx * p.Cx;
    // workspace: [ds parts + their chunk sums][dx parts (gsplit > 1)]
    float* dx_ws = p.ds_ws + (p.ds ? e4s_reduce_parts_ws_floats(per_img * gsplit, nds) : 0);
    const int rc = (p.Cx % 64 == 0) ? launch_bwd<64>(p, mtiles, gsplit, dx_ws, as_stream(stream))
                                    : launch_bwd<32>(p, mtiles, gsplit, dx_ws, as_stream(stream));
    if (rc) return rc;
    if (gsplit > 1)
        if (int e = e4s_reduce_parts_f32(dx_ws, p.dx, gsplit, ndx, 1.f, stream)) return e;
    if (!p.ds) return 0;
    return e4s_reduce_parts_f32(p.ds_ws, p.ds, per_img * gsplit, nds, 1.f, stream);
}

extern "C" int64_t e4s_conv_bwd_ws_floats(const e4s_conv_bwd_params* pp) {
    const e4s_conv_bwd_params& p = *pp;
    const int per_img = ((p.Hx + TH - 1) / TH) * ((p.Wx + TW - 1) / TW);
    const int gsplit = bwd_gsplit(p);
    int64_t n = 0;
    if (p.ds) n += e4s_reduce_parts_ws_floats(per_img * gsplit, (int64_t)p.B * (p.labels ? p.R : 1) * p.Cx);
    if (gsplit > 1) n += e4s_reduce_parts_ws_floats(gsplit, (int64_t)p.B * p.Hx * p.Wx * p.Cx);
    return n;
}

extern "C" int e4s_pack_taps_bwd_f32(const float* w, float* wt, int ncls, int cout, int cin, void* stream) {
    const int64_t n = (int64_t)ncls * 9 * cout * cin;
    if (n <= 0) return 0;
    if (cout % 32 == 0 && cin % 32 == 0 && ncls * 9 <= 65535 && cout / 32 <= 65535) {
        hipLaunchKernelGGL(pack_bwd_tiled_kernel, dim3((unsigned)(cin / 32), (unsigned)(cout / 32), (unsigned)(ncls * 9)), dim3(256), 0, as_stream(stream),
                           w, wt, cout, cin);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(pack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, wt, ncls, cout, cin);
    E4S_CHECK_LAUNCH();
    return 0;
}
