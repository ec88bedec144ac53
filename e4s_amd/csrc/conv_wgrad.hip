// Weight gradient of the 3x3 / 1x1 convolutions (SURVEY.md 8(f) N1 "wgrad"; config 5).  Two kernels: the exact-fp32 MFMA kernel right below
// (region maps, stride 2, 1x1) and, since round 6, conv_wgrad_bf16x3_kernel (3x3, stride 1, no region map: split-bf16 operands on the bf16
// matrix cores; e4s_conv_wgrad_path tells which one a launch takes).  The contraction:
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
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same contraction on the bf16 matrix cores with split operands (round 6; 3x3, input stride 1, no region map): G = hi + lo and
// X = hi + lo as bf16, three v_mfma_f32_32x32x16_bf16 per product (hi x lo, lo x hi, hi x hi; fp32 accumulate) -- the arithmetic of the
// forward kernels (stride-2 3x3 convs too: see WB<IS> below).  The contraction index is the PIXEL, and the bf16 MFMA wants 8 consecutive k per lane, so the staging TRANSPOSES:
// a thread owns one channel and 8 consecutive pixels of a row (8 coalesced 4-byte loads, 256 B per wave and pixel), scales by the sample's
// s / d, splits and writes two 16-byte LDS vectors -- channel-major rows of [16 hi | 16 lo] (G) and [24 hi | 24 lo] per halo row (X).
// A k-step is one anchor row of 16 pixels; the three column taps of a row are the same two 16-byte chunks shifted by 0 / 1 / 2 elements in
// registers (v_alignbit), so X is read once per (k-step, tap row), not once per tap.  Anchor tile 4 x 16 (K = 64), X halo 6 x 18,
// block = 4 waves = 64 (co) x 64 (ci) x 9 taps as above; 55 KB of LDS: two blocks per CU.  Split-K slabs and the ordered reduction are
// the fp32 kernel's.  Error vs fp64: the 2^-17-class rounding of the split (tests/test_gpu_train.py: <= 2e-5 of max |dW|, as before).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// IS = 1: anchor tile 4 x 16, X halo 6 x 18 as three 8-pixel chunks per row.  IS = 2 (the stride-2 3x3 convs of encoder and Discriminator): anchor tile
// 2 x 16, X halo 5 x 33 DE-INTERLEAVED per row into the even columns (17 -> three chunks) and the odd columns (16 -> two chunks): tap column 0 reads
// even[a], 1 reads odd[a], 2 reads even[a + 1] -- the same aligned-chunk-plus-register-shift scheme.
constexpr int WB_TW = 16;
template <int IS>
struct WB {
    static constexpr int TH = IS == 1 ? 4 : 2;
    static constexpr int HH = TH * IS + 2 - (IS - 1);                   // 6 | 5 halo rows
    static constexpr int NCH = IS == 1 ? 3 : 5;                         // 8-pixel chunks per halo row (IS = 2: 3 even + 2 odd)
    static constexpr int RB = NCH * 32;                                 // bytes per (ci, halo row): [NCH chunks hi | NCH chunks lo]
    static constexpr int GROW = TH * 64 + 16;                           // bytes per co: [ay][16 hi | 16 lo] + pad (bank spread)
    static constexpr int XROW = HH * RB + 16;                           // bytes per ci + pad
    static constexpr int SMEM = BC * (GROW + XROW);                     // (+ 64 bytes of region labels behind it)
    static constexpr int XTASKS = HH * NCH, XROUNDS = (XTASKS + 3) / 4, GTASKS = TH * 2, GROUNDS = (GTASKS + 3) / 4;
};
constexpr int WB_TH = WB<1>::TH;                      // (the masked form exists for IS = 1 only: 64 anchors = one ballot)

__device__ __forceinline__ void split8_store(unsigned char* hi_dst, unsigned char* lo_dst, const float (&v)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16x2 hh = __builtin_convertvector(f32x2{v[2 * j], v[2 * j + 1]}, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, hh);
        const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
        const bf16x2 ll = __builtin_convertvector(f32x2{v[2 * j] - h0, v[2 * j + 1] - h1}, bf16x2);
        h[j] = hb;
        l[j] = __builtin_bit_cast(unsigned, ll);
    }
    *reinterpret_cast<u32x4*>(hi_dst) = h;
    *reinterpret_cast<u32x4*>(lo_dst) = l;
}

// elements tx .. tx + 7 of the 16 bf16 in (c0, c1)
template <int TX>
__device__ __forceinline__ bf16x8 shifted(const u32x4 c0, const u32x4 c1) {
    u32x4 r;
    if (TX == 0) r = c0;
    else if (TX == 2) r = u32x4{c0[1], c0[2], c0[3], c1[0]};
    else r = u32x4{__builtin_amdgcn_alignbit(c0[1], c0[0], 16), __builtin_amdgcn_alignbit(c0[2], c0[1], 16),
                   __builtin_amdgcn_alignbit(c0[3], c0[2], 16), __builtin_amdgcn_alignbit(c1[0], c0[3], 16)};
    return __builtin_bit_cast(bf16x8, r);
}

template <bool MASKED, int IS>
__global__ __launch_bounds__(NTHR, 2) void conv_wgrad_bf16x3_kernel(const e4s_conv_wgrad_params p, const int nct, const int nnt, const int nsplit,
                                                                    const int tx_n, const int per_img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using T = WB<IS>;
    static_assert(!MASKED || IS == 1, "region maps: stride-1 layers only");
    unsigned char* sG = smem_raw;                       // [64 co][GROW]
    unsigned char* sX = smem_raw + BC * T::GROW;        // [64 ci][XROW]
    signed char* sgrp = reinterpret_cast<signed char*>(smem_raw + T::SMEM);       // MASKED: region of the tile's 64 anchors (-1: outside)
    const int tid = threadIdx.x, lane0 = tid & 63, wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = lane0, wave = wave0;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int nt = bid % nnt, ct = bid / nnt;
    const int co0 = ct * BC, ci0 = nt * BC;
    const int ntiles = p.B * per_img;
    const int R = MASKED ? p.R : 1;
    const bool co_ok = co0 + lane < p.Cout, ci_ok = ci0 + lane < p.Cin;
    const bool wave_live = co0 + wm * 32 < p.Cout && ci0 + wn * 32 < p.Cin;       // (Cin or Cout = 32, 96, 160: half tiles)

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = split; tile < ntiles; tile += nsplit) {
        const int tb = tile / per_img;
        const int rem = tile - tb * per_img;
        const int tyb = rem / tx_n, txb = rem - tyb * tx_n;
        int glane = 0;
        if (MASKED) {
            // The style of a product belongs to the ANCHOR (model.py:386-400), so a halo pixel is scaled differently by anchors of different
            // regions: a tile is contracted once per region PRESENT in it -- G masked to the region's anchors (x d[region]), the whole halo
            // x s[region], k-steps (anchor rows) without an anchor of the region skipped.  Interior tiles have one region: one pass, the
            // unmasked kernel's work.
            __syncthreads();                                    // the previous tile's last reader of sgrp is past its ballots
            if (tid < WB_TH * WB_TW) {
                const int ya = tyb * WB_TH + tid / WB_TW, xa = txb * WB_TW + tid % WB_TW;
                sgrp[tid] = (ya < p.Ha && xa < p.Wa) ? (signed char)label_of(p, tb, ya * p.ostride + p.py, xa * p.ostride + p.px) : (signed char)-1;
            }
            __syncthreads();
            glane = sgrp[lane];
        }
        for (int g = 0; g < R; ++g) {
            unsigned long long m = ~0ull;                       // anchors of this pass (bit = 16 ay + ax), block-uniform
            if (MASKED) {
                m = __builtin_amdgcn_ballot_w64(glane == g);
                if (m == 0) continue;
            }
            const int grp = tb * R + g;
            __syncthreads();                                    // previous tile / pass fully consumed
            // (opaque copies: everything below that depends only on the lane / wave is loop-invariant, and with three loop levels the compiler
            // hoists it all in front of the tile loop -- 100+ values that then live in scratch across the MFMAs)
            int lane = lane0, wave = wave0;
            asm volatile("" : "+v"(lane));
            asm volatile("" : "+s"(wave));
            // ---- G: lane = co; wave w takes the (anchor row, half) pairs w and w + 4: 8 anchors each.  Addresses: a wave-uniform base per sample
            // and ONE 32-bit element offset per row, stepped by a uniform stride (64-bit per-load addresses spilled 100+ registers here) ----
            const float dv = (p.d && co_ok) ? p.d[(size_t)grp * p.Cout + co0 + lane] : 1.f;
            const float* gzb = p.gz + (size_t)tb * p.Ho * p.Wo * p.Cout;
            const float* xsb = p.x + (size_t)tb * p.Hi * p.Wi * p.Cin;
            const int gstep = p.ostride * p.Cout;
            float gv[T::GROUNDS][8];
#pragma unroll
            for (int r = 0; r < T::GROUNDS; ++r) {
                const int j = wave + 4 * r, ay = j >> 1, half = j & 1;
                const int ya = tyb * T::TH + ay, xa0 = txb * WB_TW + half * 8;
                const unsigned mj = (unsigned)(m >> ((ay * 16 + half * 8) & 63)) & 0xffu;          // wave-uniform
                int off = ((ya * p.ostride + p.py) * p.Wo + xa0 * p.ostride + p.px) * p.Cout + co0 + lane;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool on = MASKED ? ((mj >> e) & 1u) != 0 : (ya < p.Ha && xa0 + e < p.Wa);
                    gv[r][e] = (co_ok && on) ? gzb[off] : 0.f;
                    off += gstep;
                }
            }
            // ---- X: lane = ci; the (halo row, 8-pixel chunk) pairs over the waves ----
            const float sv = (p.s && ci_ok) ? p.s[(size_t)grp * p.Cin + ci0 + lane] : 1.f;
#pragma unroll
            for (int r = 0; r < T::GROUNDS; ++r) {
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[r][e] *= dv;
                const int j = wave + 4 * r, ay = j >> 1, half = j & 1;
                unsigned char* d = sG + lane * T::GROW + ay * 64 + half * 16;
                split8_store(d, d + 32, gv[r]);
            }
#pragma unroll
            for (int r = 0; r < T::XROUNDS; ++r) {
                const int j = wave + 4 * r;                         // wave-uniform
                if (j < T::XTASKS) {
                    const int hy = j / T::NCH, ch = j - hy * T::NCH;
                    // halo pixel (hy, hx) = input (IS * tile origin + hy - 1 + tap_shift, ... + hx ...); IS = 2: chunk ch < 3 holds hx = 2 (8 ch + e),
                    // chunk ch >= 3 holds hx = 2 (8 (ch - 3) + e) + 1
                    const int hx0 = IS == 1 ? ch * 8 : (ch < 3 ? 16 * ch : 16 * (ch - 3) + 1);
                    const int nvalid = IS == 1 ? 18 - ch * 8 : (ch < 3 ? 17 - ch * 8 : 8);       // chunk entries that exist
                    const int iy = tyb * T::TH * IS + hy - 1 + p.tap_shift, ix0 = txb * WB_TW * IS + hx0 - 1 + p.tap_shift;
                    const bool row_ok = ci_ok && (unsigned)iy < (unsigned)p.Hi;
                    int off = (iy * p.Wi + ix0) * p.Cin + ci0 + lane;
                    float xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        xv[e] = (row_ok && e < nvalid && (unsigned)(ix0 + IS * e) < (unsigned)p.Wi) ? xsb[off] : 0.f;
                        off += IS * p.Cin;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] *= sv;
                    unsigned char* d = sX + lane * T::XROW + hy * T::RB + ch * 16;
                    split8_store(d, d + T::NCH * 16, xv);
                }
            }
            __syncthreads();
            if (!wave_live) continue;
            // ---- contraction: k-step = anchor row ay (16 anchors), 9 taps x 3 MFMAs ----
            const unsigned char* gb = sG + (wm * 32 + li) * T::GROW + kh * 16;
            const unsigned char* xb = sX + (wn * 32 + li) * T::XROW + kh * 16;
#pragma unroll
            for (int ay = 0; ay < T::TH; ++ay) {
                if (MASKED && ((m >> ((ay * 16) & 63)) & 0xffffull) == 0) continue;           // no anchor of the region in this row: G is zero (-8 % on the step's masked layers)
                const bf16x8 Gh = *reinterpret_cast<const bf16x8*>(gb + ay * 64), Gl = *reinterpret_cast<const bf16x8*>(gb + ay * 64 + 32);
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const unsigned char* xr = xb + (ay * IS + ty) * T::RB;
                    constexpr int LO = T::NCH * 16;
                    const u32x4 h0 = *reinterpret_cast<const u32x4*>(xr), h1 = *reinterpret_cast<const u32x4*>(xr + 16);
                    const u32x4 l0 = *reinterpret_cast<const u32x4*>(xr + LO), l1 = *reinterpret_cast<const u32x4*>(xr + LO + 16);
                    bf16x8 Xh[3], Xl[3];
                    if (IS == 1) {
                        Xh[0] = shifted<0>(h0, h1), Xh[1] = shifted<1>(h0, h1), Xh[2] = shifted<2>(h0, h1);
                        Xl[0] = shifted<0>(l0, l1), Xl[1] = shifted<1>(l0, l1), Xl[2] = shifted<2>(l0, l1);
                    } else {                                        // even[a], odd[a], even[a + 1]
                        const u32x4 oh = *reinterpret_cast<const u32x4*>(xr + 48), ol = *reinterpret_cast<const u32x4*>(xr + LO + 48);
                        Xh[0] = shifted<0>(h0, h1), Xh[1] = __builtin_bit_cast(bf16x8, oh), Xh[2] = shifted<1>(h0, h1);
                        Xl[0] = shifted<0>(l0, l1), Xl[1] = __builtin_bit_cast(bf16x8, ol), Xl[2] = shifted<1>(l0, l1);
                    }
                    // small products first; consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Gh, Xl[tx], acc[ty * 3 + tx], 0, 0, 0);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Gl, Xh[tx], acc[ty * 3 + tx], 0, 0, 0);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Gh, Xh[tx], acc[ty * 3 + tx], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial slab of this split: ws[split][tap][co][ci]; 4x4 quad transposes turn four 4-byte stores into one of 16 bytes (common.h) ----
    const size_t slab = (size_t)9 * p.Cout * p.Cin;
    float* out = p.ws + (size_t)split * slab;
    const int ci = ci0 + wn * 32 + (li & ~3);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a0 = acc[t][4 * g], a1 = acc[t][4 * g + 1], a2 = acc[t][4 * g + 2], a3 = acc[t][4 * g + 3];
            quad_transpose4(a0, a1, a2, a3, lane);
            const int co = co0 + wm * 32 + 8 * g + 4 * kh + (lane & 3);
            if (co < p.Cout && ci < p.Cin) *reinterpret_cast<f32x4*>(out + ((size_t)t * p.Cout + co) * p.Cin + ci) = f32x4{a0, a1, a2, a3};
        }
}

template <int IS>
constexpr int wg_smem() { return (WgTile<IS>::NA * BC + WgTile<IS>::NH * BC + MAXR * BC) * 4 + WgTile<IS>::NA * 4; }

static bool wg_bf16x3(const e4s_conv_wgrad_params& p) {            // E4S_WGRAD_BF16X3 (A/B switch): 0 = the exact-fp32 kernel everywhere, 2 = only without a region map
    static const int on = [] { const char* e = getenv("E4S_WGRAD_BF16X3"); return e ? atoi(e) : 1; }();
    static const int min_masked = [] { const char* e = getenv("E4S_WGRAD_MASKED_MIN_ANCHORS"); return e ? atoi(e) : 0; }();
    if (!on || p.ntaps != 9) return false;
    if ((int64_t)p.Ho * p.Wo * p.Cout >= (1ll << 31) || (int64_t)p.Hi * p.Wi * p.Cin >= (1ll << 31)) return false;      // 32-bit element offsets per sample
    if (p.istride == 2) return !p.labels && on != 3;                        // (3: stride 1 only, the A/B switch of the stride-2 form)
    return !p.labels || (on == 1 && p.Ha * p.Wa >= min_masked);
}

int wg_nsplit(const e4s_conv_wgrad_params& p) {
    if (wg_bf16x3(p)) {
        // two blocks per CU, but at least 4 anchor tiles (K = 256) per block where the layer has them: a block's 147 KB slab (write, read by the
        // reduction) and its 36 wide stores per lane cost about what two tiles of MFMAs do
        static const int target = [] { const char* e = getenv("E4S_WGRAD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
        const int th = p.istride == 2 ? WB<2>::TH : WB<1>::TH;
        const int ntiles = p.B * ((p.Ha + th - 1) / th) * ((p.Wa + WB_TW - 1) / WB_TW);
        const int ctiles = ((p.Cout + BC - 1) / BC) * ((p.Cin + BC - 1) / BC);
        int ns = (target + ctiles - 1) / ctiles;
        // (a masked tile is contracted once per region present in it: 2-4 passes at low resolutions; a stride-2 tile holds 32 anchors, not 64)
        const int per_block = p.labels ? 1 : p.istride == 2 ? 8 : 4;
        if (ns > (ntiles + per_block - 1) / per_block) ns = (ntiles + per_block - 1) / per_block;
        if (ns > 2048) ns = 2048;
        return ns < 1 ? 1 : ns;
    }
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

template <bool MASKED, int IS>
int launch_wg_bf16x3_t(const e4s_conv_wgrad_params& p, hipStream_t st) {
    using T = WB<IS>;
    auto kern = conv_wgrad_bf16x3_kernel<MASKED, IS>;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), T::SMEM + 64, smem_set)) return e;
    const int nct = (p.Cout + BC - 1) / BC, nnt = (p.Cin + BC - 1) / BC;
    const int tx_n = (p.Wa + WB_TW - 1) / WB_TW, per_img = ((p.Ha + T::TH - 1) / T::TH) * tx_n;
    const int ns = wg_nsplit(p);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nct * nnt * ns)), dim3(NTHR), T::SMEM + 64, st, p, nct, nnt, ns, tx_n, per_img);
    E4S_CHECK_LAUNCH();
    return 0;
}

int launch_wg_bf16x3(const e4s_conv_wgrad_params& p, hipStream_t st) {
    if (p.istride == 2) return launch_wg_bf16x3_t<false, 2>(p, st);
    return p.labels ? launch_wg_bf16x3_t<true, 1>(p, st) : launch_wg_bf16x3_t<false, 1>(p, st);
}

}  // namespace

extern "C" int64_t e4s_conv_wgrad_ws_floats(const e4s_conv_wgrad_params* p) {
    return e4s_reduce_parts_ws_floats(wg_nsplit(*p), (int64_t)p->ntaps * p->Cout * p->Cin);
}

extern "C" int e4s_conv_wgrad_path(const e4s_conv_wgrad_params* p) { return p && wg_bf16x3(*p) ? 1 : 0; }

extern "C" int e4s_conv_wgrad_f32(const e4s_conv_wgrad_params* pp, void* stream) {
    const e4s_conv_wgrad_params& p = *pp;
    if (p.Cin % 32 || p.Cout % 32 || (p.ntaps != 9 && p.ntaps != 1) || (p.istride != 1 && p.istride != 2) ||
        (p.ostride != 1 && p.ostride != 2) || !p.ws || (p.labels && (p.R < 1 || p.R > MAXR)) || p.B < 1)
        return (int)hipErrorInvalidValue;
    if ((p.Ha - 1) * p.ostride + p.py >= p.Ho || (p.Wa - 1) * p.ostride + p.px >= p.Wo) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int rc;
    if (wg_bf16x3(p)) rc = launch_wg_bf16x3(p, st);
    else if (p.istride == 1) rc = p.ntaps == 9 ? launch_wg<1, 9>(p, st) : launch_wg<1, 1>(p, st);
    else rc = p.ntaps == 9 ? launch_wg<2, 9>(p, st) : launch_wg<2, 1>(p, st);
    if (rc) return rc;
    return e4s_reduce_parts_f32(p.ws, p.dw, wg_nsplit(p), (int64_t)p.ntaps * p.Cout * p.Cin, 1.f, stream);
}
