// 3x3 stride-1 convolution as Winograd F(2,3) ALONG THE IMAGE ROWS on the bf16 matrix cores at fp32-class accuracy (split bf16, three
// MFMAs per product, as conv_bf16x3.hip): 4 transform positions per PAIR of output columns instead of 3 x 2 taps -- 12 multiply-adds per
// output pair and input channel where the direct form spends 18: 1.5x fewer MFMAs for the same convolution.  The encoder's stride-1 3x3
// convs (helpers.py:128-137: InstanceNorm -> Conv2d -> PReLU -> Conv2d; 44 % of a face swap's GPU time on the direct kernel) are the
// target: VERDICT r3 'next' 4 asked for a measured MAC-reduction prototype.  Why 1-D and not F(2x2,3x3): the 2-D form needs 16 transform
// positions per 2x2 outputs -- 16 accumulator sets per (tile, channel) tile, 128 KB of split operands per 16-channel K step of a 64 x 64
// tile, and ~100 VALU of input transform + split per (tile, channel) against 48 MFMAs; on gfx950 that kernel is bound by VALU issue and L2
// weight traffic, not by the matrix pipe.  The 1-D form keeps the direct kernel's shape: the vertical taps stay
// taps (3 K stages per channel chunk), the 4 positions are 4 accumulator sets, the output transform happens in registers.
//
//   pair j of row y: d0..d3 = x[y][2j-1 .. 2j+2]        V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3
//   taps g0 g1 g2 of weight row ky:                     U0 = g0   U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2
//   M_p[y][j][co] = sum_ky sum_ci V_p[y + ky - 1][j][ci] * U_p[ky][ci][co]
//   out[y][2j] = M0 + M1 + M2        out[y][2j+1] = M1 - M2 - M3
// Error: the transforms are exact in fp32 up to one rounding per add (coefficients +-1, 1/2); on top of the 2^-16 split the measured
// growth is 1.7x per layer (tools/winograd_error_study.py: 7.8e-6 vs 4.6e-6 of the output scale on 512 -> 512 @32^2; style vectors of the
// whole encoder 1.3e-4 vs 7.9e-5 for the 2-D form, less for this one).
//
// Block = 512 threads = 8 waves, tile = 16 x 16 output pixels (128 GEMM rows = 16 rows x 8 pairs) x 128 output channels x 4 positions.
// Wave (wm, wn) of 2 x 4 owns 64 rows x 32 channels of ALL 4 positions (8 accumulators, 128 registers), so the output transform, bias,
// PReLU, the InstanceNorm statistics of the output and the NHWC stores work on registers, nothing is exchanged between waves.
// K step = one vertical tap x 16 input channels = 24 MFMAs per wave between two barriers; a chunk = 3 such stages.
// LDS (140 KB): V of the current and the next chunk (4 positions x 144 V-pixels [18 rows x 8 pairs] x 64-byte rows [16 hi | 16 lo]),
// U of the current and the next stage (4 positions x 128 channels x 64-byte rows).  Rows are 64 bytes WITHOUT padding; the 16-byte granule
// index is XORed with (row >> 2) & 3, which makes every ds_read_b128 fragment read conflict free (the lane groups of a wave64 b128 read
// hold rows r, r+12, r+20, r+24 (mod 4 classes) whose (row >> 2) & 3 differ) -- vertical taps shift a fragment by 8 rows and keep that.
// Weights arrive by LDS-DMA (global_load_lds_dwordx4; the DMA image is lane-linear, so the swizzle is applied to the SOURCE granule).
// The input transform runs on the way into LDS, one stage behind its loads: a work item = (V-pixel, 4 channels): four 16-byte loads in
// stage s; in stage s+1, folded between the storer wave's own MFMAs: [InstanceNorm from a {mean, rstd} table in LDS: the mean cancels in
// the three differences], 16 values split to hi / lo, eight 8-byte LDS stores.  The 576 items of chunk c+1 are spread over the three
// stages of chunk c, three waves per stage, rotating through the eight.
// Persistent (grid = min(tiles, CUs); the pipeline runs through the tile boundary) and split-K for launches of <= 128 tiles.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

int e4s_launch_wino1w(const e4s_conv_params& p, int ntn, int tx_n, int per_img, int ntiles, int grid, hipStream_t st);      // conv_wino1w.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int NTHR = 512;
constexpr int TH = 16, TW = 16, NPAIR = TW / 2;
constexpr int VROWS = (TH + 2) * NPAIR;          // 144 V-pixels per position and chunk
constexpr int BN = 128, KC = 16, ROWB = 64;
constexpr int A_PLANE = VROWS * ROWB, A_BYTES = 4 * A_PLANE;        // 36 864
constexpr int B_PLANE = BN * ROWB, B_BYTES = 4 * B_PLANE;           // 32 768
#ifndef E4S_WINO_READS_FIRST
#define E4S_WINO_READS_FIRST 1
#endif
// wave tile 1 (64 x 64 x 2 positions) from this many 16-channel chunks per block on.  Measured (profiles/r05_wino_wave_tile.json, same-box
// alternation): 512 -> 512 @32^2 -2 % (six pairs of six), 256 -> 256 @64^2 -2.6 %, the K = 128 / 64 layers +2 ... +4 % (the exchange in the
// epilogue is per tile): the 512-channel layers only.  The 33 % fewer fragment reads it was built for bought far less than the ablation that
// motivated it (-11 % with a third of the reads skipped): that gain was the shorter dependency chain of the ablated variant, not LDS bandwidth.
constexpr int WT_MIN_CHUNKS = 32;
constexpr int NITEMS = VROWS * 4;                                   // 576 (V-pixel, 4-channel group) items per chunk
constexpr int SMEM_WINO = 2 * A_BYTES + 2 * B_BYTES + 4 * BN * 2 * 8;       // + Cin * 16 bytes of {mean, rstd} tables when XF == 2 (launcher)

__device__ __forceinline__ int swz(int row, int g) { return row * ROWB + ((g ^ ((row >> 2) & 3)) << 4); }

// hi / lo split of 4 floats -> two 8-byte LDS stores
__device__ __forceinline__ void split_store4(unsigned char* hi_dst, unsigned char* lo_dst, const f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const f32x4 r = v - __builtin_convertvector(h, f32x4);
    const bf16x4 l = __builtin_convertvector(r, bf16x4);
    *reinterpret_cast<bf16x4*>(hi_dst) = h;
    *reinterpret_cast<bf16x4*>(lo_dst) = l;
}

// w9 [9][Cout][Cin] (tap = ky * 3 + kx) -> U [3 ky][Cin/16][4 pos][Cout][16 hi | 16 lo] bf16
__global__ void wino_weights_kernel(const float* __restrict__ w9, unsigned char* __restrict__ out, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)3 * Cout * Cin;
    if (i >= total) return;
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int ky = (int)(i / ((int64_t)Cin * Cout));
    const float g0 = w9[((int64_t)(ky * 3 + 0) * Cout + co) * Cin + ci];
    const float g1 = w9[((int64_t)(ky * 3 + 1) * Cout + co) * Cin + ci];
    const float g2 = w9[((int64_t)(ky * 3 + 2) * Cout + co) * Cin + ci];
    const float u[4] = {g0, ((g0 + g2) + g1) * 0.5f, ((g0 + g2) - g1) * 0.5f, g2};
    const int chunk = ci / KC, k = ci % KC, nchunk = Cin / KC;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const __bf16 h = (__bf16)u[p];
        const __bf16 l = (__bf16)(u[p] - (float)h);
        unsigned char* row = out + ((((int64_t)ky * nchunk + chunk) * 4 + p) * Cout + co) * ROWB;
        reinterpret_cast<__bf16*>(row)[k] = h;
        reinterpret_cast<__bf16*>(row + 32)[k] = l;
        // the same values fragment-major (conv_wino1w.hip reads its B fragments straight from global memory): behind the image above, per
        // (ky, chunk, position, 32-channel block) 2 KB = [hi: lane -> (channel & 31, k-half)][lo], 16 bytes per lane
        if (Cout % 32 == 0) {
            unsigned char* fr = out + (int64_t)3 * nchunk * 4 * Cout * ROWB +
                                (((((int64_t)ky * nchunk + chunk) * 4 + p) * (Cout / 32) + co / 32) * 2 * 1024) + ((k >> 3) * 32 + (co & 31)) * 16;
            reinterpret_cast<__bf16*>(fr)[k & 7] = h;
            reinterpret_cast<__bf16*>(fr + 1024)[k & 7] = l;
        }
    }
}

// XF: 0 plain input, 2 InstanceNorm (x - mean) * rstd folded into the input transform (zero padding applies to the NORMALISED map)
// VAR: profiling variants (builds with -DE4S_ABLATIONS select them with env E4S_WINO_VAR; results are WRONG for VAR >= 1; product builds
// only instantiate VAR = 0): 1 no input-transform staging, 2 no weight staging, 3 neither, 4 neither and no fragment reads (MFMAs + barriers),
// 5 MFMAs only (no barriers), 6 everything but the MFMAs, 7 input items loaded but not transformed / stored, 8 transformed / stored but
// not loaded (stale registers)
struct WTile {              // one 16x16-pixel x 128-column output tile (x one K split: input-channel chunks [c_lo, c_hi))
    int n0, tb, ty0, tx0, slot, ks, c_lo, c_hi;
};

// PERSISTENT (round 4, second version): the grid is min(tiles, CUs); block b walks tiles b', b' + G, ... (b' = XCD-remapped b) and the stage
// pipeline runs straight through the tile boundary -- while the last chunk of tile t is contracted, the V of tile t+1's first chunk is
// loaded / transformed / stored and its first U stage arrives by DMA, so a tile's prologue is never exposed (it was ~4 us of the 17 us a
// K = 64 tile takes, ~4 of 34 at K = 128); only the register epilogue (output transform + stores, ~1.5 us of issue) sits between two tiles.
// WT (round 5): the wave tile.  0: wave (wm, wn) of 2 x 4 owns 64 rows x 32 channels of ALL four positions -- every A and B fragment feeds
// one or two MFMAs: 24 ds_read_b128 per 24 MFMAs, and with the weight DMA and the transform's stores the LDS is ~86 % busy when the matrix
// pipe is 100 % busy (ablation: the same kernel with a third of its fragment reads skipped runs 11 % faster on 512 -> 512 @32^2,
// profiles/r05_wino_wave_tile.json).  1: wave (pp, wm, wn) of 2 x 2 x 2 owns 64 rows x 64 channels of TWO positions {2 pp, 2 pp + 1}: 16 reads
// per 24 MFMAs, same accumulators (8 x 16 registers), same fragment registers (the B fragments of the next position roll into the registers
// the first half of the MFMAs has released).  The output transform needs M0..M3 of a pair in one place: once per tile the two waves of a
// (wm, wn) exchange one position each through the LDS buffer the finished tile has released (pp = 0 sends M1 and produces the EVEN columns
// out[2j] = (M0 + M1) + M2, pp = 1 sends M2 and produces the ODD ones out[2j+1] = (M1 - M2) - M3; two barriers).
template <int XF, int VAR = 0, int WT = 0>
__global__ __launch_bounds__(NTHR) void conv_wino_kernel(const e4s_conv_params p, const int ntn, const int tx_n, const int per_img,
                                                         const int ntiles, const int ksplit, const int cper) {
    // ksplit > 1 (launches of <= 128 tiles: batch-1 latency runs, the 16x16 maps): the 16-channel chunks of a tile are divided over ksplit
    // consecutive tile ids; each writes its raw partial OUTPUT (the output transform is linear) to p.splitk_ws[ks] and e4s_splitk_epilogue
    // adds the slabs in order and applies bias / activation (no fused statistics then: the caller runs the separate pass)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                  // [2][4 pos][144][64]
    unsigned char* sB = smem + 2 * A_BYTES;                    // [2][4 pos][128][64]
    double* s_st = reinterpret_cast<double*>(sB + 2 * B_BYTES);      // [2 wm | WT: 4 = (pp, wm)][BN][2]
    float* s_in = reinterpret_cast<float*>(s_st + 4 * BN * 2);       // XF == 2: [2 tile parities][Cin][{mean, rstd}] (read at transform time:
                                                                     // an item does not carry its statistics across the stage boundary)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = WT ? (wave >> 1) & 1 : wave >> 2, wn = WT ? wave & 1 : wave & 3;
    const int pp = WT ? wave >> 2 : 0;            // WT: this wave's positions are 2 pp and 2 pp + 1
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int mtiles = ntiles / (ntn * ksplit);
    const int nchunk = p.Cin / KC;
    const unsigned char* ub = reinterpret_cast<const unsigned char*>(p.w);
    // n-major tile order: consecutive ids (one XCD) share a column tile, i.e. one 3 * Cin * 512-byte slab of U in their L2
    auto decode = [&](int t0) -> WTile {
        WTile w;
        const int t = t0 / ksplit;
        w.ks = t0 - t * ksplit;
        w.c_lo = w.ks * cper;
        w.c_hi = min(w.c_lo + cper, nchunk);
        const int nt = t / mtiles, mt = t - nt * mtiles;
        w.n0 = nt * BN;
        w.tb = mt / per_img;
        w.slot = mt - w.tb * per_img;
        const int tyb = w.slot / tx_n;
        w.ty0 = tyb * TH;
        w.tx0 = (w.slot - tyb * tx_n) * TW;
        return w;
    };

    // ---- input-transform items ----
    struct Item {
        f32x4 d[4];
        f32x4 s0, s1;        // XF == 2: {mean, rstd} of channels c..c+1 / c+2..c+3 interleaved (filled by item_prep from LDS)
        int dst;             // byte offset of the hi half inside a position plane
        unsigned okmask;     // bit i: d[i] is inside the image
    };
    auto item_load = [&](const WTile& T, int it, int chunk, Item& I) {
        const int v = it >> 2, cq = it & 3;
        const int hy = v >> 3, j = v & 7;
        const int iy = T.ty0 + hy - 1;
        const int c = chunk * KC + cq * 4;
        const bool rowok = (unsigned)iy < (unsigned)p.Hi;
        const float* xb = p.x + (size_t)T.tb * p.Hi * p.Wi * p.Cin;
        I.okmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ix = T.tx0 + 2 * j - 1 + i;
            const bool ok = rowok && (unsigned)ix < (unsigned)p.Wi;
            I.okmask |= ok ? (1u << i) : 0u;
            const size_t off = ok ? ((size_t)iy * p.Wi + ix) * p.Cin + c : (size_t)c;
            I.d[i] = *reinterpret_cast<const f32x4*>(xb + off);
        }
        I.dst = swz(v, cq >> 1) + (cq & 1) * 8;
    };
    // padded pixels -> 0 (XF == 2: 0 in the NORMALISED map, i.e. the mean); done once per item, before the per-position parts
    auto item_prep = [&](Item& I, int it, int chunk, int par) {      // (it & 3 == lane & 3 for every item a lane ever holds)
        f32x4 fill = {0.f, 0.f, 0.f, 0.f};
        if (XF == 2) {
            const int coff = par * p.Cin * 2 + (chunk * KC + (it & 3) * 4) * 2;
            I.s0 = *reinterpret_cast<const f32x4*>(s_in + coff);
            I.s1 = *reinterpret_cast<const f32x4*>(s_in + coff + 4);
            fill = f32x4{I.s0[0], I.s0[2], I.s1[0], I.s1[2]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(I.okmask & (1u << i))) I.d[i] = fill;
    };
    // position ps of an item: V_ps of 4 channels, split to hi / lo (packed converts, shift / mask re-expansion), two 8-byte LDS stores
    auto item_part = [&](unsigned char* Abuf, const Item& I, int ps) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x4 v;
        if (ps == 0) v = I.d[0] - I.d[2];
        else if (ps == 1) {
            if (XF == 2) {
                const f32x4 mu = f32x4{I.s0[0], I.s0[2], I.s1[0], I.s1[2]};
                v = (I.d[1] - mu) + (I.d[2] - mu);
            } else {
                v = I.d[1] + I.d[2];
            }
        } else if (ps == 2) v = I.d[2] - I.d[1];
        else v = I.d[1] - I.d[3];
        if (XF == 2) v = v * f32x4{I.s0[1], I.s0[3], I.s1[1], I.s1[3]};
        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
        const f32x2 r01 = f32x2{v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
        const f32x2 r23 = f32x2{v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        unsigned char* a = Abuf + ps * A_PLANE;
        *reinterpret_cast<u32x2*>(a + I.dst) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(a + (I.dst ^ 32)) = u32x2{l01, l23};
    };
    // the {mean, rstd} table of a tile's sample -> s_in[par] (made visible by the stage barriers that follow)
    auto load_stats = [&](const WTile& T, int par) {
        if (XF == 2) {
            const float* st = p.in_stats + (size_t)T.tb * p.Cin * 2;
            for (int i = tid * 4; i < p.Cin * 2; i += NTHR * 4)
                *reinterpret_cast<f32x4*>(s_in + par * p.Cin * 2 + i) = *reinterpret_cast<const f32x4*>(st + i);
        }
    };

    // ---- weight staging by LDS-DMA: instruction q = wave * 4 + jj of a stage fills the 1 KB (16 rows) block q & 7 of position plane q >> 3;
    // the LDS image is lane-linear, so the XOR swizzle goes on the SOURCE granule: lane -> row (lane >> 2), granule (lane & 3) ^ f(row) ----
    const size_t u_pos = (size_t)p.Cout * ROWB;                          // bytes per position plane in global memory
    auto u_stage = [&](int ky, int chunk) -> size_t { return ((size_t)ky * nchunk + chunk) * 4 * u_pos; };
    // (q >> 3 = wave >> 1 and (row >> 2) & 3 = (lane >> 4) & 3 for all four jj: the four pieces are 1 KB apart on both sides)
    const int q0 = wave * 4, ps0 = q0 >> 3, blk0 = q0 & 7, row0 = blk0 * 16 + (lane >> 2);
    const size_t g_src0 = ps0 * u_pos + (size_t)row0 * ROWB + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const int g_dst0 = ps0 * B_PLANE + blk0 * 1024;
    auto glds_stage = [&](unsigned char* Bdst, size_t stage_off, int n0) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(ub + stage_off + (size_t)n0 * ROWB + g_src0 + jj * 1024),
                (__attribute__((address_space(3))) void*)(Bdst + g_dst0 + jj * 1024), 16, 0, 0);
    };

    if (first >= ntiles) return;
    WTile cur = decode(first);
    int t_next = first + G;
    bool has_next = t_next < ntiles;
    WTile nxt = decode(has_next ? t_next : first);

    // Roles of a wave in stage sg (global stage counter): LOADER: (wave - 3 sg) & 7 < 3 fetches 64 items of the V group this stage feeds;
    // STORER: the loader of stage sg - 1 transforms, splits and stores them, one position per MFMA group, under its own MFMAs.  Group
    // (ts + 1) % 3 of the NEXT chunk's 576 items is loaded in stage ts and stored in stage ts + 1, so chunk c + 1's V is complete at the
    // barrier that ends chunk c (group 0 of chunk c + 1 is loaded in the last stage of chunk c - 1, or in the prologue).  "Next chunk" runs
    // through the tile boundary: after the last chunk of a tile comes chunk 0 of the block's next tile.
    Item I;
    I.dst = 0;
    I.okmask = 0;
    // ---- prologue (once per block): V of chunk 0 (all threads), U of stage 0, the loads of group 0 of chunk 1 for the storers of stage 0 ----
    load_stats(cur, 0);
    if (XF == 2) __syncthreads();
    {
        Item I0, I1;
        item_load(cur, tid, cur.c_lo, I0);
        const bool two = tid + NTHR < NITEMS;
        if (two) item_load(cur, tid + NTHR, cur.c_lo, I1);
        glds_stage(sB, u_stage(0, cur.c_lo), cur.n0);
        item_prep(I0, tid, cur.c_lo, 0);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) item_part(sA, I0, ps);
        if (two) {
            item_prep(I1, tid + NTHR, cur.c_lo, 0);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) item_part(sA, I1, ps);
        }
        const int slot = (wave + 3) & 7;
        if (slot < 3) item_load(cur, slot * 64 + lane, cur.c_lo + 1, I);      // (every split holds >= 2 chunks)
    }
    __syncthreads();

    // ---- fragment addressing: byte offsets inside a position plane ----
    int aoff[2][3], boff, boff2[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) aoff[tm][ky] = swz(wm * 64 + tm * 32 + li + 8 * ky, kh);
    boff = swz(wn * 32 + li, kh);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) boff2[tn] = swz(wn * 64 + tn * 32 + li, kh);        // WT: channel tiles of this wave's 64 columns

    f32x16 acc[4][2];
    struct AF { bf16x8 h[2], l[2]; };
    struct BF { bf16x8 h, l; };
    unsigned sg = 0, cg = 0;         // running stage / chunk counters: the LDS buffer parities continue across tiles
    int par = 0;                     // parity of the block's tile counter: s_in[par] holds the current tile's statistics
    for (;;) {
        if (has_next) load_stats(nxt, par ^ 1);      // read from the last chunk's first stage on: >= 3 barriers away (>= 2 chunks per tile)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ps][tm][r] = 0.f;

        for (int chunk = cur.c_lo; chunk < cur.c_hi; ++chunk) {
            const bool in_tile = chunk + 1 < cur.c_hi;          // the next chunk belongs to this tile
            const bool have_nc = in_tile || has_next;
            const WTile& Tn = in_tile ? cur : nxt;              // owner of the next chunk
            const int c_n = in_tile ? chunk + 1 : nxt.c_lo;
            const int par_n = in_tile ? par : par ^ 1;
            const unsigned char* Ab = sA + (cg & 1) * A_BYTES;
            unsigned char* An = sA + ((cg + 1) & 1) * A_BYTES;
#pragma unroll
            for (int ts = 0; ts < 3; ++ts) {
                const unsigned char* Bb = sB + (sg & 1) * B_BYTES;
                unsigned char* Bn = sB + ((sg + 1) & 1) * B_BYTES;
                const bool more = ts < 2 || have_nc;
                const bool storer = have_nc && ((wave - 3 * ((int)sg - 1)) & 7) < 3 && VAR != 1 && VAR != 3 && VAR != 4 && VAR != 5 && VAR != 7;
                const int lslot = (wave - 3 * (int)sg) & 7;
                // the chunk whose group (ts + 1) % 3 is loaded now: the next one (ts < 2) or the one after it (ts == 2)
                const int lv = ts < 2 ? chunk + 1 : chunk + 2;              // >= cur.c_hi: in the next tile
                const bool l_in = lv < cur.c_hi;
                const int lc = l_in ? lv : nxt.c_lo + (lv - cur.c_hi);      // its chunk index in its own tile
                const bool l_ok = l_in || (has_next && lc < nxt.c_hi);
                const bool loader = lslot < 3 && l_ok && VAR != 1 && VAR != 3 && VAR != 4 && VAR != 5 && VAR != 8;
                auto ldA = [&](AF& F, int ps) {
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const unsigned char* a = Ab + ps * A_PLANE;
                        F.h[tm] = *reinterpret_cast<const bf16x8*>(a + aoff[tm][ts]);
                        F.l[tm] = *reinterpret_cast<const bf16x8*>(a + (aoff[tm][ts] ^ 32));
                    }
                };
                auto ldB = [&](BF& F, int ps) {
                    const unsigned char* b = Bb + ps * B_PLANE;
                    F.h = *reinterpret_cast<const bf16x8*>(b + boff);
                    F.l = *reinterpret_cast<const bf16x8*>(b + (boff ^ 32));
                };
                // the MFMA section, with the storer's transform parts folded in (STORE is a compile-time copy: a branch around the VALU
                // block would pin it outside the MFMA stream)
                auto body = [&](auto store_tag, auto load_tag) {
                    constexpr bool STORE = decltype(store_tag)::value;
                    constexpr bool LOAD = decltype(load_tag)::value;
                    AF A0, A1;
                    BF B0, B1;
                    ldB(B0, 0);
                    ldA(A0, 0);
                    if (VAR == 4 || VAR == 5) { A1 = A0; B1 = B0; }
                    // the storer consumes last stage's plain loads FIRST: with an LDS-DMA in flight hipcc waits vmcnt(0) at the next use of
                    // any plain load, which would also wait for the DMA issued a moment ago
                    if (STORE) item_prep(I, lane, c_n, par_n);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && (VAR < 2 || VAR == 6)) {
                        if (ts < 2) glds_stage(Bn, u_stage(ts + 1, chunk), cur.n0);
                        else glds_stage(Bn, u_stage(0, c_n), Tn.n0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        AF& Ac = (ps & 1) ? A1 : A0;
                        AF& Anx = (ps & 1) ? A0 : A1;
                        BF& Bc = (ps & 1) ? B1 : B0;
                        BF& Bnx = (ps & 1) ? B0 : B1;
                        if (ps + 1 < 4 && VAR != 4 && VAR != 5) {
                            // VAR 9 / 10 (ablation builds; results WRONG): the fragment traffic of a wave tile of 64 x 64 x TWO positions
                            // -- 16 instead of 24 ds_read_b128 per 24 MFMAs (9: the A fragments of odd positions are not re-read;
                            // 10: nor their B fragments, 12 reads) -- what the LDS share of this kernel's time is worth
                            if (!(VAR == 10 && ((ps + 1) & 1))) ldB(Bnx, ps + 1);
                            if (!((VAR == 9 || VAR == 10) && ((ps + 1) & 1))) ldA(Anx, ps + 1);
                            else Anx = Ac;
                            if (VAR == 10 && ((ps + 1) & 1)) Bnx = Bc;
                        }
                        if (STORE) item_part(An, I, ps);
                        // the loader's address arithmetic + four 16-byte loads ride between the first group's MFMAs instead of in front of them
                        if (LOAD && ps == 0) item_load(l_in ? cur : nxt, ((ts + 1) % 3) * 192 + lslot * 64 + lane, lc, I);
                        if (VAR != 6) {
                            acc[ps][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l[0], Bc.h, acc[ps][0], 0, 0, 0);
                            acc[ps][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l[1], Bc.h, acc[ps][1], 0, 0, 0);
                            acc[ps][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[0], Bc.l, acc[ps][0], 0, 0, 0);
                            acc[ps][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[1], Bc.l, acc[ps][1], 0, 0, 0);
                            acc[ps][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[0], Bc.h, acc[ps][0], 0, 0, 0);
                            acc[ps][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[1], Bc.h, acc[ps][1], 0, 0, 0);
                        } else {
                            asm volatile("" ::"v"(Ac.h[0]), "v"(Ac.l[0]), "v"(Ac.h[1]), "v"(Ac.l[1]), "v"(Bc.h), "v"(Bc.l));
                        }
                        // the NEXT step's fragment reads are issued before this step's MFMAs (round 5: left to itself hipcc sank them behind
                        // five of the six MFMAs and then waited lgkmcnt(0) with ONE MFMA of cover -- ISA of the plain body)
                        if (E4S_WINO_READS_FIRST && ps + 1 < 4 && VAR == 0) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                        if (E4S_WINO_READS_FIRST && !STORE && !(LOAD && ps == 0) && VAR == 0) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        if (STORE) {
                            // this position's transform + split + stores (~40 VALU, 2 DS writes) and the next group's 6 fragment reads go
                            // between the six MFMAs
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                                __builtin_amdgcn_sched_group_barrier(0x306, 9, 0);      // then up to 9 VALU / SALU / DS
                            }
                        }
                        if (LOAD && ps == 0) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                                __builtin_amdgcn_sched_group_barrier(0x126, 12, 0);     // then up to 12 VALU / SALU / VMEM-read / DS-read
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                // WT == 1: four steps (lp, tn) of six MFMAs, acc[lp * 2 + tn][tm] -- the pipeline of `body` (the fragments of step k + 1 are
                // read into the OTHER buffer before the MFMAs of step k are issued), with the A fragments of a position read once for both
                // channel tiles: 2 x 4 A reads + 4 x 2 B reads = 16 per stage
                auto body2 = [&](auto store_tag, auto load_tag) {
                    constexpr bool STORE = decltype(store_tag)::value;
                    constexpr bool LOAD = decltype(load_tag)::value;
                    auto ldBt = [&](BF& F, int ps, int tn) {
                        const unsigned char* b = Bb + ps * B_PLANE;
                        F.h = *reinterpret_cast<const bf16x8*>(b + boff2[tn]);
                        F.l = *reinterpret_cast<const bf16x8*>(b + (boff2[tn] ^ 32));
                    };
                    AF A0, A1;
                    BF B0, B1;
                    const int p0 = pp * 2;
                    ldBt(B0, p0, 0);
                    ldA(A0, p0);
                    if (STORE) item_prep(I, lane, c_n, par_n);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        if (ts < 2) glds_stage(Bn, u_stage(ts + 1, chunk), cur.n0);
                        else glds_stage(Bn, u_stage(0, c_n), Tn.n0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int lp = k >> 1, tn = k & 1;
                        AF& Ac = lp ? A1 : A0;
                        BF& Bc = (k & 1) ? B1 : B0;
                        BF& Bnx = (k & 1) ? B0 : B1;
                        if (k + 1 < 4) ldBt(Bnx, p0 + ((k + 1) >> 1), (k + 1) & 1);
                        if (k == 0) ldA(A1, p0 + 1);
                        if (STORE) item_part(An, I, k);
                        if (LOAD && k == 0) item_load(l_in ? cur : nxt, ((ts + 1) % 3) * 192 + lslot * 64 + lane, lc, I);
                        f32x16(&c)[2] = acc[lp * 2 + tn];
                        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l[0], Bc.h, c[0], 0, 0, 0);
                        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l[1], Bc.h, c[1], 0, 0, 0);
                        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[0], Bc.l, c[0], 0, 0, 0);
                        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[1], Bc.l, c[1], 0, 0, 0);
                        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[0], Bc.h, c[0], 0, 0, 0);
                        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[1], Bc.h, c[1], 0, 0, 0);
                        if (E4S_WINO_READS_FIRST && k == 0) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                        if (E4S_WINO_READS_FIRST && (k == 1 || k == 2)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (E4S_WINO_READS_FIRST && !STORE && !(LOAD && k == 0)) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        if (STORE) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                                __builtin_amdgcn_sched_group_barrier(0x306, 9, 0);      // then up to 9 VALU / SALU / DS
                            }
                        }
                        if (LOAD && k == 0) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                                __builtin_amdgcn_sched_group_barrier(0x126, 12, 0);     // then up to 12 VALU / SALU / VMEM-read / DS-read
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (WT) {
                    if (storer) body2(std::true_type{}, std::false_type{});
                    else if (loader) body2(std::false_type{}, std::true_type{});
                    else body2(std::false_type{}, std::false_type{});
                } else if (storer) {
                    body(std::true_type{}, std::false_type{});
                } else if (loader) {
                    body(std::false_type{}, std::true_type{});
                    if (VAR == 7) asm volatile("" ::"v"(I.d[0]), "v"(I.d[1]), "v"(I.d[2]), "v"(I.d[3]));
                } else {
                    body(std::false_type{}, std::false_type{});
                }
                if (VAR != 5) __syncthreads();
                ++sg;
            }
            ++cg;
        }

        // ---- epilogue of the tile (WT == 1): exchange one position with the partner wave, then bias, activation, statistics, NHWC stores of
        // this wave's column parity ----
        if (WT) {
            // the buffers of the finished tile's last chunk / last stage are free (the next tile's first chunk and first U stage sit in the
            // other parities): 4 slots of 16 KB, two in the free A buffer, two in the free B buffer; slot = (wm, wn), shared by the pair
            const int slot = wm * 2 + wn;
            f32x4* xs = reinterpret_cast<f32x4*>((slot < 2 ? sA + ((cg + 1) & 1) * A_BYTES : sB + ((sg + 1) & 1) * B_BYTES) + (slot & 1) * 16384);
            if (pp == 0) {                                            // M1 = local position 1
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            xs[((tn * 2 + tm) * 4 + g) * 64 + lane] = f32x4{acc[2 + tn][tm][4 * g], acc[2 + tn][tm][4 * g + 1],
                                                                            acc[2 + tn][tm][4 * g + 2], acc[2 + tn][tm][4 * g + 3]};
            }
            __syncthreads();
            if (pp == 1) {                                            // out[2j+1] = (M1 - M2) - M3, then send M2 = local position 0
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 m1 = xs[((tn * 2 + tm) * 4 + g) * 64 + lane];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[2 + tn][tm][4 * g + i] = (m1[i] - acc[tn][tm][4 * g + i]) - acc[2 + tn][tm][4 * g + i];
                            xs[((tn * 2 + tm) * 4 + g) * 64 + lane] = f32x4{acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2],
                                                                            acc[tn][tm][4 * g + 3]};
                        }
            }
            __syncthreads();
            if (pp == 0) {                                            // out[2j] = (M0 + M1) + M2
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 m2 = xs[((tn * 2 + tm) * 4 + g) * 64 + lane];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[tn][tm][4 * g + i] = (acc[tn][tm][4 * g + i] + acc[2 + tn][tm][4 * g + i]) + m2[i];
                        }
            }
            const bool raw = ksplit > 1;
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const bool do_act = p.act != 0 && !raw;
            const bool stats = p.stats_ws != nullptr && !raw;
            float* yb = (raw ? p.splitk_ws + (size_t)cur.ks * ((size_t)p.B * p.Ho * p.Wo * p.Cout) : p.y) + (size_t)cur.tb * p.Ho * p.Wo * p.Cout;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int co = cur.n0 + wn * 64 + tn * 32 + li;
                const float bsv = (p.bias && !raw) ? p.bias[co] : 0.f;
                const float slp = (p.act == 2) ? p.slope[co] : p.alpha;
                double st_s = 0.0, st_q = 0.0;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float y0[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float a = (pp == 0 ? acc[tn][tm][4 * g + i] : acc[2 + tn][tm][4 * g + i]) + bsv;
                            if (do_act) a = (a > 0.f ? a : a * slp) * gain;
                            y0[i] = a;
                            if (stats) {
                                st_s += (double)a;
                                st_q += (double)a * (double)a;
                            }
                        }
                        quad_transpose4(y0[0], y0[1], y0[2], y0[3], li);
                        const int m = wm * 64 + tm * 32 + (li & 3) + 8 * g + 4 * kh;
                        const int oy = cur.ty0 + (m >> 3), ox = cur.tx0 + 2 * (m & 7) + pp;
                        *reinterpret_cast<f32x4*>(yb + ((size_t)oy * p.Wo + ox) * p.Cout + (co - (li & 3))) = f32x4{y0[0], y0[1], y0[2], y0[3]};
                    }
                }
                if (stats) {
                    st_s += __shfl_xor(st_s, 32, 64);
                    st_q += __shfl_xor(st_q, 32, 64);
                    if (kh == 0) {
                        const int col = wn * 64 + tn * 32 + li;
                        s_st[((pp * 2 + wm) * BN + col) * 2] = st_s;
                        s_st[((pp * 2 + wm) * BN + col) * 2 + 1] = st_q;
                    }
                }
            }
            if (stats) {
                __syncthreads();
                if (tid < BN) {
                    // fixed order: (even columns, odd columns) of the upper rows, then of the lower rows
                    const double a = ((s_st[tid * 2] + s_st[(2 * BN + tid) * 2]) + s_st[(BN + tid) * 2]) + s_st[(3 * BN + tid) * 2];
                    const double q = ((s_st[tid * 2 + 1] + s_st[(2 * BN + tid) * 2 + 1]) + s_st[(BN + tid) * 2 + 1]) + s_st[(3 * BN + tid) * 2 + 1];
                    double* slot2 = p.stats_ws + (((size_t)cur.tb * p.Cout + cur.n0 + tid) * p.stats_slots + cur.slot) * 2;
                    slot2[0] = a;
                    slot2[1] = q;
                }
            }
            // (the exchange slots are rewritten by the next tile's staging only after its first stage barrier; s_st a whole tile later)
        } else
        // ---- epilogue of the tile: output transform in registers, bias, activation, statistics, NHWC stores ----
        {
            const int co = cur.n0 + wn * 32 + li;
            const bool raw = ksplit > 1;
            const float bsv = (p.bias && !raw) ? p.bias[co] : 0.f;
            const float slp = (p.act == 2) ? p.slope[co] : p.alpha;
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const bool do_act = p.act != 0 && !raw;
            const bool stats = p.stats_ws != nullptr && !raw;
            double st_s = 0.0, st_q = 0.0;
            float* yb = (raw ? p.splitk_ws + (size_t)cur.ks * ((size_t)p.B * p.Ho * p.Wo * p.Cout) : p.y) + (size_t)cur.tb * p.Ho * p.Wo * p.Cout;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float y0[4], y1[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const float m0 = acc[0][tm][r], m1 = acc[1][tm][r], m2 = acc[2][tm][r], m3 = acc[3][tm][r];
                        float a = (m0 + m1) + m2 + bsv;
                        float b = (m1 - m2) - m3 + bsv;
                        if (do_act) {
                            a = (a > 0.f ? a : a * slp) * gain;
                            b = (b > 0.f ? b : b * slp) * gain;
                        }
                        y0[i] = a;
                        y1[i] = b;
                        if (stats) {
                            st_s += (double)a + (double)b;
                            st_q += (double)a * (double)a + (double)b * (double)b;
                        }
                    }
                    quad_transpose4(y0[0], y0[1], y0[2], y0[3], li);
                    quad_transpose4(y1[0], y1[1], y1[2], y1[3], li);
                    const int m = wm * 64 + tm * 32 + (li & 3) + 8 * g + 4 * kh;
                    const int oy = cur.ty0 + (m >> 3), ox = cur.tx0 + 2 * (m & 7);
                    float* dst = yb + ((size_t)oy * p.Wo + ox) * p.Cout + (co - (li & 3));
                    *reinterpret_cast<f32x4*>(dst) = f32x4{y0[0], y0[1], y0[2], y0[3]};
                    *reinterpret_cast<f32x4*>(dst + p.Cout) = f32x4{y1[0], y1[1], y1[2], y1[3]};
                }
            }
            if (stats) {
                st_s += __shfl_xor(st_s, 32, 64);
                st_q += __shfl_xor(st_q, 32, 64);
                if (kh == 0) {
                    const int col = wn * 32 + li;
                    s_st[(wm * BN + col) * 2] = st_s;
                    s_st[(wm * BN + col) * 2 + 1] = st_q;
                }
                __syncthreads();
                if (tid < BN) {
                    const double a = s_st[tid * 2] + s_st[(BN + tid) * 2];
                    const double q = s_st[tid * 2 + 1] + s_st[(BN + tid) * 2 + 1];
                    double* slot = p.stats_ws + (((size_t)cur.tb * p.Cout + cur.n0 + tid) * p.stats_slots + cur.slot) * 2;
                    slot[0] = a;
                    slot[1] = q;
                }
                // (the next write of s_st is a whole tile of barriers away)
            }
        }
        if (!has_next) break;
        cur = nxt;
        par ^= 1;
        t_next += G;
        has_next = t_next < ntiles;
        if (has_next) nxt = decode(t_next);
    }
}

int wino_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (!cus[dev & 63]) {
        hipDeviceProp_t prop;
        cus[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        if (const char* g = getenv("E4S_WINO_GRID")) {          // testing aid: fewer blocks, so that small shapes walk several tiles per block
            const int v = atoi(g);
            if (v > 0) cus[dev & 63] = v;
        }
    }
    return cus[dev & 63];
}

// split-K policy (the plain kernel's, on 16-channel chunks): only when the tiles alone leave most CUs idle; every split keeps >= 2 chunks
void wino_split(const e4s_conv_params& p, int64_t tiles, int& ksplit, int& cper) {
    const int nchunk = p.Cin / KC;
    ksplit = 1;
    cper = nchunk;
    if (tiles > 128 || nchunk < 4) return;
    int want = (int)((256 + tiles - 1) / tiles);
    if (want > nchunk / 2) want = nchunk / 2;
    if (want < 2) return;
    const int c = (nchunk + want - 1) / want, k = (nchunk + c - 1) / c;
    if (k < 2 || nchunk - (k - 1) * c < 2) return;
    ksplit = k;
    cper = c;
}

bool wino_covers(const e4s_conv_params& p) {
    return p.istride == 1 && p.ostride == 1 && p.ntaps == 9 && p.ncls == 1 && !p.labels && !p.rows && !p.in_scale && !p.out_scale &&
           !p.noise && p.y_cstride == 0 && p.Hi == p.Ho && p.Wi == p.Wo && p.Ha == p.Ho && p.Wa == p.Wo && p.Hi % TH == 0 &&
           p.Wi % TW == 0 && p.Cin % KC == 0 && p.Cin >= 2 * KC && p.Cout % BN == 0 && p.B > 0 &&
           (p.stats_ws == nullptr || p.stats_slots == (p.Hi / TH) * (p.Wi / TW));
}

}  // namespace

extern "C" int64_t e4s_wino_weights_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin % KC) return -1;
    return (int64_t)3 * (Cin / KC) * 4 * Cout * ROWB * (Cout % 32 == 0 ? 2 : 1);       // plane-major image [+ fragment-major image]
}

extern "C" int e4s_wino_weights_f32(const float* w9, void* out, int Cout, int Cin, void* stream) {
    if (!w9 || !out || Cout <= 0 || Cin <= 0 || Cin % KC) return (int)hipErrorInvalidValue;
    const int64_t total = (int64_t)3 * Cout * Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w9,
                       reinterpret_cast<unsigned char*>(out), Cout, Cin);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv_wino_covers(const e4s_conv_params* p) { return p && wino_covers(*p) ? 1 : 0; }

/* floats of p->splitk_ws the launch needs (0: no split-K; then stats_ws is honoured) */
extern "C" int64_t e4s_conv_wino_ws_floats(const e4s_conv_params* pp) {
    if (!pp || !wino_covers(*pp)) return 0;
    const e4s_conv_params& p = *pp;
    int ksplit, cper;
    wino_split(p, (int64_t)p.B * (p.Hi / TH) * (p.Wi / TW) * (p.Cout / BN), ksplit, cper);
    return ksplit > 1 ? (int64_t)ksplit * p.B * p.Ho * p.Wo * p.Cout : 0;
}

extern "C" int e4s_conv_wino_bf16x3_f32(const e4s_conv_params* pp, void* stream) {
    if (!pp) return (int)hipErrorInvalidValue;
    const e4s_conv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !wino_covers(p)) return (int)hipErrorInvalidValue;
    if (p.act == 2 && !p.slope) return (int)hipErrorInvalidValue;
    const int tx_n = p.Wi / TW, per_img = (p.Hi / TH) * tx_n, ntn = p.Cout / BN;
    int ksplit, cper;
    wino_split(p, (int64_t)p.B * per_img * ntn, ksplit, cper);
    if (ksplit > 1 && !p.splitk_ws) return (int)hipErrorInvalidValue;
    const int64_t tiles = (int64_t)p.B * per_img * ntn * ksplit;
    if (tiles > 0x7fffffff) return (int)hipErrorInvalidValue;
    static std::atomic<uint64_t> m0{0}, m2{0};
    int e;
    const int ncu = wino_num_cus();
    const int64_t grid = tiles < ncu ? tiles : ncu;                               // persistent: one block per CU
#ifdef E4S_ABLATIONS
    {
        static std::atomic<uint64_t> mv[11];
        const char* ev = getenv("E4S_WINO_VAR");
        const int var = ev ? atoi(ev) : 0;
        const void* fn = nullptr;
#define WV(V) case V: fn = (const void*)conv_wino_kernel<0, V>; if ((e = e4s_ensure_dyn_smem(fn, SMEM_WINO, mv[V]))) return e; \
              hipLaunchKernelGGL((conv_wino_kernel<0, V>), dim3((unsigned)grid), dim3(NTHR), SMEM_WINO, as_stream(stream), p, ntn, tx_n, per_img, (int)tiles, ksplit, cper); \
              E4S_CHECK_LAUNCH(); return 0;
        switch (p.in_stats ? 0 : var) {
            WV(1) WV(2) WV(3) WV(4) WV(5) WV(6) WV(7) WV(8) WV(9) WV(10)
            default: break;
        }
#undef WV
    }
#endif
    // launches without a K split and with >= 8 chunks per tile (Cin >= 128): the one-wave-per-SIMD kernel (conv_wino1w.hip).  Measured
    // (profiles/r06_wino1w.json, 16 images, InstanceNorm+PReLU form / statistics form, against this file's kernel): 512 -> 512 @32^2 -12 / -14 %,
    // 256 -> 256 @64^2 -10 / -11 %, 256 -> 512 @64^2 -11 / -10 %, 128 -> 128 @128^2 -3 / -3 %, 128 -> 256 @128^2 -3 / -3 %; 64 -> 128 @256^2 +4 / +5 %
    // (four chunks per tile: its register epilogue is not covered by a SIMD partner).  E4S_WINO_1W = 0 / 1 forces one of them (read per launch,
    // so that a test can run both in one process)
    {
        const char* e1 = getenv("E4S_WINO_1W");
        const bool want = e1 ? atoi(e1) != 0 : (p.Cin / KC >= 8 && p.Cout % 32 == 0);
        if (ksplit == 1 && want) return e4s_launch_wino1w(p, ntn, tx_n, per_img, (int)tiles, (int)grid, as_stream(stream));
    }
    // wave tile (see the kernel): 64 x 64 x two positions where a tile has enough K stages to pay for the exchange in its epilogue
    // (E4S_WINO_WT = 0 / 1 forces one of them: read per launch, so that a test can run both in one process)
    const char* wt_env = getenv("E4S_WINO_WT");
    const bool wt = wt_env ? atoi(wt_env) != 0 : (p.Cin / KC) / (ksplit > 0 ? ksplit : 1) >= WT_MIN_CHUNKS;
    if (wt) {
        static std::atomic<uint64_t> w0{0}, w2{0};
        if (p.in_stats) {
            if (p.Cin > 1024) return (int)hipErrorInvalidValue;
            if ((e = e4s_ensure_dyn_smem((const void*)conv_wino_kernel<2, 0, 1>, SMEM_WINO + 16384, w2))) return e;
            hipLaunchKernelGGL((conv_wino_kernel<2, 0, 1>), dim3((unsigned)grid), dim3(NTHR), SMEM_WINO + p.Cin * 16, as_stream(stream), p, ntn,
                               tx_n, per_img, (int)tiles, ksplit, cper);
        } else {
            if ((e = e4s_ensure_dyn_smem((const void*)conv_wino_kernel<0, 0, 1>, SMEM_WINO, w0))) return e;
            hipLaunchKernelGGL((conv_wino_kernel<0, 0, 1>), dim3((unsigned)grid), dim3(NTHR), SMEM_WINO, as_stream(stream), p, ntn, tx_n, per_img,
                               (int)tiles, ksplit, cper);
        }
    } else if (p.in_stats) {
        if (p.Cin > 1024) return (int)hipErrorInvalidValue;                    // the two {mean, rstd} tables have 16 KB of LDS
        if ((e = e4s_ensure_dyn_smem((const void*)conv_wino_kernel<2>, SMEM_WINO + 16384, m2))) return e;
        hipLaunchKernelGGL(conv_wino_kernel<2>, dim3((unsigned)grid), dim3(NTHR), SMEM_WINO + p.Cin * 16, as_stream(stream), p, ntn, tx_n,
                           per_img, (int)tiles, ksplit, cper);
    } else {
        if ((e = e4s_ensure_dyn_smem((const void*)conv_wino_kernel<0>, SMEM_WINO, m0))) return e;
        hipLaunchKernelGGL(conv_wino_kernel<0>, dim3((unsigned)grid), dim3(NTHR), SMEM_WINO, as_stream(stream), p, ntn, tx_n, per_img,
                           (int)tiles, ksplit, cper);
    }
    E4S_CHECK_LAUNCH();
    if (ksplit > 1) return e4s_splitk_epilogue(p, ksplit, as_stream(stream));      // ordered slab sum + bias / activation (conv_bf16x3.hip)
    return 0;
}
