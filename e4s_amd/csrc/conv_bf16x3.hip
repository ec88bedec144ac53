// 3x3 stride-1 convolution on the gfx950 bf16 matrix cores at fp32-class accuracy ("bf16x3"):
// every fp32 operand is split into two bf16 halves, v = hi + lo (hi = rne_bf16(v), lo = rne_bf16(v - hi); the
// residual is < 2^-17 |v|), and a product is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  in three
// v_mfma_f32_32x32x16_bf16 with an fp32 accumulator: relative error per product <= ~2^-16, i.e. ~1e-5 on a 4608-term
// contraction -- inside the path's 1e-3 bound with an order of magnitude to spare -- at 3 bf16 MFMAs (3 x 32 cycles per
// 32x32x16) instead of 8 fp32 MFMAs (8 x 64 cycles) for the same 32x32x16 block: 5.3x the fp32-MFMA rate.
//
// Scope: the natural-order (halo-tiled) 3x3 stride-1 contraction with at most ONE style per sample -- the encoder's
// Conv2d+PReLU (helpers.py:128-137; ~70 % of the MACs of a face swap) and the generator's unmasked StyledConvs
// (model.py:655-657).  Masked layers need the style of the OUTPUT pixel's region on the A fragment and stay on
// e4s_conv_mfma_f32.
//
// Block = 512 threads = 8 waves, tile 256 pixels (16x16) x 128 output channels; each wave owns 64 x 64 = 2 x 2 MFMA
// blocks (64 accumulator registers), so two waves share a SIMD and one wave's address arithmetic, bf16 conversion, LDS
// traffic and barrier waits run under the other's MFMAs (a 4-wave / 128x64-per-wave variant left the matrix pipe idle
// ~45 % of the time: with one wave per SIMD every non-MFMA instruction is serial overhead).  K step = one tap x 32
// input channels.
// LDS (132 KB, one block per CU):
//   * A: the (16+2)x(16+2) halo of the CURRENT 32-channel chunk, already split, one 144-byte row per halo pixel
//        [32 hi bf16 | 32 lo bf16 | 16 pad]; the 9 taps are 9 shifted views of it.  Double buffered: while chunk c is
//        being contracted, 1/9 of chunk c+1's halo is fetched, split and stored per tap stage, so every stage has the
//        same small load/convert/store footprint and no extra barrier.
//   * B: weights pre-split on the host side of the ABI (e4s_split_bf16x2_f32) in the same 128-byte row format, so
//        the B stage is a straight 16-byte copy global -> VGPR -> LDS, double buffered.
// The 144-byte row stride makes the ds_read_b128 fragment reads (lane -> row, lane half -> +16 B) conflict free, the
// same argument as conv_mfma.hip.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32;                 // input channels per chunk
constexpr int ROWB = 144;              // LDS row bytes
constexpr int LO = 64;                 // byte offset of the lo half inside a row
constexpr int TW = 16, HALO_W = TW + 2;
constexpr int PBM = 256, PTH = 16, PHALO = (PTH + 2) * HALO_W;      // 16x16-pixel tiles, 324 halo pixels
constexpr int PNTHR = 512;

// Column-tile configurations of the plain (one style per sample) kernel.  BN = GEMM columns per tile; TPS = taps per
// pipeline stage (a stage = TPS taps x 32 input channels): the narrow tiles take three taps per stage so that a stage
// still carries >= 18 MFMAs per wave between two barriers.
//   <128, 4, 2, 1>  Cout % 128 == 0 (and the pixel-shuffled up-convs, N = 4 Cout): waves 64 x 64
//   < 64, 4, 2, 3>  Cout % 64 == 0: waves 64 x 32        < 32, 8, 1, 3>  Cout % 32 == 0: waves 32 x 32
//   <128, 4, 2, 3, 16>  the same tile on 16-CHANNEL chunks: one MFMA k-step per tap, three taps per stage = 36 MFMAs per wave
//                        between two barriers instead of 24, 80-byte LDS rows [16 hi | 16 lo | pad]; measured on the variant-rows
//                        kernel (conv_region.hip, same pipeline): 8.5 % faster than the 32-channel / one-tap stages at K = 4608
template <int BN_, int WM_, int WN_, int TPS_, int KCH_ = 32>
struct PCfg {
    static constexpr int BN = BN_, WM = WM_, WN = WN_, TPS = TPS_, KCH = KCH_;
    static constexpr int TM = PBM / (WM * 32), TN = BN / (WN * 32);
    static constexpr int ROWB = KCH == 32 ? 144 : 80;         // LDS row bytes: [KCH hi bf16 | KCH lo bf16 | 16 pad]
    static constexpr int LO = KCH * 2;                        // byte offset of the lo half inside a row
    static constexpr int QG = KCH / 8;                        // 8-channel groups per halo pixel
    static constexpr int KSTEPS = KCH / 16;                   // MFMA k-steps per tap
    static constexpr int ITEMS = PHALO * QG;                  // (halo pixel, 8-channel group) work items of one chunk
    static constexpr int A_BYTES = PHALO * ROWB;
    static constexpr int NSTG = 9 / TPS;                      // stages per chunk
    static constexpr int PIECE = ITEMS / NSTG;                // halo items of the NEXT chunk fetched per stage
    static constexpr int BPC = KCH / 4;                       // 16-byte pieces per weight row
    static constexpr int BITEMS = TPS * BN * BPC;             // 16-byte weight pieces per stage
    static constexpr int BJ = (BITEMS + PNTHR - 1) / PNTHR;
    static constexpr int B_BYTES = TPS * BN * ROWB;
    static_assert(KCH == 32 || KCH == 16, "chunk");
    static_assert(WM * WN * 64 == PNTHR && TM >= 1 && TN >= 1 && NSTG * TPS == 9, "wave layout");
    static_assert(PIECE * NSTG == ITEMS && PIECE <= PNTHR, "halo split");
};
using CfgL = PCfg<128, 4, 2, 1>;
using CfgL16 = PCfg<128, 4, 2, 3, 16>;
using CfgM = PCfg<64, 4, 2, 3>;
using CfgS = PCfg<32, 8, 1, 3>;

template <typename C, bool SHUF>
constexpr int plain_smem() {      // A x2, B x2, 3 x per-tile metadata {out offset int, noise float x (4 if SHUF)},
    return 2 * C::A_BYTES + 2 * C::B_BYTES + 3 * PBM * 4 * (1 + (SHUF ? 4 : 1))      // + fused-statistics scratch
           + (SHUF ? 0 : C::WM * C::BN * 2 * 8);
}

__device__ __forceinline__ void split_store(unsigned char* dst, const f32x8 v, const int lo = LO) {
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + lo) = l;
}

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

__global__ void splitk_epilogue_kernel(const e4s_conv_params p, const int ksplit, const int ycs, const int64_t hw,
                                       const int64_t n4);

// split-K policy shared by the plain and the region-select kernels: only when the tiles alone leave most CUs idle; every
// split keeps >= 2 input-channel chunks
inline void few_tiles_split(int64_t tiles, int nchunk, int& ksplit, int& cper) {
    ksplit = 1;
    cper = nchunk;
    if (tiles > 128 || nchunk < 4) return;        // <= 128 tiles: two K halves fill the 256 CUs (the masked 32^2 layers ran on half the chip)
    int want = (int)((256 + tiles - 1) / tiles);
    if (want > nchunk / 2) want = nchunk / 2;
    if (want < 2) return;
    cper = (nchunk + want - 1) / want;
    ksplit = (nchunk + cper - 1) / cper;
}

inline void region_split(const e4s_conv_params& p, int& ksplit, int& cper) {
    const int64_t tiles = (int64_t)p.B * ((p.Ha + 15) / 16) * ((p.Wa + 15) / 16) * p.ncls * (p.Cout / 128);
    few_tiles_split(tiles, p.Cin / 32, ksplit, cper);
}

struct TileId {          // one 16x16-pixel x BN-column output tile (x one K split: input-channel chunks [c_lo, c_hi))
    int tb, tyb, txb, n0, ks, c_lo, c_hi;
};

// Plain kernel, PERSISTENT: the grid is one block per CU (or fewer tiles); block b walks tiles b, b + grid, ... and the
// stage pipeline runs straight through the tile boundary -- while the last chunk of tile t is contracted, the halo of
// tile t+1's first chunk and the weights of its first stage are fetched, so neither the prologue loads nor the epilogue
// stores of a tile are exposed (they were ~30 % of a K = 576 layer at one block per CU).
// XF: what happens to a halo element on its way into LDS (before the hi/lo split):
//   0 nothing; 1 v * in_scale[b][c] (one style per sample: unmasked StyledConv, model.py:655-657);
//   2 (v - mean[b][c]) * rstd[b][c] (InstanceNorm2d of the encoder unit folded into its first conv, helpers.py:128-131;
//     the same two fp32 operations e4s_instnorm_apply_f32 performs, so fused == unfused bitwise)
// SHUF: polyphase up-conv (model.py:287-300 folded into 4 phase kernels) as ONE GEMM with N = 4 Cout columns
//   (column n = phase * Cout + co) over the input grid, pixel-shuffled by the epilogue: output pixel (2ay+py, 2ax+px).
// VAR: profiling variants (builds with -DE4S_ABLATIONS select them with env E4S_BF16X3_ABL; results are WRONG for VAR >= 3;
// product builds only instantiate VAR = 0): 1 s_setprio(1) around the MFMA groups, 2 no scheduling pins, 3 MFMAs + barriers
// only (no fragment reads, no staging), 4 MFMAs only, 5 everything but the MFMAs
template <typename C, int XF, bool SHUF, int VAR = 0>
__global__ __launch_bounds__(PNTHR) void conv_bf16x3_kernel(const e4s_conv_params p, const int ntn, const int tx_n,
                                                            const int per_img, const int ntiles, const int ksplit,
                                                            const int cper) {
    constexpr int BM = PBM, BN = C::BN, NTHR = PNTHR, WN = C::WN, TM = C::TM, TN = C::TN, TH = PTH, TPS = C::TPS;
    constexpr int NSTG = C::NSTG, PIECE = C::PIECE, BITEMS = C::BITEMS, BJ = C::BJ;
    constexpr int A_BYTES = C::A_BYTES, B_BYTES = C::B_BYTES, NZ = SHUF ? 4 : 1;
    constexpr int KC = C::KCH, ROWB = C::ROWB, LO = C::LO, QG = C::QG, KSTEPS = C::KSTEPS, BPC = C::BPC, PITEMS = C::ITEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][HALO][ROWB]
    unsigned char* sB = smem + 2 * A_BYTES;            // [2][TPS*BN][ROWB]
    int* s_out = reinterpret_cast<int*>(sB + 2 * B_BYTES);            // [3][BM]
    float* s_nz = reinterpret_cast<float*>(s_out + 3 * BM);           // [3][BM][NZ]
    double* s_st = reinterpret_cast<double*>(s_nz + 3 * BM * NZ);     // [WM][BN][2] (fused output statistics; !SHUF)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);

    const int nchunk = p.Cin / KC;
    // split-K (ksplit > 1; launches with too few tiles to fill the chip, e.g. batch-1 latency runs): the input-channel
    // chunks of a tile are divided over ksplit consecutive tile ids; each writes its raw partial accumulators to
    // p.splitk_ws[ks] and a second kernel adds the slabs in order and applies the epilogue
    auto decode = [&](int t0) -> TileId {
        TileId id;
        const int t = t0 / ksplit;
        id.ks = t0 - t * ksplit;
        id.c_lo = id.ks * cper;
        id.c_hi = min(id.c_lo + cper, nchunk);
        const int mt = t / ntn, nt = t - mt * ntn;
        id.n0 = nt * BN;
        id.tb = mt / per_img;
        const int rem = mt - id.tb * per_img;
        id.tyb = rem / tx_n;
        id.txb = rem - id.tyb * tx_n;
        return id;
    };
    // per-row metadata of a tile: output offset and noise term(s) of each of the BM pixels
    auto fill_meta = [&](int buf, const TileId& id) {
        if (tid < BM) {
            const int ay = id.tyb * TH + tid / TW, ax = id.txb * TW + tid % TW;
            const bool valid = ay < p.Ha && ax < p.Wa;
            const int oy = ay * p.ostride, ox = ax * p.ostride;
            s_out[buf * BM + tid] = valid ? (id.tb * p.Ho + oy) * p.Wo + ox : -1;
#pragma unroll
            for (int ph = 0; ph < NZ; ++ph) {
                float nz = 0.f;
                if (valid && p.noise)
                    nz = p.noise_w[0] * p.noise[(int64_t)id.tb * p.noise_bstride + (int64_t)(oy + (ph >> 1)) * p.Wo + ox + (ph & 1)];
                s_nz[(buf * BM + tid) * NZ + ph] = nz;
            }
        }
    };

    const int ngemm = SHUF ? 4 * p.Cout : p.Cout;
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w);
    const size_t wrow = (size_t)p.Cin * 4;             // bytes per (tap, cout) row of the split weights
    const size_t img_stride = (size_t)p.Hi * p.Wi * p.Cin;

    // halo item of tile `id` -> (global offset in floats inside the sample, or a dummy in-bounds one)
    auto item_src = [&](const TileId& id, int item, bool& ok) -> size_t {
        const int h = item / QG, q = item % QG;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = id.tyb * TH + hy - 1, ix = id.txb * TW + hx - 1;
        ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        return ok ? ((size_t)iy * p.Wi + ix) * p.Cin + q * 8 : (size_t)(q * 8);
    };
    auto item_dst = [&](int item) -> int { return (item / QG) * ROWB + (item % QG) * 16; };
    // the per-channel transform operands of 8 channels starting at channel c of sample tb
    struct XOp { f32x8 a, b; };
    auto load_xop = [&](int tb, int c) -> XOp {
        XOp o;
        if (XF == 1) {
            o.a = load8(p.in_scale + (size_t)tb * p.Cin + c);
        } else if (XF == 2) {
            const float* st = p.in_stats + ((size_t)tb * p.Cin + c) * 2;       // {mean, rstd} interleaved
            const f32x8 s0 = load8(st), s1 = load8(st + 8);
            o.a = f32x8{s0[0], s0[2], s0[4], s0[6], s1[0], s1[2], s1[4], s1[6]};
            o.b = f32x8{s0[1], s0[3], s0[5], s0[7], s1[1], s1[3], s1[5], s1[7]};
        }
        return o;
    };
    auto apply_xop = [&](f32x8 v, const XOp& o) -> f32x8 {
        if (XF == 1) return v * o.a;
        if (XF == 2) return (v - o.a) * o.b;
        return v;
    };
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // weight staging role: piece i = tid + NTHR*j -> LDS row i/8 (= tap_local*BN + n), 16-byte column i%8
    int b_dst[BJ];
    bool b_ok[BJ];
    // global byte offset of (tap 0 of the stage, tile column 0) for each staged row; SHUF: column n = phase*Cout + co
    // lives at weight set `phase` ([4][9][Cout][Cin])
    auto b_src = [&](int j, int n0, int tap0, int chunk) -> size_t {
        const int i = tid + NTHR * j;
        const int row = b_ok[j] ? i / BPC : 0, pc = i % BPC;
        const int tl = row / BN, n = n0 + (row - tl * BN);
        size_t r;
        if (SHUF) {
            const int ph = n / p.Cout, co = n - ph * p.Cout;
            r = ((size_t)(ph * 9 + tap0 + tl) * p.Cout + co);
        } else {
            r = (size_t)(tap0 + tl) * ngemm + n;
        }
        // the split weights keep 32-channel chunks [32 hi | 32 lo]; a 16-channel chunk is the (chunk & 1) half of both
        if (KC == 16) return r * wrow + (size_t)(chunk >> 1) * 128 + (chunk & 1) * 32 + (pc >> 1) * 64 + (pc & 1) * 16;
        return r * wrow + (size_t)chunk * 128 + pc * 16;
    };
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int i = tid + NTHR * j;
        b_ok[j] = i < BITEMS;
        b_dst[j] = b_ok[j] ? (i / BPC) * ROWB + (i % BPC) * 16 : 0;
    }

    if (first >= ntiles) return;
    TileId cur = decode(first);
    int t_next = first + G;
    bool has_next = t_next < ntiles;
    TileId nxt = decode(has_next ? t_next : first);

    // ---- prologue: whole halo of tile 0 / chunk 0, weights of its stage 0, its metadata ----
    {
        const float* xb = p.x + (size_t)cur.tb * img_stride + cur.c_lo * KC;
        for (int item = tid; item < PITEMS; item += NTHR) {
            bool ok;
            const size_t off = item_src(cur, item, ok);
            f32x8 v = load8(xb + off);
            if (XF) v = apply_xop(v, load_xop(cur.tb, cur.c_lo * KC + (item % QG) * 8));
            if (!ok) v = zero8;                                 // zero padding applies AFTER the transform (conv pad)
            split_store(sA + item_dst(item), v, LO);
        }
        f32x4 pb[BJ];
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wbytes + b_src(j, cur.n0, 0, cur.c_lo));
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            if (b_ok[j]) *reinterpret_cast<f32x4*>(sB + b_dst[j]) = pb[j];
        fill_meta(0, cur);
    }
    __syncthreads();

    // ---- fragment addressing ----
    int arow[TM], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = (wm * TM + tm) * 32 + li;
        arow[tm] = ((m / TW) * HALO_W + (m % TW)) * ROWB + kh * 16;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = ((wn * TN + tn) * 32 + li) * ROWB + kh * 16;

    f32x16 acc[TM][TN];
    struct Pref {
        f32x4 b[BJ];
        f32x8 a;
        XOp x;
        int dst;
        bool part, ok;     // part: this thread holds a halo item of a real next chunk; ok: the item is inside the image
    };
    struct BFrag { bf16x8 h[TN], l[TN]; };
    struct AFrag { bf16x8 h, l; };
    const bool piece_thr = tid < PIECE;
    unsigned sg = 0, cg = 0;         // running stage / chunk counters: LDS buffer parities continue across tiles
    int mbuf = 0;                    // metadata buffer of the current tile (3-deep ring)
    for (;;) {
        // metadata of the NEXT tile -> ring slot mbuf+1 (its previous reader, the epilogue two tiles back, is behind
        // at least one full tile of barriers; the epilogue one tile back reads slot mbuf-1)
        const int mnext = (mbuf + 1) % 3;
        if (has_next) fill_meta(mnext, nxt);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

        for (int chunk = cur.c_lo; chunk < cur.c_hi; ++chunk) {
            const bool last_chunk = (chunk + 1 == cur.c_hi);
            // owner of chunk cg+1 (whose halo is fetched during this chunk) and of the stage after this chunk's last
            const TileId& own = last_chunk ? nxt : cur;
            const bool have_nc = !last_chunk || has_next;
            const int c_n = last_chunk ? nxt.c_lo : chunk + 1;
            const float* xb_n = p.x + (size_t)own.tb * img_stride + c_n * KC;
#pragma clang loop unroll_count(NSTG <= 3 ? NSTG : 1)
            for (int ts = 0; ts < NSTG; ++ts) {
                const unsigned char* Ab = sA + (cg & 1) * A_BYTES;
                const unsigned char* Bb = sB + (sg & 1) * B_BYTES;
                const bool last_ts = (ts + 1 == NSTG);
                const bool more = !last_ts || have_nc;          // is there a stage after this one (weights to fetch)?
                Pref P;
                auto issue_loads = [&]() {
                    // weights of the next stage: same chunk / next tap group, or tap group 0 of chunk cg+1 (own tile)
                    const int n0_w = last_ts ? own.n0 : cur.n0;
                    const int tap_w = last_ts ? 0 : (ts + 1) * TPS;
                    const int ch_w = last_ts ? c_n : chunk;
#pragma unroll
                    for (int j = 0; j < BJ; ++j)
                        P.b[j] = *reinterpret_cast<const f32x4*>(wbytes + b_src(j, more ? n0_w : cur.n0, more ? tap_w : 0,
                                                                                more ? ch_w : 0));
                    // piece `ts` of the halo of chunk cg+1
                    const int item = ts * PIECE + (piece_thr ? tid : 0);
                    bool ok;
                    const size_t off = item_src(own, item, ok);
                    P.a = load8((have_nc ? xb_n : p.x) + off);
                    if (XF) P.x = load_xop(have_nc ? own.tb : 0, c_n * KC + (item % QG) * 8);
                    P.ok = ok;
                    P.part = have_nc && piece_thr;
                    P.dst = item_dst(item);
                };
                // fragments: B of one sub-step u = (tap_local, kk) double-buffered across sub-steps, A of one MFMA group
                // g = (u, tm) double-buffered across groups; the next group's reads are requested before the current
                // group's MFMAs are issued, and the order is pinned (left alone, the scheduler sinks each read to just
                // before its first use and the matrix pipe idles an LDS round trip per group)
                auto ldB = [&](BFrag& F, int u) {
                    const int tl = u / KSTEPS, kk = u % KSTEPS;
                    const unsigned char* Bt = Bb + tl * BN * ROWB + kk * 32;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        F.h[tn] = *reinterpret_cast<const bf16x8*>(Bt + brow[tn]);
                        F.l[tn] = *reinterpret_cast<const bf16x8*>(Bt + brow[tn] + LO);
                    }
                };
                auto ldA = [&](AFrag& F, int g) {
                    const int u = g / TM, tm = g - u * TM;
                    const int tap = ts * TPS + u / KSTEPS;
                    const unsigned char* At = Ab + ((tap / 3) * HALO_W + (tap % 3)) * ROWB + (u % KSTEPS) * 32 + arow[tm];
                    F.h = *reinterpret_cast<const bf16x8*>(At);
                    F.l = *reinterpret_cast<const bf16x8*>(At + LO);
                };
                auto mfmas = [&](const AFrag& A, const BFrag& B, int tm) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h[tn], acc[tm][tn], 0, 0, 0);
                };
                constexpr int NU = KSTEPS * TPS, NGRP = NU * TM;   // sub-steps / MFMA groups (3*TN MFMAs each) per stage
                BFrag B0, B1;
                AFrag A0, A1;
                ldB(B0, 0);
                ldA(A0, 0);
                if (VAR == 3 || VAR == 4) { B1 = B0; A1 = A0; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    const int u = g / TM, tm = g - u * TM;
                    AFrag& Ac = (g & 1) ? A1 : A0;
                    AFrag& An = (g & 1) ? A0 : A1;
                    BFrag& Bc = (u & 1) ? B1 : B0;
                    BFrag& Bn = (u & 1) ? B0 : B1;
                    if (VAR != 3 && VAR != 4) {
                        if (g + 1 < NGRP) ldA(An, g + 1);
                        if (tm == 0 && u + 1 < NU) ldB(Bn, u + 1);
                        if (g == 0) issue_loads();
                    }
                    if (VAR == 1) __builtin_amdgcn_s_setprio(1);
                    if (VAR != 5) mfmas(Ac, Bc, tm);
                    else asm volatile("" :: "v"(Ac.h), "v"(Ac.l), "v"(Bc.h[0]), "v"(Bc.l[0]));
                    if (VAR == 1) __builtin_amdgcn_s_setprio(0);
                    if (g == 0 && VAR != 2 && VAR != 3 && VAR != 4) {
                        // the global-load block (~100 address/SALU instructions) is spread between this group's MFMAs:
                        // after a barrier all waves are in the same phase and nothing else would cover it
#pragma unroll
                        for (int i = 0; i < 3 * TN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x126, 16, 0);     // then up to 16 VALU/SALU/VMEM-read/DS-read
                        }
                    }
                    if (VAR != 2) __builtin_amdgcn_sched_barrier(0);
                }
                // -- VGPR -> LDS: weights of the next stage, halo piece of the next chunk --
                if (VAR == 3 || VAR == 4) {
                    if (VAR == 3) __syncthreads();
                    ++sg;
                    continue;
                }
                if (more) {
                    unsigned char* db = sB + ((sg + 1) & 1) * B_BYTES;
#pragma unroll
                    for (int j = 0; j < BJ; ++j)
                        if (b_ok[j]) *reinterpret_cast<f32x4*>(db + b_dst[j]) = P.b[j];
                }
                if (P.part) {
                    f32x8 v = P.a;
                    if (XF) v = apply_xop(v, P.x);
                    if (!P.ok) v = zero8;
                    split_store(sA + ((cg + 1) & 1) * A_BYTES + P.dst, v, LO);
                }
                __syncthreads();
                ++sg;
            }
            ++cg;
        }

        // ---- epilogue of the tile: demod * acc + noise + bias, activation, NHWC store (pixel-shuffled if SHUF) ----
        {
            const int* so = s_out + mbuf * BM;
            const float* sn = s_nz + mbuf * BM * NZ;
            float osc[TN], bsv[TN], slp[TN];
            int coff[TN], nzi[TN];
            const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = cur.n0 + (wn * TN + tn) * 32 + li;
                int ph = 0, co = n;
                if (SHUF) { ph = n / p.Cout; co = n - ph * p.Cout; }
                osc[tn] = p.out_scale ? p.out_scale[(size_t)cur.tb * p.Cout + co] : 1.f;
                bsv[tn] = p.bias ? p.bias[co] : 0.f;
                slp[tn] = (p.act == 2) ? p.slope[co] : p.alpha;
                coff[tn] = ((ph >> 1) * p.Wo + (ph & 1)) * ycs + co;
                nzi[tn] = ph;
            }
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const bool do_act = p.act != 0;
            const bool raw = ksplit > 1;
            const bool stats = !SHUF && !raw && p.stats_ws != nullptr;
            double st_s[TN], st_q[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) st_s[tn] = st_q[tn] = 0.0;
            float* yo = raw ? p.splitk_ws + (size_t)cur.ks * ((size_t)p.B * p.Ho * p.Wo * ycs) : p.y;
            // registers 4g .. 4g+3 of an accumulator are rows i + 8g + 4kh (i = 0..3) of ONE column per lane: epilogue math per
            // element, then a 4x4 transpose across each lane quad, then one 16-byte store per (row, 4 columns) -- 4x fewer store
            // instructions (stores from the accumulators are store-issue bound: common.h)
            const bool wide = (ycs & 3) == 0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[TN][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const int row = (wm * TM + tm) * 32 + i + 8 * g + 4 * kh;
                        const bool live = so[row] >= 0;
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            float t = acc[tm][tn][r];
                            if (!raw) {
                                t = t * osc[tn] + sn[row * NZ + (SHUF ? nzi[tn] : 0)] + bsv[tn];
                                if (do_act) t = (t > 0.f ? t : t * slp[tn]) * gain;
                            }
                            v[tn][i] = t;
                            if (stats && live) { st_s[tn] += (double)t; st_q[tn] += (double)t * (double)t; }
                        }
                    }
                    if (wide) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) quad_transpose4(v[tn][0], v[tn][1], v[tn][2], v[tn][3], li);
                        const int off = so[(wm * TM + tm) * 32 + (li & 3) + 8 * g + 4 * kh];
                        if (off >= 0) {
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                *reinterpret_cast<f32x4*>(yo + (size_t)off * ycs + coff[tn] - (li & 3)) =
                                    f32x4{v[tn][0], v[tn][1], v[tn][2], v[tn][3]};
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int off = so[(wm * TM + tm) * 32 + i + 8 * g + 4 * kh];
                            if (off < 0) continue;
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) yo[(size_t)off * ycs + coff[tn]] = v[tn][i];
                        }
                    }
                }
            }
            if (!SHUF && stats) {
                // per-channel {sum, sum of squares} of the tile: lanes li / li+32 hold the two row halves of a column
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    st_s[tn] += __shfl_xor(st_s[tn], 32, 64);
                    st_q[tn] += __shfl_xor(st_q[tn], 32, 64);
                    if (kh == 0) {
                        const int col = (wn * TN + tn) * 32 + li;
                        s_st[(wm * BN + col) * 2] = st_s[tn];
                        s_st[(wm * BN + col) * 2 + 1] = st_q[tn];
                    }
                }
                __syncthreads();
                if (tid < BN) {
                    double a = 0.0, q = 0.0;
#pragma unroll
                    for (int j = 0; j < C::WM; ++j) { a += s_st[(j * BN + tid) * 2]; q += s_st[(j * BN + tid) * 2 + 1]; }
                    const int tile_in_img = cur.tyb * tx_n + cur.txb;
                    double* slot = p.stats_ws + (((size_t)cur.tb * p.Cout + cur.n0 + tid) * p.stats_slots + tile_in_img) * 2;
                    slot[0] = a;
                    slot[1] = q;
                }
                __syncthreads();
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

// ---------------------------------------------------------------------------------------------------------------
// Region-select variant: masked StyledConvs (model.py:386-400) and the polyphase up-conv (ncls = 4).
// The style that scales an A element belongs to the OUTPUT pixel's region, and neighbouring taps pair one halo pixel
// with output pixels of different regions, so the halo cannot be scaled and split once when it is staged.  Here the
// halo stays fp32 in LDS; each wave multiplies its A fragment by the style fragment s[region(row)][k] (held in
// registers for the 9 taps of a chunk) and splits the product into hi/lo bf16 on its way into the MFMAs: 32 VALU per
// 6 MFMAs, which the second wave of the SIMD overlaps.  Weights arrive pre-split as in the kernel above; the
// demodulation table d[region][co] is applied in the epilogue.
// a = x * s split into hi + lo bf16 (8 products): hi = rne(x s) with the packed convert, re-expanded with one shift / one mask per
// element, lo = rne(fma(x, s, -hi)) -- the residual of the EXACT product, so the pair is at least as close to x s as the split of
// the rounded product
__device__ __forceinline__ void scale_split(const f32x8 x, const f32x8 s, bf16x8& hi, bf16x8& lo) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hp;
    f32x8 res;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 xs = f32x2{x[2 * j], x[2 * j + 1]}, ss = f32x2{s[2 * j], s[2 * j + 1]};
        const f32x2 v = xs * ss;
        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        hp[j] = h2;
        const f32x2 hf = f32x2{__builtin_bit_cast(float, h2 << 16), __builtin_bit_cast(float, h2 & 0xffff0000u)};
        f32x2 r;                                     // x s - hi; written out because the compiler turns -hi into 2 integer XORs
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(xs), "v"(ss), "v"(hf));
        res[2 * j] = r[0];
        res[2 * j + 1] = r[1];
    }
    hi = __builtin_bit_cast(bf16x8, hp);
    lo = __builtin_convertvector(res, bf16x8);
}

constexpr int NTHR = 512;
constexpr int BM = 256, BN = 128;
constexpr int TH = 16, HALO = (TH + 2) * HALO_W;                               // 324 halo pixels
constexpr int WN = 2, TM = 2, TN = 2;                                          // 4 x 2 waves, 2 x 2 blocks (64x64) each
constexpr int BSTEP = NTHR / 8, BJ = BN / BSTEP;                               // B staging: rows br0 + BSTEP*j, j < BJ
constexpr int ITEMS = HALO * 4;
constexpr int A_BYTES = HALO * ROWB, B_BYTES = BN * ROWB;
constexpr int MAXR = 16;
constexpr int XITEMS = ITEMS + MAXR * 4;           // + the chunk's style slice s[r][32]: 16 regions x 4 groups of 8
constexpr int XPIECE = (XITEMS + 8) / 9;           // 152 items per tap stage
constexpr int S_BYTES = MAXR * ROWB;
constexpr int S_OFF = 2 * A_BYTES + 2 * B_BYTES;
constexpr int SMEM_REGION = S_OFF + 2 * S_BYTES + BM * 12;
static_assert(XPIECE * 9 >= XITEMS && XPIECE <= NTHR, "extended halo split");

// Wave layout WN_ (column waves): 1 = 8 x 1 waves of 32 rows x 128 columns (TM 1, TN 4: a scaled-and-split A fragment feeds 12
// MFMAs, 48 VALU of scale/split per stage and wave), 2 = 4 x 2 waves of 64 x 64 (TM 2, TN 2: 6 MFMAs per fragment, the same rows
// scaled by both column waves).  SPL: 1 = hand-written split (packed convert, shift/mask re-expansion: 24 VALU per 8 products),
// 0 = __builtin_convertvector round trip (the compiler converts every element twice: 32 VALU).
// SCAN: the launch BEHIND conv_region_rows_kernel (conv_region.hip): a small grid whose blocks walk the logical block ids b, b + grid, ...
// and contract only the pixel tiles that kernel flagged (too many variant rows).  On face-parsing masks no tile is flagged and every block
// leaves after reading a few flags: ~3 us, where a full grid of early-exiting 512-thread / 150 KB-LDS blocks cost 13-17 us per masked layer
// (0.23 ms of a batch-8 step, 13 launches).
template <int WN_, int SPL, bool SCAN = false>
__global__ __launch_bounds__(NTHR) void conv_bf16x3_region_kernel(const e4s_conv_params p, const int ntn,
                                                                  const int tx_n, const int per_img,
                                                                  const int tiles_per_cls, const int ksplit,
                                                                  const int cper, const int* __restrict__ only_flagged,
                                                                  const int nblocks) {
    constexpr int WN = WN_, WM = NTHR / 64 / WN, TM = BM / (WM * 32), TN = BN / (WN * 32), NG = 2 * TM;
    static_assert(TM * TN == 4 && (NG == 2 || NG == 4), "wave layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][HALO][ROWB]  fp32 x, 32 channels per row
    unsigned char* sB = smem + 2 * A_BYTES;            // [2][BN][ROWB]    split weights
    int* s_out = reinterpret_cast<int*>(smem + S_OFF + 2 * S_BYTES);
    float* s_nz = reinterpret_cast<float*>(s_out + BM);
    int* s_grp = reinterpret_cast<int*>(s_nz + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ksplit > 1 (few tiles: batch-1 latency runs): the input-channel chunks are split over ksplit blocks per tile; each
    // stores d[region] * (its partial sum) and a second kernel adds the slabs in order and applies noise / bias / activation
    int scan_b = blockIdx.x;
  for (;;) {
    if (SCAN) {
        while (scan_b < nblocks && !only_flagged[(scan_b / ksplit) / ntn]) scan_b += gridDim.x;
        if (scan_b >= nblocks) return;
    }
    const int logical0 = SCAN ? scan_b : xcd_remap(blockIdx.x, gridDim.x);
    const int ks = logical0 % ksplit;
    const int logical = logical0 / ksplit;
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const int cls = mt / tiles_per_cls;
    const int tt = mt - cls * tiles_per_cls;
    const int tb = tt / per_img;
    const int rem = tt - tb * per_img;
    const int tyb = rem / tx_n, txb = rem - tyb * tx_n;
    const int py = (p.ncls == 4) ? (cls >> 1) : 0, px = (p.ncls == 4) ? (cls & 1) : 0;
    const int R = p.labels ? p.groups_per_batch : 1;

    if (tid < BM) {
        const int ay = tyb * TH + tid / TW, ax = txb * TW + tid % TW;
        const bool valid = ay < p.Ha && ax < p.Wa;
        const int oy = ay * p.ostride + py, ox = ax * p.ostride + px;
        s_out[tid] = valid ? (tb * p.Ho + oy) * p.Wo + ox : -1;
        float nz = 0.f;
        int r = 0;
        if (valid) {
            if (p.noise) nz = p.noise_w[0] * p.noise[(int64_t)tb * p.noise_bstride + (int64_t)oy * p.Wo + ox];
            if (p.labels) {   // legacy-nearest lookup of the OUTPUT pixel (F.interpolate 'nearest', model.py:391)
                const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Ho)), p.Hm - 1);
                const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wo)), p.Wm - 1);
                r = p.labels[((size_t)tb * p.Hm + sy) * p.Wm + sx];
            }
        }
        s_nz[tid] = nz;
        s_grp[tid] = r;
    }

    const int c_lo = ks * cper, nchunk = min(p.Cin / KC - c_lo, cper), nstage = nchunk * 9;      // this block's chunks
    const float* xb = p.x + (size_t)tb * p.Hi * p.Wi * p.Cin;
    const float* stab = p.in_scale + (size_t)tb * R * p.Cin;
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w) + (size_t)cls * 9 * p.Cout * p.Cin * 4;
    const size_t wrow = (size_t)p.Cin * 4;

    // work item -> global source (floats from xb / stab; dummy in-bounds when !ok), LDS byte offset inside buffer 0
    struct Item {
        const float* src;
        int dst, bufstride;
        bool ok;
    };
    auto item_of = [&](int item) -> Item {
        Item it;
        if (item < ITEMS) {
            const int h = item >> 2, q = item & 3;
            const int hy = h / HALO_W, hx = h - hy * HALO_W;
            const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
            it.ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            it.src = xb + (it.ok ? ((size_t)iy * p.Wi + ix) * p.Cin : 0) + q * 8;
            it.dst = h * ROWB + q * 32;
            it.bufstride = A_BYTES;
        } else {
            const int idx = item - ITEMS, r = idx >> 2, q = idx & 3;
            it.ok = r < R && item < XITEMS;
            it.src = stab + (it.ok ? (size_t)r * p.Cin : 0) + q * 8;
            it.dst = S_OFF + (r & (MAXR - 1)) * ROWB + q * 32;
            it.bufstride = S_BYTES;
        }
        return it;
    };
    auto load8 = [&](const float* src) -> f32x8 {
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
        return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
    auto store8 = [&](unsigned char* dst, const f32x8 v) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 16) = f32x4{v[4], v[5], v[6], v[7]};
    };

    // ---- prologue: halo + style slice of the first chunk, weights of stage 0 ----
    for (int item = tid; item < XPIECE * 9; item += NTHR) {
        const Item it = item_of(item);
        f32x8 v = load8(it.src + c_lo * KC);
        if (!it.ok) v = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (item < XITEMS) store8(smem + it.dst, v);
    }
    const int bq = (tid & 7) * 16, br0 = tid >> 3;
    {
        const unsigned char* wp = wbytes + (size_t)n0 * wrow + (size_t)c_lo * 128 + bq;
        f32x4 pb[BJ];
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(br0 + BSTEP * j) * wrow);
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + (br0 + BSTEP * j) * ROWB + bq) = pb[j];
    }
    __syncthreads();

    int arow[TM], srow[TM], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = (wm * TM + tm) * 32 + li;
        arow[tm] = ((m / TW) * HALO_W + (m % TW)) * ROWB + kh * 32;
        srow[tm] = S_OFF + s_grp[m] * ROWB + kh * 32;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = ((wn * TN + tn) * 32 + li) * ROWB + kh * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    f32x8 sv[TM][2];                 // style fragments of the current chunk: s[region(row)][kk*16 + kh*8 .. +7]
    const bool piece_thr = tid < XPIECE;
    int tap = 0, chunk = 0, t1 = 1, c1 = c_lo;       // chunk: relative to c_lo (LDS buffer parity); c1: absolute

    for (int s = 0; s < nstage; ++s) {
        const unsigned char* Ab = sA + (chunk & 1) * A_BYTES + ((tap / 3) * HALO_W + (tap % 3)) * ROWB;
        const unsigned char* Bb = sB + (s & 1) * B_BYTES;
        auto ldraw = [&](int g) -> f32x8 {            // group g = (kk, tm): 8 fp32 of the lane's pixel
            const int kk = g / TM, tm = g % TM;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + arow[tm] + kk * 64);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + arow[tm] + kk * 64 + 16);
            return f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        };
        if (tap == 0) {
            const unsigned char* Sb = smem + (chunk & 1) * S_BYTES;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(Sb + srow[tm] + kk * 64);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(Sb + srow[tm] + kk * 64 + 16);
                    sv[tm][kk] = f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                }
        }
        bf16x8 bh[2][TN], bl[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bh[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32);
                bl[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32 + LO);
            }
        f32x8 raw = ldraw(0);
        __builtin_amdgcn_sched_barrier(0);

        // -- global -> VGPR: weights of stage s+1, one piece of chunk+1's halo / style slice (issued after the first
        // MFMA group and spread between the MFMAs of groups 0-1, see the kernel above) --
        f32x4 pb[BJ];
        f32x8 pa;
        const bool more = (s + 1 < nstage);
        const bool have_next = (chunk + 1 < nchunk);
        const int item = tap * XPIECE + (piece_thr ? tid : 0);
        const Item it = item_of(item);
        auto issue_loads = [&]() {
            const unsigned char* wp =
                wbytes + ((size_t)(more ? t1 : 0) * p.Cout + n0) * wrow + (size_t)(more ? c1 : 0) * 128 + bq;
#pragma unroll
            for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(br0 + BSTEP * j) * wrow);
            pa = load8(it.src + (have_next ? (c_lo + chunk + 1) * KC : 0));
        };

        // -- NG groups of 3 TN MFMAs; the next group's fp32 fragment is requested before the current one is converted --
        auto mfma_group = [&](int g) {
            const int kk = g / TM, tm = g % TM;
            bf16x8 ah, al;
            if (SPL) {
                scale_split(raw, sv[tm][kk], ah, al);
                if (g + 1 < NG) raw = ldraw(g + 1);
            } else {
                const f32x8 v = raw * sv[tm][kk];
                if (g + 1 < NG) raw = ldraw(g + 1);
                ah = __builtin_convertvector(v, bf16x8);
                const f32x8 res = v - __builtin_convertvector(ah, f32x8);
                al = __builtin_convertvector(res, bf16x8);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kk][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kk][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kk][tn], acc[tm][tn], 0, 0, 0);
        };
        mfma_group(0);
        issue_loads();
        if (NG == 4) mfma_group(1);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x126, 8, 0);      // then up to 8 VALU / SALU / VMEM-read / DS-read
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(NG / 2);
        if (NG == 4) mfma_group(3);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x126, 6, 1);
        }
        __builtin_amdgcn_sched_barrier(0);

        // -- VGPR -> LDS --
        if (more) {
            unsigned char* db = sB + ((s + 1) & 1) * B_BYTES + br0 * ROWB + bq;
#pragma unroll
            for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(db + BSTEP * j * ROWB) = pb[j];
        }
        if (have_next && piece_thr && item < XITEMS) {
            if (!it.ok) pa = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            store8(smem + ((chunk + 1) & 1) * it.bufstride + it.dst, pa);
        }
        __syncthreads();
        if (++tap == 9) { tap = 0; ++chunk; }
        if (++t1 == 9) { t1 = 0; ++c1; }
    }

    // ---- epilogue: d[region][co] * acc + noise + bias, activation, NHWC store ----
    float* sD = reinterpret_cast<float*>(sA);          // [R][BN]; the loop's last barrier has passed
    if (p.out_scale) {
        for (int t = tid; t < R * BN; t += NTHR) {
            const int r = t / BN, n = t - r * BN;
            sD[t] = p.out_scale[((size_t)tb * R + r) * p.Cout + n0 + n];
        }
        __syncthreads();
    }
    float bsv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bsv[tn] = p.bias ? p.bias[n0 + (wn * TN + tn) * 32 + li] : 0.f;
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0, scaled = p.out_scale != nullptr, raw = ksplit > 1;
    float* yo = raw ? p.splitk_ws + (size_t)ks * ((size_t)p.B * p.Ho * p.Wo * p.Cout) : p.y;
    // epilogue math per element, 4x4 transpose across each lane quad, one 16-byte store per (row, 4 columns): see the plain kernel
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[TN][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const int row = (wm * TM + tm) * 32 + i + 8 * g + 4 * kh;
                const float nz = s_nz[row];
                const float* drow = sD + s_grp[row] * BN;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int ncol = (wn * TN + tn) * 32 + li;
                    float t = acc[tm][tn][r] * (scaled ? drow[ncol] : 1.f);
                    if (!raw) {
                        t += nz + bsv[tn];
                        if (do_act) t = (t > 0.f ? t : t * p.alpha) * gain;
                    }
                    v[tn][i] = t;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) quad_transpose4(v[tn][0], v[tn][1], v[tn][2], v[tn][3], li);
            const int off = s_out[(wm * TM + tm) * 32 + (li & 3) + 8 * g + 4 * kh];
            if (off >= 0) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    *reinterpret_cast<f32x4*>(yo + (size_t)off * p.Cout + n0 + (wn * TN + tn) * 32 + (li & ~3)) =
                        f32x4{v[tn][0], v[tn][1], v[tn][2], v[tn][3]};
            }
        }
    }
    if (!SCAN) return;
    scan_b += gridDim.x;
    __syncthreads();               // the next flagged tile re-uses the LDS
  }
}

template <int WN_, int SPL>
int launch_region_v(const e4s_conv_params& p, const int* only_flagged, hipStream_t st) {
    auto kern = only_flagged ? conv_bf16x3_region_kernel<WN_, SPL, true> : conv_bf16x3_region_kernel<WN_, SPL, false>;
    static std::atomic<uint64_t> smem_set[2] = {{0}, {0}};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM_REGION, smem_set[only_flagged ? 1 : 0])) return e;
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + TH - 1) / TH) * tx_n;
    const int tiles_per_cls = p.B * per_img;
    int ksplit, cper;
    region_split(p, ksplit, cper);
    if (ksplit > 1 && !p.splitk_ws) return (int)hipErrorInvalidValue;
    const int64_t blocks = (int64_t)tiles_per_cls * p.ncls * ntn * ksplit;
    if (blocks <= 0) return 0;
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const unsigned grid = only_flagged ? (unsigned)(blocks < 256 ? blocks : 256) : (unsigned)blocks;      // scan mode: one block per CU at most
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), SMEM_REGION, st, p, ntn, tx_n, per_img, tiles_per_cls,
                       ksplit, cper, only_flagged, (int)blocks);
    E4S_CHECK_LAUNCH();
    if (ksplit > 1) {           // slabs already carry d[region]: the second stage adds them and applies noise / bias / act
        e4s_conv_params q = p;
        q.out_scale = nullptr;
        const int64_t n4 = (int64_t)p.B * p.Ho * p.Wo * (p.Cout / 4);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, q, ksplit, p.Cout,
                           (int64_t)p.Ho * p.Wo, n4);
        E4S_CHECK_LAUNCH();
    }
    return 0;
}

int launch_region(const e4s_conv_params& p, const int* only_flagged, hipStream_t st) {
#ifdef E4S_ABLATIONS
    static const int v = [] { const char* e = getenv("E4S_REGION_VAR"); return e ? atoi(e) : -1; }();
    switch (v) {
        case 0: return launch_region_v<2, 0>(p, only_flagged, st);       // round-2 kernel
        case 1: return launch_region_v<2, 1>(p, only_flagged, st);
        case 2: return launch_region_v<1, 0>(p, only_flagged, st);
        default: break;
    }
#endif
    return launch_region_v<1, 1>(p, only_flagged, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Gather variant: the encoder's stride-2 3x3 convs and 1x1 stride-2 shortcut convs (helpers.py:125-137).  With stride 2 an
// input pixel feeds at most four outputs, so there is no halo to reuse: every stage (one tap x 32 channels) gathers its
// own A tile -- 256 output pixels x 32 channels, one 128-byte segment per row -- splits it to hi/lo bf16 and stores it
// next to the weights, both double buffered; the MFMA side is the plain kernel's.  Tile 256 pixels x 128 channels,
// natural pixel order (tiles may straddle rows and samples), one block per CU.
constexpr int GA_BYTES = BM * ROWB;                         // one A stage: 256 rows
constexpr int SMEM_GATHER = 2 * GA_BYTES + 2 * B_BYTES + BM * 4;

__global__ __launch_bounds__(NTHR) void conv_bf16x3_gather_kernel(const e4s_conv_params p, const int ntn,
                                                                  const int npix) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][BM][ROWB]
    unsigned char* sB = smem + 2 * GA_BYTES;           // [2][BN][ROWB]
    int* s_out = reinterpret_cast<int*>(sB + 2 * B_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const int hw = p.Ha * p.Wa;

    if (tid < BM) {
        const int a = mt * BM + tid;
        s_out[tid] = a < npix ? a : -1;                // ostride == 1: the output pixel index IS the anchor index
    }
    // A staging role: rows ar0 + 128 j (j < 2), 8-channel group aq
    const int aq = tid & 3, ar0 = tid >> 2;
    int a_base[2], a_yx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int a = mt * BM + ar0 + 128 * j;
        const bool valid = a < npix;
        const int aa = valid ? a : 0;
        const int b = aa / hw, rem = aa - b * hw;
        const int ay = rem / p.Wa, ax = rem - ay * p.Wa;
        const int by = ay * p.istride, bx = ax * p.istride;
        a_base[j] = (b * p.Hi + by) * p.Wi + bx;
        a_yx[j] = valid ? ((by << 16) | bx) : 0x7fff7fff;
    }
    const int ntaps = p.ntaps, nchunk = p.Cin / KC, nstage = ntaps * nchunk;
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w);
    const size_t wrow = (size_t)p.Cin * 4;
    const int bq = (tid & 7) * 16, br0 = tid >> 3;
    // Cout = 64 (the encoder's first stride-2 unit): ONE half-used 128-column tile -- the waves of the upper 64 columns stage and synchronise
    // with the rest but issue no MFMAs and store nothing (this layer ran on the exact-fp32 gather path: 206 us at 16 images)
    const int ncols = min(BN, p.Cout - n0);
    const bool cols_live = wn * TN * 32 < ncols;
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    struct Pref {
        f32x4 b[BJ];
        f32x8 a[2];
        unsigned ok;
    };
    auto fetch = [&](Pref& P, int tap, int chunk) {
        const unsigned char* wp = wbytes + ((size_t)tap * p.Cout + n0) * wrow + (size_t)chunk * 128 + bq;
#pragma unroll
        for (int j = 0; j < BJ; ++j) P.b[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)min(br0 + BSTEP * j, ncols - 1) * wrow);       // (Cout = 64: rows past the layer repeat its last one; never stored)
        const int oy = ((ntaps == 9) ? tap / 3 - 1 : 0) + p.tap_shift, ox = ((ntaps == 9) ? tap % 3 - 1 : 0) + p.tap_shift;
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = (a_yx[j] >> 16) + oy, ix = (a_yx[j] & 0xffff) + ox;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const size_t off = ok ? (size_t)(a_base[j] + oy * p.Wi + ox) * p.Cin : 0;
            P.a[j] = load8(p.x + off + chunk * KC + aq * 8);
            okm |= (ok ? 1u : 0u) << j;
        }
        P.ok = okm;
    };
    auto store = [&](const Pref& P, int buf) {
        unsigned char* db = sB + buf * B_BYTES + br0 * ROWB + bq;
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(db + BSTEP * j * ROWB) = P.b[j];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            split_store(sA + buf * GA_BYTES + (ar0 + 128 * j) * ROWB + aq * 16, ((P.ok >> j) & 1u) ? P.a[j] : zero8);
    };

    int arow[TM], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) arow[tm] = ((wm * TM + tm) * 32 + li) * ROWB + kh * 16;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = ((wn * TN + tn) * 32 + li) * ROWB + kh * 16;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    Pref P;
    fetch(P, 0, 0);
    store(P, 0);
    __syncthreads();
    int t1 = 0, c1 = 0;                      // stage s + 1
    if (++t1 == ntaps) { t1 = 0; ++c1; }
    for (int s = 0; s < nstage; ++s) {
        const unsigned char* Ab = sA + (s & 1) * GA_BYTES;
        const unsigned char* Bb = sB + (s & 1) * B_BYTES;
        const bool more = s + 1 < nstage;
        bf16x8 bh[2][TN], bl[2][TN], ah[2], al[2];
        auto ldB = [&](int kk) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bh[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32);
                bl[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32 + LO);
            }
        };
        auto ldA = [&](int g, int slot) {
            const int kk = g / TM, tm = g % TM;
            ah[slot] = *reinterpret_cast<const bf16x8*>(Ab + arow[tm] + kk * 32);
            al[slot] = *reinterpret_cast<const bf16x8*>(Ab + arow[tm] + kk * 32 + LO);
        };
        auto mfma_group = [&](int g) {
            const int kk = g / TM, tm = g % TM, cur = g & 1;
            if (g + 1 < 2 * TM) ldA(g + 1, cur ^ 1);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur], bh[kk][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bl[kk][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bh[kk][tn], acc[tm][tn], 0, 0, 0);
        };
        if (!cols_live) {                             // (wave-uniform) staging and barriers only
            fetch(P, more ? t1 : 0, more ? c1 : 0);
            if (more) store(P, (s + 1) & 1);
            __syncthreads();
            if (++t1 == ntaps) { t1 = 0; ++c1; }
            continue;
        }
        ldB(0);
        ldA(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0);
        fetch(P, more ? t1 : 0, more ? c1 : 0);       // unconditional (dummy stage 0 at the end): waits stay counted
        ldB(1);
        mfma_group(1);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x126, 10, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(3);
        __builtin_amdgcn_sched_barrier(0);
        if (more) store(P, (s + 1) & 1);
        __syncthreads();
        if (++t1 == ntaps) { t1 = 0; ++c1; }
    }

    // ---- epilogue: bias, activation, NHWC store ----
    float bsv[TN], slp[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + min((wn * TN + tn) * 32 + li, ncols - 1);
        bsv[tn] = p.bias ? p.bias[col] : 0.f;
        slp[tn] = (p.act == 2) ? p.slope[col] : p.alpha;
    }
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0;
    const bool stats = p.stats_ws != nullptr;         // the launcher guarantees tiles that do not straddle samples
    double st_s[TN], st_q[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) st_s[tn] = st_q[tn] = 0.0;
    const int gycs = p.y_cstride ? p.y_cstride : p.Cout;
    const bool wide = (gycs & 3) == 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[TN][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const bool live = s_out[(wm * TM + tm) * 32 + i + 8 * g + 4 * kh] >= 0;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float t = acc[tm][tn][r] + bsv[tn];
                    if (do_act) t = (t > 0.f ? t : t * slp[tn]) * gain;
                    v[tn][i] = t;
                    if (stats && live) { st_s[tn] += (double)t; st_q[tn] += (double)t * (double)t; }
                }
            }
            if (wide) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) quad_transpose4(v[tn][0], v[tn][1], v[tn][2], v[tn][3], li);
                const int off = s_out[(wm * TM + tm) * 32 + (li & 3) + 8 * g + 4 * kh];
                if (off >= 0 && cols_live) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        *reinterpret_cast<f32x4*>(p.y + (size_t)off * gycs + n0 + (wn * TN + tn) * 32 + (li & ~3)) =
                            f32x4{v[tn][0], v[tn][1], v[tn][2], v[tn][3]};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int off = s_out[(wm * TM + tm) * 32 + i + 8 * g + 4 * kh];
                    if (off < 0 || !cols_live) continue;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) p.y[(size_t)off * gycs + n0 + (wn * TN + tn) * 32 + li] = v[tn][i];
                }
            }
        }
    }
    if (stats) {
        double* s_st = reinterpret_cast<double*>(sA);        // the stage loop's last barrier has passed: A is free
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            st_s[tn] += __shfl_xor(st_s[tn], 32, 64);
            st_q[tn] += __shfl_xor(st_q[tn], 32, 64);
            if (kh == 0) {
                const int col = (wn * TN + tn) * 32 + li;
                s_st[(wm * BN + col) * 2] = st_s[tn];
                s_st[(wm * BN + col) * 2 + 1] = st_q[tn];
            }
        }
        __syncthreads();
        if (tid < ncols) {
            double a = 0.0, q = 0.0;
#pragma unroll
            for (int j = 0; j < NTHR / 64 / WN; ++j) { a += s_st[(j * BN + tid) * 2]; q += s_st[(j * BN + tid) * 2 + 1]; }
            const int tiles_per_img = hw / BM;
            const int b = mt / tiles_per_img, tile_in_img = mt - b * tiles_per_img;
            double* slot = p.stats_ws + (((size_t)b * p.Cout + n0 + tid) * p.stats_slots + tile_in_img) * 2;
            slot[0] = a;
            slot[1] = q;
        }
    }
}

int launch_gather(const e4s_conv_params& p, hipStream_t st) {
    auto kern = conv_bf16x3_gather_kernel;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM_GATHER, smem_set)) return e;
    const int ntn = (p.Cout + BN - 1) / BN;               // (Cout = 64: one half-used tile)
    const int64_t npix = (int64_t)p.B * p.Ha * p.Wa;
    if (npix <= 0) return 0;
    if (npix * ntn >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const int64_t blocks = ((npix + BM - 1) / BM) * ntn;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), SMEM_GATHER, st, p, ntn, (int)npix);
    E4S_CHECK_LAUNCH();
    return 0;
}

// fp32 rows [rows][cin] -> split rows [rows][cin/32][hi x32 | lo x32] (bf16), same byte size
__global__ void split_bf16x2_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int64_t n8,
                                    int cin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 8-channel group
    if (i >= n8) return;
    const int g8 = cin / 8;
    const int64_t row = i / g8;
    const int c = (int)(i - row * g8) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + row * cin + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + row * cin + c + 4);
    const f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    unsigned short* d = out + row * (int64_t)cin * 2 + (c / 32) * 64 + (c % 32);
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + 32) = l;
}

// second stage of a split-K launch: y = act(sum_ks ws[ks] * out_scale[b] + noise_w * noise + bias), slabs added in order
__global__ void splitk_epilogue_kernel(const e4s_conv_params p, const int ksplit, const int ycs, const int64_t hw,
                                       const int64_t n4) {
    const int C4 = p.Cout / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const int64_t pix = i / C4;
    const int64_t b = pix / hw;
    const size_t slab = (size_t)p.B * hw * ycs;
    const float* src = p.splitk_ws + (size_t)pix * ycs + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(src + (size_t)k * slab);
    float nz = 0.f;
    if (p.noise) nz = p.noise_w[0] * p.noise[b * p.noise_bstride + (pix - b * hw)];
    const float gain = (p.act == 1) ? p.gain : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = v[e] * (p.out_scale ? p.out_scale[b * p.Cout + c + e] : 1.f) + nz + (p.bias ? p.bias[c + e] : 0.f);
        if (p.act) t = (t > 0.f ? t : t * (p.act == 2 ? p.slope[c + e] : p.alpha)) * gain;
        v[e] = t;
    }
    *reinterpret_cast<f32x4*>(p.y + (size_t)pix * ycs + c) = v;
}

void plain_split(const e4s_conv_params& p, int ntn, int& ksplit, int& cper) {
    const int64_t tiles = (int64_t)p.B * ((p.Ha + PTH - 1) / PTH) * ((p.Wa + TW - 1) / TW) * ntn;
    few_tiles_split(tiles, p.Cin / KC, ksplit, cper);
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

template <typename C, int XF, bool SHUF, int VAR = 0>
int launch(const e4s_conv_params& p, hipStream_t st) {
    auto kern = conv_bf16x3_kernel<C, XF, SHUF, VAR>;
    constexpr int SMEM = plain_smem<C, SHUF>();
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, smem_set)) return e;
    const int ngemm = SHUF ? 4 * p.Cout : p.Cout;
    const int ntn = ngemm / C::BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + PTH - 1) / PTH) * tx_n;
    int ksplit, cper;
    plain_split(p, ngemm / C::BN, ksplit, cper);             // policy in 32-channel chunks, whatever the kernel's chunk
    cper *= 32 / C::KCH;
    if (ksplit > 1 && !p.splitk_ws) return (int)hipErrorInvalidValue;
    const int64_t ntiles = (int64_t)p.B * per_img * ntn * ksplit;
    if (ntiles <= 0) return 0;
    if (ntiles >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());        // persistent: one block per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PNTHR), SMEM, st, p, ntn, tx_n, per_img, (int)ntiles, ksplit, cper);
    E4S_CHECK_LAUNCH();
    if (ksplit > 1) {
        const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
        const int64_t npix = (int64_t)p.B * p.Ho * p.Wo;
        const int64_t n4 = npix * (p.Cout / 4);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p, ksplit, ycs,
                           (int64_t)p.Ho * p.Wo, n4);
        E4S_CHECK_LAUNCH();
    }
    return 0;
}

template <typename C, bool SHUF>
int launch_xf(const e4s_conv_params& p, hipStream_t st) {
    if (p.in_stats) return SHUF ? (int)hipErrorInvalidValue : launch<C, 2, false>(p, st);
    return p.in_scale ? launch<C, 1, SHUF>(p, st) : launch<C, 0, SHUF>(p, st);
}

}  // namespace

// conv_region.hip: the region-select kernel as the fallback / split-K path of e4s_conv_region_bf16x3_f32
int e4s_launch_region_select(const e4s_conv_params& p, const int* only_flagged, hipStream_t st) {
    return launch_region(p, only_flagged, st);
}
void e4s_region_split_policy(const e4s_conv_params& p, int& ksplit, int& cper) { region_split(p, ksplit, cper); }

extern "C" int e4s_conv_bf16x3_f32(const e4s_conv_params* pp, void* stream) {
    const e4s_conv_params& p = *pp;
    const bool up = (p.ncls == 4);
    if (p.istride == 2 || p.ntaps == 1) {      // encoder stride-2 3x3 / 1x1 shortcut convs: per-tap gather kernel
        if (p.Cin % KC || (p.Cout % BN && p.Cout != 64) || (p.ntaps != 9 && p.ntaps != 1) || p.ncls != 1 || p.ostride != 1 || p.tiles ||
            p.labels || p.in_scale || p.out_scale || p.in_stats || p.noise || p.Ho != p.Ha || p.Wo != p.Wa ||
            p.Hi >= 32767 || p.Wi >= 32767 || (p.Ha - 1) * p.istride >= p.Hi || (p.Wa - 1) * p.istride >= p.Wi ||
            p.tap_shift < 0 || p.tap_shift > 1 ||
            (p.stats_ws && (p.act != 0 || (p.Ha * p.Wa) % BM || p.stats_slots != (p.Ha * p.Wa) / BM)))
            return (int)hipErrorInvalidValue;
        return launch_gather(p, as_stream(stream));
    }
    if (p.Cin % KC || p.Cout % 32 || p.ntaps != 9 || (p.ncls != 1 && !up) || p.istride != 1 ||
        p.ostride != (up ? 2 : 1) || p.tiles || p.noise_per_channel || p.Ha != p.Hi || p.Wa != p.Wi ||
        p.Ho != p.Hi * p.ostride || p.Wo != p.Wi * p.ostride || (p.in_stats && p.in_scale))
        return (int)hipErrorInvalidValue;
    if (p.stats_ws && (p.labels || up || p.act != 0 || p.noise || p.out_scale ||
                       p.stats_slots != ((p.Ha + PTH - 1) / PTH) * ((p.Wa + TW - 1) / TW)))
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    if (p.labels) {                       // per-pixel regions: the region-select kernel (128-wide column tiles only)
        if (p.y_cstride || !p.in_scale || p.in_stats || p.act == 2 || p.Cout % BN || p.groups_per_batch < 1 || p.groups_per_batch > MAXR)
            return (int)hipErrorInvalidValue;
        return launch_region(p, nullptr, st);
    }
#ifdef E4S_ABLATIONS      // profiling builds only (E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build): tools/bench_abl.py
    static const int abl = [] { const char* e = getenv("E4S_BF16X3_ABL"); return e ? atoi(e) : 0; }();
    if (abl && !up && p.Cout % 128 == 0 && !p.in_stats && !p.in_scale) {
        switch (abl) {
            case 1: return launch<CfgL, 0, false, 1>(p, st);
            case 2: return launch<CfgL, 0, false, 2>(p, st);
            case 3: return launch<CfgL, 0, false, 3>(p, st);
            case 4: return launch<CfgL, 0, false, 4>(p, st);
            case 5: return launch<CfgL, 0, false, 5>(p, st);
            default: break;
        }
    }
#endif
    // one style per sample (or none): the persistent plain kernel.  The polyphase up-conv (ncls = 4) is ONE GEMM with
    // N = 4 Cout columns over the input grid + a pixel-shuffling epilogue.
    if (up) {
        if ((4 * p.Cout) % 128) return (int)hipErrorInvalidValue;
        return launch_xf<CfgL, true>(p, st);
    }
#ifdef E4S_ABLATIONS
    static const int k16 = [] { const char* e = getenv("E4S_PLAIN_K16"); return e ? atoi(e) : 1; }();
    if (p.Cout % 128 == 0 && !k16) return launch_xf<CfgL, false>(p, st);
#endif
    if (p.Cout % 128 == 0) return launch_xf<CfgL16, false>(p, st);
    if (p.Cout % 64 == 0) return launch_xf<CfgM, false>(p, st);
    return launch_xf<CfgS, false>(p, st);
}

// second stage of every split-K launch (also e4s_conv_mfma_f32's): slabs p.splitk_ws[k] (raw sums x demodulation) are added in
// order, then noise / bias / activation
int e4s_splitk_epilogue(const e4s_conv_params& p, int ksplit, hipStream_t st) {
    if (!p.splitk_ws || ksplit < 2 || p.Cout % 4) return (int)hipErrorInvalidValue;
    e4s_conv_params q = p;
    q.out_scale = nullptr;
    const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
    const int64_t n4 = (int64_t)p.B * p.Ho * p.Wo * (p.Cout / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, q, ksplit, ycs,
                       (int64_t)p.Ho * p.Wo, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_conv_bf16x3_ws_floats(const e4s_conv_params* pp) {
    const e4s_conv_params& p = *pp;
    if (p.istride != 1 || p.ntaps != 9 || p.Cin % KC) return 0;                  // gather kernels: no split-K
    if (p.labels) {
        int ksplit, cper;
        region_split(p, ksplit, cper);
        return ksplit > 1 ? (int64_t)ksplit * p.B * p.Ho * p.Wo * p.Cout : 0;
    }
    const bool up = (p.ncls == 4);
    const int ngemm = up ? 4 * p.Cout : p.Cout;
    const int bn = (up || p.Cout % 128 == 0) ? 128 : (p.Cout % 64 == 0 ? 64 : 32);
    int ksplit, cper;
    plain_split(p, ngemm / bn, ksplit, cper);
    if (ksplit <= 1) return 0;
    return (int64_t)ksplit * p.B * p.Ho * p.Wo * (p.y_cstride ? p.y_cstride : p.Cout);
}

extern "C" int e4s_split_bf16x2_f32(const float* w, void* out, int64_t rows, int cin, void* stream) {
    if (cin % 32) return (int)hipErrorInvalidValue;
    const int64_t n8 = rows * (cin / 8);
    if (n8 <= 0) return 0;
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, as_stream(stream), w,
                       reinterpret_cast<unsigned short*>(out), n8, cin);
    E4S_CHECK_LAUNCH();
    return 0;
}
