// Winograd F(2,3)-along-the-rows 3x3 convolution (conv_wino.hip: algebra, V layout, epilogue forms), re-cut for ONE wave per SIMD
// (round 6; the encoder's stride-1 convs are 43 % of a face swap's GPU time).
//
// conv_wino_kernel: 8 waves of 64 rows x 32 channels x 4 positions, two per SIMD inside 256 registers: 24 ds_read_b128 per 24 MFMAs, a
// barrier every 24 MFMAs per wave, the weights by LDS-DMA, the input transform on three "storer" waves per stage.  Here a 256-thread block
// owns the whole 512-entry register file: wave w holds ALL 128 GEMM rows x channels 32 w .. 32 w + 31 x four positions = 16 accumulator
// tiles (256 AGPRs); a stage (one vertical tap x 16 channels) is 48 MFMAs per wave; the output transform still happens in registers and the
// block tile (128 GEMM rows x 128 channels x 4 positions) is unchanged.
//   * U (weights) does NOT go through the LDS: a wave needs only its own 32 channels, so it loads them straight into its B fragment
//     registers from a fragment-major image (1 KB contiguous per instruction), position p of the NEXT stage behind the MFMAs of position p
//     of this one.  History: LDS-DMA into a ring of planes with vmcnt(0) before the barrier: 23 % SLOWER than the 8-wave kernel (a DMA
//     needs ~1.1 us from issue to landed); register-staged ring (load -> ds_write_b128 -> ds_read_b128): -6 % -- the LDS write port, ~79
//     B/clk per CU, was 415 of the 1 070 LDS cycles of a stage; direct: see profiles/r06_wino1w.json.
//   * V (transformed input) in LDS, double buffered by chunk, ONE barrier per stage after position 1; the A fragments of position p + 1 (of
//     the next stage after position 3) are read under the MFMAs of position p; every wave reads the whole V tile (the same LDS read
//     traffic as A + B fragments of a 64 x 64 wave tile).
//   * Input transform: 576 items (V-pixel, 4 channels) per chunk = two per thread (A, B) + 64 left over, which are cut into their four
//     positions so that every thread takes one quarter (Q: wave w computes position w); an item is fetched in the second half of a stage
//     and transformed + stored in the second half of the next one (A and Q in the chunk's first stage, B in its second).  Addresses are a
//     per-thread constant + a wave-uniform tile / chunk base, padding is two flag words: no integer multiplies, no branches.
//   * The issue order is pinned slot by slot (one MFMA + at most one LDS / global instruction + a few VALU, sched_barrier(0)).
// Split-K launches (<= 128 tiles) and short-K layers stay on conv_wino_kernel.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NTHR = 256;
constexpr int TH = 16, TW = 16, NPAIR = TW / 2;
constexpr int VROWS = (TH + 2) * NPAIR;          // 144 V-pixels per position and chunk
constexpr int BN = 128, KC = 16, ROWB = 64;
constexpr int A_PLANE = VROWS * ROWB, A_BYTES = 4 * A_PLANE;        // 36 864
constexpr int OFF_IN = 2 * A_BYTES;                                 // XF == 2: float [2 tile parities][Cin][{mean, rstd}]
constexpr int SMEM_1W = OFF_IN;
#ifndef E4S_WINO_BPF
#define E4S_WINO_BPF 1
#endif
// B fragments are fetched this many stages ahead.  2 (three register sets) measured SLOWER in the step, same box, two alternations:
// encoder convs 6.95 vs 6.85 ms, step 16.40 vs 16.31 ms (profiles/r06_wino1w.json) -- one stage of flight covers the weight stream
constexpr int BPF = E4S_WINO_BPF;
static_assert(BPF == 1 || BPF == 2, "B prefetch distance");

__device__ __forceinline__ int swz(int row, int g) { return row * ROWB + ((g ^ ((row >> 2) & 3)) << 4); }

// LDS-only workgroup barrier: every LDS access of this wave issued so far has completed (lgkmcnt(0)); global loads stay in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct WTile {
    int n0, tb, ty0, tx0, slot;
};

struct AF { bf16x8 h[4], l[4]; };          // the four 32-row groups of a position
struct BF { bf16x8 h, l; };                // one position of this wave's 32 channels

// v = (a - m) + sg (b - m) [m = 0, sg = -1: a - b exactly; m = mean, sg = +1: position 1 of the InstanceNorm form], * rs, split to hi / lo bf16
template <int XF>
__device__ __forceinline__ void pair_split(const float a0, const float b0, const float a1, const float b1, const float m0, const float m1,
                                           const float sg, const float r0, const float r1, unsigned& hi, unsigned& lo) {
    float v0 = __builtin_fmaf(sg, b0 - m0, a0 - m0), v1 = __builtin_fmaf(sg, b1 - m1, a1 - m1);
    if (XF == 2) {
        v0 *= r0;
        v1 *= r1;
    }
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
    const float e0 = v0 - __builtin_bit_cast(float, hi << 16), e1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e0, e1}, bf16x2));
}

// XF: 0 plain input, 2 InstanceNorm (x - mean) * rstd folded into the input transform (zero padding applies to the NORMALISED map)
// VAR: profiling variants (builds with -DE4S_ABLATIONS select them with env E4S_WINO_1W_VAR; results are WRONG for VAR >= 1; product builds
// only instantiate VAR = 0): 1 no weight staging, 2 no input-transform staging, 3 neither, 4 MFMAs only (no fragment reads, no barrier)
template <int XF, int VAR = 0>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_wino1w_kernel(const e4s_conv_params p, const int ntn, const int tx_n, const int per_img, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                  // [2][4 pos][144][64]
    float* s_in = reinterpret_cast<float*>(smem + OFF_IN);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int wn = wave;                                       // this wave: all 128 GEMM rows x channels 32 wn .. 32 wn + 31, four positions
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int mtiles = ntiles / ntn;
    const int nchunk = p.Cin / KC;
    const unsigned char* ub = reinterpret_cast<const unsigned char*>(p.w);
    // n-major tile order: consecutive ids (one XCD) share a column tile, i.e. one slab of U in their L2
    auto decode = [&](int t) -> WTile {
        WTile w;
        const int nt = t / mtiles, mt = t - nt * mtiles;
        w.n0 = nt * BN;
        w.tb = mt / per_img;
        w.slot = mt - w.tb * per_img;
        const int tyb = w.slot / tx_n;
        w.ty0 = tyb * TH;
        w.tx0 = (w.slot - tyb * tx_n) * TW;
        return w;
    };

    // ---- input-transform items.  Item it = (V-pixel v = it >> 2 [row hy = v >> 3 of 18, pair j = v & 7], channels 4 (it & 3) .. + 3) reads the
    // pixels d_i = x[ty0 + hy - 1][tx0 + 2 j - 1 + i], i < 4.  Per thread and class (A: it = tid, B: tid + 256, Q: 512 + lane) the element
    // offset relative to the tile's (ty0, tx0, chunk) origin and the edge flags {hy == 0, hy == 17, j == 0, j == 7} are constants. ----
    const int cq = tid & 3;
    auto rel_of = [&](int v) -> int { return (((v >> 3) - 1) * p.Wi + 2 * (v & 7) - 1) * p.Cin + cq * 4; };
    auto flg_of = [&](int v) -> unsigned { return ((v >> 3) == 0 ? 1u : 0u) | ((v >> 3) == 17 ? 2u : 0u) | ((v & 7) == 0 ? 4u : 0u) | ((v & 7) == 7 ? 8u : 0u); };
    auto dst_of = [&](int v) -> int { return swz(v, cq >> 1) + (cq & 1) * 8; };
    const int vA = tid >> 2, vB = 64 + (tid >> 2), vQ = 128 + (lane >> 2);
    const int relA = rel_of(vA), relB = rel_of(vB), relQ = rel_of(vQ);
    const unsigned flgA = flg_of(vA), flgB = flg_of(vB), flgQ = flg_of(vQ);
    const int dstA = dst_of(vA), dstB = dst_of(vB), dstQ = dst_of(vQ) + wave * A_PLANE;      // Q: this wave's position plane
    // quarter items: position = wave: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
    const int qa = wave == 0 ? 0 : (wave == 2 ? 2 : 1), qb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float qsg = wave == 1 ? 1.f : -1.f;
    // tile edge bits {top, bottom, left, right} and the element offset of a tile's (chunk) origin
    auto edge_of = [&](const WTile& T) -> unsigned {
        return (T.ty0 == 0 ? 1u : 0u) | (T.ty0 + TH == p.Hi ? 2u : 0u) | (T.tx0 == 0 ? 4u : 0u) | (T.tx0 + TW == p.Wi ? 8u : 0u);
    };
    auto xbase_of = [&](const WTile& T, int chunk) -> const float* {
        return p.x + (size_t)T.tb * p.Hi * p.Wi * p.Cin + ((size_t)T.ty0 * p.Wi + T.tx0) * p.Cin + chunk * KC;
    };
    // pixel i of an item: valid unless its row is padding (flags 1 | 2) or it is the left-most / right-most pixel of a border pair
    auto load_px = [&](const float* xb, int rel, unsigned bad, int i, bool& ok) -> f32x4 {
        ok = !((bad & 3u) || (i == 0 && (bad & 4u)) || (i == 3 && (bad & 8u)));
        return *reinterpret_cast<const f32x4*>(xb + (ok ? rel + i * p.Cin : cq * 4));
    };
    struct Item {
        f32x4 d[4];
        unsigned okmask;
    };
    auto item_load_part = [&](Item& I, const float* xb, int rel, unsigned bad, int i) {
        bool ok;
        I.d[i] = load_px(xb, rel, bad, i, ok);
        if (i == 0) I.okmask = 0;
        I.okmask |= ok ? (1u << i) : 0u;
    };
    // {mean, rstd} of the thread's four channels (XF == 2); padded pixels -> 0 of the NORMALISED map, i.e. the mean
    auto stats_of = [&](f32x4& mu, f32x4& rs, int chunk, int par) {
        if (XF == 2) {
            const int coff = par * p.Cin * 2 + (chunk * KC + cq * 4) * 2;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(s_in + coff);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(s_in + coff + 4);
            mu = f32x4{s0[0], s0[2], s1[0], s1[2]};
            rs = f32x4{s0[1], s0[3], s1[1], s1[3]};
        }
    };
    auto item_fill = [&](Item& I, const f32x4& mu) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(I.okmask & (1u << i))) I.d[i] = XF == 2 ? mu : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // position ps (compile time) of an item, channels 2 half .. 2 half + 1: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3 (the mean
    // cancels in the differences; V1 of the InstanceNorm form = (d1 - mu) + (d2 - mu)), * rstd, split to hi / lo
    auto item_pair = [&](const Item& I, const f32x4& mu, const f32x4& rs, int ps, int half, unsigned& hi, unsigned& lo) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * half + e;
            float t;
            if (ps == 0) t = I.d[0][c] - I.d[2][c];
            else if (ps == 1) t = XF == 2 ? (I.d[1][c] - mu[c]) + (I.d[2][c] - mu[c]) : I.d[1][c] + I.d[2][c];
            else if (ps == 2) t = I.d[2][c] - I.d[1][c];
            else t = I.d[1][c] - I.d[3][c];
            if (XF == 2) t *= rs[c];
            v[e] = t;
        }
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
        const float e0 = v[0] - __builtin_bit_cast(float, hi << 16), e1 = v[1] - __builtin_bit_cast(float, hi & 0xffff0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e0, e1}, bf16x2));
    };
    auto load_stats = [&](const WTile& T, int par) {
        if (XF == 2) {
            const float* st = p.in_stats + (size_t)T.tb * p.Cin * 2;
            for (int i = tid * 4; i < p.Cin * 2; i += NTHR * 4)
                *reinterpret_cast<f32x4*>(s_in + par * p.Cin * 2 + i) = *reinterpret_cast<const f32x4*>(st + i);
        }
    };

    // ---- weights: straight from global memory into the B fragments of the wave that uses them (no LDS round trip: the LDS write port --
    // ~79 B/clk per CU -- was 415 of the 1 070 LDS cycles of a stage).  The fragment-major image (wino_weights_kernel writes it behind the
    // plane-major one): per (ky, chunk, position, 32-channel block) 2 KB = [hi: lane l -> channel l & 31, k-half l >> 5][lo: the same], so a
    // wave's fragment load is 1 KB contiguous ----
    const unsigned char* uf = ub + (size_t)3 * nchunk * 4 * p.Cout * ROWB;                          // behind the plane-major image
    auto uf_src = [&](int ky, int chunk, int ps, int n0) -> const unsigned char* {
        return uf + ((((size_t)ky * nchunk + chunk) * 4 + ps) * (p.Cout / 32) + (n0 / 32 + wn)) * 2048 + lane * 16;
    };
    if (first >= ntiles) return;
    WTile cur = decode(first);
    int t_next = first + G;
    bool has_next = t_next < ntiles;
    WTile nxt = decode(has_next ? t_next : first);

    // ---- prologue (once per block): V of chunk 0, the U planes (0, 0..2) straight to LDS, round 0 = planes (0, 3), (1, 0..2) into registers
    // (stage 0 stores it), items A and Q of chunk 1 ----
    Item I;                            // item A / B in flight
    f32x4 Qa, Qb;                      // quarter item in flight: the two pixels of this wave's position
    bool Qoka = true, Qokb = true;
    I.okmask = 0;
    load_stats(cur, 0);
    if (XF == 2) __syncthreads();
    {
        const float* xb0 = xbase_of(cur, 0);
        const unsigned edge = edge_of(cur);
        for (int it = tid; it < VROWS * 4; it += NTHR) {
            const int v = it >> 2;
            Item I0;
            f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) item_load_part(I0, xb0, rel_of(v), flg_of(v) & edge, i);
            stats_of(mu, rs, 0, 0);
            item_fill(I0, mu);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                unsigned h[2], l[2];
                item_pair(I0, mu, rs, ps, 0, h[0], l[0]);
                item_pair(I0, mu, rs, ps, 1, h[1], l[1]);
                const int a = ps * A_PLANE + dst_of(v);
                *reinterpret_cast<u32x2*>(sA + a) = u32x2{h[0], h[1]};
                *reinterpret_cast<u32x2*>(sA + (a ^ 32)) = u32x2{l[0], l[1]};
            }
        }
        const float* xb1 = xbase_of(cur, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) item_load_part(I, xb1, relA, flgA & edge, i);
        Qa = load_px(xb1, relQ, flgQ & edge, qa, Qoka);
        Qb = load_px(xb1, relQ, flgQ & edge, qb, Qokb);
    }
    __syncthreads();

    int aoff[4][3];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) aoff[tm][ky] = swz(tm * 32 + li + 8 * ky, kh);

    f32x16 acc[4][4];                 // [position][32-row group]
    AF Af[2];
    // B fragments.  BPF = 1: one set, position p re-filled for the NEXT stage as soon as its MFMAs have issued (a stage of flight); BPF = 2: three
    // sets rotating with the vertical tap, position p of the stage AFTER the next fetched under position p of this one
    BF Bf[BPF == 1 ? 1 : 3][4];
    auto ldA = [&](AF& F, const unsigned char* Abuf, int ps, int ky, int part) {      // part: 2 tm + {0: h, 1: l}
        const unsigned char* a = Abuf + ps * A_PLANE;
        const int tm = part >> 1;
        if (part & 1) F.l[tm] = *reinterpret_cast<const bf16x8*>(a + (aoff[tm][ky] ^ 32));
        else F.h[tm] = *reinterpret_cast<const bf16x8*>(a + aoff[tm][ky]);
    };
    // fragments of (stage 0, position 0); the B fragments of all of stage 0
#pragma unroll
    for (int part = 0; part < 8; ++part) ldA(Af[0], sA, 0, 0, part);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const unsigned char* b = uf_src(0, 0, ps, cur.n0);
        Bf[0][ps].h = *reinterpret_cast<const bf16x8*>(b);
        Bf[0][ps].l = *reinterpret_cast<const bf16x8*>(b + 1024);
        if (BPF == 2) {
            const unsigned char* b1 = uf_src(1, 0, ps, cur.n0);
            Bf[1][ps].h = *reinterpret_cast<const bf16x8*>(b1);
            Bf[1][ps].l = *reinterpret_cast<const bf16x8*>(b1 + 1024);
        }
    }

    unsigned sg = 0, cg = 0;         // running stage / chunk counters: the LDS buffer parities and the plane ring continue across tiles
    int par = 0;                     // parity of the block's tile counter: s_in[par] holds the current tile's statistics
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f};
    unsigned hh[2], ll[2];
    for (;;) {
        if (has_next) load_stats(nxt, par ^ 1);      // first read when this tile's last chunk stores the next tile's items: barriers in between
        const unsigned edge_c = edge_of(cur), edge_n = edge_of(nxt);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ps][tm][r] = 0.f;

        for (int chunk = 0; chunk < nchunk; ++chunk) {
            const bool in_tile = chunk + 1 < nchunk;            // the next chunk belongs to this tile
            const WTile& Tn = in_tile ? cur : nxt;              // owner of the next chunk (past the last tile: a harmless re-read of this one)
            const int c_n = in_tile ? chunk + 1 : 0;
            const int par_n = in_tile ? par : par ^ 1;
            const unsigned char* Ab = sA + (cg & 1) * A_BYTES;
            unsigned char* An = sA + ((cg + 1) & 1) * A_BYTES;
            // the chunk after the next one (its items A and Q are fetched in this chunk's last stage)
            const bool in2 = chunk + 2 < nchunk;
            const WTile& T2 = in2 ? cur : nxt;
            const int c_2 = in2 ? chunk + 2 : (chunk + 2 - nchunk);
            const unsigned edge_nn = in_tile ? edge_c : edge_n, edge_2 = in2 ? edge_c : edge_n;
            const float* xb_n = xbase_of(Tn, c_n);              // origin of the next chunk (item B)
            const float* xb_2 = xbase_of(T2, c_2);              // ... and of the one after (items A, Q)
            // (a generic lambda per vertical tap: `#pragma unroll` on a loop over this body is refused by the optimizer, and a runtime ts
            // would index aoff[][] dynamically -- scratch)
            auto stage = [&](auto ts_c) {
                constexpr int ts = decltype(ts_c)::value;
                // the next stage = (chunk, ts + 1) or (next chunk, 0): where its B fragments come from
                // the stage whose B fragments are fetched now: BPF stages ahead = (chunk, ts + BPF) or (next chunk, ts + BPF - 3)
                constexpr int ky1 = (ts + BPF) % 3;
                const int ch1 = ts + BPF < 3 ? chunk : c_n, n01 = ts + BPF < 3 ? cur.n0 : Tn.n0;
                constexpr int bs_cur = BPF == 1 ? 0 : ts, bs_ld = BPF == 1 ? 0 : (ts + 2) % 3;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    AF& Ac = Af[ps & 1];
                    AF& Ax = Af[(ps + 1) & 1];
                    const BF Bc = Bf[bs_cur][ps];
                    // next A fragments: position ps + 1 of this stage, or position 0 of the next stage (its V: this chunk's buffer with the next
                    // vertical tap, or the next chunk's)
                    const unsigned char* nA = ps < 3 ? Ab : (ts < 2 ? Ab : An);
                    const int nps = (ps + 1) & 3, nky = ps < 3 ? ts : (ts + 1) % 3;
                    const unsigned char* nb = uf_src(ky1, ch1, ps, n01);
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const int prod = k >> 2, tm = k & 3;
                        acc[ps][tm] = prod == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[tm], Bc.l, acc[ps][tm], 0, 0, 0)
                                    : prod == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l[tm], Bc.h, acc[ps][tm], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h[tm], Bc.h, acc[ps][tm], 0, 0, 0);
                        // slots 0..7: one A fragment read each
                        if (VAR == 4) {
                            if (k == 0) Ax = Ac;
                        } else if (k < 8) ldA(Ax, nA, nps, nky, k);
                        // this position's B fragment for the NEXT stage, behind the sweeps that used the current one last (lo: hi x lo, slots 0..3;
                        // hi: slots 4..11): a whole stage of flight
                        if (VAR == 0 || VAR == 2) {
                            if (k == 8) Bf[bs_ld][ps].l = *reinterpret_cast<const bf16x8*>(nb + 1024);
                            if (k == 11) Bf[bs_ld][ps].h = *reinterpret_cast<const bf16x8*>(nb);
                        }
                        if (ps >= 2 && (VAR == 0 || VAR == 1)) {
                            // ---- second half, slots n = 0..23: the input transform ----
                            const int n = (ps - 2) * 12 + k;
                            if (ts < 2 && n < 16) {              // item A (ts 0) / B (ts 1) of the next chunk: per position {pair, pair, store, store}
                                const int q = n >> 2, piece = n & 3;
                                if (n == 0) {
                                    stats_of(mu, rs, c_n, par_n);
                                    item_fill(I, mu);
                                }
                                if (piece == 0) item_pair(I, mu, rs, q, 0, hh[0], ll[0]);
                                if (piece == 1) item_pair(I, mu, rs, q, 1, hh[1], ll[1]);
                                const int a = q * A_PLANE + (ts == 0 ? dstA : dstB);
                                if (piece == 2) *reinterpret_cast<u32x2*>(An + a) = u32x2{hh[0], hh[1]};
                                if (piece == 3) *reinterpret_cast<u32x2*>(An + (a ^ 32)) = u32x2{ll[0], ll[1]};
                            }
                            if (ts == 0 && n >= 16 && n < 20) {  // the quarter item: this wave's position of a left-over item
                                const f32x4 fill = XF == 2 ? mu : f32x4{0.f, 0.f, 0.f, 0.f};
                                const f32x4 da = Qoka ? Qa : fill, db = Qokb ? Qb : fill;
                                const bool nrm = XF == 2 && wave == 1;
                                if (n == 16) pair_split<XF>(da[0], db[0], da[1], db[1], nrm ? mu[0] : 0.f, nrm ? mu[1] : 0.f, qsg, rs[0], rs[1], hh[0], ll[0]);
                                if (n == 17) pair_split<XF>(da[2], db[2], da[3], db[3], nrm ? mu[2] : 0.f, nrm ? mu[3] : 0.f, qsg, rs[2], rs[3], hh[1], ll[1]);
                                if (n == 18) *reinterpret_cast<u32x2*>(An + dstQ) = u32x2{hh[0], hh[1]};
                                if (n == 19) *reinterpret_cast<u32x2*>(An + (dstQ ^ 32)) = u32x2{ll[0], ll[1]};
                            }
                            // fetches: ts 0: item B of the next chunk; ts 2: items A and Q of the chunk after it
                            if (ts == 0 && n >= 20) item_load_part(I, xb_n, relB, flgB & edge_nn, n - 20);
                            if (ts == 2 && n >= 16 && n < 20) item_load_part(I, xb_2, relA, flgA & edge_2, n - 16);
                            if (ts == 2 && n == 20) Qa = load_px(xb_2, relQ, flgQ & edge_2, qa, Qoka);
                            if (ts == 2 && n == 21) Qb = load_px(xb_2, relQ, flgQ & edge_2, qb, Qokb);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ps == 1 && VAR != 4) {
                        lds_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                ++sg;
            };
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 2>{});
            ++cg;
        }

        // ---- epilogue of the tile: output transform in registers, bias, activation, statistics, NHWC stores.  A wave holds ALL 128 rows of its
        // 32 channels: the per-(sample, channel, tile) statistics slot is complete inside the wave (no LDS, no barrier) ----
        {
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const bool do_act = p.act != 0;
            const bool stats = p.stats_ws != nullptr;
            float* yb = p.y + (size_t)cur.tb * p.Ho * p.Wo * p.Cout;
            const int co = cur.n0 + wn * 32 + li;
            const float bsv = p.bias ? p.bias[co] : 0.f;
            const float slp = (p.act == 2) ? p.slope[co] : p.alpha;
            double st_s = 0.0, st_q = 0.0;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
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
                    const int m = tm * 32 + (li & 3) + 8 * g + 4 * kh;
                    const int oy = cur.ty0 + (m >> 3), ox = cur.tx0 + 2 * (m & 7);
                    float* dst = yb + ((size_t)oy * p.Wo + ox) * p.Cout + (co - (li & 3));
                    *reinterpret_cast<f32x4*>(dst) = f32x4{y0[0], y0[1], y0[2], y0[3]};
                    *reinterpret_cast<f32x4*>(dst + p.Cout) = f32x4{y1[0], y1[1], y1[2], y1[3]};
                }
            }
            if (stats) {
                st_s += __shfl_xor(st_s, 32, 64);            // the two k-halves of a lane pair hold rows 4 kh .. of the same channel
                st_q += __shfl_xor(st_q, 32, 64);
                if (kh == 0) {
                    double* slot = p.stats_ws + (((size_t)cur.tb * p.Cout + co) * p.stats_slots + cur.slot) * 2;
                    slot[0] = st_s;
                    slot[1] = st_q;
                }
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

}  // namespace

// conv_wino.hip's launcher: launches without a K split
int e4s_launch_wino1w(const e4s_conv_params& p, int ntn, int tx_n, int per_img, int ntiles, int grid, hipStream_t st) {
    static std::atomic<uint64_t> m0{0}, m2{0};
    int e;
#ifdef E4S_ABLATIONS
    {
        const char* ev = getenv("E4S_WINO_1W_VAR");
        const int var = ev ? atoi(ev) : 0;
        const void* fn = nullptr;
#define WV(V) case V: fn = (const void*)conv_wino1w_kernel<0, V>; if ((e = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_1W))) return e; \
              hipLaunchKernelGGL((conv_wino1w_kernel<0, V>), dim3((unsigned)grid), dim3(NTHR), SMEM_1W, st, p, ntn, tx_n, per_img, ntiles); \
              E4S_CHECK_LAUNCH(); return 0;
        switch (p.in_stats ? 0 : var) {
            WV(1) WV(2) WV(3) WV(4)
            default: break;
        }
#undef WV
    }
#endif
    if (p.in_stats) {
        if (p.Cin > 1024) return (int)hipErrorInvalidValue;
        if ((e = e4s_ensure_dyn_smem((const void*)conv_wino1w_kernel<2>, SMEM_1W + 16384, m2))) return e;
        hipLaunchKernelGGL(conv_wino1w_kernel<2>, dim3((unsigned)grid), dim3(NTHR), SMEM_1W + p.Cin * 16, st, p, ntn, tx_n, per_img, ntiles);
    } else {
        if ((e = e4s_ensure_dyn_smem((const void*)conv_wino1w_kernel<0>, SMEM_1W, m0))) return e;
        hipLaunchKernelGGL(conv_wino1w_kernel<0>, dim3((unsigned)grid), dim3(NTHR), SMEM_1W, st, p, ntn, tx_n, per_img, ntiles);
    }
    E4S_CHECK_LAUNCH();
    return 0;
}
