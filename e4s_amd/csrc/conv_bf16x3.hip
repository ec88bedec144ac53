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

constexpr int NTHR = 512;
constexpr int KC = 32;                 // input channels per stage
constexpr int ROWB = 144;              // LDS row bytes
constexpr int LO = 64;                 // byte offset of the lo half inside a row
constexpr int BM = 256, BN = 128;
constexpr int TH = 16, TW = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;     // 324 halo pixels
constexpr int WN = 2, TM = 2, TN = 2;                                          // 4 x 2 waves, 2 x 2 blocks (64x64) each
constexpr int BSTEP = NTHR / 8, BJ = BN / BSTEP;                               // B staging: rows br0 + BSTEP*j, j < BJ
constexpr int ITEMS = HALO * 4;        // (halo pixel, 8-channel group) work items of one chunk = 1296
constexpr int NPIECE = 9;               // the next chunk's halo is fetched and stored in 9 pieces, one per tap stage
constexpr int PIECE = ITEMS / NPIECE;  // items per piece = 144
constexpr int A_BYTES = HALO * ROWB, B_BYTES = BN * ROWB;
constexpr int SMEM_BYTES = 2 * A_BYTES + 2 * B_BYTES + BM * 8;
static_assert(PIECE * NPIECE == ITEMS && PIECE <= NTHR, "halo split");

__device__ __forceinline__ void split_store(unsigned char* dst, const f32x8 v) {
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + LO) = l;
}

// ABL: profiling ablations (builds with -DE4S_ABLATIONS select them with env E4S_BF16X3_ABL; results are wrong for
// ABL != 0; product builds only instantiate ABL = 0): 1 no MFMAs, 2 no fragment reads,
// 3 no global loads / LDS stores in the loop, 4 MFMAs only (no barrier either)
template <bool SCALED, int ABL>
__global__ __launch_bounds__(NTHR) void conv_bf16x3_kernel(const e4s_conv_params p, const int ntn, const int tx_n,
                                                           const int per_img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][HALO][ROWB]
    unsigned char* sB = smem + 2 * A_BYTES;            // [2][BN][ROWB]
    int* s_out = reinterpret_cast<int*>(sB + 2 * B_BYTES);
    float* s_nz = reinterpret_cast<float*>(s_out + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = logical / ntn, nt = logical - mt * ntn;
    const int n0 = nt * BN;
    const int tb = mt / per_img;
    const int rem = mt - tb * per_img;
    const int tyb = rem / tx_n, txb = rem - tyb * tx_n;

    // ---- per-row metadata: output offset and noise term of each of the 256 pixels ----
    if (tid < BM) {
        const int ay = tyb * TH + tid / TW, ax = txb * TW + tid % TW;
        const bool valid = ay < p.Ha && ax < p.Wa;
        s_out[tid] = valid ? (tb * p.Ho + ay) * p.Wo + ax : -1;
        float nz = 0.f;
        if (valid && p.noise) nz = p.noise_w[0] * p.noise[(int64_t)tb * p.noise_bstride + (int64_t)ay * p.Wo + ax];
        s_nz[tid] = nz;
    }

    const int nchunk = p.Cin / KC, nstage = nchunk * 9;
    const float* xb = p.x + (size_t)tb * p.Hi * p.Wi * p.Cin;
    const float* sc = SCALED ? p.in_scale + (size_t)tb * p.Cin : nullptr;
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w);
    const size_t wrow = (size_t)p.Cin * 4;             // bytes per (tap, cout) row of the split weights

    // halo item -> (global offset in floats or -1, LDS byte offset)
    auto item_src = [&](int item, bool& ok) -> size_t {
        const int h = item >> 2, q = item & 3;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
        ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        return ok ? ((size_t)iy * p.Wi + ix) * p.Cin + q * 8 : (size_t)(q * 8);
    };
    auto item_dst = [&](int item) -> int { return (item >> 2) * ROWB + (item & 3) * 16; };
    auto load8 = [&](const float* src) -> f32x8 {
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
        return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };

    // ---- prologue: whole halo of chunk 0 + weights of stage 0 ----
    for (int item = tid; item < ITEMS; item += NTHR) {
        bool ok;
        const size_t off = item_src(item, ok);
        f32x8 v = load8(xb + off);
        if (SCALED) v *= load8(sc + (item & 3) * 8);
        if (!ok) v = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        split_store(sA + item_dst(item), v);
    }
    const int bq = (tid & 7) * 16, br0 = tid >> 3;       // B staging role: 16-byte piece bq of rows br0 + BSTEP j
    f32x4 pb[BJ];
    {
        const unsigned char* wp = wbytes + (size_t)n0 * wrow + bq;
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(br0 + BSTEP * j) * wrow);
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + (br0 + BSTEP * j) * ROWB + bq) = pb[j];
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
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- stage loop: LDS holds stage s; the global loads of stage s+1 (weights; one piece of the next chunk's halo) are
    // issued right after the first fragment reads, fly during the 48 MFMAs (~1500 cycles) and are written to the other
    // LDS buffers at the end of the stage.  Loads are unconditional with dummy in-bounds addresses so that the waits
    // stay counted.  (A two-register-set variant with the stores delayed by one more stage measured slower: the
    // unrolled loop made the compiler shuttle the 128 accumulators between AGPRs and VGPRs every iteration.)
    struct Pref {
        f32x4 b[BJ];
        f32x8 a, s;
        int dst;
        bool part, ok;     // part: this thread holds a halo item of a real next chunk; ok: the item is inside the image
    };
    const bool piece_thr = tid < PIECE;
    int tap = 0, chunk = 0;          // stage s
    int t2 = 1, c2 = 0;              // stage s + 1

    Pref P;
    auto stage = [&](const int s) {
        const unsigned char* Ab = sA + (chunk & 1) * A_BYTES + ((tap / 3) * HALO_W + (tap % 3)) * ROWB;
        const unsigned char* Bb = sB + (s & 1) * B_BYTES;
        // One wave per SIMD: nothing else hides an LDS round trip (~120 cycles), so the fragment reads are software
        // pipelined by hand -- the A fragments of MFMA group g+1 are requested before the 6 MFMAs (192 cycles) of group
        // g are issued -- and the order is pinned with sched_barriers (left alone, the scheduler sinks each read to
        // just before its first use and the matrix pipe idles ~90 cycles per group).
        bf16x8 bh[2][TN], bl[2][TN], ah[2], al[2];
        auto ldB = [&](int kk) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bh[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32);
                bl[kk][tn] = *reinterpret_cast<const bf16x8*>(Bb + brow[tn] + kk * 32 + LO);
            }
        };
        auto ldA = [&](int g, int slot) {          // group g = (kk, tm)
            const int kk = g / TM, tm = g % TM;
            ah[slot] = *reinterpret_cast<const bf16x8*>(Ab + arow[tm] + kk * 32);
            al[slot] = *reinterpret_cast<const bf16x8*>(Ab + arow[tm] + kk * 32 + LO);
        };
        if (ABL == 2 || ABL == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = al[i] = *reinterpret_cast<const bf16x8*>(sA + arow[0]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bh[i][tn] = bl[i][tn] = ah[i];
            }
        } else {
            ldB(0);
            ldA(0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);

        // -- global -> VGPR: weights of stage s+1, halo piece `tap` of chunk+1.  Issued after the first MFMA group so the
        // matrix pipe starts right after the barrier; its ~90 address/SALU instructions are spread between the MFMAs of
        // groups 0-1 (sched_group_barrier: 1 MFMA, then up to 8 others), because after a barrier all 8 waves are in
        // the same phase and nothing else would cover them --
        auto issue_loads = [&]() {
        if (ABL < 3) {
            const bool more2 = (s + 1 < nstage);
            const unsigned char* wp =
                wbytes + ((size_t)(more2 ? t2 : 0) * p.Cout + n0) * wrow + (size_t)(more2 ? c2 : 0) * 128 + bq;
#pragma unroll
            for (int j = 0; j < BJ; ++j) P.b[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(br0 + BSTEP * j) * wrow);
            const bool doA = (tap < NPIECE) && (chunk + 1 < nchunk);
            const int item = min(tap, NPIECE - 1) * PIECE + (piece_thr ? tid : 0);
            bool ok;
            const size_t off = item_src(item, ok);
            const int cnext = doA ? (chunk + 1) * KC : 0;
            P.a = load8(xb + off + cnext);
            if (SCALED) P.s = load8(sc + cnext + (item & 3) * 8);
            P.ok = ok;
            P.part = doA && piece_thr;
            P.dst = item_dst(item);
        }
        if (ABL != 2 && ABL != 4) ldB(1);
        };
        // -- MFMAs of stage s: 4 groups of 6 --
        auto mfma_group = [&](int g) {
            const int kk = g / TM, tm = g % TM, cur = g & 1;
            if (g + 1 < 2 * TM && ABL != 2 && ABL != 4) ldA(g + 1, cur ^ 1);
            if (ABL == 1) {
                acc[tm][0][0] += (float)ah[cur][0] + (float)al[cur][1] + (float)bh[kk][0][0] + (float)bl[kk][1][1];
                return;
            }
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
        mfma_group(0);
        issue_loads();
        mfma_group(1);
        if (ABL == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x126, 8, 0);      // then up to 8 VALU / SALU / VMEM-read / DS-read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(3);
        __builtin_amdgcn_sched_barrier(0);

        // -- VGPR -> LDS --
        if (ABL < 3 && s + 1 < nstage) {
            unsigned char* db = sB + ((s + 1) & 1) * B_BYTES + br0 * ROWB + bq;
#pragma unroll
            for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(db + BSTEP * j * ROWB) = P.b[j];
        }
        if (ABL < 3 && P.part) {
            f32x8 v = P.a;
            if (SCALED) v *= P.s;
            if (!P.ok) v = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            split_store(sA + ((chunk + 1) & 1) * A_BYTES + P.dst, v);
        }
        if (ABL != 4) __syncthreads();
        if (++tap == 9) { tap = 0; ++chunk; }
        if (++t2 == 9) { t2 = 0; ++c2; }
    };
    for (int s = 0; s < nstage; ++s) stage(s);

    // ---- epilogue: demod * acc + noise + bias, activation, NHWC store ----
    float osc[TN], bsv[TN], slp[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + (wn * TN + tn) * 32 + li;
        osc[tn] = p.out_scale ? p.out_scale[(size_t)tb * p.Cout + col] : 1.f;
        bsv[tn] = p.bias ? p.bias[col] : 0.f;
        slp[tn] = (p.act == 2) ? p.slope[col] : p.alpha;
    }
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0;
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
                float v = acc[tm][tn][r] * osc[tn] + nz + bsv[tn];
                if (do_act) v = (v > 0.f ? v : v * slp[tn]) * gain;
                p.y[(size_t)off * p.Cout + n0 + (wn * TN + tn) * 32 + li] = v;
            }
        }
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
constexpr int MAXR = 16;
constexpr int XITEMS = ITEMS + MAXR * 4;           // + the chunk's style slice s[r][32]: 16 regions x 4 groups of 8
constexpr int XPIECE = (XITEMS + 8) / 9;           // 152 items per tap stage
constexpr int S_BYTES = MAXR * ROWB;
constexpr int S_OFF = 2 * A_BYTES + 2 * B_BYTES;
constexpr int SMEM_REGION = S_OFF + 2 * S_BYTES + BM * 12;
static_assert(XPIECE * 9 >= XITEMS && XPIECE <= NTHR, "extended halo split");

__global__ __launch_bounds__(NTHR) void conv_bf16x3_region_kernel(const e4s_conv_params p, const int ntn,
                                                                  const int tx_n, const int per_img,
                                                                  const int tiles_per_cls) {
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

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
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

    const int nchunk = p.Cin / KC, nstage = nchunk * 9;
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

    // ---- prologue: halo + style slice of chunk 0, weights of stage 0 ----
    for (int item = tid; item < XPIECE * 9; item += NTHR) {
        const Item it = item_of(item);
        f32x8 v = load8(it.src);
        if (!it.ok) v = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (item < XITEMS) store8(smem + it.dst, v);
    }
    const int bq = (tid & 7) * 16, br0 = tid >> 3;
    {
        const unsigned char* wp = wbytes + (size_t)n0 * wrow + bq;
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
    int tap = 0, chunk = 0, t1 = 1, c1 = 0;

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
            pa = load8(it.src + (have_next ? (chunk + 1) * KC : 0));
        };

        // -- 4 groups of 6 MFMAs; the next group's fp32 fragment is requested before the current one is converted --
        auto mfma_group = [&](int g) {
            const int kk = g / TM, tm = g % TM;
            const f32x8 v = raw * sv[tm][kk];
            if (g + 1 < 2 * TM) raw = ldraw(g + 1);
            const bf16x8 ah = __builtin_convertvector(v, bf16x8);
            const f32x8 res = v - __builtin_convertvector(ah, f32x8);
            const bf16x8 al = __builtin_convertvector(res, bf16x8);
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
        mfma_group(1);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x126, 8, 0);      // then up to 8 VALU / SALU / VMEM-read / DS-read
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(2);
        mfma_group(3);
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
    const bool do_act = p.act != 0, scaled = p.out_scale != nullptr;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int off = s_out[row];
            if (off < 0) continue;
            const float nz = s_nz[row];
            const float* drow = sD + s_grp[row] * BN;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int ncol = (wn * TN + tn) * 32 + li;
                float v = acc[tm][tn][r] * (scaled ? drow[ncol] : 1.f) + nz + bsv[tn];
                if (do_act) v = (v > 0.f ? v : v * p.alpha) * gain;
                p.y[(size_t)off * p.Cout + n0 + ncol] = v;
            }
        }
    }
}

int launch_region(const e4s_conv_params& p, hipStream_t st) {
    auto kern = conv_bf16x3_region_kernel;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM_REGION, smem_set)) return e;
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + TH - 1) / TH) * tx_n;
    const int tiles_per_cls = p.B * per_img;
    const int64_t blocks = (int64_t)tiles_per_cls * p.ncls * ntn;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), SMEM_REGION, st, p, ntn, tx_n, per_img, tiles_per_cls);
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

template <bool SCALED, int ABL = 0>
int launch(const e4s_conv_params& p, hipStream_t st) {
    auto kern = conv_bf16x3_kernel<SCALED, ABL>;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM_BYTES, smem_set)) return e;
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + TH - 1) / TH) * tx_n;
    const int64_t blocks = (int64_t)p.B * per_img * ntn;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), SMEM_BYTES, st, p, ntn, tx_n, per_img);
    E4S_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int e4s_conv_bf16x3_f32(const e4s_conv_params* pp, void* stream) {
    const e4s_conv_params& p = *pp;
    const bool up = (p.ncls == 4);
    if (p.Cin % KC || p.Cout % BN || p.ntaps != 9 || (p.ncls != 1 && !up) || p.istride != 1 ||
        p.ostride != (up ? 2 : 1) || p.tiles || p.noise_per_channel || p.Ha != p.Hi || p.Wa != p.Wi ||
        p.Ho != p.Hi * p.ostride || p.Wo != p.Wi * p.ostride)
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    if (p.labels || up) {
        if (!p.in_scale || p.act == 2 || (p.labels && (p.groups_per_batch < 1 || p.groups_per_batch > MAXR)))
            return (int)hipErrorInvalidValue;
        return launch_region(p, st);
    }
#ifdef E4S_ABLATIONS      // profiling builds only (E4S_BUILD_ABLATIONS=1 python -m e4s_amd.build): tools/bench_abl.py
    static const int abl = [] { const char* e = getenv("E4S_BF16X3_ABL"); return e ? atoi(e) : 0; }();
    switch (abl) {
        case 1: return launch<false, 1>(p, st);
        case 2: return launch<false, 2>(p, st);
        case 3: return launch<false, 3>(p, st);
        case 4: return launch<false, 4>(p, st);
        default: break;
    }
#endif
    return p.in_scale ? launch<true>(p, st) : launch<false>(p, st);
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
