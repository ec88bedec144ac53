// Masked StyledConv (model.py:386-400) on the split-bf16 matrix-core path, "variant rows" form.
//
// The style that scales an A element belongs to the OUTPUT pixel's region: a halo pixel h read through tap t by output pixel m
// must be scaled with s[region(m)].  conv_bf16x3_region_kernel (conv_bf16x3.hip) therefore keeps the halo fp32 in LDS and every
// wave scales + splits every A fragment on its way into the MFMAs: 9 taps x 2 column waves = 18x the arithmetic of scaling each
// halo element once, on the VALU port the MFMAs are issued through -- that kernel runs at ~80 % of the plain kernel's rate.
//
// Here the halo is staged ONCE per 16-channel chunk, scaled with the style of the halo pixel's OWN region (the region of the output
// pixel at its position) and already split into hi|lo bf16, exactly like the plain kernel's halo.  Away from region boundaries
// region(m) == region(m + t) for all nine taps and the A fragments are plain shifted views of it.  Where they differ -- output
// pixels within one pixel of a boundary -- the tile gets extra LDS rows: one "variant row" per (halo pixel, foreign region that
// reads it), scaled with that region's style.  Each lane holds the LDS row of its 2 x 9 (pixel, tap) pairs in registers (own row or
// a variant row), decided once per tile from the label map; the MFMA loop has no per-fragment arithmetic at all.
// On face-parsing masks a 16x16 tile of a 64x64 map needs ~70 variant rows on average (max ~170), of a 256x256 map ~17; a tile
// that needs more than VMAX = 256 is left to the region-select kernel (flag table + a second, filtered launch).
//
// K step: 16 input channels (one v_mfma_f32_32x32x16_bf16 k-step per tap), three taps per pipeline stage, so the halo + variants
// double-buffer in 2 x 36 KB and the weights in 2 x 24 KB (64-byte swizzled rows, see ROWB below).  Weights come pre-split in the
// [tap][Cin/16][Cout][16 hi | 16 lo] layout of e4s_split16_bf16x2_f32: a stage's B tile is three contiguous 8 KB runs.
// Block = 512 threads = 4 x 2 waves of 64 x 64, tile 256 pixels x 128 output channels, one block per CU (128 KB of LDS).
#include "common.h"
#include <stdlib.h>

int e4s_launch_region_select(const e4s_conv_params& p, const int* only_flagged, hipStream_t st);      // conv_bf16x3.hip
void e4s_region_split_policy(const e4s_conv_params& p, int& ksplit, int& cper);                       // conv_bf16x3.hip
bool e4s_region_rows1w_ok(const e4s_conv_params& p);                                                  // conv_region1w.hip
int e4s_launch_region_rows1w(const e4s_conv_params& p, const void* w16, int* flags, hipStream_t st, int layout);  // conv_region1w.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int NTHR = 512, BM = 256, BN = 128, TW = 16, TH = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;
constexpr int KC = 16;                    // input channels per chunk
// LDS rows are 64 bytes [16 hi bf16 | 16 lo bf16], UNPADDED; the 16-byte granule index is XORed with (row >> 2) & 3 (round 4; the 80-byte
// rows of round 3 ran at 37-39 % bank-conflict cycles).  A halo pixel (hy, hx) lives in row hy * 32 + hx: the two 16-lane halves of an
// MFMA row tile (pixels of two image rows) are then 32 rows apart, so the 16 lanes of every ds_read_b128 group touch 16 different residues
// mod 16 = 16 different (bank quad, swizzle) pairs for any tap shift -- own-row fragment reads are conflict free.  The 14 unused slots of
// each 32-row stripe hold the variant rows (18 x 14 = 252 of them).
constexpr int ROWB = 64, HSTR = 32, VSLOT = HSTR - HALO_W;
constexpr int VMAX = (TH + 2) * VSLOT, NROW = (TH + 2) * HSTR;
constexpr int TPS = 3, NSTG = 3;          // taps per stage, stages per chunk
constexpr int WN = 2, WM = 4, TM = 2, TN = 2;
constexpr int A_BYTES = NROW * ROWB, B_BYTES = TPS * BN * ROWB;
constexpr int BJ = TPS * BN * 4 / NTHR;   // 16-byte weight pieces per thread and stage (= 3: one per tap)
constexpr int MAXR = 16;
static_assert(BJ == TPS && 2 * (HALO + VMAX) <= NSTG * NTHR, "staging split");

__device__ __forceinline__ int swz(int row, int g) { return row * ROWB + ((g ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ int halo_row(int h) { return (h / HALO_W) * HSTR + h % HALO_W; }          // LDS row of halo pixel h
__device__ __forceinline__ int var_row(int v) { return (v / VSLOT) * HSTR + HALO_W + v % VSLOT; }    // LDS row of variant row v

constexpr int OFF_B = 2 * A_BYTES;
constexpr int OFF_OUT = OFF_B + 2 * B_BYTES;        // int   [BM]    output pixel offset or -1
constexpr int OFF_NZ = OFF_OUT + BM * 4;            // float [BM]    noise term
constexpr int OFF_NEED = OFF_NZ + BM * 4;           // u32   [HALO]  bit r: a foreign region r reads this halo pixel
constexpr int OFF_BASE = OFF_NEED + HALO * 4;       // u16   [HALO]  first variant row of the halo pixel
constexpr int OFF_VAR = OFF_BASE + 656;             // u16   [VMAX]  variant row -> (halo pixel << 4) | region
constexpr int OFF_LAB = OFF_VAR + VMAX * 2;         // u8    [HALO]  own region of the halo pixel, 0xFF outside the image
constexpr int OFF_GRP = OFF_LAB + 328;              // u8    [BM]    region of the output pixel
constexpr int OFF_MISC = OFF_GRP + BM;              // int   [4]
constexpr int SMEM = OFF_MISC + 16;
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(MAXR * BN * 4 <= A_BYTES, "epilogue table");

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

// a = x * s as hi + lo bf16: hi = rne(x s) (packed convert, re-expanded with a shift / a mask), lo = rne(fma(x, s, -hi))
__device__ __forceinline__ void scale_split_store(unsigned char* dst, unsigned char* dst_lo, const f32x8 x, const f32x8 s) {
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
        f32x2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(xs), "v"(ss), "v"(hf));
        res[2 * j] = r[0];
        res[2 * j + 1] = r[1];
    }
    *reinterpret_cast<u32x4*>(dst) = hp;
    *reinterpret_cast<bf16x8*>(dst_lo) = __builtin_convertvector(res, bf16x8);
}

__global__ __launch_bounds__(NTHR) void conv_region_rows_kernel(const e4s_conv_params p, const unsigned char* __restrict__ w16,
                                                                int* __restrict__ tile_flags, const int ntn, const int tx_n,
                                                                const int per_img, const int tiles_per_cls, const int ksplit,
                                                                const int cper) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][NROW][ROWB]
    unsigned char* sB = smem + OFF_B;                  // [2][TPS*BN][ROWB]
    int* s_out = reinterpret_cast<int*>(smem + OFF_OUT);
    float* s_nz = reinterpret_cast<float*>(smem + OFF_NZ);
    unsigned* s_need = reinterpret_cast<unsigned*>(smem + OFF_NEED);
    unsigned short* s_base = reinterpret_cast<unsigned short*>(smem + OFF_BASE);
    unsigned short* s_var = reinterpret_cast<unsigned short*>(smem + OFF_VAR);
    unsigned char* s_lab = smem + OFF_LAB;
    unsigned char* s_grp = smem + OFF_GRP;
    int* s_misc = reinterpret_cast<int*>(smem + OFF_MISC);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ksplit > 1 (launches of <= 128 tiles): the 16-channel chunks of a tile are divided over ksplit consecutive blocks; each stores
    // d[region] * (its partial sum) to its slab of p.splitk_ws and e4s_splitk_epilogue adds the slabs in order (+ noise / bias / act)
    const int logical0 = xcd_remap(blockIdx.x, gridDim.x);
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
    const int R = p.groups_per_batch;

    // legacy-nearest label of an OUTPUT pixel (F.interpolate 'nearest', model.py:391)
    auto label_at = [&](int oy, int ox) -> int {
        const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Ho)), p.Hm - 1);
        const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wo)), p.Wm - 1);
        return p.labels[((size_t)tb * p.Hm + sy) * p.Wm + sx];
    };

    const int nchunk_all = p.Cin / KC, c_lo = ks * cper, nchunk = min(nchunk_all - c_lo, cper);        // this block's chunks
    const float* xb = p.x + (size_t)tb * p.Hi * p.Wi * p.Cin + c_lo * KC;
    const float* stab = p.in_scale + (size_t)tb * R * p.Cin + c_lo * KC;
    // weight piece of this thread in a tap's 8 KB run: column tid / 4, 16-byte piece tid % 4
    const size_t wtap = (size_t)nchunk_all * p.Cout * 64, wchunk = (size_t)p.Cout * 64;
    const unsigned char* wb = w16 + ((size_t)cls * 9 * nchunk_all * p.Cout + n0) * 64 + (size_t)c_lo * wchunk + (size_t)tid * 16;
    const int b_dst = swz(tid >> 2, tid & 3);

    // ---- loads that do not depend on the label map, issued before the tile analysis so that its two dependent global round trips
    // (labels, then styles) overlap them: stage 0's weights, the x of the own rows among staging items 0 / 1, the d table ----
    f32x4 pb0[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) pb0[j] = *reinterpret_cast<const f32x4*>(wb + j * wtap);
    f32x8 xe[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + NTHR * i, h = e >> 1, q = e & 1;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
        const bool inside = h < HALO && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        xe[i] = load8(xb + (inside ? (iy * p.Wi + ix) * p.Cin + q * 8 : 0));
    }
    float pd[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.out_scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tid + NTHR * k, r = t / BN, n = t - r * BN;
            if (r < R) pd[k] = p.out_scale[((size_t)tb * R + r) * p.Cout + n0 + n];
        }
    }

    // ---- tile setup 1: output pixels (offset, noise, region), halo pixels (own region) ----
    if (tid < BM) {
        const int ay = tyb * TH + tid / TW, ax = txb * TW + tid % TW;
        const bool valid = ay < p.Ha && ax < p.Wa;
        const int oy = ay * p.ostride + py, ox = ax * p.ostride + px;
        s_out[tid] = valid ? (tb * p.Ho + oy) * p.Wo + ox : -1;
        float nz = 0.f;
        int r = 0;
        if (valid) {
            if (p.noise) nz = p.noise_w[0] * p.noise[(int64_t)tb * p.noise_bstride + (int64_t)oy * p.Wo + ox];
            r = label_at(oy, ox);
        }
        s_nz[tid] = nz;
        s_grp[tid] = (unsigned char)r;
    }
    if (tid < HALO) {
        const int hy = tid / HALO_W, hx = tid - hy * HALO_W;
        const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
        const bool inside = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        s_lab[tid] = inside ? (unsigned char)label_at(iy * p.ostride + py, ix * p.ostride + px) : 0xFF;
        s_need[tid] = 0u;
    }
    __syncthreads();
    // ---- 2: which foreign regions read each halo pixel ----
    if (tid < BM && s_out[tid] >= 0) {
        const int my = tid / TW, mx = tid % TW;
        const unsigned r = s_grp[tid];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int h = (my + tap / 3) * HALO_W + mx + tap % 3;
            const unsigned lab = s_lab[h];
            if (lab != 0xFFu && lab != r) atomicOr(&s_need[h], 1u << r);
        }
    }
    __syncthreads();
    // ---- 3: variant rows of halo pixel h start at HALO + base[h] (exclusive scan of the popcounts, one wave) ----
    if (wave == 0) {
        int cnt[6], tot = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int h = lane * 6 + k;
            cnt[k] = h < HALO ? __popc(s_need[h]) : 0;
            tot += cnt[k];
        }
        int inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        int ex = inc - tot;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int h = lane * 6 + k;
            if (h < HALO) s_base[h] = (unsigned short)min(ex, 65535);
            ex += cnt[k];
        }
        if (lane == 63) s_misc[0] = inc;
    }
    __syncthreads();
    const int nvar = s_misc[0];
    const bool overflow = nvar > VMAX;
    if (tid == 0 && nt == 0 && ks == 0) tile_flags[mt] = overflow ? 1 : 0;        // overflowing tiles: the region-select kernel, second launch
    if (overflow) return;
    if (tid < HALO) {
        unsigned bits = s_need[tid];
        int idx = s_base[tid];
        while (bits) {
            const int r = __ffs(bits) - 1;
            s_var[idx++] = (unsigned short)((tid << 4) | r);
            bits &= bits - 1;
        }
    }
    __syncthreads();

    // ---- 4: per-lane LDS row of each (pixel, tap) pair; per-thread staging items ----
    int ro[TM][9], brow[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = (wm * TM + tm) * 32 + li;
        const int my = m / TW, mx = m % TW;
        const unsigned r = s_grp[m];
        const bool valid = s_out[m] >= 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int h = (my + tap / 3) * HALO_W + mx + tap % 3;
            const unsigned lab = s_lab[h];
            int row = halo_row(h);
            if (valid && lab != 0xFFu && lab != r) row = var_row(s_base[h] + __popc(s_need[h] & ((1u << r) - 1u)));
            ro[tm][tap] = swz(row, kh);
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) brow[tn] = swz((wn * TN + tn) * 32 + li, kh);

    // staging item i of a chunk (fetched during stage i of the previous chunk): LDS row j = (tid + NTHR i) / 2, 8-channel half q
    int a_src[NSTG], a_sty[NSTG], a_dst[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const int e = tid + NTHR * i, j = e >> 1, q = e & 1;
        int h = -1;
        unsigned r = 0xFFu;
        if (j < HALO) {
            h = j;
            r = s_lab[h];
        } else if (j < HALO + nvar) {
            const unsigned v = s_var[j - HALO];
            h = v >> 4;
            r = v & 15u;
        }
        a_dst[i] = -1;
        a_src[i] = a_sty[i] = 0;
        if (h >= 0) {
            const int lrow = j < HALO ? halo_row(j) : var_row(j - HALO);
            if (r == 0xFFu) {                          // outside the image: zero rows, written once
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    *reinterpret_cast<f32x4*>(sA + b2 * A_BYTES + swz(lrow, q)) = z;
                    *reinterpret_cast<f32x4*>(sA + b2 * A_BYTES + (swz(lrow, q) ^ 32)) = z;
                }
            } else {
                const int hy = h / HALO_W, hx = h - hy * HALO_W;
                const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
                a_src[i] = (iy * p.Wi + ix) * p.Cin + q * 8;
                a_sty[i] = (int)r * p.Cin + q * 8;
                a_dst[i] = swz(lrow, q);
            }
        }
    }


    // ---- prologue: chunk 0's rows, stage 0's weights ----
#pragma unroll
    for (int i = 0; i < NSTG; ++i)
        if (a_dst[i] >= 0) {
            const bool early = i == 0 || (i == 1 && tid < 2 * (HALO - NTHR / 2));        // own rows fetched above
            const f32x8 x = early ? xe[i < 2 ? i : 0] : load8(xb + a_src[i]);
            scale_split_store(sA + a_dst[i], sA + (a_dst[i] ^ 32), x, load8(stab + a_sty[i]));
        }
#pragma unroll
    for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + j * BN * ROWB + b_dst) = pb0[j];
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    struct BFrag { bf16x8 h[TN], l[TN]; };
    struct AFrag { bf16x8 h, l; };
    unsigned sg = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool have_nc = chunk + 1 < nchunk;
        const unsigned char* Ab = sA + (chunk & 1) * A_BYTES;
        unsigned char* An = sA + ((chunk + 1) & 1) * A_BYTES;
        const int c_n = (chunk + 1) * KC;
#pragma unroll
        for (int ts = 0; ts < NSTG; ++ts) {
            const unsigned char* Bb = sB + (sg & 1) * B_BYTES;
            const bool last_ts = ts + 1 == NSTG;
            const bool more = !last_ts || have_nc;
            f32x4 pb[BJ];
            f32x8 pax, pas;
            const bool a_on = have_nc && a_dst[ts] >= 0;
            auto issue_loads = [&]() {
                // weights of the next stage: next tap group of this chunk, or tap group 0 of the next chunk
                const unsigned char* wp = wb + (size_t)(last_ts ? 0 : (ts + 1) * TPS) * wtap +
                                          (size_t)(more ? (last_ts ? chunk + 1 : chunk) : 0) * wchunk;
#pragma unroll
                for (int j = 0; j < BJ; ++j) pb[j] = *reinterpret_cast<const f32x4*>(wp + j * wtap);
                // item ts of the next chunk's rows
                pax = load8(xb + (a_on ? a_src[ts] + c_n : 0));
                pas = load8(stab + (a_on ? a_sty[ts] + c_n : 0));
            };
            auto ldB = [&](BFrag& F, int u) {
                const unsigned char* Bt = Bb + u * BN * ROWB;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    F.h[tn] = *reinterpret_cast<const bf16x8*>(Bt + brow[tn]);
                    F.l[tn] = *reinterpret_cast<const bf16x8*>(Bt + (brow[tn] ^ 32));
                }
            };
            auto ldA = [&](AFrag& F, int g) {
                const int u = g / TM, tm = g - u * TM;
                F.h = *reinterpret_cast<const bf16x8*>(Ab + ro[tm][ts * TPS + u]);
                F.l = *reinterpret_cast<const bf16x8*>(Ab + (ro[tm][ts * TPS + u] ^ 32));
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
            constexpr int NGRP = TPS * TM;            // MFMA groups (3 TN MFMAs each) per stage
            BFrag B0, B1;
            AFrag A0, A1;
            ldB(B0, 0);
            ldA(A0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
                const int u = g / TM, tm = g - u * TM;
                AFrag& Ac = (g & 1) ? A1 : A0;
                AFrag& Anx = (g & 1) ? A0 : A1;
                BFrag& Bc = (u & 1) ? B1 : B0;
                BFrag& Bnx = (u & 1) ? B0 : B1;
                if (g + 1 < NGRP) ldA(Anx, g + 1);
                if (tm == 0 && u + 1 < TPS) ldB(Bnx, u + 1);
                if (g == 0) issue_loads();
                mfmas(Ac, Bc, tm);
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < 3 * TN; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x126, 12, 0);     // then up to 12 VALU / SALU / VMEM-read / DS-read
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // -- VGPR -> LDS: weights of the next stage, item ts of the next chunk's rows --
            if (more) {
                unsigned char* db = sB + ((sg + 1) & 1) * B_BYTES + b_dst;
#pragma unroll
                for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(db + j * BN * ROWB) = pb[j];
            }
            if (a_on) scale_split_store(An + a_dst[ts], An + (a_dst[ts] ^ 32), pax, pas);
            __syncthreads();
            ++sg;
        }
    }

    // ---- epilogue: d[region][co] * acc + noise + bias, activation, NHWC store (as the region-select kernel) ----
    float* sD = reinterpret_cast<float*>(sA);          // [R][BN]; the loop's last barrier has passed
    if (p.out_scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tid + NTHR * k;
            if (t < R * BN) sD[t] = pd[k];
        }
        __syncthreads();
    }
    float bsv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bsv[tn] = p.bias ? p.bias[n0 + (wn * TN + tn) * 32 + li] : 0.f;
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0, scaled = p.out_scale != nullptr, raw = ksplit > 1;
    float* yo = raw ? p.splitk_ws + (size_t)ks * ((size_t)p.B * p.Ho * p.Wo * p.Cout) : p.y;
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
}

// fp32 weights [rows = ncls * 9][Cout][Cin] -> [rows][Cin/16][Cout][16 hi | 16 lo] bf16; one thread per (row, chunk, co, 8 channels)
__global__ void split16_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, const int64_t n, const int cout,
                               const int cin, const int64_t nrows, const int frag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int q = (int)(i & 1);
    const int64_t t = i >> 1;
    const int nch = cin / KC;
    const int co = (int)(t % cout);
    const int64_t t2 = t / cout;
    const int c = (int)(t2 % nch);
    const int64_t row = t2 / nch;
    const f32x8 v = load8(w + ((size_t)row * cout + co) * cin + c * KC + q * 8);
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    unsigned char* dst = out + (((size_t)row * nch + c) * cout + co) * 64 + q * 16;
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + 32) = l;
    // the same 8 + 8 values fragment-major (conv_region1w.hip loads its B fragments straight from global memory): behind the image above,
    // per (row, chunk, 32-column block) 2 KB = [hi: lane -> (column & 31, k-half q)][lo], 16 bytes per lane
    if (frag) {
        unsigned char* fr = out + (size_t)nrows * nch * cout * 64 + (((size_t)row * nch + c) * (cout / 32) + co / 32) * 2048 + (q * 32 + (co & 31)) * 16;
        *reinterpret_cast<bf16x8*>(fr) = h;
        *reinterpret_cast<bf16x8*>(fr + 1024) = l;
    }
}

}  // namespace

extern "C" int64_t e4s_split16_bytes(int64_t rows, int cout, int cin) {
    if (cin % KC || rows < 0 || cout <= 0) return -1;
    return rows * cout * cin * 4 * (cout % 32 == 0 ? 2 : 1);           // plane-major image [+ fragment-major image]
}

extern "C" int e4s_split16_bf16x2_f32(const float* w, void* out, int64_t rows, int cout, int cin, void* stream) {
    if (cin % KC || rows < 0 || cout <= 0) return (int)hipErrorInvalidValue;
    const int64_t n = rows * (cin / 8) * cout;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(split16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), w,
                       reinterpret_cast<unsigned char*>(out), n, cout, cin, rows, cout % 32 == 0 ? 1 : 0);
    E4S_CHECK_LAUNCH();
    return 0;
}

static bool region_rows_ok(const e4s_conv_params& p) {
    const bool up = (p.ncls == 4);
    return p.labels && p.in_scale && !p.in_stats && !p.y_cstride && p.act != 2 && p.Cin % 32 == 0 && p.Cout % BN == 0 &&
           p.ntaps == 9 && (p.ncls == 1 || up) && p.istride == 1 && p.ostride == (up ? 2 : 1) && !p.tiles &&
           !p.noise_per_channel && p.Ha == p.Hi && p.Wa == p.Wi && p.Ho == p.Hi * p.ostride && p.Wo == p.Wi * p.ostride &&
           !p.stats_ws && p.groups_per_batch >= 1 && p.groups_per_batch <= MAXR &&
           (int64_t)p.Hi * p.Wi * p.Cin < (1ll << 31) && (int64_t)p.B * p.Ho * p.Wo < (1ll << 31);
}

extern "C" int64_t e4s_conv_region_ws_floats(const e4s_conv_params* pp) {
    const e4s_conv_params& p = *pp;
    if (!region_rows_ok(p)) return 0;
    int ksplit, cper;
    e4s_region_split_policy(p, ksplit, cper);
    const int64_t tiles = (int64_t)p.B * ((p.Ha + TH - 1) / TH) * ((p.Wa + TW - 1) / TW) * p.ncls;
    return (ksplit > 1 ? (int64_t)ksplit * p.B * p.Ho * p.Wo * p.Cout : 0) + tiles;        // [split-K slabs][tile flags]
}

static int region_1w_mode() {
    static const int m = [] { const char* e = getenv("E4S_REGION_1W"); return e ? atoi(e) : 1; }();
    return m;
}

extern "C" int e4s_conv_region_path(const e4s_conv_params* pp) {
    if (!pp || !region_rows_ok(*pp)) return 0;
    int ksplit, cper;
    e4s_region_split_policy(*pp, ksplit, cper);
    return (region_1w_mode() && ksplit == 1 && e4s_region_rows1w_ok(*pp)) ? 2 : 1;
}

extern "C" int e4s_conv_region_bf16x3_f32(const e4s_conv_params* pp, const void* w16, void* stream) {
    const e4s_conv_params& p = *pp;
    if (!region_rows_ok(p)) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int ksplit, cper;
    e4s_region_split_policy(p, ksplit, cper);
    if (!w16) return e4s_launch_region_select(p, nullptr, st);
    if (!p.splitk_ws) return (int)hipErrorInvalidValue;
    auto kern = conv_region_rows_kernel;
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, smem_set)) return e;
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + TH - 1) / TH) * tx_n;
    const int tiles_per_cls = p.B * per_img;
    const int64_t blocks = (int64_t)tiles_per_cls * p.ncls * ntn * ksplit;
    if (blocks <= 0) return 0;
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    int* flags = reinterpret_cast<int*>(p.splitk_ws + (ksplit > 1 ? (size_t)ksplit * p.B * p.Ho * p.Wo * p.Cout : 0));
    // launches that fill the chip without a K split and have >= 256 output channels: the one-wave-per-SIMD kernel (256 x 256 tiles,
    // conv_region1w.hip; E4S_REGION_1W=0 keeps this file's kernel: the A/B switch of tools/bench_region.py)
    const int use_1w = region_1w_mode();
    if (use_1w && ksplit == 1 && e4s_region_rows1w_ok(p)) {
        if (int e = e4s_launch_region_rows1w(p, w16, flags, st, use_1w)) return e;
        return e4s_launch_region_select(p, flags, st);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NTHR), SMEM, st, p, reinterpret_cast<const unsigned char*>(w16), flags,
                       ntn, tx_n, per_img, tiles_per_cls, ksplit, cper * (32 / KC));
    E4S_CHECK_LAUNCH();
    // tiles with > VMAX variant rows (flag table written by the launch above) on the region-select kernel, same K split and slabs;
    // that launcher also runs the second stage of a split launch
    return e4s_launch_region_select(p, flags, st);
}
