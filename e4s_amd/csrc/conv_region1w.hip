// Masked StyledConv (model.py:386-400), "variant rows" form (conv_region.hip), re-cut for ONE wave per SIMD (round 6).
//
// conv_region_rows_kernel runs 8 waves of 64 x 64 at two waves per SIMD inside 256 registers each: 24 ds_read_b128 per 36 MFMAs, a
// barrier every 36 MFMAs, and every halo (+ variant) row staged once per 128 output columns.  gfx950's register file is 512 entries per
// lane and SIMD; a 256-thread block owns all of it: 16 accumulator tiles of 32 x 32 per wave (256 AGPRs) = a wave tile of 128 pixels x
// 128 channels, block tile 256 pixels x 256 channels with 2 x 2 waves.  Per tap and 16-channel k-step a wave then reads 8 A + 8 B
// fragments for 48 MFMAs (0.33 reads per MFMA instead of 0.67), the halo rows are scaled / split / stored once per 256 columns (half the
// VALU, LDS-store and L2 traffic per MFMA; the pixel tile's halo is fetched by two column tiles instead of four), and the tile analysis
// (label map -> variant rows), which no other block on the CU can hide, is paid once per 2.4 M MFMA-products instead of 1.2 M.
//
// Pipeline (per wave, everything under its own MFMAs -- there is no second wave on the SIMD to cover a stall):
//   * a pipeline step is ONE tap of a 16-channel chunk (48 MFMAs per wave); the weights of a tap are a 16 KB LDS slot in a ring of 3
//     (9 taps per chunk: slot = tap % 3 at compile time), the halo + variant rows of a chunk 36 KB, double buffered;
//   * during tap g a wave contracts the fragments it already holds, reads the A fragments of its next 32-pixel group one group ahead,
//     and in the tap's last group re-fills the B fragment registers (as their last use retires) and the first A fragment for tap g + 1;
//     stores the weights of tap g + 2 (fetched during tap g - 1) and one staging item of the next chunk's rows, and fetches the weights
//     of tap g + 3 and the next item;
//   * ONE barrier per tap, in the MIDDLE of its MFMA stream: what was stored in a tap's first group becomes visible there and is first
//     read in the NEXT tap's last group, and the only LDS reads in flight at the barrier are the two just issued for the next group --
//     a barrier at the end of the tap would expose the latency of the whole fragment refill once per tap.
// B-DIRECT layout (the default, template <WM, WN> = <1, 4>): each wave owns all 256 pixels x 64 channels (8 x 2 accumulator tiles) and loads its B
// fragments straight from a FRAGMENT-MAJOR copy of the weights (e4s_split16_bf16x2_f32 writes it behind the plane-major image: one wave
// instruction = 1 KB contiguous), a tap ahead, into one of three register sets -- no weight ring, no LDS stores or barrier dependence for the
// weights, no spills; the rows, their staging and the mid-tap barrier are the same.  Measured level with the ring form on the headline launch
// (0.334-0.342 ms both, one box) and 0-5 % ahead on the polyphase layers; E4S_REGION_1W=3 keeps the ring form selectable.
// LDS: rows 2 x 36 KB + weights 3 x 16 KB (ring form only) + d[region][256] 16 KB + tile tables = 142 KB.  Row layout, swizzle, variant-row logic, weight image
// ([tap][Cin/16][Cout][16 hi | 16 lo], e4s_split16_bf16x2_f32) and the flag table for tiles with > VMAX variant rows are conv_region.hip's.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 256, TW = 16, TH = 16, HALO_W = TW + 2, HALO = (TH + 2) * HALO_W;
constexpr int KC = 16;
constexpr int ROWB = 64, HSTR = 32, VSLOT = HSTR - HALO_W;
constexpr int VMAX = (TH + 2) * VSLOT, NROW = (TH + 2) * HSTR;
constexpr int A_BYTES = NROW * ROWB, B_SLOT = BN * ROWB, NB = 3;
constexpr int MAXR = 16;
// weight prefetch distance in taps: the weights of tap g + 2 + PFD are fetched during tap g and stored to LDS during tap g + PFD (first read
// in tap g + 1 + PFD's last group); PFD register sets rotate with the tap index (9 taps per chunk = 0 mod 3)
#ifndef E4S_1W_PFD
#define E4S_1W_PFD 1
#endif
constexpr int PFD = E4S_1W_PFD;
// B-direct layout (1 x 4 waves): B fragments of tap g + BDPF are loaded during tap g; slot of a tap where the next staging item's four loads
// start (ILD) and where the previous item's conversion + two LDS stores start (IST .. IST + 12)
#ifndef E4S_1W_BDPF
#define E4S_1W_BDPF 1
#endif
#ifndef E4S_1W_ILD
#define E4S_1W_ILD 14
#endif
#ifndef E4S_1W_IST
#define E4S_1W_IST 18
#endif
constexpr int BDPF = E4S_1W_BDPF, ILD = E4S_1W_ILD, IST = E4S_1W_IST;
static_assert((BDPF == 1 || BDPF == 2) && ILD >= 0 && ILD + 3 < 48 && IST >= 0 && IST + 12 < 48, "B-direct pipeline slots");
static_assert(PFD == 1 || PFD == 2, "weight prefetch distance");

__device__ __forceinline__ int swz(int row, int g) { return row * ROWB + ((g ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ int halo_row(int h) { return (h / HALO_W) * HSTR + h % HALO_W; }
__device__ __forceinline__ int var_row(int v) { return (v / VSLOT) * HSTR + HALO_W + v % VSLOT; }

constexpr int OFF_B = 2 * A_BYTES;
constexpr int OFF_D = OFF_B + NB * B_SLOT;          // float [MAXR][BN]  d[region][co]
constexpr int OFF_OUT = OFF_D + MAXR * BN * 4;      // int   [BM]
constexpr int OFF_NZ = OFF_OUT + BM * 4;            // float [BM]
constexpr int OFF_NEED = OFF_NZ + BM * 4;           // u32   [HALO]
constexpr int OFF_BASE = OFF_NEED + HALO * 4;       // u16   [HALO]
constexpr int OFF_VAR = OFF_BASE + 656;             // u16   [VMAX]
constexpr int OFF_LAB = OFF_VAR + VMAX * 2;         // u8    [HALO]
constexpr int OFF_GRP = OFF_LAB + 328;              // u8    [BM]
constexpr int OFF_MISC = OFF_GRP + BM;              // int   [4]
constexpr int OFF_DUMMY = (OFF_MISC + 16 + 63) / 64 * 64;      // [64 lanes][64 B]: where the staging stores of items that do not exist go
constexpr int SMEM = OFF_DUMMY + 64 * ROWB;
static_assert(OFF_DUMMY % 64 == 0, "the lo half of a row is at offset ^ 32");
static_assert(SMEM <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

// a = x * s as hi + lo bf16 (conv_region.hip's form: packed convert, shift / mask re-expansion, v_pk_fma with a negated addend)
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

// the same split in two halves of plain f32 VALU (no packed-f32 instructions: they cost ~22 cycles each beside MFMAs):
// hi = rne_bf16(x s) -> 16-byte store; lo = rne_bf16(fma(x, s, -hi)) -> 16-byte store
__device__ __forceinline__ void split_hi_store(unsigned char* dst, unsigned (&hp)[4], const f32x8 x, const f32x8 s) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = x[2 * j] * s[2 * j];
        const float v1 = x[2 * j + 1] * s[2 * j + 1];
        hp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
    }
    *reinterpret_cast<u32x4*>(dst) = u32x4{hp[0], hp[1], hp[2], hp[3]};
}
__device__ __forceinline__ void split_lo_store(unsigned char* dst, const unsigned (&hp)[4], const f32x8 x, const f32x8 s) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned lp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float h0 = __builtin_bit_cast(float, hp[j] << 16), h1 = __builtin_bit_cast(float, hp[j] & 0xffff0000u);
        const float r0 = __builtin_fmaf(x[2 * j], s[2 * j], -h0), r1 = __builtin_fmaf(x[2 * j + 1], s[2 * j + 1], -h1);
        lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    }
    *reinterpret_cast<u32x4*>(dst) = u32x4{lp[0], lp[1], lp[2], lp[3]};
}

// LDS-only workgroup barrier: every LDS access of this wave issued so far has completed (lgkmcnt(0)); global loads stay in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct Frag { bf16x8 h, l; };

// VAR: profiling variants (builds with -DE4S_ABLATIONS select them with env E4S_REGION_1W_VAR; results are WRONG for VAR >= 1; product
// builds only instantiate VAR = 0): 1 no barrier in the loop, 2 no staging (weights / rows neither fetched nor stored), 3 neither,
// 4 MFMAs only (no fragment reads either), 5 weight staging only, 6 row staging only, 7 weight fetches only, 8 weight LDS stores only
// WM x WN waves of (256 / WM) pixels x (256 / WN) channels: 2 x 2 = one wave per SIMD with 16 accumulator tiles (256 AGPRs);
// 4 x 2 = two waves per SIMD with 8 (the same block tile and pipeline; a wave's issue bubbles are covered by its SIMD partner)
template <int VAR, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN / 4, WM * WN / 4)))
void conv_region_rows1w_kernel(const e4s_conv_params p, const unsigned char* __restrict__ w16, int* __restrict__ tile_flags,
                               const int ntn, const int tx_n, const int per_img, const int tiles_per_cls) {
    constexpr int NTHR = 64 * WM * WN, TM = BM / 32 / WM, TN = BN / 32 / WN;
    constexpr int BJ = BN * 4 / NTHR;                                      // 16-byte weight pieces per thread and tap
    constexpr int NIT = (2 * (HALO + VMAX) + NTHR - 1) / NTHR;             // staging items (LDS row, 8-channel half) per thread and chunk
    // BD ("B direct", 1 x 4 waves of 256 pixels x 64 channels): a wave needs only its own 64 columns of the weights, so it loads them straight
    // into its B fragment registers from the fragment-major image behind the plane-major one (e4s_split16_bf16x2_f32 writes both), a tap
    // ahead -- no LDS ring, no ds_write_b128 of weights (the LDS write port is ~79 B/clk per CU), no B fragment reads
    constexpr bool BD = WM == 1 && WN == 4;
    static_assert(NIT + 1 <= 7 && TM >= 2 && (TN == 4 || (BD && TN == 2)), "pipeline shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][NROW][ROWB]
    unsigned char* sB = smem + OFF_B;                  // [NB][BN][ROWB]
    float* sD = reinterpret_cast<float*>(smem + OFF_D);
    int* s_out = reinterpret_cast<int*>(smem + OFF_OUT);
    float* s_nz = reinterpret_cast<float*>(smem + OFF_NZ);
    unsigned* s_need = reinterpret_cast<unsigned*>(smem + OFF_NEED);
    unsigned short* s_base = reinterpret_cast<unsigned short*>(smem + OFF_BASE);
    unsigned short* s_var = reinterpret_cast<unsigned short*>(smem + OFF_VAR);
    unsigned char* s_lab = smem + OFF_LAB;
    unsigned char* s_grp = smem + OFF_GRP;
    int* s_misc = reinterpret_cast<int*>(smem + OFF_MISC);

    const int tid = threadIdx.x;
    [[maybe_unused]] unsigned long long tstamp[5];
    if (VAR == 9) tstamp[0] = __builtin_amdgcn_s_memtime();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int R = p.groups_per_batch;

    auto label_at = [&](int oy, int ox) -> int {          // legacy-nearest label of an OUTPUT pixel (model.py:391)
        const int sy = min((int)floorf((float)oy * ((float)p.Hm / (float)p.Ho)), p.Hm - 1);
        const int sx = min((int)floorf((float)ox * ((float)p.Wm / (float)p.Wo)), p.Wm - 1);
        return p.labels[((size_t)tb * p.Hm + sy) * p.Wm + sx];
    };

    const int nchunk = p.Cin / KC;
    const float* xb = p.x + (size_t)tb * p.Hi * p.Wi * p.Cin;
    const float* stab = p.in_scale + (size_t)tb * R * p.Cin;
    // weight piece j of this thread in a tap's 16 KB run: 16-byte piece tid + NTHR j -> column (tid >> 2) + 64 j, granule tid & 3
    const size_t wtap = (size_t)nchunk * p.Cout * 64, wchunk = (size_t)p.Cout * 64;
    const unsigned char* wb = w16 + ((size_t)cls * 9 * nchunk * p.Cout + n0) * 64;       // wave-uniform; + w_voff per lane
    const unsigned w_voff = tid * 16;
    const int b_dst = swz(tid >> 2, tid & 3);             // + NTHR / 4 rows per piece: same swizzle class

    // ---- loads that do not depend on the label map: the weights of taps 0..2 of chunk 0, the d table ----
    // BD: fragment-major image: per (class, tap, chunk, 32-column block) 2 KB = [hi: lane l -> column l & 31, k-half l >> 5][lo: the same]
    const unsigned char* wf = w16 + (size_t)p.ncls * 9 * nchunk * p.Cout * 64;
    auto wf_src = [&](int tap, int chunk, int tn) -> const unsigned char* {
        return wf + ((((size_t)cls * 9 + tap) * nchunk + chunk) * (p.Cout / 32) + (n0 / 32 + wn * TN + tn)) * 2048 + lane * 16;
    };
    f32x4 pbs[3][BJ], pb1[BJ], pb2[BJ];       // pbs[k]: weights in flight, set (tap + 2) % 3 for the tap they belong to (PFD = 1 uses one set at a time)
#pragma unroll
    for (int j = 0; j < (BD ? 0 : BJ); ++j) {
        pb1[j] = *reinterpret_cast<const f32x4*>(wb + (w_voff + j * (NTHR * 16)));
        pb2[j] = *reinterpret_cast<const f32x4*>(wb + wtap + (w_voff + j * (NTHR * 16)));
        pbs[2][j] = *reinterpret_cast<const f32x4*>(wb + 2 * wtap + (w_voff + j * (NTHR * 16)));
        if (PFD == 2) pbs[0][j] = *reinterpret_cast<const f32x4*>(wb + 3 * wtap + (w_voff + j * (NTHR * 16)));
    }
    if (p.out_scale) {
        // stored as [region][column wave][lane][tn]: a lane's TN = 4 coefficients of a row are ONE 16-byte LDS read in the epilogue
        for (int t = tid; t < R * BN; t += NTHR) {
            const int r = t / BN, n = t - r * BN;
            const int cw = n / (TN * 32), tn = (n / 32) % TN, l = n % 32;
            sD[((r * WN + cw) * 32 + l) * TN + tn] = p.out_scale[((size_t)tb * R + r) * p.Cout + n0 + n];
        }
    }

    // ---- tile setup 1: output pixels (offset, noise, region), halo pixels (own region).  Straight-line: the pixel's label and noise and the
    // halo pixels' labels are ONE batch of global loads (clamped addresses, results selected afterwards), not three dependent round trips
    // under exec masks -- nothing else runs on this CU while the tile is analysed ----
    {
        constexpr int HPT = (HALO + NTHR - 1) / NTHR;              // halo pixels per thread
        const bool is_px = tid < BM;
        const int ay = tyb * TH + (tid & (BM - 1)) / TW, ax = txb * TW + (tid & (BM - 1)) % TW;
        const bool valid = is_px && ay < p.Ha && ax < p.Wa;
        const int oy = valid ? ay * p.ostride + py : 0, ox = valid ? ax * p.ostride + px : 0;
        const float nzr = p.noise ? p.noise[(int64_t)tb * p.noise_bstride + (int64_t)oy * p.Wo + ox] : 0.f;
        const int rr = label_at(oy, ox);
        int hl[HPT];
        bool hin[HPT];
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int h = tid + k * NTHR;
            const int hy = h / HALO_W, hx = h - hy * HALO_W;
            const int iy = tyb * TH + hy - 1, ix = txb * TW + hx - 1;
            hin[k] = h < HALO && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            hl[k] = label_at(hin[k] ? iy * p.ostride + py : 0, hin[k] ? ix * p.ostride + px : 0);
        }
        if (is_px) {
            s_out[tid] = valid ? (tb * p.Ho + oy) * p.Wo + ox : -1;
            s_nz[tid] = (valid && p.noise) ? p.noise_w[0] * nzr : 0.f;
            s_grp[tid] = (unsigned char)(valid ? rr : 0);
        }
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const int h = tid + k * NTHR;
            if (h < HALO) {
                s_lab[h] = hin[k] ? (unsigned char)hl[k] : 0xFF;
                s_need[h] = 0u;
            }
        }
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
    if (tid == 0 && nt == 0) tile_flags[mt] = overflow ? 1 : 0;        // overflowing tiles: the region-select kernel, second launch
    if (overflow) return;
    for (int h = tid; h < HALO; h += NTHR) {
        unsigned bits = s_need[h];
        int idx = s_base[h];
        while (bits) {
            const int r = __ffs(bits) - 1;
            s_var[idx++] = (unsigned short)((h << 4) | r);
            bits &= bits - 1;
        }
    }
    __syncthreads();

    if (VAR == 9) tstamp[1] = __builtin_amdgcn_s_memtime();
    // ---- 4: per-lane LDS row of each (pixel, tap) pair; per-thread staging items ----
    int ro[TM][9];
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
    // 1 x 4 waves (TM = 8): the 72 row offsets (< 2^16 each) packed two per register (groups 2 j and 2 j + 1) -- 36 registers instead of 72
    unsigned rop[TM / 2][9];
#pragma unroll
    for (int j = 0; j < TM / 2; ++j)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) rop[j][tap] = (unsigned)ro[2 * j][tap] | ((unsigned)ro[2 * j + 1][tap] << 16);
    auto ro_of = [&](int tm, int tap) -> int {
        if (TM == 8) {
            unsigned pk = rop[tm >> 1][tap];
            asm volatile("" : "+v"(pk));            // (opaque: otherwise the loop-invariant halves are hoisted out of the K loop -- 72 registers again)
            return (tm & 1) ? (int)(pk >> 16) : (int)(pk & 0xffffu);
        }
        return ro[tm][tap];
    };
    const int brow = swz(wn * (TN * 32) + li, kh);         // + 32 rows (2 KB) per tn: same swizzle class

    // staging item i of a chunk: LDS row j = (tid + NTHR i) / 2, 8-channel half q
    int a_src[NIT], a_sty[NIT], a_dst[NIT];
    unsigned a_srcb[NIT], a_styb[NIT];         // the same as byte offsets (32-bit voffset beside a wave-uniform base)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
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

#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        a_srcb[i] = (unsigned)a_src[i] * 4u;
        a_styb[i] = (unsigned)a_sty[i] * 4u;
    }
    // ---- prologue: chunk 0's rows, the weights of taps 0 and 1 (tap 2's stay in pb) ----
    {
        // all 2 NIT loads first (items that do not exist read offset 0 and store to the dummy rows): one round trip, not NIT under exec masks
        f32x8 px[NIT], ps[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            px[i] = load8(xb + a_src[i]);
            ps[i] = load8(stab + a_sty[i]);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int d = a_dst[i] >= 0 ? a_dst[i] : OFF_DUMMY + lane * ROWB;
            scale_split_store(smem + d, smem + (d ^ 32), px[i], ps[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < (BD ? 0 : BJ); ++j) {
        *reinterpret_cast<f32x4*>(sB + 0 * B_SLOT + b_dst + j * (NTHR / 4 * ROWB)) = pb1[j];
        *reinterpret_cast<f32x4*>(sB + 1 * B_SLOT + b_dst + j * (NTHR / 4 * ROWB)) = pb2[j];
    }
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // fragments of (chunk 0, tap 0)
    Frag Bf[TN], Af[2];
    Frag Bd[3][TN];                // BD: B fragments by tap % 3 (two sets live at a time: this tap's and the next one's in flight)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        if (BD) {
            Bd[0][tn].h = *reinterpret_cast<const bf16x8*>(wf_src(0, 0, tn));
            Bd[0][tn].l = *reinterpret_cast<const bf16x8*>(wf_src(0, 0, tn) + 1024);
            if (BDPF == 2) {
                Bd[1][tn].h = *reinterpret_cast<const bf16x8*>(wf_src(1, 0, tn));
                Bd[1][tn].l = *reinterpret_cast<const bf16x8*>(wf_src(1, 0, tn) + 1024);
            }
        } else {
            Bf[tn].h = *reinterpret_cast<const bf16x8*>(sB + brow + tn * (32 * ROWB));
            Bf[tn].l = *reinterpret_cast<const bf16x8*>(sB + (brow ^ 32) + tn * (32 * ROWB));
        }
    }
    Af[0].h = *reinterpret_cast<const bf16x8*>(sA + ro_of(0, 0));
    Af[0].l = *reinterpret_cast<const bf16x8*>(sA + (ro_of(0, 0) ^ 32));

    if (VAR == 9) tstamp[2] = __builtin_amdgcn_s_memtime();
    f32x8 ix[2], is[2];            // staging items in flight (x, style); two waves per SIMD: only [0]
    unsigned hp[4], lp[4];         // hi / lo halves of the item being stored (built over several segments)
    const int dummy = OFF_DUMMY + lane * ROWB;
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (Measured and dropped: one instantiation of the K loop per number of staging items the tile really has, 3 .. NIT -- the dummy items of the
    // branch-free staging cost nothing measurable: 0.3838 vs 0.3799 ms, and the ring layout then spills.)
    constexpr int NITB = NIT;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool have_nc = chunk + 1 < nchunk;
        const unsigned char* Ab = sA + (chunk & 1) * A_BYTES;
        const unsigned char* An = sA + ((chunk + 1) & 1) * A_BYTES;
        const int an_off = ((chunk + 1) & 1) * A_BYTES;
        const int c_n = have_nc ? (chunk + 1) * KC : 0;
        const unsigned char* wc = wb + (size_t)chunk * wchunk;
        const unsigned char* wc_n = have_nc ? wc + wchunk : wb;         // taps past the end re-read chunk 0 (never used)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // A tap is 12 TM SLOTS: one MFMA + what rides in its shadow (at most one LDS / global instruction and a few VALU), separated by
            // sched_barrier(0) so that the issue order is the source order.  History (profiles/r06a_region1w_ablations.json): the staging as one
            // block in front of a tap's MFMAs cost 20 % of the launch (one wave per SIMD: nobody covers an issue bubble); sched_group_barrier
            // pipelines meant to spread it re-ordered the MFMAs themselves; four-MFMA segments still bunched four 16-byte loads or stores
            // (13-16 issue cycles each) between two MFMAs (32 cycles apart when the matrix pipe is full).
            // Everything is BRANCH-FREE: items that do not exist (no such row, padding rows, no next chunk / tap) read an in-bounds dummy
            // and store to the wave's dummy rows or to a buffer nobody reads any more; plain f32 VALU only (packed-f32 instructions
            // beside MFMAs cost ~22 cycles each on top of their issue slot: -fno-slp-vectorize for this file, e4s_amd/build.py).
            // Barrier after the first half of the slots: what slots before it stored is first read in the NEXT tap's last group.
            const unsigned char* Bn = sB + ((t + 1) % 3) * B_SLOT;
            const unsigned char* Anx = t == 8 ? An : Ab;
            const int tnx = (t + 1) % 9;
            const int it = t - 1 >= 0 && t - 1 < NITB ? t - 1 : 0;
            const bool st_item = t >= 1 && t <= NITB && (VAR < 2 || VAR == 6 || VAR == 9);
            const bool ld_item = t < NITB && (VAR < 2 || VAR == 6 || VAR == 9);
            const int li_ = t < NITB ? t : 0;
            const int ib = TM > 2 ? (it & 1) : 0, lb = TM > 2 ? (t & 1) : 0;       // register set of the item stored / fetched in this tap
            int d_item = 0;
            if (st_item) d_item = a_dst[it] >= 0 ? an_off + a_dst[it] : dummy;
            auto hi_pair = [&](int j) {
                const float v0 = ix[ib][2 * j] * is[ib][2 * j], v1 = ix[ib][2 * j + 1] * is[ib][2 * j + 1];
                hp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
            };
            auto lo_pair = [&](int j) {
                const float h0 = __builtin_bit_cast(float, hp[j] << 16), h1 = __builtin_bit_cast(float, hp[j] & 0xffff0000u);
                const float r0 = __builtin_fmaf(ix[ib][2 * j], is[ib][2 * j], -h0), r1 = __builtin_fmaf(ix[ib][2 * j + 1], is[ib][2 * j + 1], -h1);
                lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
            };
            auto st_hi = [&]() { *reinterpret_cast<u32x4*>(smem + d_item) = u32x4{hp[0], hp[1], hp[2], hp[3]}; };
            auto st_lo = [&]() { *reinterpret_cast<u32x4*>(smem + (d_item ^ 32)) = u32x4{lp[0], lp[1], lp[2], lp[3]}; };
            // tap g stores the weights of tap g + 2 (set (t + 2) % 3, LDS slot (t + 2) % 3) and fetches those of tap g + 2 + PFD
            auto b_store = [&](int j) {
                *reinterpret_cast<f32x4*>(sB + ((t + 2) % 3) * B_SLOT + b_dst + j * (NTHR / 4 * ROWB)) = pbs[(t + 2) % 3][j];
            };
            auto b_fetch = [&](int j) {       // wave-uniform base (SALU) + this lane's 32-bit offset
                constexpr int ahead = 2 + PFD;
                const unsigned char* wp = (t + ahead >= 9 ? wc_n : wc) + (size_t)((t + ahead) % 9) * wtap + j * (NTHR * 16);
                pbs[(t + ahead) % 3][j] = *reinterpret_cast<const f32x4*>(wp + w_voff);
            };
            auto x_fetch = [&](int half) {    // 16 bytes of item t's x (half 0 / 1) into register set lb
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(xb + c_n) + (a_srcb[li_] + half * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k) ix[lb][half * 4 + k] = v[k];
            };
            auto s_fetch = [&](int half) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(stab + c_n) + (a_styb[li_] + half * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k) is[lb][half * 4 + k] = v[k];
            };
            // slot n of the tap = what rides behind its n-th MFMA (12 TM slots): at most ONE LDS / global instruction plus a few VALU
            auto slot = [&](int n) {
                if (VAR >= 2 && VAR <= 4) return;
                constexpr bool B_ST = VAR < 2 || VAR == 5 || VAR == 8 || VAR == 9, B_LD = VAR < 2 || VAR == 5 || VAR == 7 || VAR == 9,
                               ITEMS = VAR < 2 || VAR == 6 || VAR == 9;
                if (TM > 2) {                                    // 48 slots
                    if (B_ST && n >= 2 && n < 2 + BJ) b_store(n - 2);
                    if (B_LD && n >= 6 && n < 6 + BJ) {
                        b_fetch(n - 6);
                        if (!B_ST) asm volatile("" ::"v"(pbs[(t + 2 + PFD) % 3][n - 6]));
                    }
                    if (!ITEMS) return;
                    if (ld_item) {
                        if (n == 14) x_fetch(0);
                        if (n == 15) x_fetch(1);
                        if (n == 16) s_fetch(0);
                        if (n == 17) s_fetch(1);
                    }
                    if (st_item) {
                        if (n >= 18 && n < 22) hi_pair(n - 18);
                        if (n == 22) st_hi();
                        if (n >= 26 && n < 30) lo_pair(n - 26);
                        if (n == 30) st_lo();
                    }
                } else {                                         // 24 slots; one item in flight: item t - 1 is stored before item t is fetched
                    if (VAR >= 2) return;
                    if (n >= 2 && n < 2 + BJ) b_store(n - 2);
                    if (n >= 4 && n < 4 + BJ) b_fetch(n - 4);
                    if (st_item) {
                        if (n >= 6 && n < 10) hi_pair(n - 6);
                        if (n == 10) st_hi();
                        if (n >= 14 && n < 18) lo_pair(n - 14);
                        if (n == 18) st_lo();
                    }
                    if (ld_item) {
                        if (n == 20) x_fetch(0);
                        if (n == 21) x_fetch(1);
                        if (n == 22) s_fetch(0);
                        if (n == 23) s_fetch(1);
                    }
                }
            };
            if constexpr (BD) {
                // ---- 1 x 4 waves: 8 groups of 32 pixels x 6 MFMAs (2 column tiles x {hi x lo, lo x hi, hi x hi}); slot n = 6 tm + k.  The A
                // fragment of the next group rides behind the group's first two MFMAs; the four B fragment loads of the NEXT tap (set (t + 1) % 3,
                // a tap of flight) and the row staging are spread over the slots; barrier after group 3 ----
                const int tpf = (t + BDPF) % 9;
                const unsigned char* wnx0 = wf_src(tpf, t + BDPF >= 9 ? (have_nc ? chunk + 1 : 0) : chunk, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    Frag& Ac = Af[tm & 1];
                    Frag& Ax = Af[(tm + 1) & 1];
                    const bool last = tm + 1 == TM;
                    const unsigned char* ap = last ? Anx : Ab;
                    const int ao = last ? ro_of(0, tnx) : ro_of(last ? 0 : tm + 1, t);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int sweep = k >> 1, tn = k & 1, n = tm * 6 + k;
                        const Frag& Bc = Bd[t % 3][tn];
                        acc[tm][tn] = sweep == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h, Bc.l, acc[tm][tn], 0, 0, 0)
                                    : sweep == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l, Bc.h, acc[tm][tn], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h, Bc.h, acc[tm][tn], 0, 0, 0);
                        if (VAR == 4) {
                            if (k == 0) Ax = Ac;
                        } else {
                            if (k == 0) Ax.h = *reinterpret_cast<const bf16x8*>(ap + ao);
                            if (k == 1) Ax.l = *reinterpret_cast<const bf16x8*>(ap + (ao ^ 32));
                        }
                        if (VAR < 2 || VAR == 5 || VAR == 9) {                     // B fragments of tap t + BDPF: one 1 KB-contiguous load per slot
                            if (n == 2) Bd[(t + BDPF) % 3][0].h = *reinterpret_cast<const bf16x8*>(wnx0);
                            if (n == 3) Bd[(t + BDPF) % 3][0].l = *reinterpret_cast<const bf16x8*>(wnx0 + 1024);
                            if (n == 8) Bd[(t + BDPF) % 3][1].h = *reinterpret_cast<const bf16x8*>(wnx0 + 2048);
                            if (n == 9) Bd[(t + BDPF) % 3][1].l = *reinterpret_cast<const bf16x8*>(wnx0 + 3072);
                        }
                        if (VAR < 2 || VAR == 6 || VAR == 9) {
                            if (ld_item) {
                                if (n == ILD + 0) x_fetch(0);
                                if (n == ILD + 1) x_fetch(1);
                                if (n == ILD + 2) s_fetch(0);
                                if (n == ILD + 3) s_fetch(1);
                            }
                            if (st_item) {
                                if (n >= IST && n < IST + 4) hi_pair(n - IST);
                                if (n == IST + 4) st_hi();
                                if (n >= IST + 8 && n < IST + 12) lo_pair(n - IST - 8);
                                if (n == IST + 12) st_lo();
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (tm == TM / 2 - 1 && VAR != 1 && VAR != 3 && VAR != 4) {
                        lds_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                Frag& Ac = Af[tm & 1];
                Frag& Ax = Af[(tm + 1) & 1];
                const bool last = tm + 1 == TM;
                const unsigned char* ap = last ? Anx : Ab;
                const int ao = last ? ro[0][tnx] : ro[last ? 0 : tm + 1][t];
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const int sweep = k >> 2, tn = k & 3;
                    // sweep 0: hi x lo, 1: lo x hi, 2: hi x hi (small products first; consecutive MFMAs go to different accumulators)
                    acc[tm][tn] = sweep == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h, Bf[tn].l, acc[tm][tn], 0, 0, 0)
                                : sweep == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.l, Bf[tn].h, acc[tm][tn], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac.h, Bf[tn].h, acc[tm][tn], 0, 0, 0);
                    if (VAR == 4) {
                        if (k == 0) Ax = Ac;
                    } else {
                        // the A fragment of the next group (of this tap, or group 0 of the next tap) behind the first two MFMAs
                        if (k == 0) Ax.h = *reinterpret_cast<const bf16x8*>(ap + ao);
                        if (k == 1) Ax.l = *reinterpret_cast<const bf16x8*>(ap + (ao ^ 32));
                        // last group: the B fragment registers are re-filled IN PLACE for the next tap as their last use retires: the lo halves
                        // behind the lo x hi sweep (their last use was the hi x lo sweep), each hi half behind its hi x hi MFMA
                        if (last && sweep == 1) Bf[tn].l = *reinterpret_cast<const bf16x8*>(Bn + (brow ^ 32) + tn * (32 * ROWB));
                        if (last && sweep == 2) Bf[tn].h = *reinterpret_cast<const bf16x8*>(Bn + brow + tn * (32 * ROWB));
                    }
                    slot(tm * 12 + k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (tm == TM / 2 - 1 && VAR != 1 && VAR != 3 && VAR != 4) {
                    lds_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }
    }

    if (VAR == 9) tstamp[3] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: d[region][co] * acc + noise + bias, activation, NHWC store ----
    float bsv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bsv[tn] = p.bias ? p.bias[n0 + (wn * TN + tn) * 32 + li] : 0.f;
    const float gain = (p.act == 1) ? p.gain : 1.f;
    const bool do_act = p.act != 0, scaled = p.out_scale != nullptr;
    float* yo = p.y;
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
                float d4[TN];
                {
                    const float* dp = sD + ((s_grp[row] * WN + wn) * 32 + li) * TN;
                    if (TN == 4) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(dp);
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) d4[tn] = scaled ? v4[tn] : 1.f;
                    } else {
                        const float2 v2 = *reinterpret_cast<const float2*>(dp);
                        d4[0] = scaled ? v2.x : 1.f;
                        d4[TN - 1] = scaled ? v2.y : 1.f;
                    }
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float t = acc[tm][tn][r] * d4[tn];
                    t += nz + bsv[tn];
                    if (do_act) t = (t > 0.f ? t : t * p.alpha) * gain;
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
    if (VAR == 9) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[4] = __builtin_amdgcn_s_memtime();
        if ((blockIdx.x == 0 || blockIdx.x == 100) && (tid == 0 || tid == 192))
            printf("1w block %d wave %d: analysis %llu, rows+frags %llu, loop %llu, epilogue %llu cycles\n", (int)blockIdx.x, tid >> 6,
                   tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3]);
    }
}

}  // namespace

// conv_region.hip: launches the 256 x 256-tile kernel for a layer region_rows_ok() accepted, no K split; the caller runs the
// region-select fallback over the flag table afterwards.  E4S_REGION_1W selects the wave layout: 1 = 1 x 4 waves, B-direct (one per SIMD),
// 2 = 4 x 2 waves (two per SIMD), 3 = 2 x 2 waves with the weight ring
bool e4s_region_rows1w_ok(const e4s_conv_params& p) {
    return p.Cout % BN == 0 && p.Cin % 32 == 0 && p.groups_per_batch <= MAXR && (int64_t)p.Hi * p.Wi * p.Cin < (1ll << 30);       // (Cout % 32 == 0: the fragment-major weight image exists)
}

template <int WM, int WN>
static int launch1w(const e4s_conv_params& p, const void* w16, int* flags, hipStream_t st) {
    auto kern = conv_region_rows1w_kernel<0, WM, WN>;
#ifdef E4S_ABLATIONS
    static const int var = [] { const char* e = getenv("E4S_REGION_1W_VAR"); return e ? atoi(e) : 0; }();
    switch (var) {
        case 1: kern = conv_region_rows1w_kernel<1, WM, WN>; break;
        case 2: kern = conv_region_rows1w_kernel<2, WM, WN>; break;
        case 3: kern = conv_region_rows1w_kernel<3, WM, WN>; break;
        case 4: kern = conv_region_rows1w_kernel<4, WM, WN>; break;
        case 5: kern = conv_region_rows1w_kernel<5, WM, WN>; break;
        case 6: kern = conv_region_rows1w_kernel<6, WM, WN>; break;
        case 7: kern = conv_region_rows1w_kernel<7, WM, WN>; break;
        case 8: kern = conv_region_rows1w_kernel<8, WM, WN>; break;
        case 9: kern = conv_region_rows1w_kernel<9, WM, WN>; break;
        default: break;
    }
    if (int e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)) return e;
#else
    static std::atomic<uint64_t> smem_set{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, smem_set)) return e;
#endif
    const int ntn = p.Cout / BN;
    const int tx_n = (p.Wa + TW - 1) / TW, per_img = ((p.Ha + TH - 1) / TH) * tx_n;
    const int tiles_per_cls = p.B * per_img;
    const int64_t blocks = (int64_t)tiles_per_cls * p.ncls * ntn;
    if (blocks <= 0) return 0;
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WM * WN), SMEM, st, p, reinterpret_cast<const unsigned char*>(w16), flags, ntn,
                       tx_n, per_img, tiles_per_cls);
    E4S_CHECK_LAUNCH();
    return 0;
}

int e4s_launch_region_rows1w(const e4s_conv_params& p, const void* w16, int* flags, hipStream_t st, int layout) {
    // layout (E4S_REGION_1W): 1 = 1 x 4 waves, weights straight into the B fragments (the default); 2 = 4 x 2 waves, two per SIMD (spills:
    // measurement only); 3 = 2 x 2 waves with the weight ring in LDS (the round's first form)
    return layout == 2 ? launch1w<4, 2>(p, w16, flags, st) : layout == 3 ? launch1w<2, 2>(p, w16, flags, st) : launch1w<1, 4>(p, w16, flags, st);
}
