// Exact up-sampling StyledConv on the bf16 matrix cores at fp32-class accuracy: conv_transpose2d(stride 2, 3x3) followed by the
// 4x4 FIR blur, NoiseInjection and FusedLeakyReLU (src/models/stylegan2/model.py:287-300, 206-213, 396-404) in ONE kernel at
// ~1.5x the layer's MACs (9 Cin Cout per INPUT pixel; 1.52x for the tile halo).  The polyphase form on e4s_conv_bf16x3_f32
// (ncls = 4: four 3x3 phase kernels) spends 4x; the two unmasked up-convs of a face swap (128 -> 64 into 512^2, 64 -> 32 into
// 1024^2) ran 4-7x above their HBM floor on it.
//
// (1) The transposed conv as a SUB-PIXEL GEMM.  With I[q] = sum_{2u + k = q} x[u] W[k] (no flip: conv_transpose2d), the four
//     parity classes of q are four small convolutions over the INPUT grid,
//         I[2a + dy, 2b + dx] = sum over shifts (sy, sx) in {0,-1}^2 with (sy == 0 or dy == 0) and (sx == 0 or dx == 0) of
//                               x[a + sy, b + sx] . W[dy - 2 sy, dx - 2 sx]
//     class (0,0) has 4 taps, (0,1) and (1,0) two, (1,1) one: 9 (class, shift) blocks = exactly the layer's MACs, and the
//     scatter-add of the transposed conv happens in the MFMA accumulators.  Arithmetic as conv_bf16x3.hip: three
//     v_mfma_f32_32x32x16_bf16 per product on hi/lo-split fp32 operands, fp32 accumulate.
// (2) Tile = 8 x 16 anchors a (the GEMM's 128 rows) x 32 output channels x 4 classes (128 columns): a 16 x 32 patch of I for 32
//     channels = 64 KB of LDS, from which the FIR epilogue produces the 12 x 28 output pixels whose 4x4 windows lie inside
//     (the anchors overlap by 2 between tiles: 1.52x the MACs -- still 2.6x fewer than the polyphase form).
//     out = act(d[b,co] * sum_j kflip[j] I[o + j - 1] + noise_w * noise[b,o] + bias[co]) * gain.
// (3) Persistent blocks, 8 waves: wave (wm, wn) owns anchor rows [32 wm, 32 wm + 32) and classes {2 wn, 2 wn + 1}; waves w and
//     w + 4 share a SIMD and take different wn, so every SIMD carries 9 blocks although the classes have 4 / 2 / 2 / 1 taps.
//     One pipeline stage = one 32-channel chunk (all 9 blocks); halos are fetched into registers two stages ahead, weights one
//     (or the next tile's first); the I tile aliases the weight buffers, so the weights prefetched for the next tile are parked
//     in registers across the epilogue.
// Weights arrive pre-packed and pre-split by e4s_subpixel_weights_f32: [Cin/32][Cout/32][9 blocks][32 co][32 hi | 32 lo bf16].
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 32, ROWB = 144, LO = 64;
constexpr int TAH = 8, TAW = 16;                           // anchors per tile (GEMM rows: 128)
constexpr int HH = TAH + 1, HW = TAW + 1, HALO = HH * HW;  // 9 x 17 input pixels: anchors and their (-1,-1) neighbours
constexpr int OH = 2 * TAH - 4, OW = 2 * TAW - 4;          // 12 x 28 outputs per tile
constexpr int IQH = 2 * TAH, IQW = 2 * TAW;                // 16 x 32 positions of I
constexpr int BNC = 32;                                    // output channels per tile
constexpr int NTHR = 512, NBLK = 9;
constexpr int ITEMS = HALO * 4;                            // (halo pixel, 8-channel group) items of one chunk = 612
constexpr int AJ = (ITEMS + NTHR - 1) / NTHR;              // 2
constexpr int BPIECES = NBLK * BNC * 8;                    // 16-byte weight pieces of one chunk = 2304
constexpr int BJ = (BPIECES + NTHR - 1) / NTHR;            // 5
constexpr int A_BYTES = HALO * ROWB;                       // 22 032
constexpr int B_BYTES = NBLK * BNC * ROWB;                 // 41 472
constexpr int I_BYTES = IQH * IQW * BNC * 4;               // 65 536
constexpr int SMEM = 2 * A_BYTES + 2 * B_BYTES + OH * OW * 4;
static_assert(I_BYTES <= 2 * B_BYTES, "the I tile aliases the two weight buffers");
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(8 * OW * 2 <= NTHR, "FIR epilogue thread layout");

__device__ __forceinline__ void split_store(unsigned char* dst, const f32x8 v) {
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(r, bf16x8);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + LO) = l;
}

__device__ __forceinline__ f32x8 load8(const float* src) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 4);
    return f32x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
}

struct TileId { int tb, ty, tx, nt; };

// XF: 0 none, 1 v * in_scale[b][c] while the halo is staged (one style per sample: unmasked StyledConv, model.py:655-657)
template <int XF>
__global__ __launch_bounds__(NTHR) void upconv_fused_kernel(const e4s_conv_params p, const float* __restrict__ k4, const int ntn,
                                                            const int tx_n, const int per_img, const int ntiles, const int abl) {
    // abl (profiling builds only, -DE4S_ABLATIONS + env E4S_UPCONV3_ABL; results WRONG): 1 no MFMA stages, 2 no FIR / output
    // stores, 3 no epilogue at all, 4 no global loads, 5 no LDS staging of the prefetched operands
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][HALO][ROWB]
    unsigned char* sB = smem + 2 * A_BYTES;            // [2][9][32][ROWB]
    float* sI = reinterpret_cast<float*>(sB);          // [IQH][IQW][BNC]  (epilogue; aliases both weight buffers)
    float* s_nz = reinterpret_cast<float*>(sB + 2 * B_BYTES);          // [OH*OW] noise_w * noise of the tile's output pixels

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the class schedule branches on it
    const int li = lane & 31, kh = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int nchunk = p.Cin / KC;
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w);
    const size_t img_stride = (size_t)p.Hi * p.Wi * p.Cin;

    auto decode = [&](int t) -> TileId {
        TileId id;
        const int mt = t / ntn;
        id.nt = t - mt * ntn;
        id.tb = mt / per_img;
        const int rem = mt - id.tb * per_img;
        id.ty = rem / tx_n;
        id.tx = rem - id.ty * tx_n;
        return id;
    };
    // halo item -> global offset inside the sample (floats), LDS byte offset; halo pixel (hy, hx) = input (6 ty - 2 + hy, 14 tx - 2 + hx)
    auto item_src = [&](const TileId& id, int item, bool& ok) -> size_t {
        const int h = item >> 2, q = item & 3;
        const int hy = h / HW, hx = h - hy * HW;
        const int iy = id.ty * (OH / 2) - 2 + hy, ix = id.tx * (OW / 2) - 2 + hx;
        ok = item < ITEMS && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        return ok ? ((size_t)iy * p.Wi + ix) * p.Cin + q * 8 : (size_t)(q * 8);
    };
    auto item_dst = [&](int item) -> int { return (item >> 2) * ROWB + (item & 3) * 16; };
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // weights of (chunk, nt): 9 blocks x 32 rows x 128 bytes, contiguous; piece i -> LDS row i/8, 16-byte column i%8
    auto b_src = [&](int i, int chunk, int nt) -> size_t {
        return ((size_t)chunk * ntn + nt) * (NBLK * BNC * 128) + (size_t)i * 16;
    };
    auto b_dst = [&](int i) -> int { return (i >> 3) * ROWB + (i & 7) * 16; };

    struct AReg {
        f32x8 a[AJ], x[AJ];
        bool ok[AJ];
    };
    typedef f32x4 BReg[BJ];
    auto fetch_a = [&](AReg& R, const TileId& id, int chunk, bool real) {
        const float* xb = p.x + (real ? (size_t)id.tb * img_stride + chunk * KC : 0);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int item = tid + NTHR * j;
            const size_t off = item_src(id, item, R.ok[j]);
            if (abl != 4) R.a[j] = load8(xb + off);
            if (XF) R.x[j] = load8(p.in_scale + (real ? (size_t)id.tb * p.Cin + chunk * KC : 0) + (item & 3) * 8);
        }
    };
    auto fetch_b = [&](BReg& R, const TileId& id, int chunk, bool real) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j;
            if (abl != 4) R[j] = *reinterpret_cast<const f32x4*>(wbytes + b_src(i < BPIECES ? i : 0, real ? chunk : 0, real ? id.nt : 0));
        }
    };
    auto store_a = [&](const AReg& R, int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int item = tid + NTHR * j;
            if (item < ITEMS) {
                f32x8 v = R.a[j];
                if (XF) v = v * R.x[j];
                if (!R.ok[j]) v = zero8;                    // zero padding applies after the style scale
                split_store(sA + buf * A_BYTES + item_dst(item), v);
            }
        }
    };
    auto store_b = [&](const BReg& R, int buf) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + NTHR * j;
            if (i < BPIECES) *reinterpret_cast<f32x4*>(sB + buf * B_BYTES + b_dst(i)) = R[j];
        }
    };

    if (first >= ntiles) return;
    TileId cur = decode(first);
    int t_next = first + G;
    bool has_next = t_next < ntiles;
    TileId nxt = decode(has_next ? t_next : first);

    // Pipeline over the flat sequence of (tile, chunk) stages, unrolled by two (Cin % 64 == 0: an even number of chunks per
    // tile, so even stages always live in LDS buffers 0 and odd ones in buffers 1).  A stage's halo is FETCHED two stages
    // ahead (register sets RE / RO for even / odd stages) and stored one stage ahead -- one MFMA phase (~0.8 us) is shorter
    // than an HBM round trip under load, which left every wave waiting on its loads at the end of each stage; its weights
    // (L2-resident) are fetched one stage ahead.
    AReg RE, RO;
    BReg RB;
    fetch_a(RE, cur, 0, true);
    fetch_b(RB, cur, 0, true);
    store_a(RE, 0);
    store_b(RB, 0);
    fetch_a(RO, cur, 1, true);
    __syncthreads();

    // fragment rows: anchor m = 32 wm + li -> (ay, ax) = (m / 16, m % 16); shift (sy, sx) reads halo pixel (ay + 1 + sy, ax + 1 + sx)
    const int m_row = wm * 32 + li;
    const int arow = ((m_row / TAW) * HW + (m_row % TAW)) * ROWB + kh * 16;
    const int brow = li * ROWB + kh * 16;
    // FIR epilogue role: (4 output channels, output column, half of the 12 rows)
    const int e_c4 = tid & 7, e_ox = (tid >> 3) % OW, e_half = tid / (8 * OW);
    const bool e_act = tid < 8 * OW * 2;

    f32x16 acc[2];
    // block order in LDS / in the packed weights: class 0: shifts 0,1,2,3 -> 0..3; class 1: shifts 0,2 -> 4,5;
    // class 2: shifts 0,1 -> 6,7; class 3: shift 0 -> 8
    auto contract = [&](int buf) {
        const unsigned char* Ab = sA + buf * A_BYTES + arow;
        const unsigned char* Bb = sB + buf * B_BYTES + brow;
        auto ldA = [&](int shift, int kk, bf16x8& h, bf16x8& l) {
            const unsigned char* a = Ab + ((1 - (shift >> 1)) * HW + (1 - (shift & 1))) * ROWB + kk * 32;
            h = *reinterpret_cast<const bf16x8*>(a);
            l = *reinterpret_cast<const bf16x8*>(a + LO);
        };
        auto mm = [&](f32x16& c, const bf16x8& ah, const bf16x8& al, int blk, int kk) {
            const unsigned char* b = Bb + blk * (BNC * ROWB) + kk * 32;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b + LO);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        };
        if (wn == 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 h0, l0, h1, l1, h2, l2, h3, l3;
                ldA(0, kk, h0, l0); ldA(1, kk, h1, l1); ldA(2, kk, h2, l2); ldA(3, kk, h3, l3);
                mm(acc[0], h0, l0, 0, kk);
                mm(acc[1], h0, l0, 4, kk);
                mm(acc[0], h1, l1, 1, kk);
                mm(acc[1], h2, l2, 5, kk);
                mm(acc[0], h2, l2, 2, kk);
                mm(acc[0], h3, l3, 3, kk);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 h0, l0, h1, l1;
                ldA(0, kk, h0, l0); ldA(1, kk, h1, l1);
                mm(acc[0], h0, l0, 6, kk);
                mm(acc[1], h0, l0, 8, kk);
                mm(acc[0], h1, l1, 7, kk);
            }
        }
    };

    for (;;) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // epilogue operands of THIS tile, requested now so that their latency hides under the tile's MFMA stages
        float nz_reg = 0.f;
        if (tid < OH * OW && p.noise) {
            const int oy = cur.ty * OH + tid / OW, ox = cur.tx * OW + tid % OW;
            if (oy < p.Ho && ox < p.Wo) nz_reg = p.noise[(int64_t)cur.tb * p.noise_bstride + (int64_t)oy * p.Wo + ox];
        }
        f32x4 d_reg = {1.f, 1.f, 1.f, 1.f}, b_reg = {0.f, 0.f, 0.f, 0.f};
        if (e_act) {
            const int co0 = cur.nt * BNC + e_c4 * 4;
            if (p.out_scale) d_reg = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)cur.tb * p.Cout + co0);
            if (p.bias) b_reg = *reinterpret_cast<const f32x4*>(p.bias + co0);
        }
        const float nw_reg = p.noise ? p.noise_w[0] : 0.f;

        for (int c0 = 0; c0 < nchunk; c0 += 2) {
            const bool last_pair = (c0 + 2 == nchunk);
            const TileId& nid = last_pair ? nxt : cur;       // owner of the two stages after this pair
            const bool nreal = !last_pair || has_next;
            const int ce = last_pair ? 0 : c0 + 2, co = last_pair ? 1 : c0 + 3;
            // ---- even stage: chunk c0 in buffers 0 ----
            fetch_a(RE, nid, ce, nreal);                     // halo of the next even stage (two stages ahead)
            fetch_b(RB, cur, c0 + 1, true);                  // weights of the odd stage
            if (abl != 1) contract(0);
            if (abl != 5) {
                store_a(RO, 1);                              // halo of the odd stage, fetched during the previous odd stage
                store_b(RB, 1);
            }
            __syncthreads();
            // ---- odd stage: chunk c0 + 1 in buffers 1 ----
            fetch_a(RO, nid, co, nreal);
            fetch_b(RB, nid, ce, nreal);                     // weights of the next even stage
            if (abl != 1) contract(1);
            if (abl != 5) store_a(RE, 0);
            // after the tile's last stage the I tile is about to overwrite both weight buffers: the next tile's weights wait
            // in registers until the epilogue is through
            if (!last_pair && abl != 5) store_b(RB, 0);
            __syncthreads();
        }
        if (abl == 3) {
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][7]));
            if (!has_next) break;
            store_b(RB, 0);
            __syncthreads();
            cur = nxt;
            t_next += G;
            has_next = t_next < ntiles;
            if (has_next) nxt = decode(t_next);
            continue;
        }

        // ---- epilogue (a): accumulators -> I tile; anchor (ay, ax), class (dy, dx) -> I[2 ay + dy][2 ax + dx][co] ----
        {
            if (tid < OH * OW) s_nz[tid] = nw_reg * nz_reg;      // noise of the tile's output pixels (outside the aliased region)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int cls = wn * 2 + c, dy = cls >> 1, dx = cls & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int qy = 2 * (m / TAW) + dy, qx = 2 * (m % TAW) + dx;
                    sI[(qy * IQW + qx) * BNC + li] = acc[c][r];
                }
            }
        }
        __syncthreads();
        // ---- epilogue (b): FIR + demodulation + noise + bias + activation.  Output (oyl, oxl) of the tile reads I rows
        // oyl + 1 .. oyl + 4, columns oxl + 1 .. oxl + 4 (tile origin of I = 2 * (first anchor) = output origin - 2) ----
        if (e_act && abl != 2) {
            // flipped blur taps (upfirdn2d is a true convolution, upfirdn2d_kernel.cu:77): kf[ay*4+ax] = k4[3-ay][3-ax]; re-read per
            // tile (one scalar load) rather than held in 16 SGPRs across the main loop
            float kf[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) kf[a] = k4[15 - a];
            const int co0 = cur.nt * BNC + e_c4 * 4;
            const int ox = cur.tx * OW + e_ox;
            const int oyl0 = e_half * (OH / 2);
            const f32x4 d = d_reg, bs = b_reg;
            const float gain = (p.act == 1) ? p.gain : 1.f;
            const int ycs = p.y_cstride ? p.y_cstride : p.Cout;
            const float* ip = sI + ((oyl0 + 1) * IQW + e_ox + 1) * BNC + e_c4 * 4;
            f32x4 win[3][4];                            // sliding window: 3 I rows carried, one new row per output
#pragma unroll
            for (int ry = 0; ry < 3; ++ry)
#pragma unroll
                for (int ax = 0; ax < 4; ++ax) win[ry][ax] = *reinterpret_cast<const f32x4*>(ip + (ry * IQW + ax) * BNC);
#pragma unroll
            for (int r = 0; r < OH / 2; ++r) {
                f32x4 nw[4];
#pragma unroll
                for (int ax = 0; ax < 4; ++ax) nw[ax] = *reinterpret_cast<const f32x4*>(ip + ((r + 3) * IQW + ax) * BNC);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ax = 0; ax < 4; ++ax)
                    v += win[0][ax] * kf[ax] + win[1][ax] * kf[4 + ax] + win[2][ax] * kf[8 + ax] + nw[ax] * kf[12 + ax];
#pragma unroll
                for (int ax = 0; ax < 4; ++ax) { win[0][ax] = win[1][ax]; win[1][ax] = win[2][ax]; win[2][ax] = nw[ax]; }
                const int oyl = oyl0 + r, oy = cur.ty * OH + oyl;
                if (oy < p.Ho && ox < p.Wo) {
                    v = v * d + bs + s_nz[oyl * OW + e_ox];
                    if (p.act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * p.alpha) * gain;
                    }
                    *reinterpret_cast<f32x4*>(p.y + (((size_t)cur.tb * p.Ho + oy) * p.Wo + ox) * ycs + co0) = v;
                }
            }
        }
        if (!has_next) break;
        __syncthreads();                               // every FIR read of the I tile is done: the weight buffers are free again
        store_b(RB, 0);                                // the next tile's chunk-0 weights, parked in registers since their prefetch
        __syncthreads();
        cur = nxt;
        t_next += G;
        has_next = t_next < ntiles;
        if (has_next) nxt = decode(t_next);
    }
}

// w [Cout,Cin,3,3] -> the operand of the kernel above, packed AND split: [Cin/32][Cout/32][9 blocks][32 co][32 hi | 32 lo] bf16.
// block -> (class, shift): 0..3 = class 0 shifts 0..3; 4,5 = class 1 shifts 0,2; 6,7 = class 2 shifts 0,1; 8 = class 3 shift 0;
// class (dy,dx) = (cls >> 1, cls & 1), shift s = (sy, sx) = (-(s >> 1), -(s & 1)); the tap is W[dy - 2 sy][dx - 2 sx].
__global__ void subpixel_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin,
                                        int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one (row, 8-channel group)
    if (i >= n8) return;
    const int q = (int)(i & 3);
    int64_t r = i >> 2;
    const int col = (int)(r % BNC); r /= BNC;
    const int blk = (int)(r % NBLK); r /= NBLK;
    const int ntn = Cout / BNC;
    const int nt = (int)(r % ntn), chunk = (int)(r / ntn);
    const int cls_of[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3}, sh_of[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0};
    const int cls = cls_of[blk], s = sh_of[blk];
    const int ky = (cls >> 1) + 2 * (s >> 1), kx = (cls & 1) + 2 * (s & 1);
    const int co = nt * BNC + col, ci0 = chunk * KC + q * 8;
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(((size_t)co * Cin + ci0 + e) * 3 + ky) * 3 + kx];
    const bf16x8 h = __builtin_convertvector(v, bf16x8);
    const f32x8 res = v - __builtin_convertvector(h, f32x8);
    const bf16x8 l = __builtin_convertvector(res, bf16x8);
    unsigned short* d = out + (i >> 2) * 64 + q * 8;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + 32) = l;
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

}  // namespace

// out: (Cin/32) * (Cout/32) * 9 * 32 * 64 bf16 = 9 * Cout * Cin * 4 bytes (an opaque buffer only e4s_upconv_bf16x3_f32 reads)
extern "C" int e4s_subpixel_weights_f32(const float* w, void* out, int Cout, int Cin, void* stream) {
    if (Cout % BNC || Cin % KC || Cout <= 0 || Cin <= 0) return (int)hipErrorInvalidValue;
    const int64_t n8 = (int64_t)(Cin / KC) * (Cout / BNC) * NBLK * BNC * 4;
    hipLaunchKernelGGL(subpixel_weights_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, as_stream(stream), w,
                       reinterpret_cast<unsigned short*>(out), Cout, Cin, n8);
    E4S_CHECK_LAUNCH();
    return 0;
}

// p: x NHWC [B,H,W,Cin], w = e4s_subpixel_weights_f32's output, y NHWC [B,2H,2W,Cout (or y_cstride)], in_scale [B,Cin] | null,
// out_scale [B,Cout] | null, noise / noise_w / bias / act / alpha / gain as e4s_conv_bf16x3_f32; k4: DEVICE pointer to the 4x4
// blur kernel (model.py:206-213, already x4).
extern "C" int e4s_upconv_bf16x3_f32(const e4s_conv_params* pp, const float* k4, void* stream) {
    const e4s_conv_params& p = *pp;
    // Cin % 64: the stage pipeline is unrolled by two 32-channel chunks
    if (p.Cin % (2 * KC) || p.Cout % BNC || !k4 || p.labels || p.tiles || p.in_stats || p.noise_per_channel || p.act == 2 ||
        p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi || p.B <= 0 || (p.y_cstride && p.y_cstride % 4))
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    auto kern = p.in_scale ? upconv_fused_kernel<1> : upconv_fused_kernel<0>;
    static std::atomic<uint64_t> smem_set0{0}, smem_set1{0};
    if (int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(kern), SMEM, p.in_scale ? smem_set1 : smem_set0)) return e;
    const int ntn = p.Cout / BNC;
    const int tx_n = (p.Wo + OW - 1) / OW, per_img = ((p.Ho + OH - 1) / OH) * tx_n;
    const int64_t ntiles = (int64_t)p.B * per_img * ntn;
    if (ntiles >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
    int abl = 0;
#ifdef E4S_ABLATIONS
    if (const char* e = getenv("E4S_UPCONV3_ABL")) abl = atoi(e);
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHR), SMEM, st, p, k4, ntn, tx_n, per_img, (int)ntiles, abl);
    E4S_CHECK_LAUNCH();
    return 0;
}
