// Native pieces of the optimisation / training step around the generator backward (SURVEY.md 8(f) N1,
// scripts/optimization.py:209-232): the transposed LocalMLP / style-prologue contractions (x @ W, reducing over W's
// ROW index -- the weights are streamed once, coalesced), the LocalMLP weight-gradient outer products, a fused Adam
// update, and the ordered second stage of every split reduction of the backward (no floating-point atomics anywhere:
// gradients are bit-reproducible run to run).
#include "common.h"

namespace {

constexpr int TB = 16;      // samples (rows of g) whose partial sums a thread keeps in registers

// part[split][b][r][k] = sum_{o in split} g[b,r,o] * w[r,o,k]      (k contiguous in w: lanes walk k, 16 B per lane)
// grid = (R * nsplit, ceil(K / 256)); block = 4 waves, each wave takes every 4th o of the split.
template <int NB>
__global__ void grouped_linear_t_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                        float* __restrict__ part, int B, int R, int O, int K, int nsplit) {
    __shared__ f32x4 red[3][NB][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x / nsplit, split = blockIdx.x - r * nsplit;
    const int k = (blockIdx.y * 64 + lane) * 4;
    const bool live = k < K;
    const int per = (O + nsplit - 1) / nsplit;
    const int o0 = split * per, o1 = min(o0 + per, O);
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wr = w + (int64_t)r * O * K + (live ? k : 0);
    for (int o = o0 + wv; o < o1; o += 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + (int64_t)o * K);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < B) acc[b] += g[((int64_t)b * R + r) * O + o] * w4;      // wave-uniform scalar
    }
    if (wv > 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) red[wv - 1][b][lane] = acc[b];
    }
    __syncthreads();
    if (wv == 0 && live) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < B) {
                const f32x4 v = ((acc[b] + red[0][b][lane]) + red[1][b][lane]) + red[2][b][lane];
                *reinterpret_cast<f32x4*>(part + (((int64_t)split * B + b) * R + r) * K + k) = v;
            }
        }
    }
}

// out[c][i] = sum of the parts of chunk c (RCHUNK consecutive parts, added in order): first level of a long reduction
constexpr int RCHUNK = 64;
__global__ void reduce_chunks_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p0 = blockIdx.y * RCHUNK, p1 = min(p0 + RCHUNK, nparts);
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += part[(int64_t)p * n + i];
    out[(int64_t)blockIdx.y * n + i] = s;
}

// out[i] = base[i] + mul[i] * scale * gate(i) * sum_{s < nparts} part[s * n + i]      (parts added in order)
// gate: ref ? (ref[i] > 0 ? 1 : alpha) : 1  (leaky-ReLU derivative keyed on the saved activation)
// The plain sum (no base / mul / gate), four outputs per thread and eight parts in flight, added in part order like the scalar kernels
// below (the same result bit for bit; n % 4 == 0 and 16-byte aligned buffers).  One 4-byte load per dependent add ran the weight-gradient
// slabs (8 x 9.4 MB) at ~1.8 TB/s; ~300 of these launches per config-5 G step.
template <bool CHUNKS>
__global__ __launch_bounds__(256) void reduce_parts_v4_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int64_t n4,
                                                              float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int p0 = CHUNKS ? blockIdx.y * RCHUNK : 0, p1 = CHUNKS ? min(p0 + RCHUNK, nparts) : nparts;
    const f32x4* src = reinterpret_cast<const f32x4*>(part) + i;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int p = p0; p < p1; p += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p + u < p1 ? src[(int64_t)(p + u) * n4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (p + u < p1) s += v[u];
    }
    if (!CHUNKS) s *= scale;
    reinterpret_cast<f32x4*>(out)[(CHUNKS ? (int64_t)blockIdx.y * n4 : 0) + i] = s;
}

__global__ void reduce_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int64_t n,
                                    float scale, const float* __restrict__ base, const float* __restrict__ mul,
                                    const float* __restrict__ ref, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + i];
    s *= scale;
    if (ref) s *= ref[i] > 0.f ? 1.f : alpha;
    if (mul) s *= mul[i];
    if (base) s += base[i];
    out[i] = s;
}

// dw[r,o,k] = scale * sum_b g[b,r,o] * h[b,r,k]   (LocalMLP weight gradients: B outer products per region)
__global__ void grouped_outer_kernel(const float* __restrict__ g, const float* __restrict__ h, float* __restrict__ dw,
                                     int B, int R, int O, int K, float scale) {
    const int K4 = K / 4;
    const int64_t n = (int64_t)R * O * K4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = (int)(i % K4) * 4;
    const int64_t ro = i / K4;
    const int o = (int)(ro % O), r = (int)(ro / O);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b)
        acc += g[((int64_t)b * R + r) * O + o] * *reinterpret_cast<const f32x4*>(h + ((int64_t)b * R + r) * K + k);
    *reinterpret_cast<f32x4*>(dw + i * 4) = acc * scale;
}

// out[i] = sum_b x[b*n + i]
__global__ void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(int64_t)b * n + i];
    out[i] = s;
}

// part[blk][c] = sum of rows [blk*rpb, (blk+1)*rpb) of x [rows][C] (C % 4 == 0, C <= 1024): thread = (float4 column, row lane);
// every thread adds its rows in order, the row lanes are added in order -> with the ordered e4s_reduce_parts_f32 behind it the
// column sum is bit-reproducible.  (Bias gradients of the Discriminator's FusedLeakyReLUs: rows = B * H * W up to 2 M at 1024^2;
// e4s_batch_sum_f32's one-thread-per-output loop took seconds there.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t rows, int C,
                                                     int rpb) {
    __shared__ f32x4 sm[256];
    const int c4n = C >> 2, lanes = 256 / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(r0 + rpb, rows);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rl < lanes) {
        // eight rows in flight per thread, added in row order (one load per dependent add ran the 2 M-row sums of the Discriminator's 1024^2 layers
        // at 0.64 TB/s: 421 us)
        constexpr int U = 8;
        const float* xp = x + c4 * 4;
        for (int64_t r = r0 + rl; r < r1; r += (int64_t)lanes * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rr = r + (int64_t)u * lanes;
                v[u] = rr < r1 ? *reinterpret_cast<const f32x4*>(xp + rr * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r + (int64_t)u * lanes < r1) acc += v[u];
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        f32x4 s = sm[c4];
        for (int l = 1; l < lanes; ++l) s += sm[l * c4n + c4];
        *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * C + c4 * 4) = s;
    }
}

// dgrad of an unmasked StyledConv on the forward kernels (e4s_amd/autograd.py): u = conv(gz * d, W^T flipped) arrives without the
// style; one pass writes dx = u * s[b] in place and the partial sums of ds[b][c] = sum_p x[b,p,c] * u[b,p,c] (same ordered scheme
// as colsum_kernel: rows of one sample per block, part[blk][b*C + c])
__global__ __launch_bounds__(256) void scale_dot_kernel(float* __restrict__ u, const float* __restrict__ x, const float* __restrict__ s,
                                                        float* __restrict__ part, int64_t hw, int C, int rpb, int B) {
    __shared__ f32x4 sm[256];
    const int c4n = C >> 2, lanes = 256 / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const int b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(r0 + rpb, hw);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rl < lanes) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (int64_t)b * C + c4 * 4);
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const int64_t off = ((int64_t)b * hw + r) * C + c4 * 4;
            const f32x4 uv = *reinterpret_cast<const f32x4*>(u + off);
            acc += uv * *reinterpret_cast<const f32x4*>(x + off);
            *reinterpret_cast<f32x4*>(u + off) = uv * sv;
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        f32x4 t = sm[c4];
        for (int l = 1; l < lanes; ++l) t += sm[l * c4n + c4];
        *reinterpret_cast<f32x4*>(part + ((int64_t)blockIdx.x * B + b) * C + c4 * 4) = t;
    }
}

inline int colsum_rpb(int64_t rows, int C) {
    // ~2048 rows of work per block, but never more than 1024 blocks (e4s_reduce_parts_f32's two ordered levels)
    int64_t rpb = 2048;
    while ((rows + rpb - 1) / rpb > 1024) rpb *= 2;
    (void)C;
    return (int)rpb;
}

// torch.optim.Adam (no amsgrad, maximize=False), one fused pass: the arithmetic of its single-tensor path
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float step_size, float b1, float b2, float eps, float wd,
                            float omb1, float omb2, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = grad[i];
    const float pv = p[i];
    if (wd != 0.f) g += wd * pv;
    const float mi = m[i] + omb1 * (g - m[i]);                  // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + omb2 * g * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pv - step_size * (mi / denom);                       // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// capturable form (a HIP graph replays the same launch every step): the step count lives in device memory and the bias corrections are
// computed from it in double, as the host path does; `lr_dev` (optional) keeps the learning rate in device memory too, so a schedule
// (coach.py:377-381 multiplies it by 0.1 at step 100000) reaches a captured launch
__global__ void advance_i64_kernel(int64_t* steps, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) steps[i] += 1;
}

struct AdamCoef { float step_size, bc2_sqrt, b2, omb1, omb2, eps, wd; };

__device__ __forceinline__ AdamCoef adam_coef(double lr, const double* lr_dev, double beta1, double beta2, float eps, float wd, const int64_t* step) {
    const double t = (double)*step;
    if (lr_dev) lr = *lr_dev;
    AdamCoef c;
    c.step_size = (float)(lr / (1.0 - pow(beta1, t)));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
    c.b2 = (float)beta2, c.omb1 = (float)(1.0 - beta1), c.omb2 = (float)(1.0 - beta2), c.eps = eps, c.wd = wd;
    return c;
}

__device__ __forceinline__ void adam_elem(const AdamCoef& c, float& pv, float g, float& mi, float& vi) {
    if (c.wd != 0.f) g += c.wd * pv;
    mi = mi + c.omb1 * (g - mi);                                // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * c.b2 + c.omb2 * g * g;                            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    pv = pv - c.step_size * (mi / denom);                       // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, double lr, const double* __restrict__ lr_dev, double beta1, double beta2,
                                float eps, float wd, const int64_t* __restrict__ step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AdamCoef c = adam_coef(lr, lr_dev, beta1, beta2, eps, wd, step);
    float pv = p[i], mi = m[i], vi = v[i];
    adam_elem(c, pv, grad[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pv;
}

// multi-tensor form: up to MT tensors per launch, their pointers BY VALUE in the kernel arguments (no device-side table, no
// host-to-device copy: the launch is capturable and costs the eager path one launch per 48 tensors instead of two per tensor -- Net3
// has 344 of them).  Block b owns MT_ELEMS consecutive elements of the tensor whose block range holds b.
constexpr int MT = 48;
constexpr int MT_ELEMS = 4096;
struct AdamChunk {
    float* p[MT];
    const float* g[MT];
    float* m[MT];
    float* v[MT];
    const int64_t* step[MT];
    int64_t n[MT];
    int blk0[MT + 1];
};
struct EmaChunk {
    float* dst[MT];
    const float* src[MT];
    int64_t n[MT];
    int blk0[MT + 1];
};

__device__ __forceinline__ int mt_find(const int* blk0, int cnt, int b) {
    int lo = 0, hi = cnt;                                       // blk0[lo] <= b < blk0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk0[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamChunk c, int cnt, double lr, const double* __restrict__ lr_dev, double beta1,
                                                         double beta2, float eps, float wd) {
    const int t = mt_find(c.blk0, cnt, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - c.blk0[t]) * MT_ELEMS, n = c.n[t];
    float* __restrict__ p = c.p[t];
    const float* __restrict__ g = c.g[t];
    float* __restrict__ m = c.m[t];
    float* __restrict__ v = c.v[t];
    const AdamCoef k = adam_coef(lr, lr_dev, beta1, beta2, eps, wd, c.step[t]);
    const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (al && base + MT_ELEMS <= n) {
#pragma unroll
        for (int j = 0; j < MT_ELEMS / 1024; ++j) {
            const int64_t i = base + j * 1024 + threadIdx.x * 4;
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 mv = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], me = mv[e], ve = vv[e];
                adam_elem(k, pe, gv[e], me, ve);
                pv[e] = pe, mv[e] = me, vv[e] = ve;
            }
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
            *reinterpret_cast<f32x4*>(p + i) = pv;
        }
        return;
    }
    for (int64_t i = base + threadIdx.x; i < min(base + (int64_t)MT_ELEMS, n); i += 256) {
        float pv = p[i], mi = m[i], vi = v[i];
        adam_elem(k, pv, g[i], mi, vi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pv;
    }
}

__global__ __launch_bounds__(256) void ema_multi_kernel(const EmaChunk c, int cnt, float decay, float omd) {
    const int t = mt_find(c.blk0, cnt, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - c.blk0[t]) * MT_ELEMS, n = c.n[t];
    float* __restrict__ dst = c.dst[t];
    const float* __restrict__ src = c.src[t];
    const bool al = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
    if (al && base + MT_ELEMS <= n) {
#pragma unroll
        for (int j = 0; j < MT_ELEMS / 1024; ++j) {
            const int64_t i = base + j * 1024 + threadIdx.x * 4;
            f32x4 d = *reinterpret_cast<const f32x4*>(dst + i);
            const f32x4 s = *reinterpret_cast<const f32x4*>(src + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __fmaf_rn(s[e], omd, d[e] * decay);
            *reinterpret_cast<f32x4*>(dst + i) = d;
        }
        return;
    }
    for (int64_t i = base + threadIdx.x; i < min(base + (int64_t)MT_ELEMS, n); i += 256) dst[i] = __fmaf_rn(src[i], omd, dst[i] * decay);
}

// exponential moving average of the weights (src/utils/torch_utils.py:189-194, coach.py:396-398): dst = dst*decay + src*(1-decay),
// in the operation order of `par1.data.mul_(decay).add_(par2.data, alpha=1 - decay)` (two roundings)
__global__ void ema_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, float decay, float omd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = dst[i] * decay;
    dst[i] = __fmaf_rn(src[i], omd, a);
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

int linear_t_nsplit(int R, int O, int K) {
    const int kblk = (K + 255) / 256;
    int ns = 2048 / (R * kblk > 0 ? R * kblk : 1);
    if (ns < 1) ns = 1;
    if (ns > (O + 15) / 16) ns = (O + 15) / 16;
    return ns < 1 ? 1 : ns;
}


// ---- style-gradient tail of the generator backward, all layers in two launches (round 5) ---------------------------------------------
// Every StyledConv / ToRGB of the generator backward ends in the same small chain (autograd.GeneratorFn.backward): dL/ds of the
// contraction, minus the path through the demodulation coefficients (model.py:276-285: d = scale * rsqrt(scale^2 sum_ci s^2 Wsq + eps)),
// then through the modulation EqualLinear into the layer's latent slot.  Per layer that was 2 transposed contractions with their ordered
// second stages and 3-5 ATen glue launches -- ~170 launches of 5-10 us in an l2-only latent-optimisation step of 472.  The jobs travel BY VALUE
// in the kernel arguments (a HIP-graph capture bakes them in; the addresses of a captured step are stable under replay).
constexpr int SG_U = 8;          // weight rows in flight per wave in both stages
constexpr int SG_CL = 16, SG_NG = 256 / SG_CL;      // a block = 16 lanes x 4 output columns, 16 groups of lanes over the contraction
constexpr int SG_CH = 512;       // channels of coefficients staged in LDS per pass (TB x SG_CH floats inside the 60 KB reduction buffer)
static_assert(TB * SG_CH * 4 <= (SG_NG - 1) * TB * SG_CL * 16 && (SG_NG - 1) * TB * SG_CL * 16 <= 64 * 1024, "staging aliases the reduction buffer");
struct StyleGradJobs {
    e4s_style_grad_job j[E4S_STYLE_GRAD_MAX_JOBS];
    int blk0[E4S_STYLE_GRAD_MAX_JOBS + 1];      // stage 1: first block of job i (64 input channels per block)
    int n;
};

// stage 1: ds_total[g][ci] = ds_raw[g][ci] - s[g][ci] * sum_co (dd_d[g][co] d[g][co]^2) Wsq[co][ci]          (StyledConv)
//          ds_total[g][ci] = conv_scale * ((dws[g][0][ci] w3[0][ci] + dws[g][1][ci] w3[1][ci]) + dws[g][2][ci] w3[2][ci])   (ToRGB)
// block = (job, 64 input channels): 16 lanes x 4 channels, SG_NG = 16 groups of lanes walk Cout (group w takes rows w, w + 16, ...), 16 rows g per
// pass, cross-group sum in order (round 6: 64- instead of 256-channel blocks -- ~200 blocks instead of ~50 for a latency-bound loop)
__global__ __launch_bounds__(256) void style_grad_stage1_kernel(const StyleGradJobs J) {
    __shared__ f32x4 red[SG_NG - 1][TB][SG_CL];
    const int lane = threadIdx.x & (SG_CL - 1), wv = threadIdx.x / SG_CL;
    int ji = 0;
    while (ji + 1 < J.n && (int)blockIdx.x >= J.blk0[ji + 1]) ++ji;
    const e4s_style_grad_job& q = J.j[ji];
    const int k = (((int)blockIdx.x - J.blk0[ji]) * SG_CL + lane) * 4;
    const bool live = k < q.Cin;
    if (q.dws) {                                             // ToRGB: elementwise
        if (wv == 0 && live)
            for (int g = 0; g < q.G; ++g) {
                const float* dw = q.dws + (int64_t)g * 3 * q.Cin + k;
                const f32x4 a = *reinterpret_cast<const f32x4*>(dw) * *reinterpret_cast<const f32x4*>(q.w3 + k);
                const f32x4 b = *reinterpret_cast<const f32x4*>(dw + q.Cin) * *reinterpret_cast<const f32x4*>(q.w3 + q.Cin + k);
                const f32x4 c = *reinterpret_cast<const f32x4*>(dw + 2 * q.Cin) * *reinterpret_cast<const f32x4*>(q.w3 + 2 * q.Cin + k);
                *reinterpret_cast<f32x4*>(q.ds_total + (int64_t)g * q.Cin + k) = ((a + b) + c) * q.conv_scale;
            }
        return;
    }
    const float* wr = q.wsq + (live ? k : 0);
    float* sc = reinterpret_cast<float*>(&red[0][0][0]);     // [TB][SG_CH] staged coefficients; the same LDS serves the cross-wave sum afterwards
    for (int g0 = 0; g0 < q.G; g0 += TB) {
        const int nb = min(TB, q.G - g0);
        f32x4 acc[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The coefficients dd d^2 of a chunk of output channels go through LDS (coalesced loads, then broadcast reads) and SG_U weight rows are
        // in flight per wave: the loop used to be one exposed round trip per output channel (wave-uniform global loads inside it): 261 us
        // for ~50 blocks.  Products and the order of the additions are unchanged.
        for (int c0 = 0; c0 < q.Cout; c0 += SG_CH) {
            const int cw = min(SG_CH, q.Cout - c0);
            __syncthreads();                                 // previous chunk / pass consumed
            for (int i = threadIdx.x; i < nb * cw; i += 256) {
                const int b = i / cw, c = i - b * cw;
                const float dd = q.dd_d[(int64_t)(g0 + b) * q.Cout + c0 + c], dv = q.d[(int64_t)(g0 + b) * q.Cout + c0 + c];
                sc[b * SG_CH + c] = dd * (dv * dv);
            }
            __syncthreads();
            for (int cb = wv; cb < cw; cb += SG_NG * SG_U) {
                f32x4 w4[SG_U];
#pragma unroll
                for (int u = 0; u < SG_U; ++u) {
                    const int c = cb + SG_NG * u;
                    w4[u] = c < cw ? *reinterpret_cast<const f32x4*>(wr + (int64_t)(c0 + c) * q.Cin) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < SG_U; ++u) {
                    const int c = cb + SG_NG * u;
                    if (c < cw) {
#pragma unroll
                        for (int b = 0; b < TB; ++b)
                            if (b < nb) acc[b] += sc[b * SG_CH + c] * w4[u];
                    }
                }
            }
        }
        __syncthreads();                                     // (the staging buffer becomes the reduction buffer)
        if (wv > 0) {
#pragma unroll
            for (int b = 0; b < TB; ++b) red[wv - 1][b][lane] = acc[b];
        }
        __syncthreads();
        if (wv == 0 && live) {
#pragma unroll
            for (int b = 0; b < TB; ++b)
                if (b < nb) {
                    f32x4 v = acc[b];
#pragma unroll
                    for (int w = 0; w < SG_NG - 1; ++w) v += red[w][b][lane];
                    const int64_t o = (int64_t)(g0 + b) * q.Cin + k;
                    *reinterpret_cast<f32x4*>(q.ds_total + o) =
                        *reinterpret_cast<const f32x4*>(q.ds_raw + o) - *reinterpret_cast<const f32x4*>(q.s + o) * v;
                }
        }
    }
}

// stage 2: dlat[b][r][slot][:] = sum over the jobs of that slot, in job order, of mod_scale * ds_total[g] @ Wmod   (Wmod [Cin][S]).
// A masked job's row g = b * R + r feeds dlat[b][r]; an unmasked job's row g = b feeds dlat[b][0].  Every (row, slot) is written, zeros
// included.  grid = (NL * ceil(S / 64), ceil(B * R / 16)); 16 lanes x 4 columns, the SG_NG = 16 lane groups walk Cin, cross-group sum in order.
__global__ __launch_bounds__(256) void style_grad_stage2_kernel(const StyleGradJobs J, float* __restrict__ dlat, const int BR,
                                                                const int R, const int NL, const int S) {
    __shared__ f32x4 red[SG_NG - 1][TB][SG_CL];
    const int lane = threadIdx.x & (SG_CL - 1), wv = threadIdx.x / SG_CL;
    const int kblk = (S + 4 * SG_CL - 1) / (4 * SG_CL);
    const int slot = blockIdx.x / kblk;
    const int k = ((blockIdx.x - slot * kblk) * SG_CL + lane) * 4;
    const bool live = k < S;
    const int r0 = blockIdx.y * TB;                          // first dlat row (b * R + r) of this block
    f32x4 acc[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* sc = reinterpret_cast<float*>(&red[0][0][0]);     // [TB][SG_CH] staged rows of ds_total * mod_scale; the reduction buffer afterwards
    for (int ji = 0; ji < J.n; ++ji) {
        const e4s_style_grad_job& q = J.j[ji];
        if (q.slot != slot) continue;
        const float* wr = q.wmod + (live ? k : 0);
        unsigned use = 0;                                    // rows of this block the job feeds
#pragma unroll
        for (int b = 0; b < TB; ++b)
            if (r0 + b < BR && (q.masked || (r0 + b) % R == 0)) use |= 1u << b;
        // ds_total rows through LDS, SG_U rows of Wmod in flight per wave (the loop was one exposed round trip per input channel: 560 us for
        // 36 blocks); products and the order of the additions are unchanged
        for (int c0 = 0; c0 < q.Cin; c0 += SG_CH) {
            const int cw = min(SG_CH, q.Cin - c0);
            __syncthreads();
            for (int i = threadIdx.x; i < TB * cw; i += 256) {
                const int b = i / cw, c = i - b * cw;
                const int row = r0 + b;
                float v = 0.f;
                if ((use >> b) & 1u) v = q.ds_total[(int64_t)(q.masked ? row : row / R) * q.Cin + c0 + c] * q.mod_scale;
                sc[b * SG_CH + c] = v;
            }
            __syncthreads();
            for (int cb = wv; cb < cw; cb += SG_NG * SG_U) {
                f32x4 w4[SG_U];
#pragma unroll
                for (int u = 0; u < SG_U; ++u) {
                    const int c = cb + SG_NG * u;
                    w4[u] = c < cw ? *reinterpret_cast<const f32x4*>(wr + (int64_t)(c0 + c) * S) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < SG_U; ++u) {
                    const int c = cb + SG_NG * u;
                    if (c < cw) {
#pragma unroll
                        for (int b = 0; b < TB; ++b)
                            if ((use >> b) & 1u) acc[b] += sc[b * SG_CH + c] * w4[u];
                    }
                }
            }
        }
    }
    __syncthreads();                                         // (the staging buffer becomes the reduction buffer)
    if (wv > 0) {
#pragma unroll
        for (int b = 0; b < TB; ++b) red[wv - 1][b][lane] = acc[b];
    }
    __syncthreads();
    if (wv == 0 && live) {
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int row = r0 + b;
            if (row < BR) {
                f32x4 v = acc[b];
#pragma unroll
                for (int w = 0; w < SG_NG - 1; ++w) v += red[w][b][lane];
                *reinterpret_cast<f32x4*>(dlat + ((int64_t)row * NL + slot) * S + k) = v;
            }
        }
    }
}

}  // namespace

extern "C" int64_t e4s_grouped_linear_t_ws_floats(int B, int R, int O, int K) {
    return (int64_t)linear_t_nsplit(R, O, K) * B * R * K;
}

extern "C" int e4s_grouped_linear_t_f32(const float* g, const float* w, float* out, float* ws, int B, int R, int O, int K,
                                        float scale, const float* base, const float* mul, const float* ref, float alpha,
                                        void* stream) {
    if (B < 1 || B > TB || K % 4 || R < 1 || O < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int ns = linear_t_nsplit(R, O, K);
    const dim3 grid(R * ns, (K + 255) / 256);
    if (B <= 1) hipLaunchKernelGGL(grouped_linear_t_kernel<1>, grid, dim3(256), 0, st, g, w, ws, B, R, O, K, ns);
    else if (B <= 4) hipLaunchKernelGGL(grouped_linear_t_kernel<4>, grid, dim3(256), 0, st, g, w, ws, B, R, O, K, ns);
    else hipLaunchKernelGGL(grouped_linear_t_kernel<TB>, grid, dim3(256), 0, st, g, w, ws, B, R, O, K, ns);
    E4S_CHECK_LAUNCH();
    const int64_t n = (int64_t)B * R * K;
    hipLaunchKernelGGL(reduce_parts_kernel, grid1(n), dim3(256), 0, st, ws, out, ns, n, scale, base, mul, ref, alpha);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_reduce_parts_ws_floats(int nparts, int64_t n) {
    // the parts themselves + one level of chunk sums behind them (nparts > RCHUNK^2 would need a third level: refused)
    return (int64_t)nparts * n + (int64_t)((nparts + RCHUNK - 1) / RCHUNK) * n;
}

extern "C" int e4s_reduce_parts_f32(float* parts, float* out, int nparts, int64_t n, float scale, void* stream) {
    if (n <= 0) return 0;
    if (nparts > RCHUNK * RCHUNK * 4) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const float* src = parts;
    const bool v4 = n % 4 == 0 && ((uintptr_t)parts & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (nparts > RCHUNK) {      // two levels, both in part order: chunk sums first (parallel over chunks), then the chunks
        const int nch = (nparts + RCHUNK - 1) / RCHUNK;
        float* lvl = parts + (int64_t)nparts * n;
        if (v4) hipLaunchKernelGGL(reduce_parts_v4_kernel<true>, dim3((unsigned)((n / 4 + 255) / 256), nch), dim3(256), 0, st, parts, lvl, nparts, n / 4, 1.f);
        else hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256), nch), dim3(256), 0, st, parts, lvl,
                           nparts, n);
        E4S_CHECK_LAUNCH();
        src = lvl;
        nparts = nch;
    }
    if (v4) {
        hipLaunchKernelGGL(reduce_parts_v4_kernel<false>, grid1(n / 4), dim3(256), 0, st, src, out, nparts, n / 4, scale);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(reduce_parts_kernel, grid1(n), dim3(256), 0, st, src, out, nparts, n, scale,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_grouped_outer_f32(const float* g, const float* h, float* dw, int B, int R, int O, int K, float scale,
                                     void* stream) {
    if (K % 4) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)R * O * (K / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(grouped_outer_kernel, grid1(n), dim3(256), 0, as_stream(stream), g, h, dw, B, R, O, K, scale);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_batch_sum_f32(const float* x, float* out, int B, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(batch_sum_kernel, grid1(n), dim3(256), 0, as_stream(stream), x, out, B, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_colsum_ws_floats(int64_t rows, int C) {
    const int rpb = colsum_rpb(rows, C);
    return e4s_reduce_parts_ws_floats((int)((rows + rpb - 1) / rpb), C);
}

extern "C" int e4s_colsum_f32(const float* x, float* out, float* ws, int64_t rows, int C, void* stream) {
    if (C % 4 || C > 1024 || C <= 0 || !ws) return (int)hipErrorInvalidValue;
    if (rows <= 0) return (int)hipErrorInvalidValue;
    const int rpb = colsum_rpb(rows, C);
    const int nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), x, ws, rows, C, rpb);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, out, nblk, C, 1.f, stream);
}

extern "C" int64_t e4s_scale_dot_ws_floats(int B, int64_t hw, int C) {
    const int rpb = colsum_rpb(hw, C);
    return e4s_reduce_parts_ws_floats((int)((hw + rpb - 1) / rpb), (int64_t)B * C);
}

extern "C" int e4s_scale_dot_f32(float* u, const float* x, const float* s, float* ds, float* ws, int B, int64_t hw, int C,
                                 void* stream) {
    if (C % 4 || C > 1024 || C <= 0 || B <= 0 || B > 65535 || hw <= 0 || !ws) return (int)hipErrorInvalidValue;
    const int rpb = colsum_rpb(hw, C);
    const int nblk = (int)((hw + rpb - 1) / rpb);
    hipLaunchKernelGGL(scale_dot_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, as_stream(stream), u, x, s, ws, hw, C, rpb, B);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, ds, nblk, (int64_t)B * C, 1.f, stream);
}

extern "C" int e4s_adam_step_f32(float* p, const float* grad, float* m, float* v, int64_t n, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, int step, void* stream) {
    if (step < 1) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    // bias corrections in double on the host, as torch.optim.Adam computes them (python floats)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, grid1(n), dim3(256), 0, as_stream(stream), p, grad, m, v, n, (float)(lr / bc1),
                       (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)bc2_sqrt);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_ema_f32(float* dst, const float* src, int64_t n, double decay, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ema_kernel, grid1(n), dim3(256), 0, as_stream(stream), dst, src, n, (float)decay, (float)(1.0 - decay));
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_advance_i64(int64_t* steps, int64_t n, void* stream) {
    if (!steps) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(advance_i64_kernel, grid1(n), dim3(256), 0, as_stream(stream), steps, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_adam_step_dev_f32(float* p, const float* grad, float* m, float* v, int64_t n, double lr, const double* lr_dev, double beta1,
                                     double beta2, double eps, double weight_decay, const int64_t* step, void* stream) {
    if (!step) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_dev_kernel, grid1(n), dim3(256), 0, as_stream(stream), p, grad, m, v, n, lr, lr_dev, beta1, beta2, (float)eps,
                       (float)weight_decay, step);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_adam_multi_dev_f32(int count, float* const* p, const float* const* grad, float* const* m, float* const* v, const int64_t* n,
                                      const int64_t* const* step, double lr, const double* lr_dev, double beta1, double beta2, double eps,
                                      double weight_decay, void* stream) {
    if (count < 0 || (count && (!p || !grad || !m || !v || !n || !step))) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < count) {
        AdamChunk c;
        int cnt = 0;
        int64_t blocks = 0;
        c.blk0[0] = 0;
        while (i < count && cnt < MT) {
            if (n[i] > 0) {
                if (!p[i] || !grad[i] || !m[i] || !v[i] || !step[i]) return (int)hipErrorInvalidValue;
                const int64_t nb = (n[i] + MT_ELEMS - 1) / MT_ELEMS;
                if (blocks + nb > 0x7fffffff) { if (cnt) break; return (int)hipErrorInvalidValue; }
                c.p[cnt] = p[i], c.g[cnt] = grad[i], c.m[cnt] = m[i], c.v[cnt] = v[i], c.step[cnt] = step[i], c.n[cnt] = n[i];
                blocks += nb;
                c.blk0[++cnt] = (int)blocks;
            }
            ++i;
        }
        if (!cnt) continue;
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c, cnt, lr, lr_dev, beta1, beta2, (float)eps,
                           (float)weight_decay);
        E4S_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int e4s_ema_multi_f32(int count, float* const* dst, const float* const* src, const int64_t* n, double decay, void* stream) {
    if (count < 0 || (count && (!dst || !src || !n))) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < count) {
        EmaChunk c;
        int cnt = 0;
        int64_t blocks = 0;
        c.blk0[0] = 0;
        while (i < count && cnt < MT) {
            if (n[i] > 0) {
                if (!dst[i] || !src[i]) return (int)hipErrorInvalidValue;
                const int64_t nb = (n[i] + MT_ELEMS - 1) / MT_ELEMS;
                if (blocks + nb > 0x7fffffff) { if (cnt) break; return (int)hipErrorInvalidValue; }
                c.dst[cnt] = dst[i], c.src[cnt] = src[i], c.n[cnt] = n[i];
                blocks += nb;
                c.blk0[++cnt] = (int)blocks;
            }
            ++i;
        }
        if (!cnt) continue;
        hipLaunchKernelGGL(ema_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c, cnt, (float)decay, (float)(1.0 - decay));
        E4S_CHECK_LAUNCH();
    }
    return 0;
}

/* The style-gradient tail of the generator backward for every layer at once (see the kernels above): jobs[i].ds_total <- dL/ds of layer
 * i including the demodulation path; dlat [B][R][NL][S] <- every layer's dL/dstyle in its latent slot (all of dlat is written). */
extern "C" int e4s_style_grad_multi_f32(const e4s_style_grad_job* jobs, int njobs, float* dlat, int B, int R, int NL, int S,
                                        void* stream) {
    if (!jobs || !dlat || njobs < 0 || njobs > E4S_STYLE_GRAD_MAX_JOBS || B < 1 || R < 1 || NL < 1 || S < 4 || S % 4)
        return (int)hipErrorInvalidValue;
    StyleGradJobs J;
    J.n = njobs;
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const e4s_style_grad_job& q = jobs[i];
        const bool rgb = q.dws != nullptr;
        if (!q.ds_total || !q.wmod || q.G < 1 || q.Cin < 4 || q.Cin % 4 || q.slot < 0 || q.slot >= NL ||
            q.G != (q.masked ? B * R : B) || (rgb ? !q.w3 : (!q.ds_raw || !q.dd_d || !q.d || !q.s || !q.wsq || q.Cout < 1)))
            return (int)hipErrorInvalidValue;
        J.j[i] = q;
        J.blk0[i] = blocks;
        blocks += (q.Cin + 4 * SG_CL - 1) / (4 * SG_CL);
    }
    J.blk0[njobs] = blocks;
    hipStream_t st = as_stream(stream);
    if (blocks > 0) {
        hipLaunchKernelGGL(style_grad_stage1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, J);
        E4S_CHECK_LAUNCH();
    }
    const int BR = B * R;
    hipLaunchKernelGGL(style_grad_stage2_kernel, dim3((unsigned)(NL * ((S + 4 * SG_CL - 1) / (4 * SG_CL))), (unsigned)((BR + TB - 1) / TB)), dim3(256), 0,
                       st, J, dlat, BR, R, NL, S);
    E4S_CHECK_LAUNCH();
    return 0;
}
