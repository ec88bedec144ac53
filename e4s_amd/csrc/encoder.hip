// Regional style encoder support kernels (src/models/encoders/psp_encoders.py:238-309,
// src/models/encoders/helpers.py:56-72,122-144, src/models/networks.py:15-39,131).
// The 3x3 / 1x1 convolutions themselves run on e4s_conv_mfma_f32; everything here is HBM- or
// latency-bound glue on NHWC tensors: resize, the 3->64 stem conv, InstanceNorm statistics and
// apply(+gate, +residual, +PReLU), the SE gate, regional average pooling and the LocalMLPs.
#include "common.h"

namespace {

// ---- bilinear resize, align_corners=False, no antialias; NCHW in -> NHWC out -------------------
__global__ void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int Hi,
                                       int Wi, int Ho, int Wo) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const float sy = (float)Hi / Ho, sx = (float)Wi / Wo;
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = x + ((int64_t)b * C + c) * Hi * Wi;
    y[i] = hy * (hx * p[(int64_t)y0 * Wi + x0] + lx * p[(int64_t)y0 * Wi + x1]) +
           ly * (hx * p[(int64_t)y1 * Wi + x0] + lx * p[(int64_t)y1 * Wi + x1]);
}

// ---- stem conv: 3x3, pad 1, tiny Cin (3) -> Cout.  Thread = (pixel, 4 consecutive couts): 16 lanes write
// one pixel's 64 channels as a contiguous 256 B, weights [tap][ci][cout] broadcast-read from LDS -------------
__global__ void conv3x3_small_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                     int B, int H, int W, int Cin, int Cout) {
    extern __shared__ float sw[];   // [9][Cin][Cout]
    const int nw = Cout * Cin * 9;
    for (int t = threadIdx.x; t < nw; t += blockDim.x) {
        const int k = t % 9, ci = (t / 9) % Cin, co = t / (9 * Cin);        // w is [Cout][Cin][3][3]
        sw[(k * Cin + ci) * Cout + co] = w[t];
    }
    __syncthreads();
    const int cg = Cout / 4;
    const int64_t n = (int64_t)B * H * W * cg;
    // grid-stride: the block keeps its LDS copy of the weights for many pixels (staging them per 16 pixels cost more than
    // the 16 pixels' arithmetic)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cg) * 4;
        int64_t r = i / cg;
        const int ox = (int)(r % W); r /= W;
        const int oy = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float* xp = x + (((int64_t)b * H + iy) * W + ix) * Cin;
                const float* wp = sw + ((ky * 3 + kx) * Cin) * Cout + c4;
                for (int ci = 0; ci < Cin; ++ci) acc += xp[ci] * *reinterpret_cast<const f32x4*>(wp + ci * Cout);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = acc;
    }
}

// ---- stem conv, tiled form (round 4): Cin = 3 -> Cout = 64 on 16x16-pixel tiles.  The 18x18x3 halo sits in LDS as float4 pixels, each
// thread keeps the 27 x 4 weights of its 4 output channels in REGISTERS and walks one tile row: per pixel 9 broadcast ds_read_b128 +
// 108 FMAs + one 16-byte store (the 16 lanes of a pixel write its 64 channels as 256 contiguous bytes).  stats_ws != NULL: per-(sample,
// channel, tile) fp64 {sum, sum of squares} of the output in the slot layout of the conv kernels' fused statistics
// (e4s_instnorm_finalize_f32 adds the slots in order) -- the separate statistics pass over the 268 MB stem output of a batch of 16
// disappears.  The grid-stride kernel above (weights broadcast from LDS, x from global, 27 + 27 loads per 4 outputs) took 0.25 ms for
// that batch, 4x its store time.
__global__ __launch_bounds__(256) void conv3x3_stem_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           double* __restrict__ stats_ws, int H, int W, int tx_n, int per_img) {
    __shared__ f32x4 sx[18 * 18];
    __shared__ double sred[2][16][64];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / per_img, rem = blockIdx.x - b * per_img;
    const int tyb = rem / tx_n, txb = rem - tyb * tx_n;
    const int y0 = tyb * 16, x0 = txb * 16;
    for (int h = tid; h < 324; h += 256) {
        const int hy = h / 18, hx = h - hy * 18;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const float* xp = x + (((int64_t)b * H + iy) * W + ix) * 3;
            v = f32x4{xp[0], xp[1], xp[2], 0.f};
        }
        sx[h] = v;
    }
    const int g = tid & 15, py = tid >> 4;               // 4 output channels 4g .. 4g+3, tile row py
    f32x4 wr[27];                                        // wr[(ky*3+kx)*3 + ci][j] = w[4g + j][ci][ky][kx]   (w is [64][3][3][3])
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[k * 3 + ci][j] = w[((4 * g + j) * 3 + ci) * 9 + k];
    __syncthreads();
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    float* yrow = y + (((int64_t)b * H + y0 + py) * W + x0) * 64 + 4 * g;
#pragma unroll 4
    for (int px = 0; px < 16; ++px) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 v = sx[(py + ky) * 18 + px + kx];
                const int k = (ky * 3 + kx) * 3;
                acc += v[0] * wr[k] + v[1] * wr[k + 1] + v[2] * wr[k + 2];
            }
        *reinterpret_cast<f32x4*>(yrow + (int64_t)px * 64) = acc;
        if (stats_ws) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += (double)acc[j];
                q[j] += (double)acc[j] * (double)acc[j];
            }
        }
    }
    if (stats_ws) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sred[0][py][4 * g + j] = s[j];
            sred[1][py][4 * g + j] = q[j];
        }
        __syncthreads();
        if (tid < 64) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {               // tile rows in order: the statistics do not depend on thread timing
                a += sred[0][r][tid];
                c += sred[1][r][tid];
            }
            double* slot = stats_ws + (((int64_t)b * 64 + tid) * per_img + rem) * 2;
            slot[0] = a;
            slot[1] = c;
        }
    }
}

// ---- InstanceNorm statistics: single pass, fp64 partial sums (no cancellation in E[x^2]-mean^2),
// pixels split over blockIdx.z so the 64-channel early layers still fill the chip ----------------
__global__ void instnorm_partial_kernel(const float* __restrict__ x, double* __restrict__ ws, int HW, int C, int nsplit) {
    const int slabs = C / 64;
    const int split = blockIdx.x % nsplit;
    const int slab = (blockIdx.x / nsplit) % slabs;
    const int b = blockIdx.x / (nsplit * slabs);
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;       // 4 pixel groups x 64 channels
    const int c = slab * 64 + cl;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const float* xb = x + (int64_t)b * HW * C + c;
    double s = 0.0, q = 0.0;
    for (int p = p0 + pg; p < p1; p += 4) {
        const double v = (double)xb[(int64_t)p * C];
        s += v;
        q += v * v;
    }
    __shared__ double red[2][4][64];
    red[0][pg][cl] = s;
    red[1][pg][cl] = q;
    __syncthreads();
    if (pg == 0) {
        s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        q = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
        // one slot per (b, c, split): the finalize pass adds the slots in split order, so the statistics (and with them
        // the whole encoder) are bit-reproducible run to run -- no floating-point atomics
        double* slot = ws + (((int64_t)b * C + c) * nsplit + split) * 2;
        slot[0] = s;
        slot[1] = q;
    }
}

// Small maps (HW <= 1024: the 32^2 and 16^2 levels, where a split-K conv cannot emit the statistics itself): ONE launch, block =
// (sample, 64-channel slab), 16 pixel groups x 64 channels; fp64 partial sums added in a fixed order, mean / rstd / pooled
// written directly (the same formulas as instnorm_finalize_kernel).  Batch-1 latency runs paid two launches per statistic.
__global__ __launch_bounds__(1024) void instnorm_small_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                              float* __restrict__ pooled, int HW, int C, float eps) {
    const int slabs = C / 64;
    const int slab = blockIdx.x % slabs, b = blockIdx.x / slabs;
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;       // 16 pixel groups x 64 channels
    const int c = slab * 64 + cl;
    const float* xb = x + (int64_t)b * HW * C + c;
    double s = 0.0, q = 0.0;
    int p = pg;
    for (; p + 7 * 16 < HW; p += 8 * 16) {           // eight loads in flight; the sums keep their order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)(p + 16 * u) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += (double)v[u];
            q += (double)v[u] * (double)v[u];
        }
    }
    for (; p < HW; p += 16) {
        const double v = (double)xb[(int64_t)p * C];
        s += v;
        q += v * v;
    }
    __shared__ double red[2][16][64];
    red[0][pg][cl] = s;
    red[1][pg][cl] = q;
    __syncthreads();
    if (pg == 0) {
        s = 0.0;
        q = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
        const double mean = s / HW;
        double var = q / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean;
        const float rstd = rsqrtf((float)var + eps);
        const int64_t i = (int64_t)b * C + c;
        stats[i * 2] = mf;
        stats[i * 2 + 1] = rstd;
        if (pooled) pooled[i] = (float)((mean - (double)mf) * (double)rstd);
    }
}

__global__ void instnorm_finalize_kernel(const double* __restrict__ ws, float* __restrict__ stats,
                                         float* __restrict__ pooled, int n, int HW, float eps, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int k = 0; k < nsplit; ++k) {
        s += ws[((int64_t)i * nsplit + k) * 2];
        q += ws[((int64_t)i * nsplit + k) * 2 + 1];
    }
    const double mean = s / HW;
    double var = q / HW - mean * mean;                  // biased variance (InstanceNorm2d)
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean;
    const float rstd = rsqrtf((float)var + eps);
    stats[i * 2] = mf;
    stats[i * 2 + 1] = rstd;
    // AdaptiveAvgPool2d(1) of the normalised tensor: (mean(x) - mean) * rstd, i.e. the rounding
    // residue of the mean -- what SEModule actually sees after an InstanceNorm (helpers.py:64-66)
    if (pooled) pooled[i] = (float)((mean - (double)mf) * (double)rstd);
}

__global__ void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                      const float* __restrict__ gate, const float* __restrict__ res,
                                      const float* __restrict__ res_stats, const float* __restrict__ slope,
                                      float* __restrict__ y, int B, int H, int W, int C, int rs) {
    const int C4 = C / 4;
    const int64_t n = (int64_t)B * H * W * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    int64_t r = i / C4;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* st = stats + ((int64_t)b * C + c + e) * 2;
        o[e] = (v[e] - st[0]) * st[1];
        if (gate) o[e] *= gate[(int64_t)b * C + c + e];
    }
    if (res) {
        const int Hr = H * rs, Wr = W * rs;
        const f32x4 rv = *reinterpret_cast<const f32x4*>(res + (((int64_t)b * Hr + (int64_t)yy * rs) * Wr + (int64_t)xx * rs) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = rv[e];
            if (res_stats) {
                const float* st = res_stats + ((int64_t)b * C + c + e) * 2;
                t = (t - st[0]) * st[1];
            }
            o[e] += t;
        }
    }
    if (slope) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * slope[c + e];
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

// The same apply, organised like the statistics pass (block = sample x 64-channel slab x pixel range; lanes walk the
// channels, 4 pixel groups): it ALSO accumulates the fp64 partial sums of its output, i.e. the next unit's InstanceNorm
// statistics, so that pass never re-reads the tensor.
__global__ void instnorm_apply_stats_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                            const float* __restrict__ gate, const float* __restrict__ res,
                                            const float* __restrict__ res_stats, const float* __restrict__ slope,
                                            float* __restrict__ y, double* __restrict__ ws, int H, int W, int C, int rs,
                                            int nsplit) {
    // 16 lanes x float4 walk the 64 channels of the slab, 16 pixel groups: a wave instruction moves 4 pixels x 256 B
    const int slabs = C / 64, HW = H * W;
    const int split = blockIdx.x % nsplit;
    const int slab = (blockIdx.x / nsplit) % slabs;
    const int b = blockIdx.x / (nsplit * slabs);
    const int cl = (threadIdx.x & 15) * 4, pg = threadIdx.x >> 4;
    const int c = slab * 64 + cl;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const int64_t bc = (int64_t)b * C + c;
    f32x4 mean, rstd, g, rmean, rrstd, sl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mean[e] = stats[(bc + e) * 2];
        rstd[e] = stats[(bc + e) * 2 + 1];
        g[e] = gate ? gate[bc + e] : 1.f;
        rmean[e] = res_stats ? res_stats[(bc + e) * 2] : 0.f;
        rrstd[e] = res_stats ? res_stats[(bc + e) * 2 + 1] : 1.f;
        sl[e] = slope ? slope[c + e] : 0.f;
    }
    const int Wr = W * rs;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = p0 + pg; p < p1; p += 16) {
        const int64_t idx = ((int64_t)b * HW + p) * C + c;
        f32x4 o = (*reinterpret_cast<const f32x4*>(x + idx) - mean) * rstd;
        if (gate) o *= g;
        if (res) {
            const int yy = p / W, xx = p - yy * W;
            f32x4 t = *reinterpret_cast<const f32x4*>(res + (((int64_t)b * H * rs + (int64_t)yy * rs) * Wr + (int64_t)xx * rs) * C + c);
            if (res_stats) t = (t - rmean) * rrstd;
            o += t;
        }
        if (slope) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * sl[e];
        }
        *reinterpret_cast<f32x4*>(y + idx) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] += (double)o[e];
            q[e] += (double)o[e] * (double)o[e];
        }
    }
    __shared__ double red[2][16][64];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][pg][cl + e] = s[e];
        red[1][pg][cl + e] = q[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double a = 0.0, d = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {                 // pixel groups added in order
            a += red[0][k][threadIdx.x];
            d += red[1][k][threadIdx.x];
        }
        double* slot = ws + (((int64_t)b * C + slab * 64 + threadIdx.x) * nsplit + split) * 2;
        slot[0] = a;
        slot[1] = d;
    }
}

// ---- SE gate (helpers.py:56-72): fc1 -> relu -> fc2 -> sigmoid on the pooled vector; one block (512 threads) per sample.
// fc1: the Cr rows are spread over ALL threads -- row j on threads [j*tpr, (j+1)*tpr), tpr = min(64, 512 / Cr) lanes of one wave, each
// lane a strided share of the C products, combined by a butterfly over the tpr lanes (the first version walked the rows of a wave
// one after the other: 4 dependent global round trips at Cr = 32, 14 us per launch against ~5 for the launch itself).
__device__ __forceinline__ void se_body(const float* pooled, float* hidden, const float* __restrict__ fc1,
                                        const float* __restrict__ fc2, float* __restrict__ gate, int C, int Cr) {
    int tpr = 64;
    while (tpr * Cr > (int)blockDim.x && tpr > 1) tpr >>= 1;
    for (int j0 = 0; j0 < Cr; j0 += blockDim.x / tpr) {
        const int j = j0 + threadIdx.x / tpr, l = threadIdx.x % tpr;
        float a = 0.f;
        if (j < Cr) {
            // eight loads in flight per lane (the products are still added in the order c = l, l + tpr, ...): a plain loop waits
            // one global round trip per element, 32 of them at C = 512
            const float* row = fc1 + (int64_t)j * C;
            int c = l;
            for (; c + 7 * tpr < C; c += 8 * tpr) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = row[c + u * tpr];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += w[u] * pooled[c + u * tpr];
            }
            for (; c < C; c += tpr) a += row[c] * pooled[c];
        }
        for (int o = tpr >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (j < Cr && l == 0) hidden[j] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        if ((Cr & 3) == 0) {          // a thread's fc2 row is one contiguous 16..128-byte run: read it as float4s
            const f32x4* w4 = reinterpret_cast<const f32x4*>(fc2 + (int64_t)c * Cr);
#pragma unroll 8
            for (int j = 0; j < Cr; j += 4) {
                const f32x4 w = w4[j >> 2];
                a += w[0] * hidden[j];
                a += w[1] * hidden[j + 1];
                a += w[2] * hidden[j + 2];
                a += w[3] * hidden[j + 3];
            }
        } else {
            for (int j = 0; j < Cr; ++j) a += fc2[(int64_t)c * Cr + j] * hidden[j];
        }
        gate[c] = 1.f / (1.f + __expf(-a));
    }
}

__global__ __launch_bounds__(512) void se_gate_kernel(const float* __restrict__ pooled_in, const float* __restrict__ fc1,
                                                      const float* __restrict__ fc2, float* __restrict__ gate, int C, int Cr) {
    extern __shared__ float sm[];     // pooled[C], hidden[Cr]
    float* pooled = sm;
    float* hidden = sm + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) pooled[c] = pooled_in[(int64_t)b * C + c];
    __syncthreads();
    se_body(pooled, hidden, fc1, fc2, gate + (int64_t)b * C, C, Cr);
}

// second stage of the fused InstanceNorm statistics (instnorm_finalize_kernel's arithmetic) + the SE gate of the normalised tensor
// in one launch: one block per sample
__global__ __launch_bounds__(512) void finalize_se_kernel(const double* __restrict__ ws, float* __restrict__ stats,
                                                          const float* __restrict__ fc1, const float* __restrict__ fc2,
                                                          float* __restrict__ gate, int C, int Cr, int HW, float eps, int nsplit) {
    extern __shared__ float sm[];     // pooled[C], hidden[Cr]
    float* pooled = sm;
    float* hidden = sm + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int64_t i = (int64_t)b * C + c;
        double s = 0.0, q = 0.0;
#pragma unroll 8
        for (int k = 0; k < nsplit; ++k) {
            s += ws[(i * nsplit + k) * 2];
            q += ws[(i * nsplit + k) * 2 + 1];
        }
        const double mean = s / HW;
        double var = q / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean;
        const float rstd = rsqrtf((float)var + eps);
        stats[i * 2] = mf;
        stats[i * 2 + 1] = rstd;
        pooled[c] = (float)((mean - (double)mf) * (double)rstd);
    }
    __syncthreads();
    se_body(pooled, hidden, fc1, fc2, gate + (int64_t)b * C, C, Cr);
}

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// ---- regional average pooling: block = (b, 64-channel slab); each thread owns one channel for a
// quarter of the pixels and keeps R running sums in LDS columns it alone touches ------------------
__global__ void region_mean_kernel(const float* __restrict__ feats, const uint8_t* __restrict__ labels, int Hm, int Wm,
                                   float* __restrict__ out, int H, int W, int C, int R, int out_stride, int out_off) {
    extern __shared__ float sm[];          // sums[NPG][R][64], cnt[R]
    constexpr int NPG = 16;                // pixel groups per block (1024 threads)
    const int slabs = C / 64;
    const int b = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    float* sums = sm;
    int* cnt = reinterpret_cast<int*>(sm + NPG * R * 64);
    for (int t = threadIdx.x; t < NPG * R * 64; t += blockDim.x) sums[t] = 0.f;
    for (int t = threadIdx.x; t < R; t += blockDim.x) cnt[t] = 0;
    __syncthreads();
    const int HW = H * W;
    const float* fb = feats + (int64_t)b * HW * C + slab * 64 + cl;
    float* mine = sums + (pg * R) * 64 + cl;
    // eight pixels in flight per thread (feature + label loads first, then the LDS sums in pixel order: the same additions in the same order; the
    // one-load-per-dependent-add loop read the 134 MB of a 512-channel 64^2 level at 0.96 TB/s)
    constexpr int U = 8;
    for (int p0 = pg; p0 < HW; p0 += NPG * U) {
        float v[U];
        int lab[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * NPG;
            const bool ok = p < HW;
            const int pp = ok ? p : 0;
            const int yy = pp / W, xx = pp - yy * W;
            lab[u] = ok ? labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)] : -1;
            v[u] = ok ? fb[(int64_t)pp * C] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (lab[u] >= 0) {
                mine[lab[u] * 64] += v[u];
                if (cl == 0) atomicAdd(&cnt[lab[u]], 1);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < R * 64; t += blockDim.x) {
        const int r = t / 64, c = t % 64;
        float s = 0.f;
        for (int j = 0; j < NPG; ++j) s += sums[(j * R + r) * 64 + c];
        const int n = cnt[r];
        out[((int64_t)b * R + r) * out_stride + out_off + slab * 64 + c] = n > 0 ? s / (float)n : 0.f;
    }
}

// ---- LocalMLP layer: one wave per (r, o) output neuron, all B samples at once; the weight row is
// streamed once (this op is bound by reading 12 x 16 MB of fp32 weights) ---------------------------
// One wave per (region, output row): the weight row (K <= 4096 floats) is read ONCE into registers and dotted with
// every sample's input row, so the weights are streamed once per call whatever the batch.
constexpr int MAXKV = 16;            // f32x4 per lane: K <= 64 * 4 * 16
__global__ void grouped_linear_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                      const float* __restrict__ bias, const float* __restrict__ add,
                                      float* __restrict__ y, int B, int R, int K, int O, float scale, int act,
                                      float alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= (int64_t)R * O) return;
    const int r = (int)(wid / O), o = (int)(wid % O);
    const float* wrow = Wt + ((int64_t)r * O + o) * K;
    f32x4 w[MAXKV];
#pragma unroll
    for (int j = 0; j < MAXKV; ++j) {
        const int i = (j * 64 + lane) * 4;
        w[j] = i < K ? *reinterpret_cast<const f32x4*>(wrow + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bs = bias ? bias[(int64_t)r * O + o] : 0.f;
    const float ad = add ? add[o] : 0.f;
    for (int b = 0; b < B; ++b) {
        const float* xr = x + ((int64_t)b * R + r) * K;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < MAXKV; ++j) {
            const int i = (j * 64 + lane) * 4;
            if (i < K) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
                acc += w[j][0] * v[0] + w[j][1] * v[1] + w[j][2] * v[2] + w[j][3] * v[3];
            }
        }
        float a = wave_sum(acc);
        if (lane == 0) {
            a = a * scale + bs;
            if (act == 1) a = a > 0.f ? a : a * alpha;
            y[((int64_t)b * R + r) * O + o] = a + ad;
        }
    }
}

// K > 64*4*MAXKV: the weight row no longer fits in registers; it is re-read per sample (rows of <= 64 KB: L2 hits)
__global__ void grouped_linear_longk_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                            const float* __restrict__ bias, const float* __restrict__ add,
                                            float* __restrict__ y, int B, int R, int K, int O, float scale, int act,
                                            float alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= (int64_t)R * O) return;
    const int r = (int)(wid / O), o = (int)(wid % O);
    const float* wrow = Wt + ((int64_t)r * O + o) * K;
    for (int b = 0; b < B; ++b) {
        const float* xr = x + ((int64_t)b * R + r) * K;
        float acc = 0.f;
        for (int i = lane * 4; i < K; i += 256) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + i);
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            acc += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
        }
        float a = wave_sum(acc);
        if (lane == 0) {
            a = a * scale + (bias ? bias[(int64_t)r * O + o] : 0.f);
            if (act == 1) a = a > 0.f ? a : a * alpha;
            y[((int64_t)b * R + r) * O + o] = a + (add ? add[o] : 0.f);
        }
    }
}


// ---- regional style swap (scripts/face_swap.py:117-146, per sample): out[b][r] = swap bit r ? src[b][r] : tgt[b][r]; a source without
// ears (region `ear`: sum of its style vector == 0) takes the mean of both, without teeth (`teeth`) the target's; `below` >= 0: the mean
// of both for that region.  One block per (b, r); the emptiness test is an ordered block sum of the 1280 (C) values -- an absent region's
// vector is exact zeros (regional average pooling of no pixels). ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swap_styles_kernel(const float* __restrict__ tgt, const float* __restrict__ src,
                                                          float* __restrict__ out, int R, int C, unsigned sel, int ear, int teeth,
                                                          int below) {
    __shared__ float part[256];
    const int b = blockIdx.x / R, r = blockIdx.x - b * R;
    const float* a = tgt + ((int64_t)b * R + r) * C;
    const float* c = src + ((int64_t)b * R + r) * C;
    float* o = out + ((int64_t)b * R + r) * C;
    int mode = (sel >> r) & 1u;                              // 0: target, 1: source, 2: mean of both
    if (r == ear || r == teeth) {
        float s = 0.f;
        for (int i = threadIdx.x; i < C; i += 256) s += c[i];
        part[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
            __syncthreads();
        }
        if (part[0] == 0.f) mode = (r == ear) ? 2 : 0;
    }
    if (r == below) mode = 2;
    for (int i = threadIdx.x; i < C; i += 256) o[i] = mode == 2 ? (a[i] + c[i]) / 2.f : (mode == 1 ? c[i] : a[i]);
}

}  // namespace

extern "C" int e4s_resize_bilinear_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, B, C, Hi, Wi, Ho, Wo);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv3x3_small_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
    if (Cout % 4) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * H * W * (Cout / 4);
    if (n <= 0) return 0;
    const size_t smem = (size_t)Cout * Cin * 9 * sizeof(float);
    if (smem > 48 * 1024) return (int)hipErrorInvalidValue;
    int64_t nblk = (n + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(conv3x3_small_kernel, dim3((unsigned)nblk), dim3(256), smem, as_stream(stream), x, w, y, B, H, W, Cin, Cout);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv3x3_stem_f32(const float* x, const float* w, float* y, double* stats_ws, int B, int H, int W, int Cin, int Cout,
                                    void* stream) {
    if (!x || !w || !y || Cin != 3 || Cout != 64 || H % 16 || W % 16 || B <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    const int tx_n = W / 16, per_img = (H / 16) * tx_n;
    const int64_t blocks = (int64_t)B * per_img;
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(conv3x3_stem_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, w, y, stats_ws, H, W, tx_n, per_img);
    E4S_CHECK_LAUNCH();
    return 0;
}

static int instnorm_nsplit(int B, int HW, int C) {
    int nsplit = 2048 / (B * (C / 64) > 0 ? B * (C / 64) : 1);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > HW / 64) nsplit = HW / 64 > 0 ? HW / 64 : 1;
    if (nsplit > 64) nsplit = 64;            // the finalize pass adds the slots serially per (b, c)
    return nsplit;
}

extern "C" int e4s_instnorm_stats_f32(const float* x, float* stats, float* pooled, double* ws, int B, int HW, int C,
                                      float eps, void* stream) {
    if (C % 64) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    // <= 32x32 maps only: one block per (sample, 64 channels) walks HW / 16 pixels serially -- at 64x64 (256 iterations, 2-16 blocks
    // on the chip at batch 1) that measured 32 us against ~17 us for the split two-launch form (profiles/r03_b1_steps_kernel_stats.csv)
    if (HW <= 1024 && HW >= 16) {                       // policy independent of the batch: a sample's statistics never depend on it
        hipLaunchKernelGGL(instnorm_small_kernel, dim3(B * (C / 64)), dim3(1024), 0, st, x, stats, pooled, HW, C, eps);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    const int nsplit = instnorm_nsplit(B, HW, C);
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(B * (C / 64) * nsplit), dim3(256), 0, st, x, ws, HW, C, nsplit);
    E4S_CHECK_LAUNCH();
    const int n = B * C;
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, stats, pooled, n, HW, eps, nsplit);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_instnorm_finalize_f32(const double* ws, float* stats, float* pooled, int B, int HW, int C, int nslots,
                                         float eps, void* stream) {
    const int n = B * C;
    if (n <= 0 || nslots < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), ws, stats, pooled, n,
                       HW, eps, nslots);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_instnorm_apply_stats_f32(const float* x, const float* stats, const float* gate, const float* res,
                                            const float* res_stats, const float* slope, float* y, double* ws, int* nslots,
                                            int B, int H, int W, int C, int rs, void* stream) {
    if (C % 64 || rs < 1 || !nslots) return (int)hipErrorInvalidValue;
    const int ns = instnorm_nsplit(B, H * W, C);
    *nslots = ns;
    hipLaunchKernelGGL(instnorm_apply_stats_kernel, dim3(B * (C / 64) * ns), dim3(256), 0, as_stream(stream), x, stats, gate,
                       res, res_stats, slope, y, ws, H, W, C, rs, ns);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_instnorm_ws_doubles(int B, int HW, int C) {
    return (int64_t)2 * B * C * instnorm_nsplit(B, HW, C);
}

extern "C" int e4s_instnorm_apply_f32(const float* x, const float* stats, const float* gate, const float* res,
                                      const float* res_stats, const float* slope, float* y, int B, int H, int W, int C,
                                      int rs, void* stream) {
    if (C % 4 || rs < 1) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, stats, gate, res, res_stats, slope, y, B, H, W, C, rs);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_se_gate_f32(const float* pooled, const float* fc1, const float* fc2, float* gate, int B, int C,
                               int Cr, void* stream) {
    if (B <= 0 || C <= 0 || Cr <= 0 || (size_t)(C + Cr) * sizeof(float) > 48 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(512), (size_t)(C + Cr) * sizeof(float), as_stream(stream), pooled, fc1, fc2, gate, C, Cr);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_instnorm_finalize_se_f32(const double* ws, float* stats, const float* fc1, const float* fc2, float* gate, int B,
                                            int HW, int C, int Cr, int nslots, float eps, void* stream) {
    if (B <= 0 || C <= 0 || Cr <= 0 || nslots <= 0 || HW <= 0 || (size_t)(C + Cr) * sizeof(float) > 48 * 1024)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(finalize_se_kernel, dim3(B), dim3(512), (size_t)(C + Cr) * sizeof(float), as_stream(stream), ws, stats, fc1, fc2,
                       gate, C, Cr, HW, eps, nslots);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_region_mean_f32(const float* feats, const uint8_t* labels, int Hm, int Wm, float* out, int B, int H,
                                   int W, int C, int R, int out_stride, int out_off, void* stream) {
    if (C % 64 || R > 64) return (int)hipErrorInvalidValue;
    const size_t smem = (size_t)(16 * R * 64) * sizeof(float) + (size_t)R * sizeof(int);
    hipLaunchKernelGGL(region_mean_kernel, dim3(B * (C / 64)), dim3(1024), smem, as_stream(stream), feats, labels, Hm, Wm, out, H, W, C, R, out_stride, out_off);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_grouped_linear_f32(const float* x, const float* W, const float* bias, const float* add, float* y,
                                      int B, int R, int K, int O, float scale, int act, float alpha, void* stream) {
    if (K % 4) return (int)hipErrorInvalidValue;
    const int64_t nw = (int64_t)R * O;
    if (nw <= 0 || B <= 0) return 0;
    if (K > 64 * 4 * MAXKV) {
        hipLaunchKernelGGL(grouped_linear_longk_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, as_stream(stream), x, W, bias, add, y, B, R, K, O, scale, act, alpha);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(grouped_linear_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, as_stream(stream), x, W, bias, add, y, B, R, K, O, scale, act, alpha);
    E4S_CHECK_LAUNCH();
    return 0;
}

/* Regional style swap on the device in one launch (scripts/face_swap.py:117-146, applied per sample): tgt / src / out [B][R][C];
 * sel: bit r set = region r comes from src; ear / teeth / below: region indices of the three special cases (below < 0: off). */
extern "C" int e4s_swap_styles_f32(const float* tgt, const float* src, float* out, int B, int R, int C, unsigned sel, int ear, int teeth,
                                   int below, void* stream) {
    if (!tgt || !src || !out || B < 1 || R < 1 || R > 32 || C < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(swap_styles_kernel, dim3((unsigned)(B * R)), dim3(256), 0, as_stream(stream), tgt, src, out, R, C, sel, ear, teeth,
                       below);
    E4S_CHECK_LAUNCH();
    return 0;
}
