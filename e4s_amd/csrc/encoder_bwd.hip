// Backward of the regional style encoder (SURVEY.md 8(f) "C5": the Net3 encoder + StyleGAN2 joint train step,
// src/training/coach.py:340-356): the HBM-bound pieces around the conv dgrad / wgrad kernels -- InstanceNorm backward
// (two ordered reductions + one apply pass, helpers.py:128-141), PReLU forward/backward, the strided scatters of the
// stride-2 paths, and the regional average pooling's backward (psp_encoders.py:264-283).  NHWC activations; every
// reduction is two-stage with a fixed order (no floating-point atomics).
#include "common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// ---- InstanceNorm backward, stage 1: per (b, c): A = sum_p dy, Bq = sum_p dy * xhat, xhat = (x - mean) * rstd ----------
__global__ void in_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                      const float* __restrict__ stats, double* __restrict__ ws, int HW, int C, int nsplit) {
    const int slabs = C / 64;
    const int split = blockIdx.x % nsplit;
    const int slab = (blockIdx.x / nsplit) % slabs;
    const int b = blockIdx.x / (nsplit * slabs);
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = slab * 64 + cl;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const float mean = stats[((int64_t)b * C + c) * 2], rstd = stats[((int64_t)b * C + c) * 2 + 1];
    const int64_t base = (int64_t)b * HW * C + c;
    double a = 0.0, q = 0.0;
    for (int p = p0 + pg; p < p1; p += 4) {
        const float g = dy[base + (int64_t)p * C];
        const float xh = (x[base + (int64_t)p * C] - mean) * rstd;
        a += (double)g;
        q += (double)g * (double)xh;
    }
    __shared__ double red[2][4][64];
    red[0][pg][cl] = a;
    red[1][pg][cl] = q;
    __syncthreads();
    if (pg == 0) {
        a = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        q = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
        double* slot = ws + (((int64_t)b * C + c) * nsplit + split) * 2;
        slot[0] = a;
        slot[1] = q;
    }
}

__global__ void in_bwd_finalize_kernel(const double* __restrict__ ws, float* __restrict__ sums, int n, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nsplit; ++k) {
        a += ws[((int64_t)i * nsplit + k) * 2];
        q += ws[((int64_t)i * nsplit + k) * 2 + 1];
    }
    sums[i * 2] = (float)a;
    sums[i * 2 + 1] = (float)q;
}

// stage 2: dx (+)= rstd * gate * (dy - A/N - xhat * Bq/N)
__global__ void in_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ stats, const float* __restrict__ sums,
                                    const float* __restrict__ gate, float* __restrict__ dx, int HW, int C, int accumulate,
                                    int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const int64_t b = i / ((int64_t)HW * C4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 o;
    const float invn = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t bc = b * C + c + e;
        const float mean = stats[bc * 2], rstd = stats[bc * 2 + 1];
        const float xh = (xv[e] - mean) * rstd;
        const float gt = gate ? gate[bc] : 1.f;
        o[e] = rstd * gt * (g[e] - sums[bc * 2] * invn - xh * sums[bc * 2 + 1] * invn);
    }
    if (accumulate) o += *reinterpret_cast<const f32x4*>(dx + i * 4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// ---- PReLU ------------------------------------------------------------------------------------------------------
__global__ void prelu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ slope, float* __restrict__ y,
                                 int C, int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(u + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope[c + e];
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
}

// du = dy * (u > 0 ? 1 : slope[c]);  dslope partial[blk][c] = sum over the block's pixels of dy * u * [u <= 0]
// block = 64 channels x 4 pixel groups over a pixel range; grid = (C/64) * nsplit
__global__ void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                 const float* __restrict__ slope, float* __restrict__ du, float* __restrict__ part,
                                 int64_t npix, int C, int nsplit) {
    const int slabs = C / 64;
    const int split = blockIdx.x % nsplit, slab = blockIdx.x / nsplit;
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = slab * 64 + cl;
    const int64_t per = (npix + nsplit - 1) / nsplit;
    const int64_t p0 = split * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    const float a = slope[c];
    float acc = 0.f;
    for (int64_t p = p0 + pg; p < p1; p += 4) {
        const float g = dy[p * C + c], uv = u[p * C + c];
        const bool pos = uv > 0.f;
        du[p * C + c] = pos ? g : g * a;
        acc += pos ? 0.f : g * uv;
    }
    __shared__ float red[4][64];
    red[pg][cl] = acc;
    __syncthreads();
    if (pg == 0) part[(int64_t)split * C + c] = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    (void)slabs;
}

// ---- strided scatters -------------------------------------------------------------------------------------------
// out[b, y*s, x*s, c] (+)= in[b, y, x, c]; out is [B, H*s, W*s, C]; zero_fill: the other positions are set to 0
// out[b][a][c][(py*2+px)*C + ch] = in[b][2a+py][2c+px][ch]: the four output phases of an up-sampling conv side by side in the channel
// dimension of its input grid (the dgrad of the polyphase form is then ONE 3x3 convolution with 4*Cout input channels)
__global__ void pixel_unshuffle2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C / 4;
    const int ch = (int)(i % c4);
    const int ph = (int)((i / c4) & 3);
    const int64_t pix = i / (4 * c4);                      // (b, a, c) flattened
    const int cx = (int)(pix % W);
    const int64_t ba = pix / W;                            // b * H + a
    const int64_t src = (((ba * 2 + (ph >> 1)) * (2 * W)) + 2 * cx + (ph & 1)) * C + ch * 4;
    *reinterpret_cast<f32x4*>(out + (pix * 4 + ph) * C + ch * 4) = *reinterpret_cast<const f32x4*>(in + src);
}

__global__ void strided_scatter_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int s,
                                       int accumulate, int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over the OUTPUT [B, H*s, W*s, C4]
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    int64_t r = i / C4;
    const int ox = (int)(r % (W * s)); r /= (W * s);
    const int oy = (int)(r % (H * s));
    const int64_t b = r / (H * s);
    const bool hit = (oy % s == 0) && (ox % s == 0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hit) v = *reinterpret_cast<const f32x4*>(in + ((b * H + oy / s) * W + ox / s) * C + c);
    if (accumulate) {
        if (!hit) return;
        v += *reinterpret_cast<const f32x4*>(out + i * 4);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// out[b, y*s + oy, x*s + ox, c] = in[b, y, x, c] on an output grid [B, Ho, Wo, C] of any size >= the last placed index; every
// other position is 0 (the dgrad operand of a stride-s padding-0 conv: Discriminator ConvLayers, model.py:683-700)
__global__ void strided_place_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int s, int oy,
                                     int ox, int Ho, int Wo, int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over the OUTPUT [B, Ho, Wo, C4]
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    int64_t r = i / C4;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int64_t b = r / Ho;
    const int ry = y - oy, rx = x - ox;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ry >= 0 && rx >= 0 && ry % s == 0 && rx % s == 0 && ry / s < H && rx / s < W)
        v = *reinterpret_cast<const f32x4*>(in + ((b * H + ry / s) * W + rx / s) * C + c);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- regional average pooling backward: dfeat[b,p,c] (+)= dcodes[b, r(p), off + c] / count[b, r(p)] -------------------
__global__ void region_count_kernel(const uint8_t* __restrict__ labels, int Hm, int Wm, int* __restrict__ cnt, int H,
                                    int W, int R) {
    __shared__ int sc[16];
    const int b = blockIdx.x;
    if (threadIdx.x < 16) sc[threadIdx.x] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int yy = p / W, xx = p - yy * W;
        atomicAdd(&sc[labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)]], 1);   // integer
    }
    __syncthreads();
    if ((int)threadIdx.x < R) cnt[b * R + threadIdx.x] = sc[threadIdx.x];
}

__global__ void region_mean_bwd_kernel(const float* __restrict__ dcodes, const uint8_t* __restrict__ labels, int Hm,
                                       int Wm, const int* __restrict__ cnt, float* __restrict__ dfeat, int H, int W, int C,
                                       int R, int stride, int off, int accumulate, int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    int64_t r = i / C4;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int64_t b = r / H;
    const int lab = labels[(b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
    const float inv = 1.f / (float)cnt[b * R + lab];                    // the pixel itself is in the region: count >= 1
    f32x4 v = *reinterpret_cast<const f32x4*>(dcodes + (b * R + lab) * stride + off + c) * inv;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(dfeat + i * 4);
    *reinterpret_cast<f32x4*>(dfeat + i * 4) = v;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

int in_nsplit(int B, int HW, int C) {
    int nsplit = 2048 / (B * (C / 64) > 0 ? B * (C / 64) : 1);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > HW / 64) nsplit = HW / 64 > 0 ? HW / 64 : 1;
    return nsplit;
}

int prelu_nsplit(int64_t npix, int C) {
    int ns = 2048 / (C / 64 > 0 ? C / 64 : 1);
    if ((int64_t)ns > npix / 64) ns = (int)(npix / 64 > 0 ? npix / 64 : 1);
    return ns < 1 ? 1 : ns;
}

}  // namespace

extern "C" int64_t e4s_instnorm_bwd_ws_doubles(int B, int HW, int C) { return (int64_t)2 * B * C * in_nsplit(B, HW, C); }

extern "C" int e4s_instnorm_bwd_f32(const float* dy, const float* x, const float* stats, const float* gate, float* sums,
                                    float* dx, double* ws, int B, int HW, int C, int accumulate, void* stream) {
    if (C % 64) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int ns = in_nsplit(B, HW, C);
    hipLaunchKernelGGL(in_bwd_partial_kernel, dim3(B * (C / 64) * ns), dim3(256), 0, st, dy, x, stats, ws, HW, C, ns);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ws, sums, B * C, ns);
    E4S_CHECK_LAUNCH();
    const int64_t n4 = (int64_t)B * HW * (C / 4);
    hipLaunchKernelGGL(in_bwd_apply_kernel, grid1(n4), dim3(256), 0, st, dy, x, stats, sums, gate, dx, HW, C, accumulate, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

// the two ordered sums alone: sums[b,c] = {sum_p dy, sum_p dy * xhat} (frozen-statistics normalisation: the second one is
// dL/dgate of gate * norm(x), criteria.hip applies the rest)
extern "C" int e4s_instnorm_bwd_sums_f32(const float* dy, const float* x, const float* stats, float* sums, double* ws, int B,
                                         int HW, int C, void* stream) {
    if (C % 64) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int ns = in_nsplit(B, HW, C);
    hipLaunchKernelGGL(in_bwd_partial_kernel, dim3(B * (C / 64) * ns), dim3(256), 0, st, dy, x, stats, ws, HW, C, ns);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ws, sums, B * C, ns);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_prelu_f32(const float* u, const float* slope, float* y, int64_t npix, int C, void* stream) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const int64_t n4 = npix * (C / 4);
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(prelu_fwd_kernel, grid1(n4), dim3(256), 0, as_stream(stream), u, slope, y, C, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_prelu_bwd_ws_floats(int64_t npix, int C) {
    return e4s_reduce_parts_ws_floats(prelu_nsplit(npix, C), C);
}

extern "C" int e4s_prelu_bwd_f32(const float* dy, const float* u, const float* slope, float* du, float* dslope, float* ws,
                                 int64_t npix, int C, void* stream) {
    if (C % 64) return (int)hipErrorInvalidValue;
    const int ns = prelu_nsplit(npix, C);
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3((C / 64) * ns), dim3(256), 0, as_stream(stream), dy, u, slope, du, ws, npix, C, ns);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, dslope, ns, C, 1.f, stream);
}

extern "C" int e4s_pixel_unshuffle2_f32(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    if (C % 4 || B <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * H * W * 4 * (C / 4);
    hipLaunchKernelGGL(pixel_unshuffle2_kernel, grid1(n4), dim3(256), 0, as_stream(stream), in, out, H, W, C, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_strided_scatter_f32(const float* in, float* out, int B, int H, int W, int C, int s, int accumulate,
                                       void* stream) {
    if (C % 4 || s < 1) return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * H * s * W * s * (C / 4);
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(strided_scatter_kernel, grid1(n4), dim3(256), 0, as_stream(stream), in, out, H, W, C, s, accumulate, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_strided_place_f32(const float* in, float* out, int B, int H, int W, int C, int s, int oy, int ox, int Ho,
                                     int Wo, void* stream) {
    if (C % 4 || s < 1 || oy < 0 || ox < 0 || (H - 1) * s + oy >= Ho || (W - 1) * s + ox >= Wo) return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(strided_place_kernel, grid1(n4), dim3(256), 0, as_stream(stream), in, out, H, W, C, s, oy, ox, Ho, Wo, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_region_mean_bwd_f32(const float* dcodes, const uint8_t* labels, int Hm, int Wm, int* counts, float* dfeat,
                                       int B, int H, int W, int C, int R, int stride, int off, int accumulate, void* stream) {
    if (C % 4 || R < 1 || R > 16 || stride % 4 || off % 4) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(region_count_kernel, dim3(B), dim3(256), 0, st, labels, Hm, Wm, counts, H, W, R);
    E4S_CHECK_LAUNCH();
    const int64_t n4 = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(region_mean_bwd_kernel, grid1(n4), dim3(256), 0, st, dcodes, labels, Hm, Wm, counts, dfeat, H, W, C,
                       R, stride, off, accumulate, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}
