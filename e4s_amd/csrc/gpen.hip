// Small kernels around the conv kernels for GPEN's FullGenerator (src/pretrained/gpen/face_model/gpen_model.py:318-357,
// 558-690) and the Discriminator's ConvLayer / ResBlock stack (src/models/stylegan2/model.py:670-799):
// the 3 -> C 1x1 stem conv, the noise half of GPEN's concatenating StyledConv, PixelNorm, the ResBlock combine and the
// minibatch-stddev feature.  All HBM- or latency-bound.
#include "common.h"

namespace {

// y[b,p,co] = act(sum_c x[b,c,p] * w[co,c] * scale + bias[co]);  x NCHW [B,Cin,HW] (Cin <= 4), y NHWC with channel stride ycs
__global__ void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                     const float* __restrict__ bias, float* __restrict__ y, int64_t HW, int Cin,
                                     int Cout, int ycs, float scale, int act, float alpha, float gain, int64_t n) {
    const int C4 = Cout / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // over B*HW*C4
    if (i >= n) return;
    const int co = (int)(i % C4) * 4;
    const int64_t bp = i / C4;
    const int64_t b = bp / HW, p = bp - b * HW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cin; ++c) {
        const float xv = x[(b * Cin + c) * HW + p];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += xv * w[(co + e) * Cin + c];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = acc[e] * scale + (bias ? bias[co + e] : 0.f);
        if (act) v = (v > 0.f ? v : v * alpha) * gain;
        acc[e] = v;
    }
    *reinterpret_cast<f32x4*>(y + bp * ycs + co) = acc;
}

// y[b,p,coff + c] = lrelu(noise_w * feat[b,p,c] + bias[c]) * gain   (gpen_model.py:343-353: cat((out, w * noise), 1) then
// FusedLeakyReLU over the 2C channels): the second half of the concatenated tensor
__global__ void noise_half_kernel(const float* __restrict__ feat, const float* __restrict__ noise_w,
                                  const float* __restrict__ bias, float* __restrict__ y, int C, int ycs, int coff,
                                  float alpha, float gain, int64_t n) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // over B*HW*C4
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    const int64_t bp = i / C4;
    const float nw = noise_w[0];
    f32x4 v = *reinterpret_cast<const f32x4*>(feat + bp * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = nw * v[e] + bias[c + e];
        v[e] = (t > 0.f ? t : t * alpha) * gain;
    }
    *reinterpret_cast<f32x4*>(y + bp * ycs + coff + c) = v;
}

// PixelNorm (gpen_model.py:18-23 / model.py:14-19): x * rsqrt(mean(x^2, dim=1) + 1e-8), x [B, D]; one wave per row
__global__ void pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = x[(int64_t)row * D + i]; s += v * v; }
    s = wave_sum(s);
    const float r = rsqrtf(s / D + 1e-8f);
    for (int i = lane; i < D; i += 64) y[(int64_t)row * D + i] = x[(int64_t)row * D + i] * r;
}

// out = (a + b) * scale   (ResBlock: (conv2(conv1(x)) + skip(x)) / sqrt(2), model.py:733-737)
__global__ void add_scale_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                 float scale, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 u = *reinterpret_cast<const f32x4*>(a + i * 4), v = *reinterpret_cast<const f32x4*>(b + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = (u + v) * scale;
}

// Minibatch stddev (model.py:783-790, stddev_feat = 1): x NHWC [B,HW,C], groups of `group` samples (sample b belongs to
// group member b / M, set b % M, M = B / group); std[m] = mean_{p,c} sqrt(var_over_members(x[., p, c]) + 1e-8);
// y NHWC [B,HW,Cy] gets x in channels [0,C), std[b % M] in channel C and zeros up to Cy.  One block per set m.
__global__ void minibatch_stddev_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C, int Cy,
                                        int group) {
    __shared__ double red[256];
    const int M = B / group, m = blockIdx.x;
    const int64_t per = (int64_t)HW * C;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < per; i += blockDim.x) {
        float mean = 0.f;
        for (int g = 0; g < group; ++g) mean += x[((int64_t)(g * M + m)) * per + i];
        mean /= group;
        float var = 0.f;
        for (int g = 0; g < group; ++g) { const float d = x[((int64_t)(g * M + m)) * per + i] - mean; var += d * d; }
        acc += (double)sqrtf(var / group + 1e-8f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float sd = (float)(red[0] / (double)per);
    for (int g = 0; g < group; ++g) {
        const int b = g * M + m;
        for (int64_t i = threadIdx.x; i < (int64_t)HW * Cy; i += blockDim.x) {
            const int64_t p = i / Cy;
            const int c = (int)(i - p * Cy);
            y[((int64_t)b * HW + p) * Cy + c] = c < C ? x[((int64_t)b * HW + p) * C + c] : (c == C ? sd : 0.f);
        }
    }
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int e4s_conv1x1_small_f32(const float* x, const float* w, const float* bias, float* y, int B, int HW, int Cin,
                                     int Cout, int y_cstride, float scale, int act, float alpha, float gain, void* stream) {
    if (Cin < 1 || Cin > 4 || Cout % 4) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * HW * (Cout / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(conv1x1_small_kernel, grid1(n), dim3(256), 0, as_stream(stream), x, w, bias, y, (int64_t)HW, Cin,
                       Cout, y_cstride ? y_cstride : Cout, scale, act, alpha, gain, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_noise_half_f32(const float* feat, const float* noise_w, const float* bias, float* y, int64_t npix, int C,
                                  int y_cstride, int coff, float alpha, float gain, void* stream) {
    if (C % 4 || coff % 4 || y_cstride % 4) return (int)hipErrorInvalidValue;
    const int64_t n = npix * (C / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(noise_half_kernel, grid1(n), dim3(256), 0, as_stream(stream), feat, noise_w, bias, y, C, y_cstride,
                       coff, alpha, gain, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_pixelnorm_f32(const float* x, float* y, int B, int D, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(pixelnorm_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), x, y, B, D);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_add_scale_f32(const float* a, const float* b, float* out, float scale, int64_t n, void* stream) {
    if (n % 4) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_scale_kernel, grid1(n / 4), dim3(256), 0, as_stream(stream), a, b, out, scale, n / 4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_minibatch_stddev_f32(const float* x, float* y, int B, int HW, int C, int Cy, int group, void* stream) {
    if (group < 1 || B % group || Cy <= C) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(minibatch_stddev_kernel, dim3(B / group), dim3(256), 0, as_stream(stream), x, y, B, HW, C, Cy, group);
    E4S_CHECK_LAUNCH();
    return 0;
}
