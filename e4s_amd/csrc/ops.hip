// 1:1 replacements of the reference's two native ops plus the bias-grad reduction.
//   fused_bias_act : src/models/stylegan2/op/fused_bias_act_kernel.cu:19-49
//   upfirdn2d      : src/models/stylegan2/op/upfirdn2d_kernel.cu:52-137
// Written for wave64 / 256 CUs: float4 grid-stride streams for the elementwise op,
// LDS-staged input + filter tiles for the FIR.
#include "common.h"

namespace {

__device__ __forceinline__ float bias_act_one(float x, float ref, int code, float alpha, float scale) {
    float y;
    switch (code) {
        case 12: case 32: y = 0.f; break;
        case 30: y = (x > 0.f) ? x : x * alpha; break;
        case 31: y = (ref > 0.f) ? x : x * alpha; break;
        default: y = x; break;   // 10, 11 and anything else: linear (fused_bias_act_kernel.cu:37-39)
    }
    return y * scale;
}

// vectorised: requires step_b % 4 == 0 (or no bias) and 16-byte aligned pointers
__global__ void fused_bias_act_v4(const f32x4* __restrict__ x, const float* __restrict__ b,
                                  const f32x4* __restrict__ ref, f32x4* __restrict__ y, int64_t n4,
                                  int step_b, int size_b, int code, float alpha, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = x[i];
        if (b) {
            const float bv = b[((i * 4) / step_b) % size_b];
            v += bv;
        }
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (ref) r = ref[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bias_act_one(v[e], r[e], code, alpha, scale);
        y[i] = o;
    }
}

__global__ void fused_bias_act_scalar(const float* __restrict__ x, const float* __restrict__ b,
                                      const float* __restrict__ ref, float* __restrict__ y, int64_t n,
                                      int step_b, int size_b, int code, float alpha, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        y[i] = bias_act_one(v, ref ? ref[i] : 0.f, code, alpha, scale);
    }
}

// out[c] = sum_{i : (i/step_b)%size_b == c} g[i].  One block per channel; data for a channel is
// `outer` runs of step_b contiguous floats.
__global__ void channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n,
                                   int step_b, int size_b) {
    const int c = blockIdx.x;
    const int64_t outer = n / ((int64_t)step_b * size_b);
    float acc = 0.f;
    for (int64_t o = 0; o < outer; ++o) {
        const float* base = g + (o * size_b + c) * (int64_t)step_b;
        for (int i = threadIdx.x; i < step_b; i += blockDim.x) acc += base[i];
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}

// ---- upfirdn2d ---------------------------------------------------------------------------
// Block = 256 threads -> a TOH x TOW tile of one (major, minor) plane.  The flipped filter and
// the input patch that feeds the tile are staged in LDS; each thread then walks only the taps
// that hit a real (non zero-inserted) input sample.
constexpr int TOH = 16, TOW = 64;

__device__ __forceinline__ int floordiv(int a, int b) {
    int c = a / b;
    if (c * b > a) --c;
    return c;
}

__global__ void upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y,
                                 int major, int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y,
                                 int down_x, int down_y, int pad_x0, int pad_y0, int out_h, int out_w,
                                 int tin_h, int tin_w, int tiles_x, int tiles_y) {
    extern __shared__ float sm[];
    float* sk = sm;                 // kh*kw, flipped
    float* sx = sm + kh * kw;       // tin_h * tin_w
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int mi = bid % minor;
    const int mj = bid / minor;
    const int oy0 = ty * TOH, ox0 = tx * TOW;
    for (int t = threadIdx.x; t < kh * kw; t += blockDim.x) {
        const int ky = t / kw, kx = t - ky * kw;
        sk[t] = k[(kh - 1 - ky) * kw + (kw - 1 - kx)];            // upfirdn2d_kernel.cu:77
    }
    // first up-sampled coordinate the tile touches, and the first real input sample at/after it
    const int mid_y0 = oy0 * down_y - pad_y0, mid_x0 = ox0 * down_x - pad_x0;
    const int in_y0 = floordiv(mid_y0 + up_y - 1, up_y), in_x0 = floordiv(mid_x0 + up_x - 1, up_x);
    for (int t = threadIdx.x; t < tin_h * tin_w; t += blockDim.x) {
        const int ry = t / tin_w, rx = t - ry * tin_w;
        const int iy = in_y0 + ry, ix = in_x0 + rx;
        float v = 0.f;
        if (iy >= 0 && ix >= 0 && iy < in_h && ix < in_w)
            v = x[(((int64_t)mj * in_h + iy) * in_w + ix) * minor + mi];
        sx[t] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TOH * TOW; t += blockDim.x) {
        const int ry = t / TOW, rx = t - ry * TOW;
        const int oy = oy0 + ry, ox = ox0 + rx;
        if (oy >= out_h || ox >= out_w) continue;
        // out[oy] = sum_j up[oy*down - pad0 + j] * kflip[j];  up[q] = x[q/up] iff q % up == 0
        const int my = oy * down_y - pad_y0, mx = ox * down_x - pad_x0;
        const int iy_first = floordiv(my + up_y - 1, up_y), ix_first = floordiv(mx + up_x - 1, up_x);
        float acc = 0.f;
        for (int iy = iy_first, jy = iy_first * up_y - my; jy < kh; ++iy, jy += up_y) {
            const float* row = sx + (iy - in_y0) * tin_w;
            for (int ix = ix_first, jx = ix_first * up_x - mx; jx < kw; ++ix, jx += up_x)
                acc += row[ix - in_x0] * sk[jy * kw + jx];
        }
        y[(((int64_t)mj * out_h + oy) * out_w + ox) * minor + mi] = acc;
    }
}

}  // namespace

extern "C" int e4s_abi_version(void) { return 14; }
extern "C" const char* e4s_build_arch(void) { return "gfx950"; }

extern "C" int e4s_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t n,
                                      int step_b, int size_b, int act, int grad, float alpha, float scale,
                                      void* stream) {
    if (n <= 0) return 0;
    const int code = act * 10 + grad;
    if (code == 31 && !ref) return (int)hipErrorInvalidValue;
    if (b && (step_b <= 0 || size_b <= 0)) return (int)hipErrorInvalidValue;
    const bool vec = (n % 4 == 0) && (!b || step_b % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                       reinterpret_cast<uintptr_t>(ref)) % 16 == 0);
    const int block = 256;
    if (vec) {
        const int64_t n4 = n / 4;
        const int grid = (int)((n4 + block - 1) / block < 8192 ? (n4 + block - 1) / block : 8192);
        hipLaunchKernelGGL(fused_bias_act_v4, dim3(grid), dim3(block), 0, as_stream(stream),
                           reinterpret_cast<const f32x4*>(x), b, reinterpret_cast<const f32x4*>(ref),
                           reinterpret_cast<f32x4*>(y), n4, step_b, size_b, code, alpha, scale);
    } else {
        const int grid = (int)((n + block - 1) / block < 8192 ? (n + block - 1) / block : 8192);
        hipLaunchKernelGGL(fused_bias_act_scalar, dim3(grid), dim3(block), 0, as_stream(stream), x, b, ref, y, n,
                           step_b, size_b, code, alpha, scale);
    }
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_channel_sum_f32(const float* g, float* out, int64_t n, int step_b, int size_b, void* stream) {
    if (size_b <= 0 || step_b <= 0 || n % ((int64_t)step_b * size_b)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(size_b), dim3(256), 0, as_stream(stream), g, out, n, step_b, size_b);
    E4S_CHECK_LAUNCH();
    return 0;
}

// up = down = 1 on a channels-last view (minor = C, C % 4 == 0): the Blur of every ConvLayer / ResBlock when the activations are
// NHWC (Discriminator, GPEN: model.py:683-689).  The generic kernel above takes ONE minor index per block -- right for the
// reference's [N*C, H, W, 1] view, but on [B, H, W, C] every 4-byte read is its own cache line: 19.7 % of the GPU time of the
// config-5 legs (profiles/r03_train_kernel_stats.csv: 377 us average, 2.6 ms at 1024^2).  Here a thread owns 4 channels of one
// output column and a strip of 4 output rows: (kh + 3) x kw 16-byte loads, contiguous across the lanes of a pixel.
__global__ __launch_bounds__(256) void fir_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y,
                                                       int in_h, int in_w, int C, int kh, int kw, int pad_x0, int pad_y0,
                                                       int out_h, int out_w) {
    __shared__ float sk[64];
    if (threadIdx.x < kh * kw) {
        const int ky = threadIdx.x / kw, kx = threadIdx.x - ky * kw;
        sk[threadIdx.x] = k[(kh - 1 - ky) * kw + (kw - 1 - kx)];      // flipped: true convolution (upfirdn2d_kernel.cu:77)
    }
    __syncthreads();
    const int c4n = C >> 2, ppb = 256 / c4n;
    const int c4 = threadIdx.x % c4n, pxl = threadIdx.x / c4n;
    const int ox = blockIdx.x * ppb + pxl, oy0 = blockIdx.y * 4;
    if (pxl >= ppb || ox >= out_w) return;
    const float* xb = x + (size_t)blockIdx.z * in_h * in_w * C + c4 * 4;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ry = 0; ry < kh + 3; ++ry) {
        const int iy = oy0 + ry - pad_y0;
        if ((unsigned)iy >= (unsigned)in_h) continue;
        for (int jx = 0; jx < kw; ++jx) {
            const int ix = ox + jx - pad_x0;
            if ((unsigned)ix >= (unsigned)in_w) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * in_w + ix) * C);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jy = ry - r;                      // output row oy0 + r reads input row (oy0 + r) + jy - pad_y0
                if (jy >= 0 && jy < kh) acc[r] += v * sk[jy * kw + jx];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (oy0 + r < out_h)
            *reinterpret_cast<f32x4*>(y + (((size_t)blockIdx.z * out_h + oy0 + r) * out_w + ox) * C + c4 * 4) = acc[r];
}

// The same for the 4 x 4 blur every caller uses (make_kernel([1, 3, 3, 1])): taps known at compile time, so the 7 x 4 loads of a thread are
// issued together instead of one per dependent FMA (the runtime-bound loops above ran the Discriminator's 1024^2 x 32 blur at 1.4 TB/s:
// 377 us, 3 ms of a config-5 G step).  Same products, same order of additions per output.
template <int KH, int KW>
__global__ __launch_bounds__(256) void fir_nhwc_fixed_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y,
                                                             int in_h, int in_w, int C, int pad_x0, int pad_y0, int out_h, int out_w) {
    __shared__ float sk[KH * KW];
    if (threadIdx.x < KH * KW) {
        const int ky = threadIdx.x / KW, kx = threadIdx.x - ky * KW;
        sk[threadIdx.x] = k[(KH - 1 - ky) * KW + (KW - 1 - kx)];
    }
    __syncthreads();
    const int c4n = C >> 2, ppb = 256 / c4n;
    const int c4 = threadIdx.x % c4n, pxl = threadIdx.x / c4n;
    const int ox = blockIdx.x * ppb + pxl, oy0 = blockIdx.y * 4;
    if (pxl >= ppb || ox >= out_w) return;
    const float* xb = x + (size_t)blockIdx.z * in_h * in_w * C + c4 * 4;
    f32x4 v[KH + 3][KW];
#pragma unroll
    for (int ry = 0; ry < KH + 3; ++ry) {
        const int iy = oy0 + ry - pad_y0;
#pragma unroll
        for (int jx = 0; jx < KW; ++jx) {
            const int ix = ox + jx - pad_x0;
            v[ry][jx] = ((unsigned)iy < (unsigned)in_h && (unsigned)ix < (unsigned)in_w)
                            ? *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * in_w + ix) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (oy0 + r >= out_h) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ry = r; ry < r + KH; ++ry) {               // the generic kernel's order: input rows ascending, columns ascending; padding is skipped there
            const int iy = oy0 + ry - pad_y0;
            if ((unsigned)iy >= (unsigned)in_h) continue;
#pragma unroll
            for (int jx = 0; jx < KW; ++jx) {
                const int ix = ox + jx - pad_x0;
                if ((unsigned)ix < (unsigned)in_w) acc += v[ry][jx] * sk[(ry - r) * KW + jx];
            }
        }
        *reinterpret_cast<f32x4*>(y + (((size_t)blockIdx.z * out_h + oy0 + r) * out_w + ox) * C + c4 * 4) = acc;
    }
}

extern "C" int e4s_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                                 int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                 int pad_y0, int pad_y1, void* stream) {
    if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1 || kh > 16 || kw > 16)
        return (int)hipErrorInvalidValue;
    const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;     // upfirdn2d_kernel.cu:167-168
    const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (out_h <= 0 || out_w <= 0 || major <= 0 || minor <= 0) return 0;
    if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && minor % 4 == 0 && minor >= 8 && minor <= 1024 && kh * kw <= 64 &&
        major <= 65535) {
        const int ppb = 256 / (minor / 4);
        if (kh == 4 && kw == 4) {
            hipLaunchKernelGGL((fir_nhwc_fixed_kernel<4, 4>), dim3((unsigned)((out_w + ppb - 1) / ppb), (unsigned)((out_h + 3) / 4), (unsigned)major),
                               dim3(256), 0, as_stream(stream), x, k, y, in_h, in_w, minor, pad_x0, pad_y0, out_h, out_w);
            E4S_CHECK_LAUNCH();
            return 0;
        }
        hipLaunchKernelGGL(fir_nhwc_kernel, dim3((unsigned)((out_w + ppb - 1) / ppb), (unsigned)((out_h + 3) / 4), (unsigned)major),
                           dim3(256), 0, as_stream(stream), x, k, y, in_h, in_w, minor, kh, kw, pad_x0, pad_y0, out_h, out_w);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    const int tin_h = ((TOH - 1) * down_y + kh - 1) / up_y + 2;
    const int tin_w = ((TOW - 1) * down_x + kw - 1) / up_x + 2;
    const int tiles_x = (out_w + TOW - 1) / TOW, tiles_y = (out_h + TOH - 1) / TOH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * major * minor;
    if (blocks > 0x7fffffff) return (int)hipErrorInvalidValue;
    const size_t smem = (size_t)(kh * kw + tin_h * tin_w) * sizeof(float);
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((unsigned)blocks), dim3(256), smem, as_stream(stream), x, k, y, major,
                       in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, tin_h,
                       tin_w, tiles_x, tiles_y);
    E4S_CHECK_LAUNCH();
    return 0;
}
