// ToRGB (src/models/stylegan2/model.py:422-448): 1x1 modulated conv to 3 channels (no demodulation)
// + bias + 2x FIR upsample of the previous RGB skip, fused into one HBM-bound pass: the activation
// is read once (NHWC, 16 B per lane), the skip's polyphase FIR needs 4 taps per output pixel.
// Region-select: a masked ToRGB uses ws[b*R + label(pixel)], one pass instead of 12.
#include "common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// LP lanes cooperate on one pixel (each 4 channels per step); 64/LP pixels per wave step.
template <int LP>
__global__ void torgb_kernel(const float* __restrict__ x, const float* __restrict__ ws, const float* __restrict__ bias,
                             const float* __restrict__ skip, const float* __restrict__ k4,
                             const uint8_t* __restrict__ labels, int Hm, int Wm, int R, float* __restrict__ out,
                             int B, int H, int W, int Cin) {
    constexpr int PPW = 64 / LP;                   // pixels per wave step
    const int lane = threadIdx.x & 63;
    const int sub = lane % LP, pslot = lane / LP;
    const int64_t HW = (int64_t)H * W, npix = (int64_t)B * HW;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    __shared__ float skf[16];
    if (threadIdx.x < 16) skf[threadIdx.x] = k4 ? k4[15 - threadIdx.x] : 0.f;    // flipped (true convolution)
    __syncthreads();
    for (int64_t p0 = wave_id * PPW; p0 < npix; p0 += nwaves * PPW) {
        const int64_t p = p0 + pslot;
        const bool live = p < npix;
        const int64_t pp = live ? p : 0;
        const int b = (int)(pp / HW);
        const int rem = (int)(pp - (int64_t)b * HW);
        const int yy = rem / W, xx = rem - yy * W;
        int g = b;
        if (labels) g = b * R + labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
        const float* xp = x + pp * Cin;
        const float* wp = ws + (size_t)g * 3 * Cin;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = sub * 4; c < Cin; c += LP * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + c);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + c);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + Cin + c);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wp + 2 * Cin + c);
            a0 += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
            a1 += v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
            a2 += v[0] * w2[0] + v[1] * w2[1] + v[2] * w2[2] + v[3] * w2[3];
        }
#pragma unroll
        for (int o = LP >> 1; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o, 64);
            a1 += __shfl_xor(a1, o, 64);
            a2 += __shfl_xor(a2, o, 64);
        }
        if (live && sub < 3) {
            const int ch = sub;
            float v = (ch == 0 ? a0 : (ch == 1 ? a1 : a2)) + bias[ch];
            if (skip) {
                // upfirdn2d(skip, k4, up=2, pad=(2,1)): out[y] = sum_j z[y + j - 2] * kflip[j], z = zero-inserted skip
                const int Hs = H >> 1, Ws = W >> 1;
                const float* sp = skip + ((int64_t)b * 3 + ch) * Hs * Ws;
                float acc = 0.f;
#pragma unroll
                for (int jy = 0; jy < 4; ++jy) {
                    const int qy = yy + jy - 2;
                    if (qy < 0 || (qy & 1) || (qy >> 1) >= Hs) continue;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const int qx = xx + jx - 2;
                        if (qx < 0 || (qx & 1) || (qx >> 1) >= Ws) continue;
                        acc += sp[(qy >> 1) * Ws + (qx >> 1)] * skf[jy * 4 + jx];
                    }
                }
                v += acc;
            }
            out[((int64_t)b * 3 + ch) * HW + rem] = v;
        }
    }
}

// Unmasked ToRGB at the high resolutions (Cin <= 128, 256^2..1024^2): pure HBM streaming.
// A block walks 256-pixel groups; 32-channel slabs are read fully coalesced (8 lanes x 16 B = one 128-B line
// per pixel), transposed through LDS (rows padded to 36 floats), then ONE lane owns ONE pixel: 3 dot products
// against the style-scaled weights (broadcast LDS reads), the 4-tap polyphase upsample of the skip, and stores
// that are contiguous across lanes in each NCHW colour plane.
__global__ __launch_bounds__(256) void torgb_pixel_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                          const float* __restrict__ bias, const float* __restrict__ skip,
                                                          const float* __restrict__ k4, float* __restrict__ out,
                                                          int B, int H, int W, int Cin) {
    __shared__ __attribute__((aligned(16))) float sx[256 * 36];
    __shared__ __attribute__((aligned(16))) float sw[3 * 128];
    __shared__ float skf[16];
    const int tid = threadIdx.x;
    const int64_t HW = (int64_t)H * W, npix = (int64_t)B * HW;
    if (tid < 16) skf[tid] = k4 ? k4[15 - tid] : 0.f;
    const int c4 = (tid & 7) * 4, prow = tid >> 3;           // staging role: 32 pixel rows x 8 chunks per pass
    int cur_b = -1;
    for (int64_t p0 = (int64_t)blockIdx.x * 256; p0 < npix; p0 += (int64_t)gridDim.x * 256) {
        const int b = (int)(p0 / HW);                         // HW % 256 == 0: a group never straddles samples
        if (b != cur_b) {
            __syncthreads();
            for (int t = tid; t < 3 * Cin; t += 256) sw[t] = ws[(size_t)b * 3 * Cin + t];
            cur_b = b;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c0 = 0; c0 < Cin; c0 += 32) {
            __syncthreads();                                  // previous slab fully consumed
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pl = prow + 32 * j;
                *reinterpret_cast<f32x4*>(sx + pl * 36 + c4) =
                    *reinterpret_cast<const f32x4*>(x + (p0 + pl) * Cin + c0 + c4);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sx + tid * 36 + q * 4);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(sw + c0 + q * 4);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(sw + Cin + c0 + q * 4);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(sw + 2 * Cin + c0 + q * 4);
                a0 += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
                a1 += v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
                a2 += v[0] * w2[0] + v[1] * w2[1] + v[2] * w2[2] + v[3] * w2[3];
            }
        }
        const int rem = (int)(p0 - (int64_t)b * HW) + tid;
        const int yy = rem / W, xx = rem - yy * W;
        float o[3] = {a0 + bias[0], a1 + bias[1], a2 + bias[2]};
        if (skip) {
            const int Hs = H >> 1, Ws = W >> 1;
#pragma unroll
            for (int jy = 0; jy < 4; ++jy) {
                const int qy = yy + jy - 2;
                if (qy < 0 || (qy & 1) || (qy >> 1) >= Hs) continue;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int qx = xx + jx - 2;
                    if (qx < 0 || (qx & 1) || (qx >> 1) >= Ws) continue;
                    const float kv = skf[jy * 4 + jx];
                    const int64_t so = (int64_t)(qy >> 1) * Ws + (qx >> 1);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) o[ch] += skip[((int64_t)b * 3 + ch) * Hs * Ws + so] * kv;
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[((int64_t)b * 3 + ch) * HW + rem] = o[ch];
    }
}

// Soft-mask fallback (reference formulation, model.py:391-398): out (+)= y * nearest(mask)[:, r].
// channels_last = 1: y/out are NHWC [B,H,W,C];  0: NCHW [B,C,H,W].
__global__ void mask_mul_add_kernel(const float* __restrict__ y, const float* __restrict__ mask, float* __restrict__ out,
                                    int r, int B, int H, int W, int C, int R, int Hm, int Wm, int channels_last,
                                    int accumulate) {
    const int64_t n = (int64_t)B * H * W * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b, yy, xx;
    if (channels_last) {
        int64_t p = i / C;
        xx = (int)(p % W); p /= W;
        yy = (int)(p % H);
        b = (int)(p / H);
    } else {
        int64_t p = i;
        xx = (int)(p % W); p /= W;
        yy = (int)(p % H); p /= H;
        b = (int)(p / C);
    }
    const float m = mask[(((int64_t)b * R + r) * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
    const float v = y[i] * m;
    out[i] = accumulate ? out[i] + v : v;
}

// NoiseInjection + FusedLeakyReLU on an NHWC tensor (only the soft-mask fallback needs it as a separate pass;
// the fused path does this in the conv epilogue): y = lrelu(x + nw*noise[b,p] + bias[c]) * gain
__global__ void noise_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                      const float* __restrict__ noise_w, int64_t noise_bstride,
                                      const float* __restrict__ bias, float* __restrict__ y, int64_t HW, int C,
                                      int64_t n, float alpha, float gain) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int64_t b = pix / HW, p = pix - b * HW;
    float v = x[i] + (bias ? bias[c] : 0.f);
    if (noise) v += noise_w[0] * noise[b * noise_bstride + p];
    y[i] = (v > 0.f ? v : v * alpha) * gain;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t HW, int64_t n,
                                    int src_bstride_zero) {
    // 32x32 LDS transpose per (b): tile over (c, p)
    __shared__ float t[32][33];
    const int64_t ptiles = (HW + 31) / 32;
    const int ctiles = (C + 31) / 32;
    int64_t bid = blockIdx.x;
    const int64_t pt = bid % ptiles; bid /= ptiles;
    const int ct = (int)(bid % ctiles);
    const int64_t b = bid / ctiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const float* xb = x + (src_bstride_zero ? 0 : b * C * HW);
    for (int j = ty; j < 32; j += 8) {
        const int c = ct * 32 + j;
        const int64_t p = pt * 32 + tx;
        t[j][tx] = (c < C && p < HW) ? xb[(int64_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t p = pt * 32 + j;
        const int c = ct * 32 + tx;
        if (c < C && p < HW) y[(b * HW + p) * C + c] = t[tx][j];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int64_t HW) {
    __shared__ float t[32][33];
    const int64_t ptiles = (HW + 31) / 32;
    const int ctiles = (C + 31) / 32;
    int64_t bid = blockIdx.x;
    const int64_t pt = bid % ptiles; bid /= ptiles;
    const int ct = (int)(bid % ctiles);
    const int64_t b = bid / ctiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int64_t p = pt * 32 + j;
        const int c = ct * 32 + tx;
        t[j][tx] = (c < C && p < HW) ? x[(b * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = ct * 32 + j;
        const int64_t p = pt * 32 + tx;
        if (c < C && p < HW) y[(b * C + c) * HW + p] = t[tx][j];
    }
}

}  // namespace

extern "C" int e4s_torgb_f32(const float* x, const float* ws, const float* bias, const float* skip, const float* k4,
                             const uint8_t* labels, int Hm, int Wm, int R, float* out, int B, int H, int W, int Cin,
                             void* stream) {
    if (Cin % 4 || (skip && ((H | W) & 1)) || (skip && !k4)) return (int)hipErrorInvalidValue;
    const int64_t npix = (int64_t)B * H * W;
    if (npix <= 0) return 0;
    hipStream_t st = as_stream(stream);
    if (!labels && Cin <= 128 && Cin % 32 == 0 && ((int64_t)H * W) % 256 == 0) {
        int64_t nb = npix / 256;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(torgb_pixel_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ws, bias, skip, k4, out, B, H, W, Cin);
        E4S_CHECK_LAUNCH();
        return 0;
    }
    const int lp = Cin >= 256 ? 64 : (Cin >= 128 ? 32 : (Cin >= 64 ? 16 : 8));
    const int ppw = 64 / lp;
    int64_t blocks = (npix + (int64_t)ppw * 4 - 1) / ((int64_t)ppw * 4);
    if (blocks > 16384) blocks = 16384;
    dim3 grid((unsigned)blocks), block(256);
#define E4S_TORGB(LP) hipLaunchKernelGGL(torgb_kernel<LP>, grid, block, 0, st, x, ws, bias, skip, k4, labels, Hm, Wm, R, out, B, H, W, Cin)
    if (lp == 64) E4S_TORGB(64);
    else if (lp == 32) E4S_TORGB(32);
    else if (lp == 16) E4S_TORGB(16);
    else E4S_TORGB(8);
#undef E4S_TORGB
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    const int64_t HW = (int64_t)H * W;
    const int64_t blocks = (int64_t)B * ((C + 31) / 32) * ((HW + 31) / 32);
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, C, HW, 0, 0);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_const_input_f32(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    const int64_t HW = (int64_t)H * W;
    const int64_t blocks = (int64_t)B * ((C + 31) / 32) * ((HW + 31) / 32);
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, C, HW, 0, 1);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    const int64_t HW = (int64_t)H * W;
    const int64_t blocks = (int64_t)B * ((C + 31) / 32) * ((HW + 31) / 32);
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, C, HW);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_mask_mul_add_f32(const float* y, const float* mask, float* out, int r, int B, int H, int W, int C,
                                    int R, int Hm, int Wm, int channels_last, int accumulate, void* stream) {
    const int64_t n = (int64_t)B * H * W * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mask_mul_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), y, mask,
                       out, r, B, H, W, C, R, Hm, Wm, channels_last, accumulate);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_noise_bias_act_nhwc_f32(const float* x, const float* noise, const float* noise_w,
                                           int64_t noise_bstride, const float* bias, float* y, int B, int HW, int C,
                                           float alpha, float gain, void* stream) {
    const int64_t n = (int64_t)B * HW * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(noise_bias_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, noise,
                       noise_w, noise_bstride, bias, y, (int64_t)HW, C, n, alpha, gain);
    E4S_CHECK_LAUNCH();
    return 0;
}
