// HBM-bound pieces of the generator backward: segmented (per-region) reductions for the
// demodulation and ToRGB-weight gradients, and the ToRGB input gradient.  All on NHWC activations.
#include "common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// dd[b,r,co] = (1/d[b,r,co]) * sum_{p in r} gz[p,co] * (z[p,co] - nw*noise[p] - bias[co]),  z = lrelu^-1(y)/gain
// (out_pre = d * c  =>  dL/dd = sum gz * c = sum gz * out_pre / d).  Block = (b, 64-channel slab, pixel split).
constexpr int NPG = 4;
__global__ void demod_grad_kernel(const float* __restrict__ gz, const float* __restrict__ y,
                                  const float* __restrict__ noise, const float* __restrict__ noise_w,
                                  int64_t noise_bstride, const float* __restrict__ bias, float alpha, float gain,
                                  const uint8_t* __restrict__ labels, int Hm, int Wm, int R,
                                  float* __restrict__ dd, int H, int W, int C, int nsplit, int64_t pstride) {
    extern __shared__ float sums[];        // [NPG][R][64]
    const int cw = C >= 64 ? 64 : C;       // channels per slab (C = 32 at the 1024^2 layers)
    const int slabs = C / cw;
    const int split = blockIdx.x % nsplit;
    const int slab = (blockIdx.x / nsplit) % slabs;
    const int b = blockIdx.x / (nsplit * slabs);
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < NPG * R * 64; t += blockDim.x) sums[t] = 0.f;
    __syncthreads();
    const bool live = cl < cw;
    const int HW = H * W, c = slab * cw + (live ? cl : 0);
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(p0 + per, HW);
    const float nw = noise ? noise_w[0] : 0.f, bs = bias ? bias[c] : 0.f;
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * alpha);
    float* mine = sums + (pg * R) * 64 + cl;
    for (int p = p0 + pg; p < p1; p += NPG) {
        const int yy = p / W, xx = p - yy * W;
        int lab = 0;
        if (labels) {
            lab = labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
            lab = lab < R ? lab : R - 1;                    // never index the LDS table with a label outside [0, R)
        }
        const int64_t idx = ((int64_t)b * HW + p) * C + c;
        const float yv = y[idx];
        const float z = yv > 0.f ? yv * inv_pos : yv * inv_neg;
        const float nz = noise ? nw * noise[(int64_t)b * noise_bstride + p] : 0.f;
        if (live) mine[lab * 64] += gz[idx] * (z - nz - bs);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < R * 64; t += blockDim.x) {
        const int r = t / 64, cc = t % 64;
        if (cc >= cw) continue;
        float s = 0.f;
        for (int j = 0; j < NPG; ++j) s += sums[(j * R + r) * 64 + cc];
        dd[(int64_t)split * pstride + ((int64_t)b * R + r) * C + slab * cw + cc] = s;     // slot of this pixel split
    }
}

// FusedLeakyReLU backward + dL/dd in ONE pass (round 4; replaces e4s_fused_bias_act_f32(grad = 1) followed by demod_grad_kernel, which re-read
// gz and y with 4-byte lanes and a read-modify-write of LDS per pixel: 256 us for the 268 MB of a 1024^2 layer, 4x its traffic time):
//   gz[p,c] = dy[p,c] * (y[p,c] > 0 ? gain : gain * alpha)                         (op/fused_act.py:18-31, fused_bias_act_kernel.cu:43)
//   part[split][(b R + r) C + c] = sum_{p in split, region(p) = r} gz[p,c] * (z[p,c] - nw noise[p] - bias[c]),  z = lrelu^-1(y) / gain
// Thread = (4 channels, pixel lane): 16-byte loads, a wave reads 1 KB of contiguous NHWC; the sum of the CURRENT region stays in registers
// and goes to the thread's own LDS slot only when the label changes along its pixel walk (labels are piecewise constant); pixel lanes are
// combined in lane order and the splits by e4s_reduce_parts_f32: bit-reproducible.
__global__ __launch_bounds__(256) void act_bwd_demod_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ gz,
                                                            const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                            int64_t noise_bstride, const float* __restrict__ bias, float alpha, float gain,
                                                            const uint8_t* __restrict__ labels, int Hm, int Wm, int R,
                                                            float* __restrict__ part, int H, int W, int C, int nsplit, int64_t pstride) {
    extern __shared__ f32x4 tab[];          // [lanes][R][C / 4]
    const int c4n = C >> 2, lanes = 256 / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const int b = blockIdx.x / nsplit, split = blockIdx.x - b * nsplit;
    const int HW = H * W;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(p0 + per, HW);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int t = threadIdx.x; t < lanes * R * c4n; t += 256) tab[t] = zero4;
    __syncthreads();
    const float nw = noise ? noise_w[0] : 0.f;
    const f32x4 bs = bias ? *reinterpret_cast<const f32x4*>(bias + c4 * 4) : zero4;
    const float g_pos = gain, inv_pos = 1.f / gain, inv_neg = 1.f / (gain * alpha);
    f32x4* mine = tab + (size_t)rl * R * c4n + c4;
    f32x4 acc = zero4;
    int cur = 0;
    for (int p = p0 + rl; p < p1; p += lanes) {
        int lab = 0;
        if (labels) {
            const int yy = p / W, xx = p - yy * W;
            lab = labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
            // a label outside [0, R) (a 255 "ignore" value, a num_regions mismatch) must not index the LDS table: its pixels still get
            // their gz, their demodulation-gradient term goes to the last region's slot (e4s_mask_labels never produces such labels;
            // the forward kernels clamp the same way)
            lab = lab < R ? lab : R - 1;
        }
        if (lab != cur) {
            mine[cur * c4n] += acc;
            acc = zero4;
            cur = lab;
        }
        const int64_t idx = ((int64_t)b * HW + p) * C + c4 * 4;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + idx);
        const float nz = noise ? nw * noise[(int64_t)b * noise_bstride + p] : 0.f;
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool pos = yv[e] > 0.f;
            g[e] = pos ? dv[e] * g_pos : (dv[e] * alpha) * gain;          // the operation order of fused_bias_act_kernel (bit-equal gz)
            const float z = yv[e] * (pos ? inv_pos : inv_neg);
            acc[e] += g[e] * (z - nz - bs[e]);
        }
        *reinterpret_cast<f32x4*>(gz + idx) = g;
    }
    mine[cur * c4n] += acc;
    __syncthreads();
    for (int t = threadIdx.x; t < R * c4n; t += 256) {
        f32x4 s4 = tab[t];
        for (int l = 1; l < lanes; ++l) s4 += tab[(size_t)l * R * c4n + t];
        const int r = t / c4n, cc = t - r * c4n;
        *reinterpret_cast<f32x4*>(part + (int64_t)split * pstride + ((int64_t)b * R + r) * C + cc * 4) = s4;
    }
}

// dws[b*R+r, c, ci] += sum_{p in r} drgb[b,c,p] * x[p,ci]
__global__ void torgb_bwd_w_kernel(const float* __restrict__ drgb, const float* __restrict__ x,
                                   const uint8_t* __restrict__ labels, int Hm, int Wm, int R,
                                   float* __restrict__ dws, int H, int W, int C, int nsplit, int64_t pstride) {
    extern __shared__ float sums[];        // [NPG][R][3][64]
    const int cw = C >= 64 ? 64 : C;
    const int slabs = C / cw;
    const int split = blockIdx.x % nsplit;
    const int slab = (blockIdx.x / nsplit) % slabs;
    const int b = blockIdx.x / (nsplit * slabs);
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < NPG * R * 3 * 64; t += blockDim.x) sums[t] = 0.f;
    __syncthreads();
    const bool live = cl < cw;
    const int HW = H * W, c = slab * cw + (live ? cl : 0);
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(p0 + per, HW);
    float* mine = sums + (pg * R) * 3 * 64 + cl;
    for (int p = p0 + pg; p < p1; p += NPG) {
        const int yy = p / W, xx = p - yy * W;
        int lab = 0;
        if (labels) lab = labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
        const float xv = x[((int64_t)b * HW + p) * C + c];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            if (live) mine[(lab * 3 + ch) * 64] += drgb[((int64_t)b * 3 + ch) * HW + p] * xv;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < R * 3 * 64; t += blockDim.x) {
        const int cc = t % 64, ch = (t / 64) % 3, r = t / 192;
        if (cc >= cw) continue;
        float s = 0.f;
        for (int j = 0; j < NPG; ++j) s += sums[((j * R + r) * 3 + ch) * 64 + cc];
        dws[(int64_t)split * pstride + (((int64_t)b * R + r) * 3 + ch) * C + slab * cw + cc] = s;
    }
}

// dx[p, ci] (+)= sum_c drgb[b,c,p] * ws[g(p), c, ci]
__global__ void torgb_bwd_x_kernel(const float* __restrict__ drgb, const float* __restrict__ ws,
                                   const uint8_t* __restrict__ labels, int Hm, int Wm, int R, float* __restrict__ dx,
                                   int B, int H, int W, int C, int accumulate) {
    const int C4 = C / 4;
    const int64_t HW = (int64_t)H * W, n = (int64_t)B * HW * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    const int64_t pix = i / C4;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (int64_t)b * HW);
    int g = b;
    if (labels) {
        const int yy = rem / W, xx = rem - yy * W;
        g = b * R + labels[((int64_t)b * Hm + nearest_src(yy, Hm, H)) * Wm + nearest_src(xx, Wm, W)];
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) o = *reinterpret_cast<const f32x4*>(dx + i * 4);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float gch = drgb[((int64_t)b * 3 + ch) * HW + rem];
        o += gch * *reinterpret_cast<const f32x4*>(ws + ((size_t)g * 3 + ch) * C + c);
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// Weight-gradient operands (config 5, train_G=True): the contraction itself is a plain [C1 x P] x [P x C2] GEMM over
// all pixels and goes to the BLAS; these kernels build its two operands with the region-dependent scales folded in.
//   out[b, a, c] = tab[(b*R + lab(out pixel of anchor a)) * C + c] * in[b, a*is + (dy, dx), c]     (0 outside the image)
// anchors a on a [Ha, Wa] grid; out pixel = a*os + (py, px) on the [Ha*os, Wa*os] grid decides the region.
__global__ void shift_scale_kernel(const float* __restrict__ in, const float* __restrict__ tab,
                                   const uint8_t* __restrict__ labels, int Hm, int Wm, int R, float* __restrict__ out,
                                   int B, int Ha, int Wa, int Hi, int Wi, int C, int istride, int dy, int dx, int os,
                                   int py, int px) {
    const int C4 = C / 4;
    const int64_t n = (int64_t)B * Ha * Wa * C4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    int64_t r = i / C4;
    const int ax = (int)(r % Wa); r /= Wa;
    const int ay = (int)(r % Ha);
    const int b = (int)(r / Ha);
    const int iy = ay * istride + dy, ix = ax * istride + dx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
        v = *reinterpret_cast<const f32x4*>(in + (((int64_t)b * Hi + iy) * Wi + ix) * C + c);
        if (tab) {
            int g = b;
            if (labels) {
                const int oy = ay * os + py, ox = ax * os + px;
                g = b * R + labels[((int64_t)b * Hm + nearest_src(oy, Hm, Ha * os)) * Wm + nearest_src(ox, Wm, Wa * os)];
            }
            v *= *reinterpret_cast<const f32x4*>(tab + (size_t)g * C + c);
        }
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

int seg_nsplit(int B, int H, int W, int C) {
    const int slabs = C >= 64 ? C / 64 : 1;
    int nsplit = 1024 / (B * slabs);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > (H * W) / 64) nsplit = (H * W) / 64 > 0 ? (H * W) / 64 : 1;
    return nsplit;
}

}  // namespace

/* number of pixel splits (= partial slots per output element) of e4s_demod_grad_f32 / e4s_torgb_bwd_w_f32 */
extern "C" int e4s_seg_reduce_nsplit(int B, int H, int W, int C) { return seg_nsplit(B, H, W, C); }

extern "C" int e4s_shift_scale_f32(const float* in, const float* tab, const uint8_t* labels, int Hm, int Wm, int R,
                                   float* out, int B, int Ha, int Wa, int Hi, int Wi, int C, int istride, int dy, int dx,
                                   int os, int py, int px, void* stream) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * Ha * Wa * (C / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(shift_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), in, tab,
                       labels, Hm, Wm, R, out, B, Ha, Wa, Hi, Wi, C, istride, dy, dx, os, py, px);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_demod_grad_f32(const float* gz, const float* y, const float* noise, const float* noise_w,
                                  int64_t noise_bstride, const float* bias, float alpha, float gain,
                                  const uint8_t* labels, int Hm, int Wm, int R, float* dd, float* ws, int B, int H,
                                  int W, int C, void* stream) {
    if (C % 32 || (C > 64 && C % 64) || R < 1 || R > 16) return (int)hipErrorInvalidValue;
    if (!ws) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int slabs = C >= 64 ? C / 64 : 1;
    const int nsplit = seg_nsplit(B, H, W, C);
    const int64_t n = (int64_t)B * R * C;
    hipLaunchKernelGGL(demod_grad_kernel, dim3(B * slabs * nsplit), dim3(64 * NPG), sizeof(float) * NPG * R * 64,
                       st, gz, y, noise, noise_w, noise_bstride, bias, alpha, gain, labels, Hm, Wm, R, ws, H, W, C, nsplit, n);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, dd, nsplit, n, 1.f, stream);
}

extern "C" int e4s_act_bwd_demod_nsplit(int B, int H, int W, int C) {
    // every thread walks >= 8 pixels; <= 1024 splits (e4s_reduce_parts_f32's two ordered levels), ~2048 blocks in all
    const int lanes = C >= 4 ? 256 / (C / 4 > 256 ? 256 : C / 4) : 1;
    int nsplit = (H * W) / (8 * (lanes > 0 ? lanes : 1));
    const int cap = 2048 / (B > 0 ? B : 1);
    if (nsplit > cap) nsplit = cap;
    if (nsplit > 1024) nsplit = 1024;
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}

extern "C" int e4s_act_bwd_demod_f32(const float* dy, const float* y, float* gz, const float* noise, const float* noise_w,
                                     int64_t noise_bstride, const float* bias, float alpha, float gain, const uint8_t* labels, int Hm,
                                     int Wm, int R, float* dd, float* ws, int B, int H, int W, int C, void* stream) {
    if (!dy || !y || !gz || !dd || !ws || C % 4 || C < 4 || C > 1024 || 256 % (C / 4) || R < 1 || R > 16 || B <= 0)
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int nsplit = e4s_act_bwd_demod_nsplit(B, H, W, C);
    const int64_t n = (int64_t)B * R * C;
    hipLaunchKernelGGL(act_bwd_demod_kernel, dim3(B * nsplit), dim3(256), (size_t)4096 * R, st, dy, y, gz, noise, noise_w, noise_bstride,
                       bias, alpha, gain, labels, Hm, Wm, R, ws, H, W, C, nsplit, n);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, dd, nsplit, n, 1.f, stream);
}

extern "C" int e4s_torgb_bwd_w_f32(const float* drgb, const float* x, const uint8_t* labels, int Hm, int Wm, int R,
                                   float* dws, float* ws, int B, int H, int W, int C, void* stream) {
    if (C % 32 || (C > 64 && C % 64) || R < 1 || R > 16) return (int)hipErrorInvalidValue;
    if (!ws) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int slabs = C >= 64 ? C / 64 : 1;
    const int nsplit = seg_nsplit(B, H, W, C);
    const int64_t n = (int64_t)B * R * 3 * C;
    hipLaunchKernelGGL(torgb_bwd_w_kernel, dim3(B * slabs * nsplit), dim3(64 * NPG),
                       sizeof(float) * NPG * R * 3 * 64, st, drgb, x, labels, Hm, Wm, R, ws, H, W, C, nsplit, n);
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, dws, nsplit, n, 1.f, stream);
}

extern "C" int e4s_torgb_bwd_x_f32(const float* drgb, const float* ws, const uint8_t* labels, int Hm, int Wm, int R,
                                   float* dx, int B, int H, int W, int C, int accumulate, void* stream) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(torgb_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), drgb, ws,
                       labels, Hm, Wm, R, dx, B, H, W, C, accumulate);
    E4S_CHECK_LAUNCH();
    return 0;
}
