// Style prologue of the modulated convolution (src/models/stylegan2/model.py:276-281):
//   s[g, ci]  = EqualLinear(style[g])            -> e4s_rowdot_f32 mode 0
//   d[g, co]  = rsqrt(sum_{ci,k} (scale*W*s)^2 + 1e-8) = rsqrt(scale^2 * sum_ci s^2 * Wsq[co,ci] + 1e-8)
//                                                -> e4s_rowdot_f32 mode 1 (returns scale*d)
// The reference materialises a [B,Cout,Cin,3,3] tensor per (sample, region, layer) for this
// (9.4 MB at 512x512); here it is two tiny wave-reduced GEMVs per layer over cached Wsq.
#include "common.h"

namespace {

// one wave per (output o, chunk of 8 groups): the wave reads M[o, :] once and reuses it for its groups
constexpr int GCHUNK = 8;
constexpr int RD_OW = 4;          // outputs per wave in rowdot_multi_kernel
template <int MODE>
__global__ void rowdot_kernel(const float* __restrict__ in, int64_t in_stride, const float* __restrict__ M,
                              const float* __restrict__ bias, float* __restrict__ out, int G, int O, int K,
                              float scale) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= O) return;
    const int g0 = blockIdx.y * GCHUNK;
    const float* mrow = M + (size_t)o * K;
    float acc[GCHUNK];
#pragma unroll
    for (int j = 0; j < GCHUNK; ++j) acc[j] = 0.f;
    for (int i = lane * 4; i < K; i += 256) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mrow + i);
#pragma unroll
        for (int j = 0; j < GCHUNK; ++j) {
            if (g0 + j < G) {
                f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)(g0 + j) * in_stride + i);
                if (MODE == 1) x *= x;
                acc[j] += m[0] * x[0] + m[1] * x[1] + m[2] * x[2] + m[3] * x[3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GCHUNK; ++j) {
        if (g0 + j < G) {
            const float a = wave_sum(acc[j]);
            if (lane == 0) {
                float r;
                if (MODE == 0) r = a * scale + (bias ? bias[o] : 0.f);
                else r = scale * rsqrtf(scale * scale * a + 1e-8f);
                out[(size_t)(g0 + j) * O + o] = r;
            }
        }
    }
}

// The same two GEMVs for EVERY styled layer of the generator in one launch each (blockIdx.z = job): the styles of all layers are
// known before the first conv runs (model.py:576-667 reads latent[:, :, i] per layer), so the ~43 rowdot launches of a forward
// (17 s + 17 d + 9 ToRGB s; ~10 us each, 0.46 ms of a batch-8 step, 0.3 ms of a 5.9 ms single swap) collapse into two.
// in = in_base + job.in_off, out = out_base + job.out_off; blocks beyond a job's O / G exit.
template <int MODE>
__global__ void rowdot_multi_kernel(const e4s_rowdot_job* __restrict__ jobs, const float* __restrict__ in_base,
                                    float* __restrict__ out_base) {
    // a wave owns RD_OW consecutive outputs o and GCHUNK rows g: every input vector it loads is used RD_OW times (one output per wave re-read the
    // 96 x 512 inputs of a job 512 times out of L2: 2.6 GB per launch, 114 us); each output's sum is formed exactly as before
    const e4s_rowdot_job jb = jobs[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int o0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RD_OW;
    const int g0 = blockIdx.y * GCHUNK;
    if (o0 >= jb.O || g0 >= jb.G) return;
    const float* in = in_base + jb.in_off;
    float* out = out_base + jb.out_off;
    const int G = jb.G, O = jb.O, K = jb.K;
    const float scale = jb.scale;
    float acc[RD_OW][GCHUNK];
#pragma unroll
    for (int w = 0; w < RD_OW; ++w)
#pragma unroll
        for (int j = 0; j < GCHUNK; ++j) acc[w][j] = 0.f;
    for (int i = lane * 4; i < K; i += 256) {
        f32x4 m[RD_OW];
#pragma unroll
        for (int w = 0; w < RD_OW; ++w)
            m[w] = o0 + w < O ? *reinterpret_cast<const f32x4*>(jb.M + (size_t)(o0 + w) * K + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GCHUNK; ++j) {
            if (g0 + j < G) {
                f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)(g0 + j) * jb.in_stride + i);
                if (MODE == 1) x *= x;
#pragma unroll
                for (int w = 0; w < RD_OW; ++w) acc[w][j] += m[w][0] * x[0] + m[w][1] * x[1] + m[w][2] * x[2] + m[w][3] * x[3];
            }
        }
    }
#pragma unroll
    for (int w = 0; w < RD_OW; ++w) {
        const int o = o0 + w;
        if (o >= O) break;
#pragma unroll
        for (int j = 0; j < GCHUNK; ++j) {
            if (g0 + j < G) {
                const float a = wave_sum(acc[w][j]);
                if (lane == 0) {
                    float r;
                    if (MODE == 0) r = a * scale + (jb.bias ? jb.bias[o] : 0.f);
                    else r = scale * rsqrtf(scale * scale * a + 1e-8f);
                    out[(size_t)(g0 + j) * O + o] = r;
                }
            }
        }
    }
}

__global__ void weight_sqsum_kernel(const float* __restrict__ w, float* __restrict__ wsq, int64_t n, int taps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) {
        const float v = w[i * taps + t];
        acc += v * v;
    }
    wsq[i] = acc;
}

__global__ void rgb_weights_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ ws,
                                   int G, int cin, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)G * 3 * cin;
    if (i >= n) return;
    const int ci = (int)(i % cin);
    const int c = (int)((i / cin) % 3);
    const int g = (int)(i / (3 * (int64_t)cin));
    ws[i] = scale * w[c * cin + ci] * s[(size_t)g * cin + ci];
}

// Polyphase re-packing of an up-sampling 3x3 weight: conv_transpose2d(stride 2) followed by the 4x4 blur
// (model.py:287-300) == for output phase (py,px) a 3x3 correlation over the input grid with
//   out[ph][ey*3+ex][co][ci] = E[py - 2(ey-1)][px - 2(ex-1)],  E[ty][tx] = sum_j kflip[jy][jx] W[ty+jy-1][tx+jx-1]
__global__ void polyphase_weights_kernel(const float* __restrict__ w, const float* __restrict__ k4,
                                         float* __restrict__ out, int cout, int cin) {
    const int64_t n = (int64_t)cout * cin;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wv[9], kf[16];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = w[i * 9 + t];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = k4[15 - t];                       // flipped: upfirdn2d is a true convolution
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
#pragma unroll
        for (int ey = 0; ey < 3; ++ey) {
#pragma unroll
            for (int ex = 0; ex < 3; ++ex) {
                const int ty = py - 2 * (ey - 1), tx = px - 2 * (ex - 1);
                float acc = 0.f;
#pragma unroll
                for (int jy = 0; jy < 4; ++jy) {
                    const int ky = ty + jy - 1;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const int kx = tx + jx - 1;
                        if (kx < 0 || kx > 2) continue;
                        acc += kf[jy * 4 + jx] * wv[ky * 3 + kx];
                    }
                }
                out[((int64_t)(ph * 9 + ey * 3 + ex)) * n + i] = acc;
            }
        }
    }
}

// The TRANSPOSE of polyphase_weights_kernel (config 5, train_G: the weight gradient of an up-sampling StyledConv arrives as the gradient of
// its 4 x 9 polyphase kernels, one e4s_conv_wgrad_f32 call per output phase): deff [4*9][Cout][Cin] -> dw [Cout][Cin][9] with
//   dw[k] = sum_{ph, e} C[ph, e, k] * deff[ph * 9 + e],   C[ph, e, k] = kflip[jy][jx] where W[ky][kx] enters E[ty][tx] (same index algebra).
// One thread per (co, ci), every term added in a fixed order.  (Was a [9 x 36] x [36 x Cout*Cin] library GEMM.)
__global__ void polyphase_fold_kernel(const float* __restrict__ deff, const float* __restrict__ k4, float* __restrict__ dw, int cout,
                                      int cin) {
    const int64_t n = (int64_t)cout * cin;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float kf[16], acc[9];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = k4[15 - t];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
#pragma unroll
        for (int ey = 0; ey < 3; ++ey) {
#pragma unroll
            for (int ex = 0; ex < 3; ++ex) {
                const int ty = py - 2 * (ey - 1), tx = px - 2 * (ex - 1);
                const float g = deff[((int64_t)(ph * 9 + ey * 3 + ex)) * n + i];
#pragma unroll
                for (int jy = 0; jy < 4; ++jy) {
                    const int ky = ty + jy - 1;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const int kx = tx + jx - 1;
                        if (kx < 0 || kx > 2) continue;
                        acc[ky * 3 + kx] += kf[jy * 4 + jx] * g;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[i * 9 + t] = acc[t];
}

// [Cout,Cin,kh*kw] -> [kh*kw][Cout][Cin]
__global__ void pack_taps_kernel(const float* __restrict__ w, float* __restrict__ out, int64_t n, int taps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int t = 0; t < taps; ++t) out[(int64_t)t * n + i] = w[i * taps + t];
}

}  // namespace

extern "C" int e4s_polyphase_weights_f32(const float* w, const float* k4, float* out, int cout, int cin, void* stream) {
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(polyphase_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, k4, out, cout, cin);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_polyphase_fold_f32(const float* deff, const float* k4, float* dw, int cout, int cin, void* stream) {
    if (!deff || !k4 || !dw || cout <= 0 || cin <= 0) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(polyphase_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), deff, k4, dw, cout, cin);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_pack_taps_f32(const float* w, float* out, int cout, int cin, int taps, void* stream) {
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(pack_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, out, n, taps);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_rowdot_f32(const float* in, int64_t in_stride, const float* M, const float* bias, float* out,
                              int G, int O, int K, int mode, float scale, void* stream) {
    if (K % 4 || (mode != 0 && mode != 1)) return (int)hipErrorInvalidValue;
    if (G <= 0 || O <= 0) return 0;
    const int waves = 4;
    dim3 grid((O + waves - 1) / waves, (G + GCHUNK - 1) / GCHUNK), block(64 * waves);
    if (mode == 0)
        hipLaunchKernelGGL(rowdot_kernel<0>, grid, block, 0, as_stream(stream), in, in_stride, M, bias, out, G, O, K, scale);
    else
        hipLaunchKernelGGL(rowdot_kernel<1>, grid, block, 0, as_stream(stream), in, in_stride, M, bias, out, G, O, K, scale);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_rowdot_multi_f32(const e4s_rowdot_job* jobs, int njobs, const float* in_base, float* out_base, int max_O,
                                    int max_G, int mode, void* stream) {
    if (!jobs || njobs <= 0 || (mode != 0 && mode != 1) || max_O <= 0 || max_G <= 0 || njobs > 65535) return (int)hipErrorInvalidValue;
    const int waves = 4;
    dim3 grid((max_O + waves * RD_OW - 1) / (waves * RD_OW), (max_G + GCHUNK - 1) / GCHUNK, njobs), block(64 * waves);
    if (mode == 0) hipLaunchKernelGGL(rowdot_multi_kernel<0>, grid, block, 0, as_stream(stream), jobs, in_base, out_base);
    else hipLaunchKernelGGL(rowdot_multi_kernel<1>, grid, block, 0, as_stream(stream), jobs, in_base, out_base);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_weight_sqsum_f32(const float* w, float* wsq, int cout, int cin, int taps, void* stream) {
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(weight_sqsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, wsq, n, taps);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_rgb_weights_f32(const float* w, const float* s, float* ws, int G, int cin, float scale, void* stream) {
    const int64_t n = (int64_t)G * 3 * cin;
    hipLaunchKernelGGL(rgb_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, s, ws, G, cin, scale);
    E4S_CHECK_LAUNCH();
    return 0;
}
