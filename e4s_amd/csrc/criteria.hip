// Loss networks of the optimisation loop (SURVEY.md 8(f) "N3": scripts/optimization.py:88-122 -- LPIPS-AlexNet at three
// scales, src/criteria/lpips/lpips.py:29-35, and the IR-SE50 identity loss, src/criteria/id_loss.py:24-57): the pieces
// that are not convolutions.  The 3x3 / 5x5 / 1x1 contractions of both networks run on e4s_conv_mfma_f32 /
// e4s_conv_bf16x3_f32; here are the image pooling in front of them, the 3-channel stem convs (forward, and the gradient
// back to the image), max pooling, the LPIPS distance layer, frozen-statistics normalisation backward, and the cosine
// terms.  NHWC activations; every reduction has a fixed order (no floating-point atomics), so the image gradient is
// bit-reproducible.
#include "common.h"

namespace {

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- adaptive average pooling of a crop (F.adaptive_avg_pool2d bins: start = floor(o*in/out), end = ceil((o+1)*in/out)) --
__device__ __forceinline__ int bin_lo(int o, int in, int out) { return (int)(((int64_t)o * in) / out); }
__device__ __forceinline__ int bin_hi(int o, int in, int out) { return (int)((((int64_t)o + 1) * in + out - 1) / out); }

struct PoolGeom {
    int B, C, Hi, Wi, y0, x0, Hc, Wc, Ho, Wo, in_nchw;
};

__device__ __forceinline__ int64_t img_index(const PoolGeom& g, int b, int c, int y, int x) {
    return g.in_nchw ? (((int64_t)b * g.C + c) * g.Hi + y) * g.Wi + x : (((int64_t)b * g.Hi + y) * g.Wi + x) * g.C + c;
}

// y NHWC [B,Ho,Wo,C] = mean over the bin * scale[c] + shift[c]
__global__ void adaptive_pool_kernel(const float* __restrict__ x, float* __restrict__ y, PoolGeom g,
                                     const float* __restrict__ scale, const float* __restrict__ shift, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % g.C);
    const int ox = (int)((i / g.C) % g.Wo);
    const int oy = (int)((i / ((int64_t)g.C * g.Wo)) % g.Ho);
    const int b = (int)(i / ((int64_t)g.C * g.Wo * g.Ho));
    const int ys = bin_lo(oy, g.Hc, g.Ho), ye = bin_hi(oy, g.Hc, g.Ho);
    const int xs = bin_lo(ox, g.Wc, g.Wo), xe = bin_hi(ox, g.Wc, g.Wo);
    float acc = 0.f;
    for (int yy = ys; yy < ye; ++yy)
        for (int xx = xs; xx < xe; ++xx) acc += x[img_index(g, b, c, g.y0 + yy, g.x0 + xx)];
    float v = acc / (float)((ye - ys) * (xe - xs));
    if (scale) v = v * scale[c] + (shift ? shift[c] : 0.f);
    y[i] = v;
}

// dx[b,c,y,x] (+)= sum over the bins that contain the pixel of dy * scale[c] / |bin|; 0 outside the crop
__global__ void adaptive_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, PoolGeom g,
                                         const float* __restrict__ scale, int accumulate, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b, c, y, x;
    if (g.in_nchw) {
        x = (int)(i % g.Wi);
        y = (int)((i / g.Wi) % g.Hi);
        c = (int)((i / ((int64_t)g.Wi * g.Hi)) % g.C);
        b = (int)(i / ((int64_t)g.Wi * g.Hi * g.C));
    } else {
        c = (int)(i % g.C);
        x = (int)((i / g.C) % g.Wi);
        y = (int)((i / ((int64_t)g.C * g.Wi)) % g.Hi);
        b = (int)(i / ((int64_t)g.C * g.Wi * g.Hi));
    }
    const int yc = y - g.y0, xc = x - g.x0;
    float acc = 0.f;
    if ((unsigned)yc < (unsigned)g.Hc && (unsigned)xc < (unsigned)g.Wc) {
        const int oy_e = (int)(((int64_t)yc * g.Ho) / g.Hc), ox_e = (int)(((int64_t)xc * g.Wo) / g.Wc);
        for (int oy = max(oy_e - 1, 0); oy <= min(oy_e + 1, g.Ho - 1); ++oy) {
            const int ys = bin_lo(oy, g.Hc, g.Ho), ye = bin_hi(oy, g.Hc, g.Ho);
            if (yc < ys || yc >= ye) continue;
            for (int ox = max(ox_e - 1, 0); ox <= min(ox_e + 1, g.Wo - 1); ++ox) {
                const int xs = bin_lo(ox, g.Wc, g.Wo), xe = bin_hi(ox, g.Wc, g.Wo);
                if (xc < xs || xc >= xe) continue;
                acc += dy[(((int64_t)b * g.Ho + oy) * g.Wo + ox) * g.C + c] / (float)((ye - ys) * (xe - xs));
            }
        }
        if (scale) acc *= scale[c];
    }
    dx[i] = accumulate ? dx[i] + acc : acc;
}

// ---- stem convs with 3 (<= 4) image channels ------------------------------------------------------------------
// forward: x NHWC [B,Hi,Wi,Cin<=4], wp [k*k*Cin][Cout] -> y NHWC [B,Ho,Wo,Cout], y = act(conv + bias).  One wave = 64
// output pixels x 16 output channels; the packed weights live in LDS and are read as broadcasts.
__global__ void conv_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                     float* __restrict__ y, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int k,
                                     int stride, int pad, int relu) {
    extern __shared__ float s_w[];                     // [k*k*Cin][Cout]
    const int K = k * k * Cin;
    for (int i = threadIdx.x; i < K * Cout / 4; i += blockDim.x)
        reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(wp)[i];
    __syncthreads();
    const int groups = Cout / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int64_t npix = (int64_t)B * Ho * Wo;
    // block -> 64 pixels; its waves stride over the channel groups
    const int64_t pix = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = pix < npix;
    const int64_t pp = ok ? pix : 0;
    const int ox = (int)(pp % Wo), oy = (int)((pp / Wo) % Ho), b = (int)(pp / ((int64_t)Wo * Ho));
    for (int g = wave; g < groups; g += waves) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[g * 16 + j] : 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad + ky;
            if ((unsigned)iy >= (unsigned)Hi) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad + kx;
                const bool in = (unsigned)ix < (unsigned)Wi;
                const float* xp = x + (((int64_t)b * Hi + iy) * Wi + (in ? ix : 0)) * Cin;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float xv = in ? xp[ci] : 0.f;
                    const float* wr = s_w + (size_t)((ky * k + kx) * Cin + ci) * Cout + g * 16;
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + j4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j4 * 4 + e] = fmaf(xv, w4[e], acc[j4 * 4 + e]);
                    }
                }
            }
        }
        if (ok) {
            float* yp = y + pix * Cout + g * 16;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[j4 * 4 + e];
                    o[e] = (relu && v < 0.f) ? 0.f : v;
                }
                *reinterpret_cast<f32x4*>(yp + j4 * 4) = o;
            }
        }
    }
}

// gradient back to the image: dx[b,y,x,ci] = sum_{oy,ox,co} dy[b,oy,ox,co] * w[(ky,kx,ci)][co], ky = y + pad - oy*stride.
// One thread per image pixel; only the <= ceil(k/stride)^2 output positions that see the pixel are visited.
__global__ void conv_smallcin_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wp, float* __restrict__ dx,
                                         int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int k, int stride, int pad,
                                         int64_t npix) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int x = (int)(i % Wi), y = (int)((i / Wi) % Hi), b = (int)(i / ((int64_t)Wi * Hi));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // oy*stride <= y + pad <= oy*stride + k - 1
    const int oy_hi = min((y + pad) / stride, Ho - 1), ox_hi = min((x + pad) / stride, Wo - 1);
    const int oy_lo = max((y + pad - k + stride) / stride, 0), ox_lo = max((x + pad - k + stride) / stride, 0);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int ky = y + pad - oy * stride;
        if (ky < 0 || ky >= k) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int kx = x + pad - ox * stride;
            if (kx < 0 || kx >= k) continue;
            const float* gp = dy + (((int64_t)b * Ho + oy) * Wo + ox) * Cout;
            const float* wr = wp + (size_t)((ky * k + kx) * Cin) * Cout;
            for (int co = 0; co < Cout; co += 4) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp + co);
                for (int ci = 0; ci < Cin; ++ci) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + (size_t)ci * Cout + co);
                    acc[ci] += g4[0] * w4[0] + g4[1] * w4[1] + g4[2] * w4[2] + g4[3] * w4[3];
                }
            }
        }
    }
    for (int ci = 0; ci < Cin; ++ci) dx[i * Cin + ci] = acc[ci];
}

// col2im of the GEMM form of the same gradient (large kernels: AlexNet's 11x11 / 4 stem): z[b,oy,ox,(ky*k+kx)*Cin+ci] = dy[b,oy,ox,:] . w[(ky,kx,ci)][:]
// comes out of ONE 1x1 contraction on the MFMA conv kernel (M = positions, N = k*k*Cin padded to a multiple of 32, K = Cout); here every image
// pixel adds the <= ceil(k/stride)^2 entries that land on it, in (oy, ox) order.  conv_smallcin_bwd_kernel re-reads dy and the weights per
// pixel through L1 (0.85 ms at 1024^2); this pair takes ~0.1 ms.
__global__ void conv_smallcin_col2im_kernel(const float* __restrict__ z, float* __restrict__ dx, int B, int Hi, int Wi, int Cin, int Ho,
                                            int Wo, int ZC, int k, int stride, int pad, int64_t npix) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int x = (int)(i % Wi), y = (int)((i / Wi) % Hi), b = (int)(i / ((int64_t)Wi * Hi));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy_hi = min((y + pad) / stride, Ho - 1), ox_hi = min((x + pad) / stride, Wo - 1);
    const int oy_lo = max((y + pad - k + stride) / stride, 0), ox_lo = max((x + pad - k + stride) / stride, 0);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int ky = y + pad - oy * stride;
        if (ky < 0 || ky >= k) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int kx = x + pad - ox * stride;
            if (kx < 0 || kx >= k) continue;
            const float* zp = z + (((int64_t)b * Ho + oy) * Wo + ox) * ZC + (ky * k + kx) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc[ci] += zp[ci];
        }
    }
    for (int ci = 0; ci < Cin; ++ci) dx[i * Cin + ci] = acc[ci];
}

// ---- MaxPool2d(3, 2) (torchvision AlexNet features[2], [5]) -----------------------------------------------------
__global__ void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int Hi,
                                  int Wi, int Ho, int Wo, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C / 4;
    const int c = (int)(i % C4) * 4;
    const int ox = (int)((i / C4) % Wo), oy = (int)((i / ((int64_t)C4 * Wo)) % Ho);
    const int64_t b = i / ((int64_t)C4 * Wo * Ho);
    f32x4 m;
    unsigned char am[4];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((b * Hi + oy * 2 + ky) * Wi + ox * 2 + kx) * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((ky | kx) == 0 || v[e] > m[e]) {      // first maximum in scan order (ATen's rule)
                    m[e] = v[e];
                    am[e] = (unsigned char)(ky * 3 + kx);
                }
        }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
    *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4(am[0], am[1], am[2], am[3]);
}

__global__ void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                      float* __restrict__ dx, int Hi, int Wi, int Ho, int Wo, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C / 4;
    const int c = (int)(i % C4) * 4;
    const int x = (int)((i / C4) % Wi), y = (int)((i / ((int64_t)C4 * Wi)) % Hi);
    const int64_t b = i / ((int64_t)C4 * Wi * Hi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = max((y - 1) / 2, 0); oy <= min(y / 2, Ho - 1); ++oy) {
        const int ky = y - 2 * oy;
        if (ky > 2) continue;
        for (int ox = max((x - 1) / 2, 0); ox <= min(x / 2, Wo - 1); ++ox) {
            const int kx = x - 2 * ox;
            if (kx > 2) continue;
            const int64_t o = ((b * Ho + oy) * Wo + ox) * C + c;
            const uchar4 a = *reinterpret_cast<const uchar4*>(idx + o);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
            const unsigned char me = (unsigned char)(ky * 3 + kx);
            if (a.x == me) acc[0] += g[0];
            if (a.y == me) acc[1] += g[1];
            if (a.z == me) acc[2] += g[2];
            if (a.w == me) acc[3] += g[3];
        }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
}

// ---- MaxPool2d(2) (parsing UNet encoder, src/criteria/face_parsing/unet.py:27-36) ------------------------------------
__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int Hi,
                                int Wi, int Ho, int Wo, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C / 4;
    const int c = (int)(i % C4) * 4;
    const int ox = (int)((i / C4) % Wo), oy = (int)((i / ((int64_t)C4 * Wo)) % Ho);
    const int64_t b = i / ((int64_t)C4 * Wo * Ho);
    f32x4 m;
    unsigned char am[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((b * Hi + oy * 2 + (k >> 1)) * Wi + ox * 2 + (k & 1)) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k == 0 || v[e] > m[e]) {                  // first maximum in scan order
                m[e] = v[e];
                am[e] = (unsigned char)k;
            }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
    *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4(am[0], am[1], am[2], am[3]);
}

// windows do not overlap: each input element belongs to exactly one (or, on an odd trailing row / column, to none)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                    int Hi, int Wi, int Ho, int Wo, int C, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int C4 = C / 4;
    const int c = (int)(i % C4) * 4;
    const int x = (int)((i / C4) % Wi), y = (int)((i / ((int64_t)C4 * Wi)) % Hi);
    const int64_t b = i / ((int64_t)C4 * Wi * Hi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy = y >> 1, ox = x >> 1;
    if (oy < Ho && ox < Wo) {
        const int64_t o = ((b * Ho + oy) * Wo + ox) * C + c;
        const uchar4 a = *reinterpret_cast<const uchar4*>(idx + o);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
        const unsigned char me = (unsigned char)((y & 1) * 2 + (x & 1));
        if (a.x == me) acc[0] = g[0];
        if (a.y == me) acc[1] = g[1];
        if (a.z == me) acc[2] = g[2];
        if (a.w == me) acc[3] = g[3];
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
}

// dx (+)= dy * [y > 0]   (ReLU backward from the OUTPUT; any channel count % 4)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                int accumulate, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4), v = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? g[e] : 0.f;
    if (accumulate) o += *reinterpret_cast<const f32x4*>(dx + i * 4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// ---- LPIPS distance layer (lpips.py:32-33, utils.py normalize_activation) ---------------------------------------------
// d(pixel) = sum_c w_c (fx_c/(|fx|+eps) - fy_c/(|fy|+eps))^2; one wave per pixel, <= 8 channels per lane (C <= 512).
constexpr int LP_MAXPL = 8;
constexpr float LP_EPS = 1e-10f;
constexpr int LP_PIX_PER_WAVE = 4;       // (round 6: was 16 -- a wave walks its pixels one exposed round trip after the other: 30 us per launch, 15 launches per step)
constexpr int LP_PIX_PER_BLOCK = 4 * LP_PIX_PER_WAVE;     // 4 waves x LP_PIX_PER_WAVE pixels each, summed in pixel order

__global__ void lpips_layer_kernel(const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ w,
                                   double* __restrict__ part, int HW, int C, int blocks_per_img) {
    const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x % blocks_per_img;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double s_d[4];
    double dsum = 0.0;
    for (int k = 0; k < LP_PIX_PER_WAVE; ++k) {
        const int p = blk * LP_PIX_PER_BLOCK + wave * LP_PIX_PER_WAVE + k;
        if (p >= HW) break;
        const float* px = fx + ((int64_t)b * HW + p) * C;
        const float* py = fy + ((int64_t)b * HW + p) * C;
        float vx[LP_MAXPL], vy[LP_MAXPL];
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int j = 0; j < LP_MAXPL; ++j) {
            const int c = lane + j * 64;
            vx[j] = c < C ? px[c] : 0.f;
            vy[j] = c < C ? py[c] : 0.f;
            sx += vx[j] * vx[j];
            sy += vy[j] * vy[j];
        }
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        const float ix = 1.f / (sqrtf(sx) + LP_EPS), iy = 1.f / (sqrtf(sy) + LP_EPS);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < LP_MAXPL; ++j) {
            const int c = lane + j * 64;
            if (c < C) {
                const float t = vx[j] * ix - vy[j] * iy;
                d += w[c] * t * t;
            }
        }
        d = wave_sum(d);
        dsum += (double)d;
    }
    if (lane == 0) s_d[wave] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s_d[0] + s_d[1] + s_d[2] + s_d[3];
}

__global__ void lpips_finalize_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int blocks_per_img,
                                      float scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a = 0.0;
    for (int k = 0; k < blocks_per_img; ++k) a += part[(int64_t)b * blocks_per_img + k];
    out[b] = (float)(a * (double)scale);
}

// dfx (+)= g * d(d)/d(fx):  n = fx/(r+eps);  dn_c = 2 w_c (nx_c - ny_c) g;  dfx = dn/(r+eps) - fx <dn,fx> / (r (r+eps)^2)
__global__ void lpips_layer_bwd_kernel(const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ w,
                                       const float* __restrict__ gout, float gmul, float* __restrict__ dfx, int HW, int C,
                                       int accumulate, int64_t npix) {
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const int lane = threadIdx.x & 63;
    const float g = gout[0] * gmul;
    const float* px = fx + pix * C;
    const float* py = fy + pix * C;
    float vx[LP_MAXPL], vy[LP_MAXPL];
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int j = 0; j < LP_MAXPL; ++j) {
        const int c = lane + j * 64;
        vx[j] = c < C ? px[c] : 0.f;
        vy[j] = c < C ? py[c] : 0.f;
        sx += vx[j] * vx[j];
        sy += vy[j] * vy[j];
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const float rx = sqrtf(sx);
    const float ix = 1.f / (rx + LP_EPS), iy = 1.f / (sqrtf(sy) + LP_EPS);
    float dn[LP_MAXPL];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < LP_MAXPL; ++j) {
        const int c = lane + j * 64;
        dn[j] = c < C ? 2.f * w[c] * (vx[j] * ix - vy[j] * iy) * g : 0.f;
        dot += dn[j] * vx[j];
    }
    dot = wave_sum(dot);
    const float k2 = rx > 0.f ? dot * ix * ix / rx : 0.f;
#pragma unroll
    for (int j = 0; j < LP_MAXPL; ++j) {
        const int c = lane + j * 64;
        if (c < C) {
            const float v = dn[j] * ix - vx[j] * k2;
            dfx[pix * C + c] = accumulate ? dfx[pix * C + c] + v : v;
        }
    }
}

// ---- frozen-statistics normalisation backward (BatchNorm2d in eval mode, helpers.py:108-113 inside IDLoss) ------------
// dx (+)= rstd[b,c] * (gate[b,c] * dy + extra[b,c])
__global__ void norm_bwd_frozen_kernel(const float* __restrict__ dy, const float* __restrict__ stats,
                                       const float* __restrict__ gate, const float* __restrict__ extra, float* __restrict__ dx,
                                       int HW, int C, int accumulate, int64_t n4) {
    const int C4 = C / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const int64_t b = i / ((int64_t)HW * C4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t bc = b * C + c + e;
        o[e] = stats[bc * 2 + 1] * ((gate ? gate[bc] : 1.f) * g[e] + (extra ? extra[bc] : 0.f));
    }
    if (accumulate) o += *reinterpret_cast<const f32x4*>(dx + i * 4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}

// ---- cosine terms of the identity loss (id_loss.py:41-52 on l2-normalised features, helpers.py:14-17) ----------------
// elements per block: 4096, grown so that a row never needs more than 256 partial slots (the finalize pass adds a row's
// slots serially; the parsing UNet's first map has 8.4 M elements per row)
__host__ __device__ inline int64_t cos_chunk(int64_t D) {
    int64_t c = (D + 255) / 256;
    c = (c + 255) / 256 * 256;
    return c < 4096 ? 4096 : c;
}

__global__ void cosine_partial_kernel(const float* __restrict__ a, const float* __restrict__ bv, double* __restrict__ part,
                                      int64_t D, int nchunk) {
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int64_t chunk = cos_chunk(D);
    const int64_t lo = (int64_t)ch * chunk, hi = (lo + chunk < D) ? lo + chunk : D;
    double ab = 0.0, aa = 0.0, bb = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double x = (double)a[(int64_t)b * D + i], y = (double)bv[(int64_t)b * D + i];
        ab += x * y;
        aa += x * x;
        bb += y * y;
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = ab;
    red[1][threadIdx.x] = aa;
    red[2][threadIdx.x] = bb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] += red[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + (int64_t)blockIdx.x * 3;
        o[0] = red[0][0];
        o[1] = red[1][0];
        o[2] = red[2][0];
    }
}

// out[b] = {sim, alpha, beta}: sim = <a,b>/(|a||b|); d(sim)/da = alpha * b + beta * a
__global__ void cosine_finalize_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int nchunk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double ab = 0.0, aa = 0.0, bb = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        const double* p = part + ((int64_t)b * nchunk + k) * 3;
        ab += p[0];
        aa += p[1];
        bb += p[2];
    }
    const double na = sqrt(aa), nb = sqrt(bb);
    out[b * 3] = (float)(ab / (na * nb));
    out[b * 3 + 1] = (float)(1.0 / (na * nb));
    out[b * 3 + 2] = (float)(-ab / (na * na * na * nb));
}

// da (+)= g[b] * (coef[b].alpha * b + coef[b].beta * a)
__global__ void cosine_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bv, const float* __restrict__ coef,
                                  const float* __restrict__ gout, float gmul, float* __restrict__ da, int64_t D, int accumulate,
                                  int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / D;
    const float g = gout[0] * gmul;
    const float v = g * (coef[b * 3 + 1] * bv[i] + coef[b * 3 + 2] * a[i]);
    da[i] = accumulate ? da[i] + v : v;
}

PoolGeom make_geom(int B, int C, int Hi, int Wi, int y0, int x0, int Hc, int Wc, int Ho, int Wo, int in_nchw) {
    PoolGeom g;
    g.B = B; g.C = C; g.Hi = Hi; g.Wi = Wi; g.y0 = y0; g.x0 = x0; g.Hc = Hc; g.Wc = Wc; g.Ho = Ho; g.Wo = Wo;
    g.in_nchw = in_nchw;
    return g;
}

bool geom_ok(const PoolGeom& g) {
    return g.B > 0 && g.C > 0 && g.Ho > 0 && g.Wo > 0 && g.Hc > 0 && g.Wc > 0 && g.y0 >= 0 && g.x0 >= 0 &&
           g.y0 + g.Hc <= g.Hi && g.x0 + g.Wc <= g.Wi;
}

}  // namespace

extern "C" int e4s_adaptive_pool_f32(const float* x, float* y, int B, int C, int Hi, int Wi, int y0, int x0, int Hc, int Wc,
                                     int Ho, int Wo, int in_nchw, const float* scale, const float* shift, void* stream) {
    const PoolGeom g = make_geom(B, C, Hi, Wi, y0, x0, Hc, Wc, Ho, Wo, in_nchw);
    if (!geom_ok(g)) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(adaptive_pool_kernel, grid1(n), dim3(256), 0, as_stream(stream), x, y, g, scale, shift, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_adaptive_pool_bwd_f32(const float* dy, float* dx, int B, int C, int Hi, int Wi, int y0, int x0, int Hc,
                                         int Wc, int Ho, int Wo, int in_nchw, const float* scale, int accumulate,
                                         void* stream) {
    const PoolGeom g = make_geom(B, C, Hi, Wi, y0, x0, Hc, Wc, Ho, Wo, in_nchw);
    if (!geom_ok(g)) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)B * Hi * Wi * C;
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel, grid1(n), dim3(256), 0, as_stream(stream), dy, dx, g, scale, accumulate, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv_smallcin_f32(const float* x, const float* wp, const float* bias, float* y, int B, int Hi, int Wi,
                                     int Cin, int Ho, int Wo, int Cout, int k, int stride, int pad, int relu, void* stream) {
    if (Cin < 1 || Cin > 4 || Cout % 16 || k < 1 || stride < 1) return (int)hipErrorInvalidValue;
    const int smem = k * k * Cin * Cout * (int)sizeof(float);
    if (smem > 160 * 1024 - 512 || (k * k * Cin * Cout) % 4) return (int)hipErrorInvalidValue;
    static std::atomic<uint64_t> attr_mask{0};
    if (smem > 64 * 1024) {
        const int e = e4s_ensure_dyn_smem(reinterpret_cast<const void*>(conv_smallcin_kernel), smem, attr_mask);
        if (e) return e;
    }
    const int64_t npix = (int64_t)B * Ho * Wo;
    hipLaunchKernelGGL(conv_smallcin_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), smem, as_stream(stream), x, wp,
                       bias, y, B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, pad, relu);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv_smallcin_bwd_f32(const float* dy, const float* wp, float* dx, int B, int Hi, int Wi, int Cin, int Ho,
                                         int Wo, int Cout, int k, int stride, int pad, void* stream) {
    if (Cin < 1 || Cin > 4 || Cout % 4 || k < 1 || stride < 1) return (int)hipErrorInvalidValue;
    const int64_t npix = (int64_t)B * Hi * Wi;
    hipLaunchKernelGGL(conv_smallcin_bwd_kernel, grid1(npix), dim3(256), 0, as_stream(stream), dy, wp, dx, B, Hi, Wi, Cin, Ho,
                       Wo, Cout, k, stride, pad, npix);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_conv_smallcin_col2im_f32(const float* z, float* dx, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int ZC, int k,
                                            int stride, int pad, void* stream) {
    if (!z || !dx || Cin < 1 || Cin > 4 || k < 1 || stride < 1 || ZC < k * k * Cin) return (int)hipErrorInvalidValue;
    const int64_t npix = (int64_t)B * Hi * Wi;
    if (npix <= 0) return 0;
    hipLaunchKernelGGL(conv_smallcin_col2im_kernel, grid1(npix), dim3(256), 0, as_stream(stream), z, dx, B, Hi, Wi, Cin, Ho, Wo, ZC, k,
                       stride, pad, npix);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_maxpool3s2_f32(const float* x, float* y, unsigned char* idx, int B, int Hi, int Wi, int C, void* stream) {
    if (C % 4 || Hi < 3 || Wi < 3) return (int)hipErrorInvalidValue;
    const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3s2_kernel, grid1(n4), dim3(256), 0, as_stream(stream), x, y, idx, Hi, Wi, Ho, Wo, C, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_maxpool3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int B, int Hi, int Wi, int C,
                                      void* stream) {
    if (C % 4 || Hi < 3 || Wi < 3) return (int)hipErrorInvalidValue;
    const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, grid1(n4), dim3(256), 0, as_stream(stream), dy, idx, dx, Hi, Wi, Ho, Wo, C, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_maxpool2_f32(const float* x, float* y, unsigned char* idx, int B, int Hi, int Wi, int C, void* stream) {
    if (C % 4 || Hi < 2 || Wi < 2) return (int)hipErrorInvalidValue;
    const int Ho = Hi / 2, Wo = Wi / 2;
    const int64_t n4 = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool2_kernel, grid1(n4), dim3(256), 0, as_stream(stream), x, y, idx, Hi, Wi, Ho, Wo, C, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_maxpool2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int B, int Hi, int Wi, int C,
                                    void* stream) {
    if (C % 4 || Hi < 2 || Wi < 2) return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, grid1(n4), dim3(256), 0, as_stream(stream), dy, idx, dx, Hi, Wi, Hi / 2, Wi / 2, C,
                       n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, int accumulate, void* stream) {
    if (n % 4) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(relu_bwd_kernel, grid1(n / 4), dim3(256), 0, as_stream(stream), dy, y, dx, accumulate, n / 4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_lpips_layer_ws_doubles(int B, int HW) {
    return (int64_t)B * ((HW + LP_PIX_PER_BLOCK - 1) / LP_PIX_PER_BLOCK);
}

extern "C" int e4s_lpips_layer_f32(const float* fx, const float* fy, const float* w, float* out, double* ws, int B, int HW,
                                   int C, void* stream) {
    if (C < 1 || C > 64 * LP_MAXPL || B < 1 || HW < 1) return (int)hipErrorInvalidValue;
    const int bpi = (HW + LP_PIX_PER_BLOCK - 1) / LP_PIX_PER_BLOCK;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(lpips_layer_kernel, dim3((unsigned)(B * bpi)), dim3(256), 0, st, fx, fy, w, ws, HW, C, bpi);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(lpips_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, st, ws, out, B, bpi, 1.f / (float)HW);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_lpips_layer_bwd_f32(const float* fx, const float* fy, const float* w, const float* gout, float gmul,
                                       float* dfx, int B, int HW, int C, int accumulate, void* stream) {
    if (C < 1 || C > 64 * LP_MAXPL || B < 1 || HW < 1) return (int)hipErrorInvalidValue;
    const int64_t npix = (int64_t)B * HW;
    hipLaunchKernelGGL(lpips_layer_bwd_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, as_stream(stream), fx, fy, w,
                       gout, gmul / (float)HW, dfx, HW, C, accumulate, npix);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_norm_bwd_frozen_f32(const float* dy, const float* stats, const float* gate, const float* extra, float* dx,
                                       int B, int HW, int C, int accumulate, void* stream) {
    if (C % 4) return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * HW * (C / 4);
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(norm_bwd_frozen_kernel, grid1(n4), dim3(256), 0, as_stream(stream), dy, stats, gate, extra, dx, HW, C,
                       accumulate, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t e4s_cosine_ws_doubles(int B, int64_t D) {
    const int64_t chunk = cos_chunk(D);
    return (int64_t)3 * B * ((D + chunk - 1) / chunk);
}

extern "C" int e4s_cosine_f32(const float* a, const float* b, float* out, double* ws, int B, int64_t D, void* stream) {
    if (B < 1 || D < 1) return (int)hipErrorInvalidValue;
    const int nchunk = (int)((D + cos_chunk(D) - 1) / cos_chunk(D));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(cosine_partial_kernel, dim3((unsigned)(B * nchunk)), dim3(256), 0, st, a, b, ws, D, nchunk);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(cosine_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, st, ws, out, B, nchunk);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_cosine_bwd_f32(const float* a, const float* b, const float* coef, const float* gout, float gmul, float* da,
                                  int B, int64_t D, int accumulate, void* stream) {
    const int64_t n = (int64_t)B * D;
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cosine_bwd_kernel, grid1(n), dim3(256), 0, as_stream(stream), a, b, coef, gout, gmul, da, D, accumulate,
                       n);
    E4S_CHECK_LAUNCH();
    return 0;
}
