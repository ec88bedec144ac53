// Device-side stitching of the swapped face back onto the target (the part round 2 left on the host):
//   default path   scripts/face_swap.py:81-97, 306-310  smooth_face_boundry: cv2.erode(11x11, border 255) ->
//                  cv2.GaussianBlur(11x11, sigma 0 -> 2.0) of the uint8 mask -> PIL alpha_composite
//   --lap_bld      src/utils/multi_band_blending.py:4-75  Laplacian pyramid blending (cv2.pyrDown / pyrUp, 10 levels)
// All of it is integer / short fp32 arithmetic on 1024^2 images: HBM-bound streaming kernels, one thread per pixel (or per
// pixel-channel), uint8 HWC in and out as the pipeline's tensor2im images are.
//
// OpenCV (opencv-python 4.7.0.72 in e4s_env.yaml:96) is a third-party dependency that is absent here: its algorithms are
// RESTATED -- erode trivially; GaussianBlur's CV_8U fixed-point path (smooth.dispatch.cpp: 8.8 fixed-point kernel from
// getGaussianKernelFixedPoint_ED, exact integer row pass, exact integer column pass, one rounding (+ 2^15) >> 16); pyrDown /
// pyrUp (pyramids.cpp: [1 4 6 4 1] taps, uint8: (sum + 128) >> 8; fp32: row pass, column pass, * 1/256 resp. 1/64;
// BORDER_REFLECT_101, pyrUp's far edge replicated).  PIL's alpha_composite (libImaging/AlphaComposite.c) IS available and pins
// the composite in the CPU tests.  The OpenCV restatements are "parity unpinned" (oracle/e4s_oracle.py says the same).
#include "common.h"

// Every fp32 expression in this file restates an operation ORDER of OpenCV / numpy (one rounding per operation).  hipcc's default
// -ffp-contract=fast-honor-pragmas may fuse a*b+c into one FMA, so contraction is switched off for the whole translation unit --
// and the arithmetic goes through the helpers BELOW the pragma, not through __fmul_rn / __fadd_rn: those are plain operators in this
// ROCm's headers (__clang_hip_math.h:271) whose inline bodies were parsed BEFORE the pragma, so they carry the `contract` flag and
// `add_rn(a, mul_rn(b, 6.f))` compiled to v_fmamk_f32 (one rounding instead of two).  That was round 4's "unexplained 4 ulp"
// between e4s_pyrup_f32 and the numpy statement of cv2.pyrUp (found in round 5 by reading the ISA for v_fmamk_f32, not only v_fma_f32).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }     // one IEEE rounding each, never fused (see above)
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

__device__ __forceinline__ int reflect101(int i, int n) {        // cv::BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// mask [B,Hm,Wm] fp32 -> uint8 [B,H,W] = 255 * uint8(bilinear(mask))  (face_swap.py:291-294: F.interpolate(..., 'bilinear',
// align_corners=False), then `255 * m.astype(np.uint8)`: the cast truncates BEFORE the multiplication, so only pixels whose
// resized value reaches 1.0 survive; the product wraps modulo 256 as numpy's uint8 arithmetic does)
__global__ void mask_to_u8_kernel(const float* __restrict__ mask, uint8_t* __restrict__ out, int H, int W, int Hm, int Wm,
                                  int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % W);
    const int64_t r = i / W;
    const int oy = (int)(r % H);
    const int64_t b = r / H;
    const float sy = (float)Hm / H, sx = (float)Wm / W;
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hm - 1 ? 1 : 0), x1 = x0 + (x0 < Wm - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* mp = mask + b * Hm * Wm;
    const float m = hy * (hx * mp[(int64_t)y0 * Wm + x0] + lx * mp[(int64_t)y0 * Wm + x1]) +
                    ly * (hx * mp[(int64_t)y1 * Wm + x0] + lx * mp[(int64_t)y1 * Wm + x1]);
    const float c = m < 0.f ? 0.f : (m > 255.f ? 255.f : m);
    out[i] = (uint8_t)(255u * (unsigned)(uint8_t)c);
}

// cv2.erode with a flat (2r+1)^2 element, BORDER_CONSTANT / borderValue: min over the window, outside = border
__global__ void erode_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int r, int border,
                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % W);
    const int64_t q = i / W;
    const int y = (int)(q % H);
    const uint8_t* s = src + (q / H) * (int64_t)H * W;
    int m = 255;
    for (int dy = -r; dy <= r; ++dy) {
        const int yy = y + dy;
        for (int dx = -r; dx <= r; ++dx) {
            const int xx = x + dx;
            const int v = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? s[(int64_t)yy * W + xx] : border;
            m = v < m ? v : m;
        }
    }
    dst[i] = (uint8_t)m;
}

// cv2.GaussianBlur on CV_8U: dst = (sum_j k[j] * (sum_i k[i] * src[y+j-r][x+i-r]) + 2^15) >> 16 with the 8.8 fixed-point taps k
// (sum 256), BORDER_REFLECT_101.  Separable inside the block: 32x8 outputs, (8 + 2r) rows of horizontal sums in LDS.
constexpr int GK_MAX = 31;
struct GaussTaps { int k[GK_MAX]; };

__global__ __launch_bounds__(256) void gauss_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                       int r, const GaussTaps taps) {
    __shared__ int rows[(8 + GK_MAX - 1) * 32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = blockIdx.x * 32 + tx, y0 = blockIdx.y * 8;
    const uint8_t* s = src + (int64_t)blockIdx.z * H * W;
    const int nrows = 8 + 2 * r;
    for (int rr = ty; rr < nrows; rr += 8) {
        const int yy = reflect101(y0 + rr - r, H);
        int acc = 0;
        if (x < W)
            for (int i = -r; i <= r; ++i) acc += taps.k[i + r] * (int)s[(int64_t)yy * W + reflect101(x + i, W)];
        rows[rr * 32 + tx] = acc;                          // <= 255 * 256: the 8.8 fixed-point row sum, exact
    }
    __syncthreads();
    const int y = y0 + ty;
    if (x >= W || y >= H) return;
    unsigned acc = 0;
    for (int j = 0; j <= 2 * r; ++j) acc += (unsigned)taps.k[j] * (unsigned)rows[(ty + j) * 32 + tx];
    const unsigned v = (acc + (1u << 15)) >> 16;
    dst[(int64_t)blockIdx.z * H * W + (int64_t)y * W + x] = (uint8_t)(v > 255u ? 255u : v);
}

// PIL Image.alpha_composite(dst RGBA with alpha 255, src RGBA with alpha a) -> RGB of the result (its alpha is 255):
// libImaging/AlphaComposite.c, integer path with PRECISION_BITS = 7
__device__ __forceinline__ unsigned div255(unsigned a) { return ((a >> 8) + a) >> 8; }

__global__ void alpha_composite_u8_kernel(const uint8_t* __restrict__ face, const uint8_t* __restrict__ target,
                                          const uint8_t* __restrict__ alpha, uint8_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B*H*W pixels
    if (i >= n) return;
    const unsigned a = alpha[i];
    if (a == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = target[i * 3 + c];
        return;
    }
    const unsigned blend = 255u * (255u - a);
    const unsigned outa255 = a * 255u + blend;
    const unsigned coef1 = a * 255u * 255u * (1u << 7) / outa255;
    const unsigned coef2 = 255u * (1u << 7) - coef1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned t = (unsigned)face[i * 3 + c] * coef1 + (unsigned)target[i * 3 + c] * coef2;
        out[i * 3 + c] = (uint8_t)(div255(t + (0x80u << 7)) >> 7);
    }
}

// ---- pyramids (cv::pyrDown / cv::pyrUp, [1 4 6 4 1]) on HWC images ----------------------------------------------------
template <typename T> __device__ __forceinline__ float ld(const T* p) { return (float)*p; }

// dst [B,(H+1)/2,(W+1)/2,C]; uint8: integer (sum + 128) >> 8; fp32: row pass then column pass then * (1/256), as pyramids.cpp
template <typename T>
__global__ void pyrdown_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int C, int Ho, int Wo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int x = (int)(q % Wo);
    q /= Wo;
    const int y = (int)(q % Ho);
    const T* s = src + (q / Ho) * (int64_t)H * W * C + c;
    int xs[5], ys[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { xs[k] = reflect101(2 * x + k - 2, W); ys[k] = reflect101(2 * y + k - 2, H); }
    if (sizeof(T) == 1) {
        const int wt[5] = {1, 4, 6, 4, 1};
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int row = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) row += wt[k] * (int)s[((int64_t)ys[j] * W + xs[k]) * C];
            acc += wt[j] * row;
        }
        dst[i] = (T)((acc + 128) >> 8);
    } else {
        float row[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const T* rp = s + (int64_t)ys[j] * W * C;
            // row[x] = src[2x]*6 + (src[2x-1] + src[2x+1])*4 + src[2x-2] + src[2x+2]
            const float a = mul_rn(ld(rp + (int64_t)xs[2] * C), 6.f);
            const float b4 = mul_rn(add_rn(ld(rp + (int64_t)xs[1] * C), ld(rp + (int64_t)xs[3] * C)), 4.f);
            row[j] = add_rn(add_rn(add_rn(a, b4), ld(rp + (int64_t)xs[0] * C)), ld(rp + (int64_t)xs[4] * C));
        }
        // dst = (row2*6 + (row1 + row3)*4 + row0 + row4) * (1/256)
        const float v = add_rn(add_rn(add_rn(mul_rn(row[2], 6.f), mul_rn(add_rn(row[1], row[3]), 4.f)), row[0]), row[4]);
        dst[i] = (T)mul_rn(v, 1.f / 256.f);
    }
}

// fp32 pyrUp: dst [B,2h,2w,C].  Horizontal: even x: s[x-1] + 6 s[x] + s[x+1], odd: 4 (s[x] + s[x+1]); the near edge reflects
// (s[-1] = s[1]), the far edge replicates (s[w] = s[w-1]) -- both written as pyramids.cpp writes its two edge columns;
// vertical: the generic form on rows (y-1, y, y+1) mapped the same way; scale 1/64.
__device__ __forceinline__ float up_row(const float* rp, int x2, int w, int C) {
    const int x = x2 >> 1;
    const bool odd = x2 & 1;
    if (x == 0) {                                           // pyramids.cpp: t0 = src[x]*6 + src[x+cn]*2, t1 = (src[x] + src[x+cn])*4
        const float s0 = rp[0], s1 = rp[(int64_t)(w > 1 ? 1 : 0) * C];
        return odd ? mul_rn(add_rn(s0, s1), 4.f) : add_rn(mul_rn(s0, 6.f), mul_rn(s1, 2.f));
    }
    if (x == w - 1) {                                       // t0 = src[sx-cn] + src[sx]*7, t1 = src[sx]*8
        const float sm = rp[(int64_t)(x - 1) * C], s0 = rp[(int64_t)x * C];
        return odd ? mul_rn(s0, 8.f) : add_rn(sm, mul_rn(s0, 7.f));
    }
    const float sm = rp[(int64_t)(x - 1) * C], s0 = rp[(int64_t)x * C], sp = rp[(int64_t)(x + 1) * C];
    return odd ? mul_rn(add_rn(s0, sp), 4.f) : add_rn(add_rn(sm, mul_rn(s0, 6.f)), sp);
}

__global__ void pyrup_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int C, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    int64_t q = i / C;
    const int x2 = (int)(q % (2 * w));
    q /= 2 * w;
    const int y2 = (int)(q % (2 * h));
    const float* s = src + (q / (2 * h)) * (int64_t)h * w * C + c;
    const int y = y2 >> 1;
    const int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp = y < h - 1 ? y + 1 : h - 1;
    float v;
    if (y2 & 1) {
        v = mul_rn(add_rn(up_row(s + (int64_t)y * w * C, x2, w, C), up_row(s + (int64_t)yp * w * C, x2, w, C)), 4.f);
    } else {
        v = add_rn(add_rn(up_row(s + (int64_t)ym * w * C, x2, w, C), mul_rn(up_row(s + (int64_t)y * w * C, x2, w, C), 6.f)),
                      up_row(s + (int64_t)yp * w * C, x2, w, C));
    }
    dst[i] = mul_rn(v, 1.f / 64.f);
}

__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ s, float* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}

// one level of the blend (multi_band_blending.py:28-46): ls = (a - ua) * m + (b - ub) * (1 - m); ua / ub null at the coarsest
// level (ls = a * m + b * (1 - m)); acc (the up-sampled running reconstruction) is added when given: out = acc + ls
template <typename T>
__global__ void lap_level_kernel(const T* __restrict__ a, const float* __restrict__ ua, const T* __restrict__ b,
                                 const float* __restrict__ ub, const float* __restrict__ m, const float* __restrict__ acc,
                                 float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float la = ld(a + i), lb = ld(b + i);
    if (ua) { la = sub_rn(la, ua[i]); lb = sub_rn(lb, ub[i]); }
    const float gm = m[i];
    const float ls = add_rn(mul_rn(la, gm), mul_rn(lb, sub_rn(1.f, gm)));
    out[i] = acc ? add_rn(acc[i], ls) : ls;
}

// np.uint8(np.clip(img, 0, 255)) (multi_band_blending.py:73-74)
__global__ void clip_u8_kernel(const float* __restrict__ s, uint8_t* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = s[i];
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    d[i] = (uint8_t)v;
}

}  // namespace

extern "C" int e4s_mask_to_u8(const float* mask, uint8_t* out, int B, int H, int W, int Hm, int Wm, void* stream) {
    const int64_t n = (int64_t)B * H * W;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mask_to_u8_kernel, grid1(n), dim3(256), 0, as_stream(stream), mask, out, H, W, Hm, Wm, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_erode_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int radius, int border_value, void* stream) {
    const int64_t n = (int64_t)B * H * W;
    if (radius < 0 || radius > 15 || border_value < 0 || border_value > 255 || src == dst) return (int)hipErrorInvalidValue;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(erode_u8_kernel, grid1(n), dim3(256), 0, as_stream(stream), src, dst, H, W, radius, border_value, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int ksize, const int* taps_fixed8,
                                    void* stream) {
    if (ksize < 1 || ksize > GK_MAX || !(ksize & 1) || !taps_fixed8 || src == dst) return (int)hipErrorInvalidValue;
    int sum = 0;
    GaussTaps t;
    for (int i = 0; i < GK_MAX; ++i) t.k[i] = i < ksize ? taps_fixed8[i] : 0;
    for (int i = 0; i < ksize; ++i) {
        if (t.k[i] < 0) return (int)hipErrorInvalidValue;
        sum += t.k[i];
    }
    if (sum != 256) return (int)hipErrorInvalidValue;                 // 8.8 fixed point, normalised (the row sums stay < 2^16)
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    hipLaunchKernelGGL(gauss_u8_kernel, dim3((W + 31) / 32, (H + 7) / 8, B), dim3(256), 0, as_stream(stream), src, dst, H, W,
                       ksize / 2, t);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_alpha_composite_u8(const uint8_t* face, const uint8_t* target, const uint8_t* alpha, uint8_t* out, int B,
                                      int H, int W, void* stream) {
    const int64_t n = (int64_t)B * H * W;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(alpha_composite_u8_kernel, grid1(n), dim3(256), 0, as_stream(stream), face, target, alpha, out, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_pyrdown_u8(const uint8_t* src, uint8_t* dst, int B, int H, int W, int C, void* stream) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pyrdown_kernel<uint8_t>, grid1(n), dim3(256), 0, as_stream(stream), src, dst, H, W, C, Ho, Wo, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_pyrdown_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pyrdown_kernel<float>, grid1(n), dim3(256), 0, as_stream(stream), src, dst, H, W, C, Ho, Wo, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_pyrup_f32(const float* src, float* dst, int B, int h, int w, int C, void* stream) {
    const int64_t n = (int64_t)B * 2 * h * 2 * w * C;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pyrup_f32_kernel, grid1(n), dim3(256), 0, as_stream(stream), src, dst, h, w, C, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_u8_to_f32(const uint8_t* src, float* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(u8_to_f32_kernel, grid1(n), dim3(256), 0, as_stream(stream), src, dst, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

// a / b: uint8 (a_is_u8 != 0: the finest level, whose Gaussian level is the image itself) or fp32
extern "C" int e4s_lap_level_f32(const void* a, const float* ua, const void* b, const float* ub, const float* m, const float* acc,
                                 float* out, int64_t n, int a_is_u8, void* stream) {
    if (n <= 0) return 0;
    if ((ua == nullptr) != (ub == nullptr)) return (int)hipErrorInvalidValue;
    if (a_is_u8)
        hipLaunchKernelGGL(lap_level_kernel<uint8_t>, grid1(n), dim3(256), 0, as_stream(stream), (const uint8_t*)a, ua,
                           (const uint8_t*)b, ub, m, acc, out, n);
    else
        hipLaunchKernelGGL(lap_level_kernel<float>, grid1(n), dim3(256), 0, as_stream(stream), (const float*)a, ua,
                           (const float*)b, ub, m, acc, out, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_clip_u8(const float* src, uint8_t* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(clip_u8_kernel, grid1(n), dim3(256), 0, as_stream(stream), src, dst, n);
    E4S_CHECK_LAUNCH();
    return 0;
}
