// Device-side pre/post-processing of the face-swap pipeline -- the numpy / OpenCV / PIL glue of
// scripts/face_swap.py:226-312 that sits between the parser and the generator, and between the generator and the
// all-gather, as HBM-bound kernels on label maps and images:
//   labelMap2OneHot                       src/utils/torch_utils.py:166-172
//   swap_head_mask_revisit_considerGlass  src/utils/swap_face_mask.py:33-82
//   foreground mask                       scripts/face_swap.py:280-284
//   create_masks (dilation / erosion)     scripts/face_swap.py:30-48, src/utils/morphology.py:23-198 (flat SE, geodesic)
//   tensor2im                             src/utils/torch_utils.py:63-69
//   paste with a content mask             scripts/face_swap.py:291-304 (F.interpolate bilinear of the mask + lerp)
// Everything is integer / comparison / single-rounding fp32 arithmetic, so the results are bit-exact w.r.t. the reference
// -- except the bilinear mask resize inside the paste, whose association order (and FMA use) is implementation-defined
// in ATen itself: there a uint8 result may differ by one step where the mask is fractional.
#include "common.h"

// numpy-order fp32 arithmetic (one rounding per operation): no FMA contraction in this file (see stitch.hip)
#pragma clang fp contract(off)

namespace {

__global__ void onehot_kernel(const uint8_t* __restrict__ labels, float* __restrict__ out, int R, int64_t hw, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B*R*HW outputs
    if (i >= n) return;
    const int64_t p = i % hw;
    const int64_t br = i / hw;
    const int r = (int)(br % R);
    const int64_t b = br / R;
    out[i] = labels[b * hw + p] == r ? 1.f : 0.f;
}

// swap_face_mask.py:33-82 with hair_first=True: target background / neck / ears / ear-rings / hair are kept, the
// driven face's inner regions are pasted wherever the target is not background, the target's glasses go on top and
// the holes become skin.
__global__ void swap_head_mask_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ tgt,
                                      uint8_t* __restrict__ out, uint8_t* __restrict__ hole, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = src[i], t = tgt[i];
    int res = 0;
    if (t == 0) res = 99;
    else if (t == 8 || t == 7 || t == 11 || t == 4) res = t;
    if (res != 99 && (s == 1 || s == 2 || s == 3 || s == 5 || s == 6 || s == 9)) res = s;
    if (t == 10) res = 10;
    const bool h = (res == 0);
    if (h) res = 6;
    if (res == 99) res = 0;
    out[i] = (uint8_t)res;
    hole[i] = h ? 255 : 0;
}

__global__ void foreground_kernel(const uint8_t* __restrict__ lab, const uint8_t* __restrict__ hole,
                                  float* __restrict__ fg, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = lab[i];
    const bool bg = (l == 0 || l == 11 || l == 4);
    fg[i] = (!bg || hole[i] == 255) ? 1.f : 0.f;
}

// Flat (2r+1)^2 structuring element, geodesic border (samples outside the image are ignored): window max / min.
// One block = 32x8 outputs; the (32+2r) x (8+2r) input patch is staged in LDS.
template <int MAXR>
__global__ void morph_kernel(const float* __restrict__ x, float* __restrict__ dil, float* __restrict__ ero, int H,
                             int W, int r) {
    __shared__ float tile[(8 + 2 * MAXR) * (32 + 2 * MAXR)];
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    const int tw = 32 + 2 * r, th = 8 + 2 * r;
    const float* xb = x + (int64_t)n * H * W;
    for (int t = threadIdx.x; t < tw * th; t += blockDim.x) {
        const int ty = t / tw, tx = t - ty * tw;
        const int iy = y0 + ty - r, ix = x0 + tx - r;
        // NaN marks "outside": skipped by the comparisons below
        tile[t] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xb[(int64_t)iy * W + ix] : __int_as_float(0x7fc00000);
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int ox = x0 + lx, oy = y0 + ly;
    if (ox >= W || oy >= H) return;
    float mx = -3.0e38f, mn = 3.0e38f;
    for (int dy = 0; dy <= 2 * r; ++dy)
        for (int dx = 0; dx <= 2 * r; ++dx) {
            const float v = tile[(ly + dy) * tw + lx + dx];
            if (v > mx) mx = v;
            if (v < mn) mn = v;
        }
    const int64_t o = (int64_t)n * H * W + (int64_t)oy * W + ox;
    if (dil) dil[o] = mx;
    if (ero) ero[o] = mn;
}

// scripts/face_swap.py:30-48: operation 0 dilation (full = dil, border = full - m), 1 erosion (full = ero,
// border = m - full), 2 expansion (full = dil, border = dil - ero); border clipped to [0, 1].
__global__ void create_masks_kernel(const float* __restrict__ m, const float* __restrict__ dil,
                                    const float* __restrict__ ero, float* __restrict__ border,
                                    float* __restrict__ full, int op, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f, b;
    if (op == 0) { f = dil[i]; b = f - m[i]; }
    else if (op == 1) { f = ero[i]; b = m[i] - f; }
    else { f = dil[i]; b = f - ero[i]; }
    border[i] = fminf(fmaxf(b, 0.f), 1.f);
    full[i] = f;
}

__device__ __forceinline__ uint8_t to_u8(float v) {       // tensor2im: ((v + 1) / 2) clipped to [0, 1], * 255, astype(uint8)
    float t = (v + 1.f) / 2.f;
    t = t < 0.f ? 0.f : t;
    t = t > 1.f ? 1.f : t;
    return (uint8_t)(t * 255.f);
}

// NCHW fp32 [B,3,H,W] in [-1,1] -> HWC uint8 [B,H,W,3]
__global__ void tensor2im_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int64_t hw, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B*HW pixels
    if (i >= n) return;
    const int64_t b = i / hw, p = i - b * hw;
    const float* s = img + b * 3 * hw + p;
    uint8_t* d = out + i * 3;
    d[0] = to_u8(s[0]);
    d[1] = to_u8(s[hw]);
    d[2] = to_u8(s[2 * hw]);
}

// out = uint8(face * m + target * (1 - m)); m = bilinear (align_corners=False) resize of mask [B,Hm,Wm] to [H,W];
// face / target / out: HWC uint8 [B,H,W,3]   (face_swap.py:291-292,301-303: numpy float32 arithmetic, np.uint8 truncation)
__global__ void paste_kernel(const uint8_t* __restrict__ face, const uint8_t* __restrict__ target,
                             const float* __restrict__ mask, uint8_t* __restrict__ out, int H, int W, int Hm, int Wm,
                             int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B*H*W pixels
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
    const float im = 1.f - m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // two rounded products and a rounded sum, as numpy evaluates it (no fused multiply-add)
        // (plain operators of THIS translation unit, which is under `fp contract(off)`; __fmul_rn / __fadd_rn are header inlines parsed
        // before the pragma and were fused into one FMA -- see csrc/stitch.hip)
        const float pf = (float)face[i * 3 + c] * m, pt = (float)target[i * 3 + c] * im;
        const float v = pf + pt;
        out[i * 3 + c] = (uint8_t)v;
    }
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int e4s_onehot_u8_f32(const uint8_t* labels, float* out, int B, int R, int H, int W, void* stream) {
    const int64_t hw = (int64_t)H * W, n = (int64_t)B * R * hw;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(onehot_kernel, grid1(n), dim3(256), 0, as_stream(stream), labels, out, R, hw, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_swap_head_mask_u8(const uint8_t* src, const uint8_t* tgt, uint8_t* out, uint8_t* hole, int64_t n,
                                     void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(swap_head_mask_kernel, grid1(n), dim3(256), 0, as_stream(stream), src, tgt, out, hole, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_foreground_mask_f32(const uint8_t* labels, const uint8_t* hole, float* fg, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(foreground_kernel, grid1(n), dim3(256), 0, as_stream(stream), labels, hole, fg, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_morph_f32(const float* x, float* dil, float* ero, int N, int H, int W, int radius, void* stream) {
    if (radius < 0 || radius > 16) return (int)hipErrorInvalidValue;
    if (N <= 0) return 0;
    hipLaunchKernelGGL(morph_kernel<16>, dim3((W + 31) / 32, (H + 7) / 8, N), dim3(256), 0, as_stream(stream), x, dil, ero,
                       H, W, radius);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_create_masks_f32(const float* mask, float* border, float* full, float* ws, int N, int H, int W,
                                    int radius, int operation, void* stream) {
    if (operation < 0 || operation > 2) return (int)hipErrorInvalidValue;
    const int64_t n = (int64_t)N * H * W;
    if (n <= 0) return 0;
    float* dil = ws;
    float* ero = ws + n;
    if (int e = e4s_morph_f32(mask, operation != 1 ? dil : nullptr, operation != 0 ? ero : nullptr, N, H, W, radius, stream))
        return e;
    hipLaunchKernelGGL(create_masks_kernel, grid1(n), dim3(256), 0, as_stream(stream), mask, dil, ero, border, full,
                       operation, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

// A copy with a CHOSEN number of workgroups: the stand-in for the RCCL kernels that land the peers' shards in a rank's all-gather output
// (bench.py `gather_contention`: RCCL moves data with a few persistent workgroups per channel, which share the CUs with the step's own
// persistent one-block-per-CU kernels; with one reachable GPU that term of the N = 8 efficiency can only be bounded this way).
__global__ __launch_bounds__(512) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int e4s_stream_copy_u8(const void* src, void* dst, int64_t bytes, int blocks, void* stream) {
    if (!src || !dst || bytes < 0 || bytes % 16 || blocks <= 0) return (int)hipErrorInvalidValue;
    if (bytes == 0) return 0;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(512), 0, as_stream(stream), reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<uint4*>(dst), bytes / 16);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_tensor2im_u8(const float* img, uint8_t* out, int B, int H, int W, void* stream) {
    const int64_t hw = (int64_t)H * W, n = (int64_t)B * hw;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(tensor2im_kernel, grid1(n), dim3(256), 0, as_stream(stream), img, out, hw, n);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_paste_u8(const uint8_t* face, const uint8_t* target, const float* mask, uint8_t* out, int B, int H,
                            int W, int Hm, int Wm, void* stream) {
    const int64_t n = (int64_t)B * H * W;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(paste_kernel, grid1(n), dim3(256), 0, as_stream(stream), face, target, mask, out, H, W, Hm, Wm, n);
    E4S_CHECK_LAUNCH();
    return 0;
}
