// Input gradient of the MASKED StyledConvs (autograd of model.py:386-400 with :276-320) in scatter form, so that the
// contraction runs on the plain split-bf16 matrix-core kernel instead of the exact-fp32 dx + ds kernel (6.0 ms of a 16.9 ms optimisation
// step, ~10 ms of the config-5 G step; VERDICT r4 'missing' 4 / 'next' 5).
//
// Forward (plain 3x3):  c[m, co] = sum_t sum_ci x[m + t - 1, ci] * s[r(m), ci] * W[t][co][ci],   y = d[r(m), co] * c + ...,  r(m) = region of the
// OUTPUT pixel m.  With u[m, co] = dL/dz[m, co] * d[r(m), co]:
//     dx[h, ci]   = sum_t  s[r(m_t), ci] * G[m_t, t, ci],        m_t = h - (t - 1)
//     ds[rho, ci] = sum_{m in rho} sum_t x[m + t - 1, ci] * G[m, t, ci]
//     G[m, t, ci] = sum_co u[m, co] * W[t][co][ci]
// In GATHER form (rows = h) the style factor s[r(h - t + 1), ci] sits on the accumulator of every tap separately (it depends on the row, the
// tap AND the column), which is why conv_bwd_kernel contracts in exact fp32 with per-tap handling.  In SCATTER form (rows = m) the row owns ONE
// region: G is a plain 1x1 contraction [B H W, Cy] x [Cy, 9 Cx] with no halo, no per-pixel operand scaling and no region logic at all --
// e4s_conv_bf16x3_f32 as it is (same MACs as the 3x3 conv: 9 Cy Cx per pixel).  The region-dependent parts are two streaming passes here:
//     e4s_region_scale_f32     u = gz * d[r(m)]            (16-byte lanes; for the polyphase up-convs written phase-major, one map per phase)
//     e4s_col2im_region_f32    dx = sum_t s[r(m_t)] * G[m_t, t],  ds[rho] = sum_{m in rho} sum_t x[m + t - 1] * G[m, t]   (ordered sums)
// G is 9x the activation (18 KB per pixel at Cx = 512): 75 MB at 64^2 x 512, 302 MB at 256^2 x 128 per sample -- written once, read ~twice;
// at the batch sizes of configs 3 / 5 (1 / 2) that traffic costs less than the fp32 contraction it replaces by a factor of ~5.
//
// Polyphase up-convs (model.py:287-300 folded to four 3x3 phase kernels over the INPUT grid): rows = (anchor a, phase ph), region = region of
// the OUTPUT pixel 2a + ph; G_ph[a, e, ci] = sum_co u[2a + ph, co] * Weff[ph][e][co][ci] is one 1x1 contraction per phase, and
//     dx[h, ci] = sum_ph sum_e s[r(2 a_e + ph), ci] * G_ph[a_e, e, ci],   a_e = h - (e - 1)
//     ds[rho, ci] = sum_ph sum_{a: r(2a + ph) = rho} sum_e x[a + e - 1, ci] * G_ph[a, e, ci].
#include "common.h"

namespace {

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {        // F.interpolate(mode='nearest'), model.py:391
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// u[(ph), b, a, c] = gz[b, p, c] * d[b * R + r(p), c];  NCLS == 1: p = a (same layout);  NCLS == 4: gz is [B, 2H, 2W, C], u is [4][B, H, W, C]
__global__ __launch_bounds__(256) void region_scale_kernel(const float* __restrict__ gz, const float* __restrict__ d,
                                                           const uint8_t* __restrict__ labels, int Hm, int Wm, int R, float* __restrict__ u,
                                                           int B, int H, int W, int C, int ncls, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one (output pixel, 4 channels)
    if (i >= n4) return;
    const int c4n = C >> 2;
    const int c4 = (int)(i % c4n);
    int64_t q = i / c4n;
    const int os = ncls == 4 ? 2 : 1;
    const int Wo = W * os, Ho = H * os;
    const int ox = (int)(q % Wo);
    q /= Wo;
    const int oy = (int)(q % Ho);
    const int b = (int)(q / Ho);
    int lab = labels[((int64_t)b * Hm + nearest_src(oy, Hm, Ho)) * Wm + nearest_src(ox, Wm, Wo)];
    lab = lab < R ? lab : R - 1;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gz + i * 4);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(d + ((int64_t)b * R + lab) * C + c4 * 4);
    int64_t dst;
    if (ncls == 4) {
        const int ph = (oy & 1) * 2 + (ox & 1);
        dst = ((((int64_t)ph * B + b) * H + (oy >> 1)) * W + (ox >> 1)) * C + c4 * 4;
    } else {
        dst = i * 4;
    }
    *reinterpret_cast<f32x4*>(u + dst) = g * dv;
}

// Block = (sample b, pixel split); thread = (4 channels, pixel lane) as act_bwd_demod_kernel.  Per pixel p of the x grid:
//   dx[p] = sum_ph sum_t s[r_ph(m_t)] * G_ph[m_t, t]          (9 NCLS 16-byte loads of G, the style rows from an LDS table)
//   the ds term of p as a SOURCE row: sum_t x[p + t - 1] * G_ph[p, t] added to the thread's own LDS slot of region r_ph(p)
// lanes are combined in lane order, the splits by e4s_reduce_parts_f32: bit-reproducible.
template <int NCLS>
__global__ __launch_bounds__(256) void col2im_region_kernel(const float* __restrict__ G, const float* __restrict__ x, const float* __restrict__ s,
                                                            const uint8_t* __restrict__ labels, int Hm, int Wm, int R,
                                                            float* __restrict__ dx, float* __restrict__ part, int B, int H, int W, int C,
                                                            int nsplit, int64_t pstride) {
    extern __shared__ f32x4 tab[];          // [lanes][R][C / 4] ds accumulators, then [R][C / 4] the sample's style rows
    const int c4n = C >> 2, lanes = 256 / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const int b = blockIdx.x / nsplit, split = blockIdx.x - b * nsplit;
    const int HW = H * W;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(p0 + per, HW);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4* stab = tab + (size_t)lanes * R * c4n;
    for (int t = threadIdx.x; t < lanes * R * c4n; t += 256) tab[t] = zero4;
    for (int t = threadIdx.x; t < R * c4n; t += 256) stab[t] = *reinterpret_cast<const f32x4*>(s + (int64_t)b * R * C + (int64_t)t * 4);
    __syncthreads();
    constexpr int OS = NCLS == 4 ? 2 : 1;
    const int Ho = H * OS, Wo = W * OS;
    auto label_of = [&](int ay, int ax, int ph) -> int {            // region of the OUTPUT pixel that row (anchor, phase) produces
        const int oy = ay * OS + (NCLS == 4 ? (ph >> 1) : 0), ox = ax * OS + (NCLS == 4 ? (ph & 1) : 0);
        const int lab = labels[((int64_t)b * Hm + nearest_src(oy, Hm, Ho)) * Wm + nearest_src(ox, Wm, Wo)];
        return lab < R ? lab : R - 1;
    };
    const int64_t row = (int64_t)9 * C;                               // floats of one G row
    const int64_t gcls = (int64_t)B * HW * row;                       // floats of one phase's G
    const float* Gb = G + (int64_t)b * HW * row + c4 * 4;
    const float* xb = x + (int64_t)b * HW * C + c4 * 4;
    f32x4* mine = tab + (size_t)rl * R * c4n + c4;
    if (rl < lanes)
        for (int p = p0 + rl; p < p1; p += lanes) {
            const int py = p / W, px = p - py * W;
            f32x4 dxv = zero4;
#pragma unroll
            for (int ph = 0; ph < NCLS; ++ph) {
                const float* Gp = Gb + ph * gcls;
                f32x4 dsv = zero4;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ty = t / 3 - 1, tx = t % 3 - 1;
                    // dx: the source row whose tap t lands on p
                    const int my = py - ty, mx = px - tx;
                    if ((unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W) {
                        const f32x4 g = *reinterpret_cast<const f32x4*>(Gp + ((int64_t)my * W + mx) * row + (int64_t)t * C);
                        dxv += stab[label_of(my, mx, ph) * c4n + c4] * g;
                    }
                    // ds: p as the source row, its tap t reads x[p + t - 1] (zero outside the image)
                    const int ny = py + ty, nx = px + tx;
                    if ((unsigned)ny < (unsigned)H && (unsigned)nx < (unsigned)W) {
                        const f32x4 g = *reinterpret_cast<const f32x4*>(Gp + (int64_t)p * row + (int64_t)t * C);
                        dsv += *reinterpret_cast<const f32x4*>(xb + ((int64_t)ny * W + nx) * C) * g;
                    }
                }
                mine[label_of(py, px, ph) * c4n] += dsv;              // the thread's own slot: a fixed order of additions
            }
            *reinterpret_cast<f32x4*>(dx + ((int64_t)b * HW + p) * C + c4 * 4) = dxv;
        }
    __syncthreads();
    for (int t = threadIdx.x; t < R * c4n; t += 256) {
        f32x4 s4 = tab[t];
        for (int l = 1; l < lanes; ++l) s4 += tab[(size_t)l * R * c4n + t];
        const int r = t / c4n, cc = t - r * c4n;
        *reinterpret_cast<f32x4*>(part + (int64_t)split * pstride + ((int64_t)b * R + r) * C + cc * 4) = s4;
    }
}

int col2im_nsplit(int B, int H, int W, int C) {
    const int lanes = 256 / (C / 4);
    int nsplit = (H * W) / (4 * (lanes > 0 ? lanes : 1));             // every thread walks >= 4 pixels (36 - 144 loads each)
    const int cap = 4096 / (B > 0 ? B : 1);
    if (nsplit > cap) nsplit = cap;
    if (nsplit > 1024) nsplit = 1024;
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}

}  // namespace

extern "C" int e4s_region_scale_f32(const float* gz, const float* d, const uint8_t* labels, int Hm, int Wm, int R, float* u, int B, int H,
                                    int W, int C, int ncls, void* stream) {
    if (!gz || !d || !labels || !u || C % 4 || C <= 0 || R < 1 || R > 16 || B <= 0 || H <= 0 || W <= 0 || (ncls != 1 && ncls != 4))
        return (int)hipErrorInvalidValue;
    const int64_t n4 = (int64_t)B * H * W * (ncls == 4 ? 4 : 1) * (C / 4);
    hipLaunchKernelGGL(region_scale_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), gz, d, labels, Hm, Wm, R,
                       u, B, H, W, C, ncls, n4);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_col2im_region_nsplit(int B, int H, int W, int C) {
    if (C % 4 || C < 4 || C > 1024 || 256 % (C / 4)) return 0;
    return col2im_nsplit(B, H, W, C);
}

extern "C" int e4s_col2im_region_f32(const float* G, const float* x, const float* s, const uint8_t* labels, int Hm, int Wm, int R, float* dx,
                                     float* ds, float* ws, int B, int H, int W, int C, int ncls, void* stream) {
    if (!G || !x || !s || !labels || !dx || !ds || !ws || C % 4 || C < 4 || C > 1024 || 256 % (C / 4) || R < 1 || R > 16 || B <= 0 ||
        (ncls != 1 && ncls != 4))
        return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    const int nsplit = col2im_nsplit(B, H, W, C);
    const int64_t n = (int64_t)B * R * C;
    const int c4n = C / 4, lanes = 256 / c4n;
    const size_t smem = ((size_t)lanes * R * c4n + (size_t)R * c4n) * sizeof(f32x4);        // 4 KB * R + 4 R C bytes <= 128 KB
    if (ncls == 4) {
        static std::atomic<uint64_t> m4{0};
        if (int e = e4s_ensure_dyn_smem((const void*)col2im_region_kernel<4>, 128 * 1024, m4)) return e;
        hipLaunchKernelGGL(col2im_region_kernel<4>, dim3(B * nsplit), dim3(256), smem, st, G, x, s, labels, Hm, Wm, R, dx, ws, B, H, W, C,
                           nsplit, n);
    } else {
        static std::atomic<uint64_t> m1{0};
        if (int e = e4s_ensure_dyn_smem((const void*)col2im_region_kernel<1>, 128 * 1024, m1)) return e;
        hipLaunchKernelGGL(col2im_region_kernel<1>, dim3(B * nsplit), dim3(256), smem, st, G, x, s, labels, Hm, Wm, R, dx, ws, B, H, W, C,
                           nsplit, n);
    }
    E4S_CHECK_LAUNCH();
    return e4s_reduce_parts_f32(ws, ds, nsplit, n, 1.f, stream);
}
