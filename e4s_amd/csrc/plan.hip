// Region-select plan: turns the one-hot parsing mask into per-layer row lists so that every output
// pixel of a masked layer is computed ONCE, with the style of its own region.
// The reference instead runs the full modulated conv once per region (12x) and multiplies by the
// nearest-resized one-hot mask (src/models/stylegan2/model.py:386-400, 426-439); because the mask is
// one-hot (labelMap2OneHot, src/utils/torch_utils.py:166-172) the two are the same function.
#include "common.h"
#include <stdlib.h>

namespace {

__global__ void mask_labels_kernel(const float* __restrict__ mask, uint8_t* __restrict__ labels, int* flags,
                                   int B, int R, int64_t HW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * HW) return;
    const int64_t b = i / HW, p = i - b * HW;
    const float* m = mask + b * R * HW + p;
    float best = m[0];
    int arg = 0, ones = 0;
    bool clean = true;
    for (int r = 0; r < R; ++r) {
        const float v = m[(int64_t)r * HW];
        if (v > best) { best = v; arg = r; }
        if (v == 1.f) ++ones;
        else if (v != 0.f) clean = false;
    }
    labels[i] = (uint8_t)arg;
    if (!clean || ones != 1) atomicOr(flags, 1);
}

// legacy 'nearest' source index (F.interpolate(mode='nearest')): min(floor(dst * in/out), in-1)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

struct PlanGeom {
    int B, R, Hm, Wm, Ha, Wa, nphase, BM;
};

__device__ __forceinline__ int row_key(const PlanGeom& g, const uint8_t* labels, int64_t q, int* anchor_out) {
    const int phase = (int)(q % g.nphase);
    const int anchor = (int)(q / g.nphase);
    const int hw = g.Ha * g.Wa;
    const int b = anchor / hw, rem = anchor - b * hw;
    const int ay = rem / g.Wa, ax = rem - ay * g.Wa;
    const int os = (g.nphase == 4) ? 2 : 1;
    const int oy = ay * os + (phase >> 1), ox = ax * os + (phase & 1);
    const int sy = nearest_src(oy, g.Hm, g.Ha * os), sx = nearest_src(ox, g.Wm, g.Wa * os);
    const int lab = labels[((int64_t)b * g.Hm + sy) * g.Wm + sx];
    *anchor_out = anchor;
    return (b * g.R + lab) * g.nphase + phase;
}

// counts / cursor <- 0, rows <- -1 (padding sentinel).  A KERNEL, not hipMemsetAsync: this file used to hold the library's only
// memset calls, and graphs that contained them (captured memset nodes) died with a GPU memory-access fault on replay while the
// same launches ran clean eagerly -- consumers read garbage row anchors when the 0xFF fill of `rows` does not
// happen.  Kernel nodes replay like every other launch of the library; the consumers additionally bound-check every anchor and
// take a tile's row count from the tile table instead of the sentinel (conv_mfma.hip), so a stale `rows` can no longer fault.
__global__ void plan_init_kernel(int* __restrict__ work, int nwork, int* __restrict__ rows, int64_t rows_cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwork) work[i] = 0;
    if (i < rows_cap) rows[i] = -1;
}

__global__ void plan_hist_kernel(PlanGeom g, const uint8_t* __restrict__ labels, int* __restrict__ counts, int nkeys) {
    extern __shared__ int lh[];
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    const int64_t nrows = (int64_t)g.B * g.Ha * g.Wa * g.nphase;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nrows) {
        int anchor;
        atomicAdd(&lh[row_key(g, labels, q, &anchor)], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x)
        if (lh[k]) atomicAdd(&counts[k], lh[k]);
}

// single block: exclusive scan of padded group sizes, then the tile table
__global__ void plan_scan_kernel(PlanGeom g, const int* __restrict__ counts, int* __restrict__ row_off,
                                 int* __restrict__ tiles, int* __restrict__ meta, int nkeys, int tiles_cap) {
    extern __shared__ int sh[];   // tile_off[nkeys]
    if (threadIdx.x == 0) {
        int rc = 0, tc = 0;
        for (int k = 0; k < nkeys; ++k) {
            const int nt = (counts[k] + g.BM - 1) / g.BM;
            row_off[k] = rc;
            sh[k] = tc;
            rc += nt * g.BM;
            tc += nt;
        }
        meta[0] = tc < tiles_cap ? tc : tiles_cap;
        meta[1] = rc;
        meta[2] = 0;
        meta[3] = 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x) {
        const int cnt = counts[k], nt = (cnt + g.BM - 1) / g.BM;
        for (int t = 0; t < nt; ++t) {
            const int ti = sh[k] + t;
            if (ti >= tiles_cap) break;
            tiles[ti * 4 + 0] = row_off[k] + t * g.BM;
            tiles[ti * 4 + 1] = k / g.nphase;
            tiles[ti * 4 + 2] = k % g.nphase;
            const int left = cnt - t * g.BM;
            tiles[ti * 4 + 3] = left < g.BM ? left : g.BM;
        }
    }
}

__global__ void plan_scatter_kernel(PlanGeom g, const uint8_t* __restrict__ labels, const int* __restrict__ row_off,
                                    int* __restrict__ cursor, int* __restrict__ rows, int nkeys, int rows_cap) {
    extern __shared__ int sh[];          // cnt[nkeys], base[nkeys]
    int* lc = sh;
    int* lb = sh + nkeys;
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x) lc[k] = 0;
    __syncthreads();
    const int64_t nrows = (int64_t)g.B * g.Ha * g.Wa * g.nphase;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int key = -1, anchor = 0, local = 0;
    if (q < nrows) {
        key = row_key(g, labels, q, &anchor);
        local = atomicAdd(&lc[key], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x)
        if (lc[k]) lb[k] = atomicAdd(&cursor[k], lc[k]);
    __syncthreads();
    if (key >= 0) {
        const int pos = row_off[key] + lb[key] + local;
        if (pos < rows_cap) rows[pos] = anchor;
    }
}

}  // namespace

extern "C" int e4s_mask_labels(const float* mask, uint8_t* labels, int* flags, int B, int R, int Hm, int Wm, void* stream) {
    const int64_t n = (int64_t)B * Hm * Wm;
    if (n <= 0 || R <= 0 || R > 255) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mask_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), mask,
                       labels, flags, B, R, (int64_t)Hm * Wm);
    E4S_CHECK_LAUNCH();
    return 0;
}

extern "C" int e4s_region_plan(const uint8_t* labels, int B, int R, int Hm, int Wm, int Ha, int Wa, int nphase, int BM,
                               int* rows, int* tiles, int* meta, int* work, int rows_cap, int tiles_cap, void* stream) {
    if ((nphase != 1 && nphase != 4) || BM <= 0) return (int)hipErrorInvalidValue;
    const int nkeys = B * R * nphase;
    const int64_t nrows = (int64_t)B * Ha * Wa * nphase;
    if (rows_cap < nrows + (int64_t)nkeys * BM || (int64_t)tiles_cap * BM < rows_cap) return (int)hipErrorInvalidValue;
    if (nkeys * 2 * sizeof(int) > 60000) return (int)hipErrorInvalidValue;
    hipStream_t st = as_stream(stream);
    int* counts = work;
    int* cursor = work + nkeys;
    int* row_off = work + 2 * nkeys;
    // E4S_PLAN_MEMSET=1: the pre-round-3 initialisation (two memset nodes), kept only so that the replay fault can be
    // reproduced A/B on a GPU box (tests/test_gpu_optim.py); never set in a product run
    static const bool use_memset = [] { const char* v = getenv("E4S_PLAN_MEMSET"); return v && v[0] == '1'; }();
    if (use_memset) {
        hipError_t e = hipMemsetAsync(work, 0, sizeof(int) * 2 * nkeys, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(rows, 0xFF, sizeof(int) * (size_t)rows_cap, st);
        if (e != hipSuccess) return (int)e;
    } else {
        const int64_t n = rows_cap > 2 * nkeys ? rows_cap : 2 * nkeys;
        hipLaunchKernelGGL(plan_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, 2 * nkeys, rows,
                           (int64_t)rows_cap);
        E4S_CHECK_LAUNCH();
    }
    PlanGeom g{B, R, Hm, Wm, Ha, Wa, nphase, BM};
    const unsigned nblk = (unsigned)((nrows + 255) / 256);
    hipLaunchKernelGGL(plan_hist_kernel, dim3(nblk), dim3(256), nkeys * sizeof(int), st, g, labels, counts, nkeys);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(256), nkeys * sizeof(int), st, g, counts, row_off, tiles, meta,
                       nkeys, tiles_cap);
    E4S_CHECK_LAUNCH();
    hipLaunchKernelGGL(plan_scatter_kernel, dim3(nblk), dim3(256), 2 * nkeys * sizeof(int), st, g, labels, row_off,
                       cursor, rows, nkeys, rows_cap);
    E4S_CHECK_LAUNCH();
    return 0;
}
