// Shared helpers for the gfx950 kernels of libe4s_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "e4s_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define E4S_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// wave64 all-lanes sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 4x4 transpose inside each quad of lanes (two DPP quad_perm exchanges): in: a_i = M[i][lane & 3]; out: a_k = M[lane & 3][k].
// MFMA 32x32 accumulators hold one COLUMN per lane and rows (r & 3) + 8 (r >> 2) + 4 kh in register r: transposing registers
// 4g .. 4g+3 across the quad gives each lane four consecutive columns of ONE row -- a 16-byte NHWC store instead of four 4-byte
// ones.  (Stores straight from the accumulators are store-ISSUE bound on gfx950: ~70 cycles per store instruction whatever its
// width; measured on conv_c32.hip: 128 -> 32 store instructions per tile = 0.86 -> 0.70 ms.)
__device__ __forceinline__ float e4s_dpp_f32(float v, const int ctrl_b1_or_4e) {
    return ctrl_b1_or_4e == 0xB1
               ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true))     // quad_perm [1,0,3,2]
               : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
}

__device__ __forceinline__ void quad_transpose4(float& a0, float& a1, float& a2, float& a3, const int lane) {
    const bool odd = lane & 1, hi = lane & 2;
    const float r01 = e4s_dpp_f32(odd ? a0 : a1, 0xB1), r23 = e4s_dpp_f32(odd ? a2 : a3, 0xB1);
    if (odd) { a0 = r01; a2 = r23; } else { a1 = r01; a3 = r23; }
    const float ra = e4s_dpp_f32(hi ? a0 : a2, 0x4E), rb = e4s_dpp_f32(hi ? a1 : a3, 0x4E);
    if (hi) { a0 = ra; a1 = rb; } else { a2 = ra; a3 = rb; }
}

// XCD-aware bijective remap of a 1-D block id: consecutive logical ids land on the same XCD
// (hardware places block b on XCD b % 8), so tiles that share an A panel share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Kernels that need more than 64 KB of dynamic LDS must raise hipFuncAttributeMaxDynamicSharedMemorySize first.  The
// attribute belongs to the function as loaded on ONE device, so it is set once per (kernel, device): `mask` (one static
// per launch site) has a bit per device ordinal; atomics keep concurrent host threads safe.
static inline int e4s_ensure_dyn_smem(const void* fn, int bytes, std::atomic<uint64_t>& mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

// conv_bf16x3.hip: second stage of a split-K conv launch (shared with conv_mfma.hip)
int e4s_splitk_epilogue(const e4s_conv_params& p, int ksplit, hipStream_t st);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
