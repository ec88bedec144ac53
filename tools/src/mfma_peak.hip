// tools/mfma_peak: what rate does a register-resident v_mfma_f32_32x32x16_bf16 stream sustain on this MI355X, and at what clock?
// (VERDICT r3 "Next round" 3a: settle the ceiling the split-bf16 conv kernels are priced against.)  Standalone HIP program, no torch.
//   mfma_peak <random|zero> <waves_per_simd 1|2|4> <nacc 1|2|4|8> <iters> <reps>     (nacc = independent accumulators per wave: the
//   distance between two MFMAs on the same accumulator)
// Every wave runs `iters` rounds of NACC independent accumulators x 4 (A,B) operand pairs; operands are random bf16 (or zeros) and stay in
// registers: no LDS, no memory traffic in the loop.  Prints one JSON line: TFLOP/s from HIP events over `reps` launches, and the effective
// shader clock = s_memtime ticks of the loop / its wall time (the guide: tick = shader cycle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* ticks, int iters, int zero) {
    extern __shared__ char lds_pad[];              // occupancy control only
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t ha = hash32(tid * 64 + s * 16 + e), hb = hash32(tid * 64 + s * 16 + 8 + e);
            const float fa = zero ? 0.f : ((int)(ha & 0xffff) - 32768) * (1.f / 32768.f);
            const float fb = zero ? 0.f : ((int)(hb & 0xffff) - 32768) * (1.f / 32768.f);
            a[s][e] = (__bf16)fa;
            b[s][e] = (__bf16)fb;
        }
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[(s + j) & 3], acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[tid] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int zero = argc > 1 && !strcmp(argv[1], "zero");
    const int wps = argc > 2 ? atoi(argv[2]) : 1;
    const int nacc = argc > 3 ? atoi(argv[3]) : 4;
    const int iters = argc > 4 ? atoi(argv[4]) : 20000;
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, grid = cus * wps;
    const int lds = wps == 1 ? 96 * 1024 : (wps == 2 ? 64 * 1024 : 32 * 1024);      // at most `wps` blocks of 4 waves per CU
    float* out;
    unsigned long long* ticks;
    CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CK(hipMalloc(&ticks, (size_t)grid * 8));
    void (*fn)(float*, unsigned long long*, int, int) = nacc == 8 ? mfma_loop<8> : nacc == 2 ? mfma_loop<2> : nacc == 1 ? mfma_loop<1> : mfma_loop<4>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, out, ticks, iters, zero);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, out, ticks, iters, zero);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 8);
    CK(hipMemcpy(h, ticks, (size_t)grid * 8, hipMemcpyDeviceToHost));
    double tk = 0;
    for (int i = 0; i < grid; ++i) tk += (double)h[i];
    tk /= grid;
    const int na = nacc == 8 ? 8 : nacc == 2 ? 2 : nacc == 1 ? 1 : 4;
    const double nmfma = (double)grid * 4 * iters * 4 * na;
    const double tflops = nmfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD as the wave saw them (ticks are shader cycles if s_memtime counts them; reported raw as well)
    printf("{\"operands\": \"%s\", \"waves_per_simd\": %d, \"nacc\": %d, \"iters\": %d, \"cus\": %d, \"ms_per_launch\": %.4f, \"tflops\": %.1f, "
           "\"frac_of_2500\": %.4f, \"ticks_per_launch\": %.0f, \"ticks_per_mfma_per_simd\": %.2f, \"tick_rate_ghz\": %.4f, "
           "\"implied_clock_ghz_at_32cyc_per_mfma\": %.4f}\n",
           zero ? "zero" : "random", wps, na, iters, cus, ms, tflops, tflops / 2500.0, tk,
           tk / ((double)iters * 4 * na * wps), tk / (ms * 1e-3) / 1e9,
           nmfma / (cus * 4.0) * 32.0 / (ms * 1e-3) / 1e9);
    return 0;
}
