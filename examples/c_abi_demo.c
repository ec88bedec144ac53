/* Plain-C host program over the C-ABI of libe4s_hip.so: no Python, no torch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Le4s_amd -le4s_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/e4s_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 *   /tmp/c_abi_demo            (needs an MI355X; the HIP runtime is used only for hipMalloc/hipMemcpy/streams)
 *
 * It runs the two entry points that replace the reference's native ops
 * (src/models/stylegan2/op/fused_bias_act.cpp:11-21, op/upfirdn2d.cpp:12-22) on the SURVEY.md 8(c)
 * known-answer vectors KAT5 and KAT2 and compares with the literal expected values. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "e4s_hip.h"

#define CHECK(call)                                                               \
    do {                                                                          \
        int rc__ = (int)(call);                                                   \
        if (rc__ != 0) { fprintf(stderr, "%s failed: %d\n", #call, rc__); return 1; } \
    } while (0)

static int close_to(const float* got, const float* want, int n, float tol) {
    for (int i = 0; i < n; ++i)
        if (fabsf(got[i] - want[i]) > tol) { fprintf(stderr, "mismatch at %d: %g vs %g\n", i, got[i], want[i]); return 0; }
    return 1;
}

int main(void) {
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));

    /* KAT5: fused_leaky_relu([[-1,0,2]], b=[0.5,-0.5,0.5]) = sqrt(2)*lrelu_0.2(x+b) */
    const float x5[3] = {-1.f, 0.f, 2.f}, b5[3] = {0.5f, -0.5f, 0.5f};
    const float want5[3] = {-0.141421f, -0.141421f, 3.535534f};
    float *dx, *db, *dy, y5[3];
    CHECK(hipMalloc((void**)&dx, sizeof x5));
    CHECK(hipMalloc((void**)&db, sizeof b5));
    CHECK(hipMalloc((void**)&dy, sizeof y5));
    CHECK(hipMemcpy(dx, x5, sizeof x5, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b5, sizeof b5, hipMemcpyHostToDevice));
    /* input [1,3]: step_b = 1 (no spatial axes), size_b = 3; act 3 = leaky-relu, grad 0 = forward */
    CHECK(e4s_fused_bias_act_f32(dx, db, NULL, dy, 3, 1, 3, 3, 0, 0.2f, 1.41421356f, stream));
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(y5, dy, sizeof y5, hipMemcpyDeviceToHost));
    if (!close_to(y5, want5, 3, 1e-5f)) return 2;

    /* KAT2: upfirdn2d(arange(1..9) 3x3, 4*outer([1,3,3,1])/64, pad=(1,1)) = [[11.8125,13.5625],[17.0625,18.8125]] */
    float x2[9], k2[16], y2[4];
    const float want2[4] = {11.8125f, 13.5625f, 17.0625f, 18.8125f}, k1[4] = {1.f, 3.f, 3.f, 1.f};
    for (int i = 0; i < 9; ++i) x2[i] = (float)(i + 1);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) k2[i * 4 + j] = 4.f * k1[i] * k1[j] / 64.f;
    float *dx2, *dk2, *dy2;
    CHECK(hipMalloc((void**)&dx2, sizeof x2));
    CHECK(hipMalloc((void**)&dk2, sizeof k2));
    CHECK(hipMalloc((void**)&dy2, sizeof y2));
    CHECK(hipMemcpy(dx2, x2, sizeof x2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dk2, k2, sizeof k2, hipMemcpyHostToDevice));
    /* [major=1, H=3, W=3, minor=1], 4x4 kernel, up 1, down 1, pad (1,1) on both axes -> 2x2 */
    CHECK(e4s_upfirdn2d_f32(dx2, dk2, dy2, 1, 3, 3, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, stream));
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(y2, dy2, sizeof y2, hipMemcpyDeviceToHost));
    if (!close_to(y2, want2, 4, 1e-5f)) return 3;

    printf("c_abi_demo: KAT5 and KAT2 reproduced through the C ABI\n");
    hipFree(dx); hipFree(db); hipFree(dy); hipFree(dx2); hipFree(dk2); hipFree(dy2);
    hipStreamDestroy(stream);
    return 0;
}
