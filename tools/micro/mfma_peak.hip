// Sustained bf16 MFMA rate of the box: bare v_mfma_f32_16x16x32_bf16 loops, operands in registers, W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {        // 256-thread blocks: 1 to 4 waves per SIMD
        const int blocks = 256 * wg_per_cu, iters = 20000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1000);
        hipDeviceSynchronize();
        hipEventRecord(s);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double flops = (double)blocks * 4 * iters * 16 * 16384.0;
        printf("%d wave(s) per SIMD: %.1f TFLOP/s  (%.3f ms; %.2f cycles per MFMA at 2.4 GHz)\n", wg_per_cu, flops / ms / 1e9, ms,
               ms * 1e-3 * 2.4e9 / (iters * 16.0 * wg_per_cu));
    }
    return 0;
}
