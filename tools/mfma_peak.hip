// dev probe: sustained fp32 MFMA rate of the part (v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave, no memory traffic)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    f32x16 s = a0 + a1 + a2 + a3;
    float t = 0.f;
    for (int i = 0; i < 16; i++) t += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++) {
        for (int ms_target = 0; ms_target < 2; ms_target++) {
            const int iters = ms_target ? 200000 : 20000, grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 1000);
            hipDeviceSynchronize();
            hipEventRecord(s);
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * 4096.0;
            printf("waves/SIMD %d  %.1f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
