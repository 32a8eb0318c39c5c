// dev probe: the layer kernel's inner loop in isolation -- per k-block 2 A + WNB(2)x... LDS operand reads (ds_read_b128) feeding
// 16 MFMAs (2x2 accumulator tiles x 4 k-steps), no global traffic, no barriers.  Variants: rolled with wait-before-use
// (as the kernel), and software-pipelined (operands of k-block i+1 requested before the MFMAs of i).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ALD 36
template <int VARIANT>
__global__ __launch_bounds__(256) void loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128 * ALD];
    __shared__ __attribute__((aligned(16))) float Bs[4 * 4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, h = lane >> 5, j = lane & 31;
    for (int i = tid; i < 128 * ALD; i += 256) As[i] = i * 1e-6f;
    for (int i = tid; i < 4 * 4 * 256; i += 256) Bs[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const float* a_base = &As[(wm * 64 + j) * ALD + 4 * h];
    const float* b_base = &Bs[wn * 2 * 1024 + lane * 4];
    auto rd = [&](int kbl, float4 (&a)[2], float4 (&b)[2]) {
        a[0] = *reinterpret_cast<const float4*>(a_base + kbl * 8);
        a[1] = *reinterpret_cast<const float4*>(a_base + 32 * ALD + kbl * 8);
        b[0] = *reinterpret_cast<const float4*>(b_base + kbl * 256);
        b[1] = *reinterpret_cast<const float4*>(b_base + 1024 + kbl * 256);
    };
    auto mm = [&](const float4 (&a)[2], const float4 (&b)[2]) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b[n].x, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b[n].y, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, b[n].z, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, b[n].w, acc[r][n], 0, 0, 0);
    };
    if (VARIANT == 0) {
        for (int it = 0; it < iters; it++)
            for (int kbl = 0; kbl < 4; kbl++) { float4 a[2], b[2]; rd(kbl, a, b); mm(a, b); }
    } else {
        float4 a[2], b[2];
        rd(0, a, b);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kbl = 0; kbl < 4; kbl++) {
                float4 an[2], bn[2];
                rd((kbl + 1) & 3, an, bn);
                __builtin_amdgcn_sched_barrier(0);
                mm(a, b);
                __builtin_amdgcn_sched_barrier(0);
                a[0] = an[0]; a[1] = an[1]; b[0] = bn[0]; b[1] = bn[1];
            }
        }
    }
    f32x16 s = acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1];
    float t = 0.f;
    for (int i = 0; i < 16; i++) t += s[i];
    out[blockIdx.x * 256 + tid] = t;
}
template <int V> void run(const char* name, float* out) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wgs = 1; wgs <= 3; wgs++) {
        const int iters = 20000, grid = 256 * wgs;
        hipLaunchKernelGGL(loop<V>, dim3(grid), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(s);
        hipLaunchKernelGGL(loop<V>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        printf("%s  WGs/CU %d  %.1f ms  %.1f TFLOP/s\n", name, wgs, ms, (double)grid * 4 * iters * 64.0 * 4096.0 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
    run<0>("rolled   ", out);
    run<1>("pipelined", out);
    return 0;
}
