// store_width_probe.hip -- what bounds a block-copy kernel's write rate on gfx950: 4096 workgroups each writing a contiguous
// 272384-byte block (roipool3d config 5: 512 rows x 133 floats), with 4-byte or 16-byte stores per lane, values from registers or
// gathered from random 520-byte rows.   hipcc --offload-arch=gfx950 -O3 tools/store_width_probe.hip -o tools/bin/store_width_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define BLK_E (512 * 133)
__global__ __launch_bounds__(256) void k_dword(float* __restrict__ out) {
    float* o = out + (size_t)blockIdx.x * BLK_E;
    const float v = (float)threadIdx.x;
    for (int e = threadIdx.x; e < BLK_E; e += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; u++) if (e + u * 256 < BLK_E) o[e + u * 256] = v;
    }
}
__global__ __launch_bounds__(256) void k_vec4(float* __restrict__ out) {
    float4* o = reinterpret_cast<float4*>(out + (size_t)blockIdx.x * BLK_E);
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int e = threadIdx.x; e < BLK_E / 4; e += 4 * 256) {
#pragma unroll
        for (int u = 0; u < 4; u++) if (e + u * 256 < BLK_E / 4) o[e + u * 256] = v;
    }
}
// gathered rows (sel: 512 row indices per block), dword loads + dword stores (the roipool3d copy loop)
__global__ __launch_bounds__(256) void k_gather_dword(const float* __restrict__ feat, const int* __restrict__ sel_all, float* __restrict__ out) {
    __shared__ int sel[512];
    for (int i = threadIdx.x; i < 512; i += 256) sel[i] = sel_all[blockIdx.x * 512 + i];
    __syncthreads();
    float* o = out + (size_t)blockIdx.x * BLK_E;
    const int W = 133, qstep = 256 / W, rstep = 256 - qstep * W;
    int srow = threadIdx.x / W, scol = threadIdx.x - srow * W;
    for (int e = threadIdx.x; e < BLK_E; e += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (e + u * 256 < BLK_E) v[u] = feat[(size_t)sel[srow] * W + scol];
            srow += qstep; scol += rstep;
            if (scol >= W) { scol -= W; srow++; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (e + u * 256 < BLK_E) o[e + u * 256] = v[u];
    }
}
// the same rows staged through LDS: dword gathers into a flat 32-row slab, 16-byte stores out of it (double-buffered)
#define SLAB_ROWS 32
#define SLAB_E (SLAB_ROWS * 133)
__global__ __launch_bounds__(256) void k_gather_vec4(const float* __restrict__ feat, const int* __restrict__ sel_all, float* __restrict__ out) {
    __shared__ int sel[512];
    __shared__ __attribute__((aligned(16))) float slab[2][SLAB_E];
    for (int i = threadIdx.x; i < 512; i += 256) sel[i] = sel_all[blockIdx.x * 512 + i];
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(out + (size_t)blockIdx.x * BLK_E);
    const int W = 133;
    constexpr int PER = (SLAB_E + 255) / 256;       // 17 dwords per thread and slab
    float v[PER];
    auto load = [&](int s) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = u * 256 + threadIdx.x;
            if (e < SLAB_E) { const int r = e / W, c = e - r * W; v[u] = feat[(size_t)sel[s * SLAB_ROWS + r] * W + c]; }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; u++) { const int e = u * 256 + threadIdx.x; if (e < SLAB_E) slab[buf][e] = v[u]; }
    };
    load(0);
    for (int s = 0; s < 512 / SLAB_ROWS; s++) {
        stash(s & 1);
        __syncthreads();
        if (s + 1 < 512 / SLAB_ROWS) load(s + 1);
        const float4* src = reinterpret_cast<const float4*>(slab[s & 1]);
        for (int q = threadIdx.x; q < SLAB_E / 4; q += 256) o[(size_t)s * (SLAB_E / 4) + q] = src[q];
    }
}
int main() {
    const int NB = 4096, N = 65536 * 8;
    float *out, *feat; int* sel;
    hipMalloc(&out, (size_t)NB * BLK_E * 4); hipMalloc(&feat, (size_t)N * 133 * 4); hipMalloc(&sel, NB * 512 * 4);
    hipMemset(feat, 0, (size_t)N * 133 * 4);
    std::vector<int> h(NB * 512);
    srand(1);
    for (int b = 0; b < NB; b++) { int base = (b / 512) * 65536; for (int i = 0; i < 512; i++) h[b * 512 + i] = base + rand() % 65536; }
    hipMemcpy(sel, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double mb = (double)NB * BLK_E * 4 / 1e6;
    for (int which = 0; which < 4; which++) {
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(k_dword, dim3(NB), dim3(256), 0, 0, out);
            if (which == 1) hipLaunchKernelGGL(k_vec4, dim3(NB), dim3(256), 0, 0, out);
            if (which == 2) hipLaunchKernelGGL(k_gather_dword, dim3(NB), dim3(256), 0, 0, feat, sel, out);
            if (which == 3) hipLaunchKernelGGL(k_gather_vec4, dim3(NB), dim3(256), 0, 0, feat, sel, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (it > 0 && ms < best) best = ms;
        }
        const char* nm[] = {"dword stores (registers)", "float4 stores (registers)", "gather dword -> dword stores", "gather dword -> LDS slab -> float4 stores"};
        printf("%-44s %8.1f us  %6.2f TB/s written (%.0f MB)\n", nm[which], best * 1e3, mb / best / 1e3 / 1e3, mb);
    }
    return 0;
}
