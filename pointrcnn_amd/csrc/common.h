// common.h -- shared helpers for the gfx950 point-ops kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/prcnn_pointops.h"

#define PRCNN_API extern "C" __attribute__((visibility("default")))

int prcnn_fail(int code, const char* fmt, ...);

#define PRCNN_REQUIRE(cond, ...)                              \
    do {                                                      \
        if (!(cond)) return prcnn_fail(PRCNN_EINVAL, __VA_ARGS__); \
    } while (0)

#define PRCNN_LAUNCH_CHECK(name)                                                                        \
    do {                                                                                                \
        hipError_t e_ = hipGetLastError();                                                              \
        if (e_ != hipSuccess) return prcnn_fail(PRCNN_EHIP, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int prcnn_divup(long a, long b) { return (int)((a + b - 1) / b); }

// Counters and small workspaces the kernels depend on are cleared by a KERNEL on the launch stream, not by hipMemsetAsync: a
// captured hipMemsetAsync becomes a memset node of the hipGraph, and round 4 traced a memory fault in group_compact_kernel
// (list offsets taken from a counter that was not zero when the kernel ran) to replays of graphs holding such nodes
// (tools/graph_fault_probe2.py, rocgdb); a kernel node is ordered like every other launch of the step.
static __global__ void prcnn_fill_words_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
static inline hipError_t prcnn_fill_words(void* p, uint32_t v, size_t nwords, hipStream_t s) {
    if (nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(prcnn_fill_words_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, s, (uint32_t*)p, v, nwords);
    return hipGetLastError();
}

// hipFuncSetAttribute applies to the function ON THE CURRENT DEVICE only, and one process may drive several devices from
// several threads (the reference's nn.DataParallel convention, SURVEY 8(b) "Threading").  The "already raised" state is
// therefore one bit per device, updated atomically; two threads racing on the same device both set the (idempotent)
// attribute.  Devices beyond 63 set it on every call.
struct PrcnnLdsLimit {
    std::atomic<uint64_t> done{0};
    bool raise(const void* fn, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const uint64_t bit = dev >= 0 && dev < 64 ? (1ull << dev) : 0;
        if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        if (bit) done.fetch_or(bit, std::memory_order_release);
        return true;
    }
};

// ---- wave64 cross-lane reductions on the VALU (DPP), no LDS ---------------------------------
// row_shr:1,2,4,8 builds an inclusive scan inside each 16-lane row, row_bcast:15 / row_bcast:31
// fold the rows; lane 63 ends up with the wave-wide result, returned in an SGPR (wave-uniform).
#define PRCNN_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rmask), 0xf, false)

__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, PRCNN_DPP(v, 0x111, 0xf));
    v = max(v, PRCNN_DPP(v, 0x112, 0xf));
    v = max(v, PRCNN_DPP(v, 0x114, 0xf));
    v = max(v, PRCNN_DPP(v, 0x118, 0xf));
    v = max(v, PRCNN_DPP(v, 0x142, 0xa));
    v = max(v, PRCNN_DPP(v, 0x143, 0xc));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, PRCNN_DPP(v, 0x111, 0xf));
    v = min(v, PRCNN_DPP(v, 0x112, 0xf));
    v = min(v, PRCNN_DPP(v, 0x114, 0xf));
    v = min(v, PRCNN_DPP(v, 0x118, 0xf));
    v = min(v, PRCNN_DPP(v, 0x142, 0xa));
    v = min(v, PRCNN_DPP(v, 0x143, 0xc));
    return __builtin_amdgcn_readlane(v, 63);
}
// reductions over the first 16 lanes only (one DPP row); result read from lane 15
__device__ __forceinline__ int row0_max_i32(int v) {
    v = max(v, PRCNN_DPP(v, 0x111, 0xf));
    v = max(v, PRCNN_DPP(v, 0x112, 0xf));
    v = max(v, PRCNN_DPP(v, 0x114, 0xf));
    v = max(v, PRCNN_DPP(v, 0x118, 0xf));
    return __builtin_amdgcn_readlane(v, 15);
}
__device__ __forceinline__ int row0_min_i32(int v) {
    v = min(v, PRCNN_DPP(v, 0x111, 0xf));
    v = min(v, PRCNN_DPP(v, 0x112, 0xf));
    v = min(v, PRCNN_DPP(v, 0x114, 0xf));
    v = min(v, PRCNN_DPP(v, 0x118, 0xf));
    return __builtin_amdgcn_readlane(v, 15);
}

// squared distance under the canonical arithmetic contract: individually rounded, left to right.
// (the library is also built with -ffp-contract=off; the intrinsics make the intent explicit)
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
