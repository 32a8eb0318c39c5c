// prcnn_ref_shim.h -- TEST INFRASTRUCTURE (oracle/_ref build only).
// Minimal host stand-ins for the torch / CUDA-runtime names the reference's four native
// source files use, so that those files can be compiled WHERE THEY LIE under /root/reference
// by plain g++ (no torch, no nvcc) and executed on the CPU as the parity pin for the oracle.
// Kernels are run by a sequential grid emulator (PRCNN_LAUNCH); every block is executed in
// two passes so that `__shared__` (mapped to `static`) staging followed by __syncthreads()
// sees fully populated data on the second pass (all reference kernels are idempotent).
#pragma once
#include <cmath>
#include <math.h>   // brings the float overloads of cos/sin/atan2/fabs into the global namespace, as nvcc and torch headers do
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

using std::min;
using std::max;

// ---- CUDA language / runtime surface -------------------------------------------------
#define __device__
#define __global__
#define __host__
#define __shared__ static
#define __syncthreads() ((void)0)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static dim3 blockIdx, threadIdx, blockDim, gridDim;

#define PRCNN_LAUNCH(kernel, grid, block, ...)                                            \
    do {                                                                                  \
        dim3 g_(grid), b_(block);                                                         \
        gridDim = g_; blockDim = b_;                                                      \
        for (unsigned bz_ = 0; bz_ < g_.z; bz_++)                                         \
        for (unsigned by_ = 0; by_ < g_.y; by_++)                                         \
        for (unsigned bx_ = 0; bx_ < g_.x; bx_++) {                                       \
            blockIdx = dim3(bx_, by_, bz_);                                               \
            for (int pass_ = 0; pass_ < 2; pass_++)                                       \
            for (unsigned tz_ = 0; tz_ < b_.z; tz_++)                                     \
            for (unsigned ty_ = 0; ty_ < b_.y; ty_++)                                     \
            for (unsigned tx_ = 0; tx_ < b_.x; tx_++) {                                   \
                threadIdx = dim3(tx_, ty_, tz_);                                          \
                kernel(__VA_ARGS__);                                                      \
            }                                                                             \
        }                                                                                 \
    } while (0)

typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return *p ? 0 : 2; }
static inline cudaError_t cudaFree(void* p) { free(p); return 0; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
static inline const char* cudaGetErrorString(cudaError_t) { return "emulated"; }

// ---- torch surface --------------------------------------------------------------------
namespace at {
struct TensorTypeStub { bool is_cuda() const { return true; } };
class Tensor {
  public:
    Tensor() : ptr_(nullptr) {}
    Tensor(void* p, std::vector<long> sizes) : ptr_(p), sizes_(std::move(sizes)) {}
    long size(int d) const { return sizes_.at(d); }
    template <class T> T* data() const { return (T*)ptr_; }
    bool is_contiguous() const { return true; }
    TensorTypeStub type() const { return TensorTypeStub(); }
  private:
    void* ptr_;
    std::vector<long> sizes_;
};
}  // namespace at

#define AT_CHECK(cond, ...) do { if (!(cond)) throw std::runtime_error("AT_CHECK failed: " #cond); } while (0)

struct prcnn_pybind_stub { template <class F> void def(const char*, F, const char*) {} };
#define PRCNN_CAT2(a, b) a##b
#define PRCNN_CAT(a, b) PRCNN_CAT2(a, b)
#define TORCH_EXTENSION_NAME prcnn_ref
#define PYBIND11_MODULE(name, var) static void __attribute__((unused)) PRCNN_CAT(prcnn_pybind_, name)(prcnn_pybind_stub& var)
