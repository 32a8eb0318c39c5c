// gather.hip -- gather_operation / grouping_operation / three_interpolate and their backward passes,
// in the reference op surface's channel-first layout (B,C,N).
//
// Replaces pointnet2_cuda.{gather_points,group_points,three_interpolate}{,_grad}_wrapper [UPSTREAM, not in
// tree]; semantics per SURVEY Appendix A.2 / A.4 / A.6.  These are the drop-in ops for callers that use
// the op surface directly (and for training); the inference fast path never materialises grouped tensors
// (mlp.hip gathers straight into the MFMA A-tile).
//
// HBM-bound copies: one thread per output (m[,s]) position, looping over channels with the index held
// in a register, so index traffic is read once (upstream re-reads idx for every channel) and the writes
// are fully coalesced; the gathered reads are scattered by nature of the (B,C,N) layout.
#include "common.h"

#define G_THREADS 256

__global__ __launch_bounds__(G_THREADS) void gather_kernel(const float* __restrict__ feat,
                                                           const int32_t* __restrict__ idx, int C, int N, int J,
                                                           float* __restrict__ out) {
    // J = outputs per (b,c) plane: M for gather, M*ns for group
    const int b = blockIdx.y;
    const int j = blockIdx.x * G_THREADS + threadIdx.x;
    if (j >= J) return;
    const int id = idx[(size_t)b * J + j];
    const float* f = feat + (size_t)b * C * N + id;
    float* o = out + (size_t)b * C * J + j;
#pragma unroll 4
    for (int c = 0; c < C; c++) o[(size_t)c * J] = f[(size_t)c * N];
}

__global__ __launch_bounds__(G_THREADS) void gather_grad_kernel(const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx, int C, int N, int J,
                                                                float* __restrict__ grad_feat) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * G_THREADS + threadIdx.x;
    if (j >= J) return;
    const int id = idx[(size_t)b * J + j];
    const float* g = grad_out + (size_t)b * C * J + j;
    float* gf = grad_feat + (size_t)b * C * N + id;
    for (int c = 0; c < C; c++) atomicAdd(gf + (size_t)c * N, g[(size_t)c * J]);
}

__global__ __launch_bounds__(G_THREADS) void three_interp_kernel(const float* __restrict__ feat,
                                                                 const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ w, int C, int m, int n,
                                                                 float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * G_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t o3 = ((size_t)b * n + i) * 3;
    const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
    const float w0 = w[o3], w1 = w[o3 + 1], w2 = w[o3 + 2];
    const float* f = feat + (size_t)b * C * m;
    float* o = out + (size_t)b * C * n + i;
#pragma unroll 4
    for (int c = 0; c < C; c++) {
        const float* fc = f + (size_t)c * m;
        float v = __fadd_rn(__fadd_rn(__fmul_rn(w0, fc[i0]), __fmul_rn(w1, fc[i1])), __fmul_rn(w2, fc[i2]));
        o[(size_t)c * n] = v;
    }
}

__global__ __launch_bounds__(G_THREADS) void three_interp_grad_kernel(const float* __restrict__ grad_out,
                                                                      const int32_t* __restrict__ idx,
                                                                      const float* __restrict__ w, int C, int n, int m,
                                                                      float* __restrict__ grad_feat) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * G_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t o3 = ((size_t)b * n + i) * 3;
    const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
    const float w0 = w[o3], w1 = w[o3 + 1], w2 = w[o3 + 2];
    const float* g = grad_out + (size_t)b * C * n + i;
    float* gf = grad_feat + (size_t)b * C * m;
    for (int c = 0; c < C; c++) {
        float gv = g[(size_t)c * n];
        atomicAdd(gf + (size_t)c * m + i0, gv * w0);
        atomicAdd(gf + (size_t)c * m + i1, gv * w1);
        atomicAdd(gf + (size_t)c * m + i2, gv * w2);
    }
}

__global__ __launch_bounds__(G_THREADS) void gather_rows_kernel(const float* __restrict__ in, int ld_in,
                                                                const int32_t* __restrict__ idx, int N, int M, int C,
                                                                float* __restrict__ out) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * G_THREADS + threadIdx.x;      // element of the (M,C) output plane
    if (e >= (long)M * C) return;
    const int m = (int)(e / C), c = (int)(e - (long)m * C);
    out[((size_t)b * M + m) * C + c] = in[((size_t)b * N + idx[(size_t)b * M + m]) * ld_in + c];
}

static int check_bcn(const char* op, const void* a, const void* b, const void* c, int B, int C, int N, long J) {
    if (B < 0 || C < 0 || N <= 0 || J < 0) return prcnn_fail(PRCNN_EINVAL, "%s: bad shape B=%d C=%d N=%d J=%ld", op, B, C, N, J);
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;      // empty problem: pointers may legitimately be null
    if (!a || !b || !c) return prcnn_fail(PRCNN_EINVAL, "%s: null pointer", op);
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather(const float* feat, const int32_t* idx, int B, int C, int N, int M, float* out,
                           prcnn_stream_t stream) {
    int rc = check_bcn("prcnn_gather", feat, idx, out, B, C, N, M);
    if (rc) return rc;
    if (B == 0 || C == 0 || M == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_kernel, dim3(prcnn_divup(M, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream, feat,
                       idx, C, N, M, out);
    PRCNN_LAUNCH_CHECK("prcnn_gather");
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, float* grad_feat,
                                prcnn_stream_t stream) {
    int rc = check_bcn("prcnn_gather_grad", grad_out, idx, grad_feat, B, C, N, M);
    if (rc) return rc;
    if (B == 0 || C == 0 || M == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_grad_kernel, dim3(prcnn_divup(M, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                       grad_out, idx, C, N, M, grad_feat);
    PRCNN_LAUNCH_CHECK("prcnn_gather_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_group(const float* feat, const int32_t* idx, int B, int C, int N, int M, int nsample, float* out,
                          prcnn_stream_t stream) {
    long J = (long)M * nsample;
    int rc = check_bcn("prcnn_group", feat, idx, out, B, C, N, J);
    if (rc) return rc;
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_kernel, dim3(prcnn_divup(J, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream, feat,
                       idx, C, N, (int)J, out);
    PRCNN_LAUNCH_CHECK("prcnn_group");
    return PRCNN_OK;
}

PRCNN_API int prcnn_group_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                               float* grad_feat, prcnn_stream_t stream) {
    long J = (long)M * nsample;
    int rc = check_bcn("prcnn_group_grad", grad_out, idx, grad_feat, B, C, N, J);
    if (rc) return rc;
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_grad_kernel, dim3(prcnn_divup(J, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                       grad_out, idx, C, N, (int)J, grad_feat);
    PRCNN_LAUNCH_CHECK("prcnn_group_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_interp(const float* feat, const int32_t* idx, const float* weight, int B, int C, int m, int n,
                                 float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && C >= 0 && m > 0 && n >= 0, "prcnn_three_interp: bad shape B=%d C=%d m=%d n=%d", B, C, m, n);
    if (B == 0 || C == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(feat && idx && weight && out, "prcnn_three_interp: null pointer");
    hipLaunchKernelGGL(three_interp_kernel, dim3(prcnn_divup(n, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                       feat, idx, weight, C, m, n, out);
    PRCNN_LAUNCH_CHECK("prcnn_three_interp");
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_interp_grad(const float* grad_out, const int32_t* idx, const float* weight, int B, int C,
                                      int n, int m, float* grad_feat, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && C >= 0 && m > 0 && n >= 0, "prcnn_three_interp_grad: bad shape");
    if (B == 0 || C == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grad_out && idx && weight && grad_feat, "prcnn_three_interp_grad: null pointer");
    hipLaunchKernelGGL(three_interp_grad_kernel, dim3(prcnn_divup(n, G_THREADS), B), dim3(G_THREADS), 0,
                       (hipStream_t)stream, grad_out, idx, weight, C, n, m, grad_feat);
    PRCNN_LAUNCH_CHECK("prcnn_three_interp_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather_rows(const float* in_cl, int ld_in, const int32_t* idx, int B, int N, int M, int C,
                                float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0 && ld_in >= C, "prcnn_gather_rows: bad shape");
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(in_cl && idx && out, "prcnn_gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(prcnn_divup((long)M * C, G_THREADS), B), dim3(G_THREADS), 0,
                       (hipStream_t)stream, in_cl, ld_in, idx, N, M, C, out);
    PRCNN_LAUNCH_CHECK("prcnn_gather_rows");
    return PRCNN_OK;
}
