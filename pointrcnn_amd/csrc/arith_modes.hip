// arith_modes.hip -- COMPARISON-MODE kernels for the PointNet++ index operators: selectable tie order and selectable squared-distance
// arithmetic (include/prcnn_pointops.h: prcnn_fps_mode, prcnn_ball_query_arith, prcnn_three_nn_arith).
//
// Why this exists ("what's missing" #3): the upstream kernels (sshaoshuai/Pointnet2.PyTorch, absent from the reference
// tree: .gitmodules:1-4) write the squared distance as  (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)  and are built by nvcc,
// whose default -fmad=true contracts  a*a + b*b + c*c  into  fma(c,c, fma(b,b, a*a)).  The contract of this library (and of every
// fast kernel in fps.hip / neighbor.hip / grid.hip) is three individually rounded products summed left to right.  The two agree
// except in the last bit of some distances -- and an index operator turns a last-bit difference into a different index wherever two
// candidates are that close.  PRCNN_ARITH_UPSTREAM evaluates the contracted form, so that (a) an actual upstream build can be compared
// index for index, and (b) the rate at which the two arithmetics disagree can be MEASURED (tools/arith_disagreement.py).
//
// These are straightforward kernels -- one workgroup per frame for FPS with the running minima in HBM, one thread per query with a
// serial scan for ball_query / three_nn -- held to oracle/prcnn_oracle.c (prcnn_cpu_fps_mode, prcnn_cpu_ball_query_arith,
// prcnn_cpu_three_nn_arith) bit for bit in BOTH arithmetics.  Not a fast path and not used by any module.
#include "common.h"

template <int ARITH>
__device__ __forceinline__ float sqdist_mode(float ax, float ay, float az, float bx, float by, float bz) {
    if (ARITH == PRCNN_ARITH_CANONICAL) return sqdist3(ax, ay, az, bx, by, bz);
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// FPS: T threads scan k = t, t+T, ... keeping their first maximum; among equal maxima the winner is the lowest point index
// (canonical order) or the lowest thread (upstream order: argmin (k mod T, k)).
template <int ARITH>
__global__ __launch_bounds__(1024) void fps_mode_kernel(const float* __restrict__ xyz, int N, int npoint, int T, int upstream_order,
                                                        float* __restrict__ tmp, int32_t* __restrict__ idx_out) {
    constexpr int NWMAX = 16;
    __shared__ int sval[2][NWMAX], skey[2][NWMAX], sidx[2][NWMAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float* __restrict__ t = tmp + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    for (int k = tid; k < N; k += blockDim.x) t[k] = 1e10f;
    if (tid == 0 && npoint > 0) out[0] = 0;
    __syncthreads();
    int old = 0;
    for (int j = 1; j < npoint; j++) {
        const float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
        float best = -2.0f;
        int bk = 0x7fffffff;
        if (tid < T) {
            for (int k = tid; k < N; k += T) {
                const float d = sqdist_mode<ARITH>(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
                float v = t[k];
                v = d < v ? d : v;
                t[k] = v;
                if (v > best) { best = v; bk = k; }          // strict: the thread's FIRST maximum
            }
        }
        // running minima are >= 0 (or the -2 of an idle thread): integer order == float order
        const int vb = __float_as_int(best);
        const int key = upstream_order ? tid : bk;            // what breaks a tie between threads
        const int wmax = wave_max_i32(vb);
        const int wkey = wave_min_i32(vb == wmax ? key : 0x7fffffff);
        const int wlane = __builtin_ctzll(__ballot(vb == wmax && key == wkey));
        const int wk = __builtin_amdgcn_readlane(bk, wlane);
        if (lane == 0) { sval[j & 1][wave] = wmax; skey[j & 1][wave] = wkey; sidx[j & 1][wave] = wk; }
        __syncthreads();
        const int v = lane < nw ? sval[j & 1][lane] : (int)0x80000000;
        const int ky = lane < nw ? skey[j & 1][lane] : 0x7fffffff;
        const int id = lane < nw ? sidx[j & 1][lane] : 0;
        const int gmax = row0_max_i32(v);
        const int gkey = row0_min_i32(v == gmax ? ky : 0x7fffffff);
        const int win = __builtin_ctzll(__ballot(v == gmax && ky == gkey));
        old = __builtin_amdgcn_readlane(id, win);
        if (tid == 0) out[j] = old;
    }
}

template <int ARITH>
__global__ __launch_bounds__(256) void ball_query_arith_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int M,
                                                               float r2, int nsample, int32_t* __restrict__ idx) {
    const int b = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const float* q = new_xyz + ((size_t)b * M + m) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int32_t* o = idx + ((size_t)b * M + m) * nsample;
    int cnt = 0, first = 0;
    for (int k = 0; k < N && cnt < nsample; k++) {
        const float d2 = sqdist_mode<ARITH>(qx, qy, qz, p[k * 3], p[k * 3 + 1], p[k * 3 + 2]);
        if (d2 < r2) {
            if (cnt == 0) first = k;
            o[cnt++] = k;
        }
    }
    for (int s = cnt; s < nsample; s++) o[s] = first;          // pad with the first hit (0 if none)
}

template <int ARITH>
__global__ __launch_bounds__(256) void three_nn_arith_kernel(const float* __restrict__ unknown, const float* __restrict__ known, int n, int m,
                                                             float* __restrict__ dist2, int32_t* __restrict__ idx) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* __restrict__ kn = known + (size_t)b * m * 3;
    const float* u = unknown + ((size_t)b * n + i) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; k++) {
        const float d = sqdist_mode<ARITH>(ux, uy, uz, kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
        else if (d < b3) { b3 = d; i3 = k; }
    }
    float* od = dist2 + ((size_t)b * n + i) * 3;
    int32_t* oi = idx + ((size_t)b * n + i) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
}

#define PRCNN_ARITH_OK(a) ((a) == PRCNN_ARITH_CANONICAL || (a) == PRCNN_ARITH_UPSTREAM)

PRCNN_API int prcnn_fps_mode(const float* xyz, int B, int N, int npoint, int order, int arith, float* tmp, int32_t* idx,
                             prcnn_stream_t stream) {
    PRCNN_REQUIRE(order == PRCNN_FPS_ORDER_CANONICAL || order == PRCNN_FPS_ORDER_UPSTREAM, "prcnn_fps_mode: unknown order %d", order);
    PRCNN_REQUIRE(PRCNN_ARITH_OK(arith), "prcnn_fps_mode: unknown arithmetic %d", arith);
    PRCNN_REQUIRE(B >= 0 && N > 0 && npoint >= 0 && npoint <= N, "prcnn_fps_mode: bad shape B=%d N=%d npoint=%d", B, N, npoint);
    if (B == 0 || npoint == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && idx && tmp, "prcnn_fps_mode: null pointer (the (B,N) tmp buffer is required)");
    int T = 1;
    while (T * 2 <= N && T < 1024) T <<= 1;
    const int block = T < 64 ? 64 : T;
    if (arith == PRCNN_ARITH_UPSTREAM)
        hipLaunchKernelGGL(fps_mode_kernel<PRCNN_ARITH_UPSTREAM>, dim3(B), dim3(block), 0, (hipStream_t)stream, xyz, N, npoint, T, order, tmp, idx);
    else
        hipLaunchKernelGGL(fps_mode_kernel<PRCNN_ARITH_CANONICAL>, dim3(B), dim3(block), 0, (hipStream_t)stream, xyz, N, npoint, T, order, tmp, idx);
    PRCNN_LAUNCH_CHECK("prcnn_fps_mode");
    return PRCNN_OK;
}

PRCNN_API int prcnn_ball_query_arith(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample, int arith,
                                     int32_t* idx, prcnn_stream_t stream) {
    PRCNN_REQUIRE(PRCNN_ARITH_OK(arith), "prcnn_ball_query_arith: unknown arithmetic %d", arith);
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample > 0, "prcnn_ball_query_arith: bad shape B=%d N=%d M=%d nsample=%d", B, N, M, nsample);
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && new_xyz && idx, "prcnn_ball_query_arith: null pointer");
    const float r2 = radius * radius;                   // in float, as upstream (host code: no contraction possible with one product)
    dim3 grid(prcnn_divup(M, 256), B);
    if (arith == PRCNN_ARITH_UPSTREAM)
        hipLaunchKernelGGL(ball_query_arith_kernel<PRCNN_ARITH_UPSTREAM>, grid, dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, M, r2, nsample, idx);
    else
        hipLaunchKernelGGL(ball_query_arith_kernel<PRCNN_ARITH_CANONICAL>, grid, dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, M, r2, nsample, idx);
    PRCNN_LAUNCH_CHECK("prcnn_ball_query_arith");
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_nn_arith(const float* unknown, const float* known, int B, int n, int m, int arith, float* dist2, int32_t* idx,
                                   prcnn_stream_t stream) {
    PRCNN_REQUIRE(PRCNN_ARITH_OK(arith), "prcnn_three_nn_arith: unknown arithmetic %d", arith);
    PRCNN_REQUIRE(B >= 0 && n >= 0 && m > 0, "prcnn_three_nn_arith: bad shape B=%d n=%d m=%d", B, n, m);
    if (B == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(unknown && known && dist2 && idx, "prcnn_three_nn_arith: null pointer");
    dim3 grid(prcnn_divup(n, 256), B);
    if (arith == PRCNN_ARITH_UPSTREAM)
        hipLaunchKernelGGL(three_nn_arith_kernel<PRCNN_ARITH_UPSTREAM>, grid, dim3(256), 0, (hipStream_t)stream, unknown, known, n, m, dist2, idx);
    else
        hipLaunchKernelGGL(three_nn_arith_kernel<PRCNN_ARITH_CANONICAL>, grid, dim3(256), 0, (hipStream_t)stream, unknown, known, n, m, dist2, idx);
    PRCNN_LAUNCH_CHECK("prcnn_three_nn_arith");
    return PRCNN_OK;
}
