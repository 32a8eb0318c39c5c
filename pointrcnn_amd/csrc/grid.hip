// grid.hip -- neighbour search over a per-frame uniform grid: ball_query / three_nn with the SAME results as the
// brute-force scans of neighbor.hip (and of the reference's CUDA ops), but touching only the cells a query can reach.
//
// The reference (and neighbor.hip) test every query against every point of the frame: 67 M distance evaluations per
// frame for SA1's ball query and again for FP0's three_nn -- brute force already runs near the packed
// fp32 VALU limit here, so the only way further is to not do the work.  Points are binned once per frame into a 64x64
// (three_nn) or 128x128 (ball query) grid over the x-z plane (the LiDAR ground plane; y spans a few metres) and stored cell-contiguous as float4
// (x, y, z, original index); a query then visits the 3x3-ish block of cells its ball overlaps (ball_query) or grows a
// square ring by ring until the third-best distance is provably final (three_nn): ~30 candidates instead of 16 384.
//
// Exactness: every candidate is tested with the canonical distance arithmetic of neighbor.hip (common.h sqdist3), the
// candidate cell range is a superset of the ball by monotonicity of the cell function, and both result rules are
// restated order-independently --
//   ball_query: "first nsample hits in index order, padded with the first hit" == the nsample SMALLEST indices among
//               the hits, ascending, padded with the smallest;
//   three_nn  : strict-'<' insertion in index order == the three smallest (d2, index) pairs, lexicographically
// -- so the output is bit-identical to the scan regardless of the (arbitrary) order of points inside a cell.
#include "grid_layout.h"

#define GRID_BUILD_THREADS 1024

__device__ __forceinline__ const int32_t* grid_cells(const void* g, size_t fb, int b) { return (const int32_t*)((const char*)g + fb * b + sizeof(GridHeader)); }
__device__ __forceinline__ const float4* grid_points(const void* g, size_t fb, int b) {
    size_t off = (sizeof(GridHeader) + (size_t)(GRID_CELLS_MAX + 1) * 4 + 15) & ~(size_t)15;
    return (const float4*)((const char*)g + fb * b + off);
}

// monotone non-decreasing in v for fixed (o, inv): fp32 subtract, multiply by a positive constant, truncate, clamp
__device__ __forceinline__ int cell_coord(float v, float o, float inv, int dim) {
    float t = (v - o) * inv;
    t = fminf(fmaxf(t, 0.0f), (float)(dim - 1));            // NaN -> 0
    return (int)t;
}

__device__ __forceinline__ float block_reduce(float v, bool is_min, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        float o = __shfl_xor(v, d, 64);
        v = is_min ? fminf(v, o) : fmaxf(v, o);
    }
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nw; w++) r = is_min ? fminf(r, scratch[w]) : fmaxf(r, scratch[w]);
    return r;
}

__device__ __forceinline__ float block_reduce_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < nw; w++) r += scratch[w];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(GRID_BUILD_THREADS) void grid_build_kernel(const float* __restrict__ xyz, int N, float min_cell, int dim,
                                                                        void* grid, size_t fb) {
    extern __shared__ int cnt[];                  // dim * dim counters / scatter cursors
    const int ncell = dim * dim, per_thread = ncell / GRID_BUILD_THREADS;       // 4 or 16
    __shared__ float red[16];
    __shared__ int wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    GridHeader* H = (GridHeader*)grid_header(grid, fb, b);
    int32_t* cstart = (int32_t*)grid_cells(grid, fb, b);
    float4* sorted = (float4*)grid_points(grid, fb, b);
    // bounds over the finite points
    float mnx = INFINITY, mxx = -INFINITY, mnz = INFINITY, mxz = -INFINITY;
    for (int i = tid; i < N; i += GRID_BUILD_THREADS) {
        float x = p[i * 3], z = p[i * 3 + 2];
        if (fabsf(x) < INFINITY && fabsf(z) < INFINITY) {
            mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
        }
    }
    mnx = block_reduce(mnx, true, red); mxx = block_reduce(mxx, false, red);
    mnz = block_reduce(mnz, true, red); mxz = block_reduce(mxz, false, red);
    if (!(mnx <= mxx)) { mnx = mxx = 0.f; mnz = mxz = 0.f; }           // no finite point at all
    float cs = fmaxf(fmaxf(mxx - mnx, mxz - mnz) / (float)dim * 1.0001f, fmaxf(min_cell, 1e-3f));
    const float inv = 1.0f / cs;
    for (int c = tid; c < ncell; c += GRID_BUILD_THREADS) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += GRID_BUILD_THREADS) {
        int c = cell_coord(p[i * 3 + 2], mnz, inv, dim) * dim + cell_coord(p[i * 3], mnx, inv, dim);
        atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: per_thread consecutive cells per thread, wave scan, wave offsets
    int ssum = 0, occ = 0;
    for (int u = 0; u < per_thread; u++) {
        const int c = cnt[tid * per_thread + u];
        ssum += c;
        occ += c > 0;
    }
    const float occupied = block_reduce_sum((float)occ, red);
    int inc = ssum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    int run = woff + inc - ssum;
    for (int u = 0; u < per_thread; u++) {
        const int c = cnt[tid * per_thread + u];
        cstart[tid * per_thread + u] = run;
        cnt[tid * per_thread + u] = run;         // becomes the scatter cursor
        run += c;
    }
    if (tid == GRID_BUILD_THREADS - 1) cstart[ncell] = run;
    if (tid == 0) {
        H->x0 = mnx; H->z0 = mnz; H->inv_cs = inv; H->cs = cs; H->dim = dim;
        const float side = 2.0f * min_cell * inv + 1.0f;                       // cells a ball of radius min_cell spans per axis
        H->cand = (float)N / fmaxf(occupied, 1.0f) * side * side;
        H->pad1 = H->pad2 = 0;
    }
    __syncthreads();
    for (int i = tid; i < N; i += GRID_BUILD_THREADS) {
        float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
        int c = cell_coord(z, mnz, inv, dim) * dim + cell_coord(x, mnx, inv, dim);
        int pos = atomicAdd(&cnt[c], 1);
        sorted[pos] = make_float4(x, y, z, __int_as_float(i));
    }
}

// ---- ball query over the grid: one thread per centroid, per-thread sorted hit lists in LDS (k-major) ----------
#define GBQ_THREADS 64

// keep the `cap` smallest indices, ascending; returns the new count
__device__ __forceinline__ int insert_sorted(int* list, int stride, int cnt, int cap, int k) {
    if (cnt == cap) {
        if (k >= list[(cap - 1) * stride]) return cnt;
        cnt = cap - 1;                            // the largest falls off
    }
    int pos = cnt;
    while (pos > 0 && list[(pos - 1) * stride] > k) { list[pos * stride] = list[(pos - 1) * stride]; pos--; }
    list[pos * stride] = k;
    return cnt + 1;
}

template <bool DUAL>
__global__ __launch_bounds__(GBQ_THREADS) void grid_ball_query_kernel(const void* __restrict__ grid, size_t fb,
                                                                     const float* __restrict__ new_xyz, int M, float ra,
                                                                     float r2a, int nsa, int32_t* __restrict__ idxa, float rb,
                                                                     float r2b, int nsb, int32_t* __restrict__ idxb, int N,
                                                                     int skip_dense) {
    extern __shared__ int lists[];               // [nsa + nsb][GBQ_THREADS]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int m = blockIdx.x * GBQ_THREADS + tid;
    if (m >= M) return;
    const GridHeader H = *grid_header(grid, fb, b);
    if (skip_dense && grid_frame_dense(H, N)) return;          // the scan kernel takes this frame (same results)
    const int32_t* __restrict__ cstart = grid_cells(grid, fb, b);
    const float4* __restrict__ pts = grid_points(grid, fb, b);
    const float* q = new_xyz + ((size_t)b * M + m) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int* la = lists + tid;
    int* lb = lists + nsa * GBQ_THREADS + tid;
    int cnta = 0, cntb = 0;
    const float rmax = DUAL ? fmaxf(ra, rb) : ra;
    const float r2max = DUAL ? fmaxf(r2a, r2b) : r2a;
    // a hit has |dx| < r (its squared distance, a sum of rounded non-negative terms, is below r2): widen r by a
    // relative margin that dwarfs the rounding of r*r and of the subtraction; cell_coord is monotone
    const float rs = rmax * 1.001f + 1e-6f;
    if (qx == qx && qz == qz && r2max > 0.0f) {   // NaN centroid: no hits
        const int dim = H.dim;
        const int cx0 = cell_coord(qx - rs, H.x0, H.inv_cs, dim), cx1 = cell_coord(qx + rs, H.x0, H.inv_cs, dim);
        const int cz0 = cell_coord(qz - rs, H.z0, H.inv_cs, dim), cz1 = cell_coord(qz + rs, H.z0, H.inv_cs, dim);
        for (int cz = cz0; cz <= cz1; cz++) {
            const int beg = cstart[cz * dim + cx0], end = cstart[cz * dim + cx1 + 1];   // one contiguous run per row
            for (int t = beg; t < end; t++) {
                const float4 c = pts[t];
                const float d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
                if (d < r2max) {
                    const int k = __float_as_int(c.w);
                    if (d < r2a) cnta = insert_sorted(la, GBQ_THREADS, cnta, nsa, k);
                    if (DUAL && d < r2b) cntb = insert_sorted(lb, GBQ_THREADS, cntb, nsb, k);
                }
            }
        }
    }
    int32_t* oa = idxa + ((size_t)b * M + m) * nsa;
    const int fa = cnta ? la[0] : 0;
    for (int s = 0; s < nsa; s++) oa[s] = s < cnta ? la[s * GBQ_THREADS] : fa;
    if (DUAL) {
        int32_t* ob = idxb + ((size_t)b * M + m) * nsb;
        const int fbk = cntb ? lb[0] : 0;
        for (int s = 0; s < nsb; s++) ob[s] = s < cntb ? lb[s * GBQ_THREADS] : fbk;
    }
}

template <typename T> struct __attribute__((packed, aligned(4))) Trip { T a, b, c; };

// ---- three_nn over the grid: one thread per query, square rings until the third-best distance is final -------
#define GNN_THREADS 256
__global__ __launch_bounds__(GNN_THREADS) void grid_three_nn_kernel(const void* __restrict__ grid, size_t fb,
                                                                    const float* __restrict__ unknown, int n,
                                                                    float* __restrict__ dist2, int32_t* __restrict__ idx,
                                                                    float* __restrict__ weight) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * GNN_THREADS + threadIdx.x;
    if (i >= n) return;
    const GridHeader H = *grid_header(grid, fb, b);
    const int32_t* __restrict__ cstart = grid_cells(grid, fb, b);
    const float4* __restrict__ pts = grid_points(grid, fb, b);
    const float* u = unknown + ((size_t)b * n + i) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    const int GD = H.dim;
    // always_inline: as an out-of-line call the by-reference captures (the running top-3) live in scratch memory
    auto visit = [&](int cz, int cxa, int cxb) __attribute__((always_inline)) {          // cells [cxa, cxb] of row cz (all inside the grid)
        const int beg = cstart[cz * GD + cxa], end = cstart[cz * GD + cxb + 1];
        for (int t = beg; t < end; t++) {
            const float4 c = pts[t];
            const float dd = sqdist3(ux, uy, uz, c.x, c.y, c.z);
            const int k = __float_as_int(c.w);
            // lexicographic (distance, index): what strict-'<' insertion in index order produces
            // (written with selects: as an if / else-if ladder the compiler turned the six running values into a
            //  dynamically indexed array in scratch memory)
            if (dd < b3 || (dd == b3 && k < i3 && dd < INFINITY)) {
                const bool lt1 = dd < b1 || (dd == b1 && k < i1);
                const bool lt2 = lt1 || dd < b2 || (dd == b2 && k < i2);
                b3 = lt2 ? b2 : dd; i3 = lt2 ? i2 : k;
                b2 = lt1 ? b1 : (lt2 ? dd : b2); i2 = lt1 ? i1 : (lt2 ? k : i2);
                b1 = lt1 ? dd : b1; i1 = lt1 ? k : i1;
            }
        }
    };
    const bool finite_q = fabsf(ux) < INFINITY && fabsf(uz) < INFINITY;
    if (finite_q) {
        // unclamped cell coordinates of the query (it may lie outside the grid's bounding box)
        const float fx = (ux - H.x0) * H.inv_cs, fz = (uz - H.z0) * H.inv_cs;
        // A query outside the points' bounding box starts from the ring just outside the grid ([-1, GD]) on that side: the
        // bounds below measure from the query's real coordinates to cell sides, so they stay valid, and a far-away query
        // costs at most GD rings instead of one ring per cell of its distance.
        const int qcx = (int)floorf(fminf(fmaxf(fx, -1.0f), (float)GD)), qcz = (int)floorf(fminf(fmaxf(fz, -1.0f), (float)GD));
        for (int R = 0;; R++) {
            const int x0 = qcx - R, x1 = qcx + R, z0 = qcz - R, z1 = qcz + R;
            const int cxa = max(x0, 0), cxb = min(x1, GD - 1);
            if (cxa <= cxb) {
                if (z0 >= 0 && z0 < GD) visit(z0, cxa, cxb);                      // bottom row of the ring
                if (R > 0 && z1 >= 0 && z1 < GD) visit(z1, cxa, cxb);             // top row
            }
            if (R > 0) {                                                                  // left / right columns
                const int za = max(z0 + 1, 0), zb = min(z1 - 1, GD - 1);
                for (int cz = za; cz <= zb; cz++) {
                    if (x0 >= 0 && x0 < GD) visit(cz, x0, x0);
                    if (x1 >= 0 && x1 < GD) visit(cz, x1, x1);
                }
            }
            if (x0 <= 0 && z0 <= 0 && x1 >= GD - 1 && z1 >= GD - 1) break;    // whole grid visited
            // every point not yet visited lies outside the square of cells [x0,x1] x [z0,z1]; points are binned by a
            // CLAMPED coordinate, so only sides strictly inside the grid bound anything.  Planar distance from the
            // query to the nearest such side, shrunk by a margin far above the rounding of the cell arithmetic:
            float lb = INFINITY;
            if (x0 > 0) lb = fminf(lb, ux - (H.x0 + (float)x0 * H.cs));
            if (x1 < GD - 1) lb = fminf(lb, (H.x0 + (float)(x1 + 1) * H.cs) - ux);
            if (z0 > 0) lb = fminf(lb, uz - (H.z0 + (float)z0 * H.cs));
            if (z1 < GD - 1) lb = fminf(lb, (H.z0 + (float)(z1 + 1) * H.cs) - uz);
            lb = lb * 0.999f - 1e-4f * H.cs;
            if (lb > 0.0f && b3 < lb * lb) break;
        }
    }
    // a query's three values are 12 contiguous bytes of each output: ONE global_store_dwordx3 per array (a wave then writes
    // 768 contiguous bytes per instruction; nine 4-byte stores at a 12-byte stride measured 10x write amplification)
    const size_t o = ((size_t)b * n + i) * 3;
    *reinterpret_cast<Trip<float>*>(dist2 + o) = Trip<float>{b1, b2, b3};
    *reinterpret_cast<Trip<int32_t>*>(idx + o) = Trip<int32_t>{i1, i2, i3};
    if (weight) {                                 // same expressions as three_nn_kernel (neighbor.hip)
        float r0 = 1.0f / (sqrtf(b1) + 1e-8f);
        float r1 = 1.0f / (sqrtf(b2) + 1e-8f);
        float r2 = 1.0f / (sqrtf(b3) + 1e-8f);
        float s = (r0 + r1) + r2;
        *reinterpret_cast<Trip<float>*>(weight + o) = Trip<float>{r0 / s, r1 / s, r2 / s};
    }
}

// ---- host ------------------------------------------------------------------------------------------------
PRCNN_API size_t prcnn_grid_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return grid_frame_bytes(N) * (size_t)B;
}

PRCNN_API int prcnn_grid_build(const float* xyz, int B, int N, float min_cell, int cells_per_axis, void* grid, size_t grid_bytes,
                               prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && min_cell >= 0.0f, "prcnn_grid_build: bad shape B=%d N=%d min_cell=%g", B, N, (double)min_cell);
    PRCNN_REQUIRE(cells_per_axis == 64 || cells_per_axis == 128, "prcnn_grid_build: cells_per_axis must be 64 or 128, got %d", cells_per_axis);
    if (B == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && grid, "prcnn_grid_build: null pointer");
    PRCNN_REQUIRE(((uintptr_t)grid & 15) == 0, "prcnn_grid_build: grid buffer must be 16-byte aligned");
    PRCNN_REQUIRE(grid_bytes >= prcnn_grid_bytes(B, N), "prcnn_grid_build: buffer %zu < %zu bytes", grid_bytes, prcnn_grid_bytes(B, N));
    static PrcnnLdsLimit attr;
    if (!attr.raise((const void*)grid_build_kernel, GRID_CELLS_MAX * 4))
        return prcnn_fail(PRCNN_EHIP, "prcnn_grid_build: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(GRID_BUILD_THREADS), (size_t)cells_per_axis * cells_per_axis * 4, (hipStream_t)stream,
                       xyz, N, min_cell, cells_per_axis, grid, grid_frame_bytes(N));
    PRCNN_LAUNCH_CHECK("prcnn_grid_build");
    return PRCNN_OK;
}

PRCNN_API int prcnn_ball_query2_grid(const void* grid, const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a,
                                     int nsample_a, int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, prcnn_stream_t stream) {
    const bool dual = nsample_b > 0;
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample_a > 0 && nsample_b >= 0, "prcnn_ball_query2_grid: bad shape");
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grid && new_xyz && idx_a && (!dual || idx_b), "prcnn_ball_query2_grid: null pointer");
    const size_t lds = (size_t)(nsample_a + nsample_b) * GBQ_THREADS * sizeof(int);
    PRCNN_REQUIRE(lds <= 64 * 1024, "prcnn_ball_query2_grid: nsample %d+%d too large for the LDS hit lists", nsample_a, nsample_b);
    dim3 g(prcnn_divup(M, GBQ_THREADS), B);
    const float r2a = radius_a * radius_a, r2b = radius_b * radius_b;      // fp32 products, as the oracle
    if (dual)
        hipLaunchKernelGGL(grid_ball_query_kernel<true>, g, dim3(GBQ_THREADS), lds, (hipStream_t)stream, grid, grid_frame_bytes(N),
                           new_xyz, M, radius_a, r2a, nsample_a, idx_a, radius_b, r2b, nsample_b, idx_b, N, xyz ? 1 : 0);
    else
        hipLaunchKernelGGL(grid_ball_query_kernel<false>, g, dim3(GBQ_THREADS), lds, (hipStream_t)stream, grid, grid_frame_bytes(N),
                           new_xyz, M, radius_a, r2a, nsample_a, idx_a, 0.f, 0.f, 0, (int32_t*)nullptr, N, xyz ? 1 : 0);
    PRCNN_LAUNCH_CHECK("prcnn_ball_query2_grid");
    // dense frames (GridHeader::cand): the index-order scan, restricted to exactly the frames the grid kernel skipped
    if (xyz) return prcnn_launch_ball_query_scan(xyz, new_xyz, B, N, M, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, grid,
                                                 (hipStream_t)stream);
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_nn_grid(const void* grid, const float* unknown, int B, int n, int m, float* dist2, int32_t* idx,
                                  float* weight, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && n >= 0 && m > 0, "prcnn_three_nn_grid: bad shape B=%d n=%d m=%d", B, n, m);
    if (B == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grid && unknown && dist2 && idx, "prcnn_three_nn_grid: null pointer");
    hipLaunchKernelGGL(grid_three_nn_kernel, dim3(prcnn_divup(n, GNN_THREADS), B), dim3(GNN_THREADS), 0, (hipStream_t)stream, grid,
                       grid_frame_bytes(m), unknown, n, dist2, idx, weight);
    PRCNN_LAUNCH_CHECK("prcnn_three_nn_grid");
    return PRCNN_OK;
}
