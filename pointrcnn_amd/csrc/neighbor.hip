// neighbor.hip -- ball_query (one or two radii per scan) and three_nn for gfx950.
//
// Replaces pointnet2_cuda.ball_query_wrapper / three_nn_wrapper [UPSTREAM, not in tree]; semantics per
// SURVEY Appendix A.3 / A.5 and oracle prcnn_cpu_ball_query / prcnn_cpu_three_nn.
//
// Both are brute-force scans (compulsory HBM traffic is tiny: N*12 B per frame; the work is VALU):
// a workgroup owns 64 (ball query) or 256 (three_nn) query points (one per lane) of one frame and streams the frame's candidate
// points through LDS in coalesced chunks, expanded to float4 so that the inner loop is ONE
// ds_read_b128 broadcast (all lanes read the same address: conflict-free) per candidate.  Scanning
// candidates in ascending index order makes the "first nsample hits in index order" contract fall
// out by construction -- no sort, no compaction pass.  The MSG levels query two radii around the
// same centroids: ball_query2 evaluates both in the same pass (half the scans).
#include "grid_layout.h"

#define BQ_THREADS 64       // ball query: ONE wave per workgroup -> wave-local staging, 4x more workgroups than 256-thread
                            // blocks (the op only has B*M threads in total), no cross-wave barrier in the scan
#define NN_THREADS 256      // three_nn has n >= 4x more query points: 256-thread blocks share each staged chunk
#define NB_CHUNK 1024       // candidates staged per LDS chunk: 512 pairs * 32 B = 16 KB

// On CDNA4 only PACKED fp32 VALU ops (v_pk_add/mul_f32) run at the full 64 lanes x 2 rate; plain v_add/v_mul_f32
// measured ~3.6 cycles per wave64 instruction.  Candidates are therefore staged as PAIRS -- LDS record of pair p =
// {x0,x1,y0,y1 | z0,z1,-,-} (two ds_read_b128 broadcasts) -- and every distance evaluation below is a float2
// expression the compiler lowers to v_pk_*: same individually rounded operations per component, half the VALU issue.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS record of candidate pair p: floats [8p .. 8p+7] = {x0,x1,y0,y1,z0,z1,-,-}; point i of a chunk goes to pair i>>1, slot i&1.
// squared distances of the query to the two candidates of pair record `rec` (canonical arithmetic per component)
__device__ __forceinline__ f32x2 pair_sqdist(const float* rec, f32x2 qx, f32x2 qy, f32x2 qz) {
    float4 a = *reinterpret_cast<const float4*>(rec);          // x0 x1 y0 y1
    f32x2 z = *reinterpret_cast<const f32x2*>(rec + 4);        // z0 z1
    f32x2 dx = qx - (f32x2){a.x, a.y}, dy = qy - (f32x2){a.z, a.w}, dz = qz - z;
    return (dx * dx + dy * dy) + dz * dz;                       // -ffp-contract=off: no FMA, left to right
}

template <bool DUAL>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float* __restrict__ xyz,
                                                                const float* __restrict__ new_xyz, int N, int M,
                                                                float r2a, int nsa, int32_t* __restrict__ idxa,
                                                                float r2b, int nsb, int32_t* __restrict__ idxb,
                                                                const void* __restrict__ grid, size_t grid_fb) {
    __shared__ __attribute__((aligned(16))) float spts[NB_CHUNK * 4];
    const int b = blockIdx.y, tid = threadIdx.x;
    // launched behind the grid kernel (prcnn_ball_query2_grid): only the frames that kernel left to the scan
    if (grid && !grid_frame_dense(*grid_header(grid, grid_fb, b), N)) return;
    const int m = blockIdx.x * BQ_THREADS + tid;
    const bool valid = m < M;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        const float* q = new_xyz + ((size_t)b * M + m) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    int32_t* oa = idxa + ((size_t)b * M + (valid ? m : 0)) * nsa;
    int32_t* ob = DUAL ? idxb + ((size_t)b * M + (valid ? m : 0)) * nsb : nullptr;
    int cnta = 0, cntb = 0, firsta = 0, firstb = 0;
    const float r2max = DUAL ? fmaxf(r2a, r2b) : r2a;
    bool done = !valid;

    // The chunk for iteration c+1 is fetched into registers while chunk c is scanned: with one wave per workgroup
    // nothing else would hide the global-load latency of the staging.
    constexpr int PER_LANE = NB_CHUNK * 3 / BQ_THREADS;       // 48 floats of the flat xyz stream per lane per chunk
    float pre[PER_LANE];
    auto fetch_chunk = [&](int c0) {
        const int cnt3 = min(NB_CHUNK, N - c0) * 3;
        const float* src = p + (size_t)c0 * 3;
#pragma unroll
        for (int u = 0; u < PER_LANE; u++) {
            int i = tid + u * BQ_THREADS;
            pre[u] = i < cnt3 ? src[i] : INFINITY;          // slots past the frame end: +inf, never a hit
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < PER_LANE; u++) {
            int i = tid + u * BQ_THREADS;
            int pt = i / 3, c = i - pt * 3;
            spts[(pt >> 1) * 8 + c * 2 + (pt & 1)] = pre[u];
        }
    };
    fetch_chunk(0);
    for (int c0 = 0; c0 < N; c0 += NB_CHUNK) {
        if (__all(done)) break;                    // single-wave workgroup: the whole block is finished
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c0 + NB_CHUNK < N) fetch_chunk(c0 + NB_CHUNK);
        if (!done) {
            for (int i0 = 0; i0 < NB_CHUNK; i0 += 8) {
                // 8 candidates = 4 packed pairs, ONE test for "any of them inside the larger ball"
                f32x2 d2[4];
                bool any = false;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    d2[u] = pair_sqdist(spts + (i0 / 2 + u) * 8, qx2, qy2, qz2);     // padded slots give +inf
                    any |= (d2[u].x < r2max) | (d2[u].y < r2max);
                }
                if (any) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        float d = (u & 1) ? d2[u >> 1].y : d2[u >> 1].x;
                        if (d < r2max) {
                            int k = c0 + i0 + u;
                            if (d < r2a && cnta < nsa) {
                                if (cnta == 0) firsta = k;
                                oa[cnta++] = k;
                            }
                            if (DUAL && d < r2b && cntb < nsb) {
                                if (cntb == 0) firstb = k;
                                ob[cntb++] = k;
                            }
                        }
                    }
                    if (cnta >= nsa && (!DUAL || cntb >= nsb)) { done = true; break; }
                }
            }
        }
    }
    if (valid) {
        for (int s = cnta; s < nsa; s++) oa[s] = firsta;      // pad with the first hit (0 if none)
        if (DUAL)
            for (int s = cntb; s < nsb; s++) ob[s] = firstb;
    }
}

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(const float* __restrict__ unknown,
                                                              const float* __restrict__ known, int n, int m,
                                                              float* __restrict__ dist2, int32_t* __restrict__ idx,
                                                              float* __restrict__ weight) {
    __shared__ __attribute__((aligned(16))) float spts[NB_CHUNK * 4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * NN_THREADS + tid;
    const bool valid = i < n;
    const float* __restrict__ p = known + (size_t)b * m * 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (valid) {
        const float* u = unknown + ((size_t)b * n + i) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const f32x2 ux2 = {ux, ux}, uy2 = {uy, uy}, uz2 = {uz, uz};
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    // next chunk prefetched into registers while the current one is scanned (hides the staging latency)
    constexpr int PER_LANE = NB_CHUNK * 3 / NN_THREADS;       // 12 floats of the flat xyz stream per lane per chunk
    float pre[PER_LANE];
    auto fetch_chunk = [&](int c0) {
        const int cnt3 = min(NB_CHUNK, m - c0) * 3;
        const float* src = p + (size_t)c0 * 3;
#pragma unroll
        for (int u = 0; u < PER_LANE; u++) {
            int e = tid + u * NN_THREADS;
            pre[u] = e < cnt3 ? src[e] : INFINITY;          // slots past the end: +inf, never < b3
        }
    };
    fetch_chunk(0);
    for (int c0 = 0; c0 < m; c0 += NB_CHUNK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER_LANE; u++) {
            int e = tid + u * NN_THREADS;
            int pt = e / 3, c = e - pt * 3;
            spts[(pt >> 1) * 8 + c * 2 + (pt & 1)] = pre[u];
        }
        __syncthreads();
        if (c0 + NB_CHUNK < m) fetch_chunk(c0 + NB_CHUNK);
        for (int j0 = 0; j0 < NB_CHUNK; j0 += 8) {
            f32x2 d[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                d[u] = pair_sqdist(spts + (j0 / 2 + u) * 8, ux2, uy2, uz2);          // padded slots: +inf, never < b3
                any |= (d[u].x < b3) | (d[u].y < b3);
            }
            if (any) {                              // rare after warm-up
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float dd = (u & 1) ? d[u >> 1].y : d[u >> 1].x;
                    if (dd < b3) {
                        int k = c0 + j0 + u;
                        if (dd < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = dd; i1 = k; }
                        else if (dd < b2) { b3 = b2; i3 = i2; b2 = dd; i2 = k; }
                        else { b3 = dd; i3 = k; }
                    }
                }
            }
        }
    }
    if (valid) {
        size_t o = ((size_t)b * n + i) * 3;
        dist2[o] = b1; dist2[o + 1] = b2; dist2[o + 2] = b3;
        idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
        if (weight) {
            // PointnetFPModule: w = 1/(dist+1e-8), normalised
            // sqrtf and '/' are IEEE correctly rounded here (-fhip-fp32-correctly-rounded-divide-sqrt, set
            // explicitly in build.py); the __fsqrt_rn/__fdiv_rn intrinsics are NOT (they map to native ops).
            float r0 = 1.0f / (sqrtf(b1) + 1e-8f);
            float r1 = 1.0f / (sqrtf(b2) + 1e-8f);
            float r2 = 1.0f / (sqrtf(b3) + 1e-8f);
            float s = (r0 + r1) + r2;
            weight[o] = r0 / s; weight[o + 1] = r1 / s; weight[o + 2] = r2 / s;
        }
    }
}

static int ball_query_impl(const float* xyz, const float* new_xyz, int B, int N, int M, float ra, int nsa, int32_t* ia,
                           float rb, int nsb, int32_t* ib, bool dual, hipStream_t s, const void* dense_grid = nullptr) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsa > 0 && (!dual || nsb > 0),
                  "prcnn_ball_query: bad shape B=%d N=%d M=%d nsample=%d/%d", B, N, M, nsa, nsb);
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && new_xyz && ia && (!dual || ib), "prcnn_ball_query: null pointer");
    dim3 grid(prcnn_divup(M, BQ_THREADS), B);
    float r2a = ra * ra, r2b = rb * rb;        // fp32 product, as the oracle
    const size_t gfb = grid_frame_bytes(N);
    if (dual)
        hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(BQ_THREADS), 0, s, xyz, new_xyz, N, M, r2a, nsa, ia, r2b,
                           nsb, ib, dense_grid, gfb);
    else
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(BQ_THREADS), 0, s, xyz, new_xyz, N, M, r2a, nsa, ia,
                           0.f, 0, (int32_t*)nullptr, dense_grid, gfb);
    PRCNN_LAUNCH_CHECK("prcnn_ball_query");
    return PRCNN_OK;
}

int prcnn_launch_ball_query_scan(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a,
                                 int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, const void* grid, hipStream_t s) {
    return ball_query_impl(xyz, new_xyz, B, N, M, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, nsample_b > 0, s, grid);
}

PRCNN_API int prcnn_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample,
                               int32_t* idx, prcnn_stream_t stream) {
    return ball_query_impl(xyz, new_xyz, B, N, M, radius, nsample, idx, 0.f, 0, nullptr, false, (hipStream_t)stream);
}

PRCNN_API int prcnn_ball_query2(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a,
                                int nsample_a, int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b,
                                prcnn_stream_t stream) {
    return ball_query_impl(xyz, new_xyz, B, N, M, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, true,
                           (hipStream_t)stream);
}

PRCNN_API int prcnn_three_nn(const float* unknown, const float* known, int B, int n, int m, float* dist2, int32_t* idx,
                             float* weight, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && n >= 0 && m > 0, "prcnn_three_nn: bad shape B=%d n=%d m=%d", B, n, m);
    if (B == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(unknown && known && dist2 && idx, "prcnn_three_nn: null pointer");
    hipLaunchKernelGGL(three_nn_kernel, dim3(prcnn_divup(n, NN_THREADS), B), dim3(NN_THREADS), 0, (hipStream_t)stream,
                       unknown, known, n, m, dist2, idx, weight);
    PRCNN_LAUNCH_CHECK("prcnn_three_nn");
    return PRCNN_OK;
}
