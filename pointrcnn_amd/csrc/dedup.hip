// dedup.hip -- skip the padding rows of ball-query groups.
//
// ball_query pads a group that has fewer than nsample neighbours by repeating its FIRST hit (the
// reference's CUDA op does the same), and the set-abstraction MLP then pushes every padded copy through all its layers
// before max-pooling them back into one value.  A max over copies of the same row is that row, so only the `cnt` real
// rows of a group matter -- and on sparse clouds cnt is tiny: on the benchmark's uniform clouds 94-97 % of all grouped
// rows are padding (mean cnt = 1.0-1.3 of 16 / 32), on a LiDAR-like density ~90 %.
//
// The split is done once per (level, radius), on the device, with no host round trip:
//   group_compact_kernel : cnt = number of leading distinct indices of the group (real hits are strictly ascending, the
//                          padding repeats idx[0]).  Groups with cnt <= T go to the SPARSE list: their cnt real rows are
//                          appended to one flat row list (global point index + the group's centroid per row), the group
//                          remembers (first row, cnt).  Groups with cnt > T go to the DENSE list with their full nsample
//                          rows, padding included -- a full tile pooled in the MLP epilogue is cheaper for them than a
//                          round trip of cnt result rows through HBM.  Lists are appended with ONE atomic per 1024-thread
//                          block and list; their order is arbitrary, the results are not (every group's output row is
//                          written exactly once).  With valid_n (roipool3d-padded clouds) a hit that is a wrap-copy of an
//                          earlier point ends the group's real rows exactly like padding does.
//   the MLP kernels take the list lengths as DEVICE-side row counts (MlpParams::rows_dev): launched for the worst case,
//   workgroups past the end exit at once;
//   segmax_scatter_kernel : max over each sparse group's result rows -> the group's row of the level's output;
//   scatter_rows_kernel   : pooled dense-list rows -> the groups' rows of the level's output.
// Bit-identical to the un-split path: same rows, same arithmetic, max over a multiset == max over its support.
#include "common.h"

struct CompactParams {
    const int32_t* idx;      // (B, M, ns) ball-query result
    const float* new_xyz;    // (B, M, 3)
    int32_t* ridx;           // (G*T)    sparse list: global index (b*N + p) of every real row, G = B*M
    float* rnx;              // (G*T, 3) the row's group centroid
    int32_t* slist;          // (G)      sparse groups: group id,
    int32_t* soff;           // (G)        first row in the flat list,
    int32_t* scnt;           // (G)        row count (1..T)
    int32_t* idxn;           // (G, ns)  dense list: global indices of the groups' nsample rows
    float* nxn;              // (G, 3)
    int32_t* listn;          // (G)
    int32_t* counts;         // [0] flat rows, [1] dense groups, [2] sparse groups (zeroed by the launcher)
    const int32_t* valid_n;  // (B) or NULL: points >= valid_n[b] of frame b are wrap-copies of earlier points (roipool3d)
    int G, N, M, ns, T;
};

constexpr int COMPACT_THREADS = 1024;
constexpr int DEDUP_GRID_CAP = 8192;           // workgroups of the grid-stride scatter kernels (32 per CU)

// NS > 0: nsample known at compile time (a multiple of 4) -- the group's index row is fetched with 16-byte loads, all in
// flight at once, instead of a dependent scalar walk
template <int NS>
__global__ __launch_bounds__(COMPACT_THREADS) void group_compact_kernel(CompactParams P) {
    const int g = blockIdx.x * COMPACT_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool ok = g < P.G;
    const int ns = NS > 0 ? NS : P.ns;
    int cnt = 0, first = 0;
    const int32_t* row = P.idx + (size_t)(ok ? g : 0) * ns;
    if (ok) {
        if constexpr (NS > 0) {
            int4 v[NS / 4];
#pragma unroll
            for (int q = 0; q < NS / 4; q++) v[q] = reinterpret_cast<const int4*>(row)[q];
            first = v[0].x;
            cnt = NS;
#pragma unroll
            for (int q = NS / 4 - 1; q >= 0; q--) {          // downwards: the smallest repeat position wins
                if (v[q].w == first) cnt = 4 * q + 3;
                if (v[q].z == first) cnt = 4 * q + 2;
                if (v[q].y == first) cnt = 4 * q + 1;
                if (q > 0 && v[q].x == first) cnt = 4 * q;
            }
        } else {
            first = row[0];
            cnt = ns;
            for (int s = 1; s < ns; s++)
                if (row[s] == first) { cnt = s; break; }      // real hits are strictly ascending: a repeat of idx[0] is padding
        }
    }
    const int b = (ok ? g : 0) / P.M;
    const int vn = P.valid_n ? P.valid_n[b] : P.N;
    if (ok && P.valid_n) {
        // ascending hits: every first copy (index < vn) precedes every wrap-copy, so the copies end the real rows
        int c2 = 0;
        while (c2 < cnt && row[c2] < vn) c2++;
        cnt = c2 > 0 ? c2 : 1;                       // (a group always holds its own centroid's first copy; 1 = defensive)
    }
    const bool sparse = ok && cnt <= P.T, dense = ok && cnt > P.T;
    // sparse: wave-wide exclusive scan of the row counts, one atomic per wave and list
    const int v = sparse ? cnt : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int total = __shfl(incl, 63);
    const unsigned long long bs = __ballot(sparse), bd = __ballot(dense);
    // one atomic per BLOCK and list (same-address atomics serialise in L2: per-wave appends made this kernel latency-bound)
    __shared__ int wsum[COMPACT_THREADS / 64][3];
    __shared__ int bbase[3];
    const int wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave][0] = total; wsum[wave][1] = (int)__popcll(bd); wsum[wave][2] = (int)__popcll(bs); }
    __syncthreads();
    if (threadIdx.x < 3) {
        int t = 0;
        for (int w = 0; w < COMPACT_THREADS / 64; w++) t += wsum[w][threadIdx.x];
        bbase[threadIdx.x] = t > 0 ? atomicAdd(P.counts + threadIdx.x, t) : 0;
    }
    __syncthreads();
    int base_rows = bbase[0], base_d = bbase[1], base_s = bbase[2];
    for (int w = 0; w < wave; w++) { base_rows += wsum[w][0]; base_d += wsum[w][1]; base_s += wsum[w][2]; }
    const unsigned long long below = (1ULL << lane) - 1ULL;
    float c3[3] = {0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int c = 0; c < 3; c++) c3[c] = P.new_xyz[(size_t)g * 3 + c];
    }
    if (sparse) {
        const int j = base_s + (int)__popcll(bs & below), r0 = base_rows + incl - v;
        P.slist[j] = g; P.soff[j] = r0; P.scnt[j] = cnt;
        for (int s = 0; s < cnt; s++) {
            P.ridx[r0 + s] = b * P.N + row[s];
#pragma unroll
            for (int c = 0; c < 3; c++) P.rnx[(size_t)(r0 + s) * 3 + c] = c3[c];
        }
    }
    if (dense) {
        const int pn = base_d + (int)__popcll(bd & below);
        P.listn[pn] = g;
        for (int s = 0; s < ns; s++) P.idxn[(size_t)pn * ns + s] = b * P.N + (row[s] < vn ? row[s] : first);
#pragma unroll
        for (int c = 0; c < 3; c++) P.nxn[(size_t)pn * 3 + c] = c3[c];
    }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int ld_src, const int32_t* __restrict__ list,
                                                           const int32_t* __restrict__ count, int C, float* __restrict__ dst, int ld_dst,
                                                           int col_off) {
    const long total = (long)(*count) * C;             // grid-stride: the grid is capped, not sized for the list's capacity
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        dst[(size_t)list[r] * ld_dst + col_off + c] = src[(size_t)r * ld_src + c];
    }
}

__global__ __launch_bounds__(256) void segmax_scatter_kernel(const float* __restrict__ src, int ld_src, const int32_t* __restrict__ list,
                                                             const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
                                                             const int32_t* __restrict__ count, int C, float* __restrict__ dst,
                                                             int ld_dst, int col_off) {
    const long total = (long)(*count) * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long j = e / C;
        const int c = (int)(e - j * C);
        const float* p = src + (size_t)off[j] * ld_src + c;
        const int rows = cnt[j];
        float m = p[0];
        for (int r = 1; r < rows; r++) m = fmaxf(m, p[(size_t)r * ld_src]);
        dst[(size_t)list[j] * ld_dst + col_off + c] = m;
    }
}

PRCNN_API int prcnn_group_compact(const int32_t* idx, const float* new_xyz, int B, int N, int M, int nsample, int sparse_max,
                                  const int32_t* valid_n, int32_t* ridx, float* rnx, int32_t* slist, int32_t* soff, int32_t* scnt,
                                  int32_t* idxn, float* nxn, int32_t* listn, int32_t* counts, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample > 0, "prcnn_group_compact: bad shape B=%d N=%d M=%d nsample=%d", B, N, M, nsample);
    PRCNN_REQUIRE(sparse_max >= 1 && sparse_max <= nsample, "prcnn_group_compact: sparse_max=%d (1..nsample)", sparse_max);
    PRCNN_REQUIRE(counts, "prcnn_group_compact: null counts");
    hipStream_t s = (hipStream_t)stream;
    if (prcnn_fill_words(counts, 0u, 3, s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_group_compact: clearing the counters failed");
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE((long)B * N < 2147483647L && (long)B * M * sparse_max < 2147483647L, "prcnn_group_compact: 32-bit row index overflow");
    PRCNN_REQUIRE(idx && new_xyz && ridx && rnx && slist && soff && scnt, "prcnn_group_compact: null pointer");
    PRCNN_REQUIRE(sparse_max == nsample || (idxn && nxn && listn), "prcnn_group_compact: null dense-list pointer");
    CompactParams P;
    P.idx = idx; P.new_xyz = new_xyz; P.ridx = ridx; P.rnx = rnx; P.slist = slist; P.soff = soff; P.scnt = scnt;
    P.idxn = idxn; P.nxn = nxn; P.listn = listn;
    P.counts = counts; P.valid_n = valid_n; P.G = B * M; P.N = N; P.M = M; P.ns = nsample; P.T = sparse_max;
    const dim3 grid(prcnn_divup(P.G, COMPACT_THREADS));
    const bool vec = ((uintptr_t)idx % 16) == 0;
    if (vec && nsample == 16) hipLaunchKernelGGL(group_compact_kernel<16>, grid, dim3(COMPACT_THREADS), 0, s, P);
    else if (vec && nsample == 32) hipLaunchKernelGGL(group_compact_kernel<32>, grid, dim3(COMPACT_THREADS), 0, s, P);
    else if (vec && nsample == 64) hipLaunchKernelGGL(group_compact_kernel<64>, grid, dim3(COMPACT_THREADS), 0, s, P);
    else hipLaunchKernelGGL(group_compact_kernel<0>, grid, dim3(COMPACT_THREADS), 0, s, P);
    PRCNN_LAUNCH_CHECK("prcnn_group_compact");
    return PRCNN_OK;
}

PRCNN_API int prcnn_segmax_scatter(const float* src, int ld_src, const int32_t* list, const int32_t* off, const int32_t* cnt,
                                   const int32_t* count, int max_groups, int C, float* dst, int ld_dst, int col_off,
                                   prcnn_stream_t stream) {
    PRCNN_REQUIRE(max_groups >= 0 && C > 0 && ld_src >= C && ld_dst >= col_off + C, "prcnn_segmax_scatter: bad shape");
    if (max_groups == 0) return PRCNN_OK;
    PRCNN_REQUIRE(src && list && off && cnt && count && dst, "prcnn_segmax_scatter: null pointer");
    // (capped grid + stride instead of one workgroup per 256 elements of CAPACITY; on the RCNN stage's lists the launch is its
    //  traffic, ~37 us, either way)
    hipLaunchKernelGGL(segmax_scatter_kernel, dim3(min(prcnn_divup((long)max_groups * C, 256), DEDUP_GRID_CAP)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       list, off, cnt, count, C, dst, ld_dst, col_off);
    PRCNN_LAUNCH_CHECK("prcnn_segmax_scatter");
    return PRCNN_OK;
}

PRCNN_API int prcnn_scatter_rows(const float* src, int ld_src, const int32_t* list, const int32_t* count, int max_rows, int C, float* dst,
                                 int ld_dst, int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(max_rows >= 0 && C > 0 && ld_src >= C && ld_dst >= col_off + C, "prcnn_scatter_rows: bad shape");
    if (max_rows == 0) return PRCNN_OK;
    PRCNN_REQUIRE(src && list && count && dst, "prcnn_scatter_rows: null pointer");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(min(prcnn_divup((long)max_rows * C, 256), DEDUP_GRID_CAP)), dim3(256), 0, (hipStream_t)stream, src, ld_src, list,
                       count, C, dst, ld_dst, col_off);
    PRCNN_LAUNCH_CHECK("prcnn_scatter_rows");
    return PRCNN_OK;
}
