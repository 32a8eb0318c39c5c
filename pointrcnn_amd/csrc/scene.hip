// scene.hip -- the RPN input builder on the device: raw velodyne scans -> the (B, npoints, 3)
// rect-camera clouds the backbone consumes, for a whole batch in two launches and with no host round trip.
//
// Replaces lib/datasets/kitti_rcnn_dataset.py:246-310 (get_rpn_sample, inference branch), which does per frame, in numpy on
// one dataloader core: lidar_to_rect (calibration.py:51-59), rect_to_img (:61-70), get_valid_flag (image bounds, depth >= 0,
// PC_AREA_SCOPE; kitti_rcnn_dataset.py:198-219), boolean-mask compaction, then the npoints sampling of :285-306 -- every
// point at depth >= 40 m kept, the rest drawn without replacement, the result shuffled; scans with fewer valid points
// than npoints are topped up with a draw without replacement from themselves.
//
// Arithmetic contract (shared with oracle/prcnn_oracle.c, canonical like the other ops): fp32, individually rounded,
// left to right, no FMA:  rect_c = ((x*M[0][c] + y*M[1][c]) + z*M[2][c]) + M[3][c]   with M = V2C^T . R0^T formed by the
// host in fp32 exactly as calibration.py:57 does;  hom_r = ((rx*P[r][0] + ry*P[r][1]) + rz*P[r][2]) + P[r][3];
// u = hom_0 / rz, v = hom_1 / rz (IEEE division; the reference divides by the RECT z, calibration.py:68);
// depth = hom_2 - P[2][3];  the PC_AREA_SCOPE comparisons are done in double against the double bounds, as numpy does.
// (The reference's np.dot goes through a BLAS sgemm whose summation order is unspecified: its values can differ from the
// canonical ones in the last ulp -- tests/test_oracle_scene.py pins the oracle against the reference's own output.)
//
// Randomness: numpy's global Mersenne-Twister stream cannot be reproduced by a parallel kernel (nor does the reference seed
// it); the draw is re-specified with a counter-based generator so that GPU and oracle agree bit for bit:
//   r(stream, frame, i) = mix(i ^ mix(frame * 0x9E3779B9 + mix(seed + stream * 0x85EBCA6B))),  mix = the 32-bit finaliser below;
//   "draw k of a candidate set without replacement" = the k candidates with the smallest (r(0,.) >> 2, raw index);
//   "shuffle" = ascending order of (r(1,.), raw index) (top-up copies use r(2,.)).
// Both are exact uniform draws / permutations when r is uniform; the selected SET and the output ORDER depend only on
// (seed, frame, raw index), never on the order in which the kernels' atomics append.
//
//   scene_flag_kernel   : one thread per raw point: transform, project, flags; valid points are appended (unordered, one
//                         atomic per 1024-thread block) to the frame's candidate list as (class | 30-bit key, raw index).
//   scene_sample_kernel : one workgroup per frame: 3-pass radix select of the k-th smallest key among the candidates
//                         (LDS histograms), ties by raw index, selected entries appended to LDS, sorted by their shuffle
//                         key with the shared bitonic sort (lds_sort.h), rows recomputed and written in that order.
#include "lds_sort.h"

constexpr int SCENE_THREADS = 1024;
constexpr int SCENE_MAX_TIES = 1024;
constexpr unsigned SCENE_FAR = 1u << 30;

struct SceneParams {
    const float4* raw;          // (total, 4) x y z intensity, lidar frame
    const int64_t* off;         // (B+1) first raw point of every frame
    const float* calib;         // (B, 24): M (4x3 row-major), P2 (3x4 row-major)
    const int32_t* img_hw;      // (B, 2) image height, width
    double scope[6];            // x0 x1 y0 y1 z0 z1 (PC_AREA_SCOPE)
    int use_scope;
    int B, npoints, NP;
    unsigned seed;
    uint2* list;                // (total) candidate entries, frame b at off[b]
    int32_t* counters;          // (B, 2) valid, far -- zeroed by the launcher
    float* out_xyz;             // (B, npoints, 3)
    float* out_int;             // (B, npoints) intensity - 0.5
    int32_t* out_src;           // (B, npoints) raw index of every output point
    int32_t* nvalid;            // (B)
    int32_t* status;            // (B) 0 ok, 1 outside the reference's domain (it raises), 2 no valid point
};

__host__ __device__ __forceinline__ unsigned scene_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned scene_rand(unsigned seed, unsigned stream, unsigned frame, unsigned i) {
    return scene_mix(i ^ scene_mix(frame * 0x9E3779B9U + scene_mix(seed + stream * 0x85EBCA6BU)));
}

struct RectPoint { float x, y, z; bool valid; };

__device__ __forceinline__ RectPoint scene_project(const float4 p, const float* __restrict__ c, int H, int W, const double* scope,
                                                   int use_scope) {
    RectPoint r;
    r.x = ((p.x * c[0] + p.y * c[3]) + p.z * c[6]) + c[9];
    r.y = ((p.x * c[1] + p.y * c[4]) + p.z * c[7]) + c[10];
    r.z = ((p.x * c[2] + p.y * c[5]) + p.z * c[8]) + c[11];
    const float* P = c + 12;
    const float h0 = ((r.x * P[0] + r.y * P[1]) + r.z * P[2]) + P[3];
    const float h1 = ((r.x * P[4] + r.y * P[5]) + r.z * P[6]) + P[7];
    const float h2 = ((r.x * P[8] + r.y * P[9]) + r.z * P[10]) + P[11];
    const float u = h0 / r.z, v = h1 / r.z, depth = h2 - P[11];
    bool ok = (u >= 0.f) && (u < (float)W) && (v >= 0.f) && (v < (float)H) && (depth >= 0.f);
    if (use_scope)
        ok = ok && ((double)r.x >= scope[0]) && ((double)r.x <= scope[1]) && ((double)r.y >= scope[2]) && ((double)r.y <= scope[3]) &&
             ((double)r.z >= scope[4]) && ((double)r.z <= scope[5]);
    r.valid = ok;
    return r;
}

__global__ __launch_bounds__(SCENE_THREADS) void scene_flag_kernel(SceneParams P) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t o = P.off[b];
    const int n = (int)(P.off[b + 1] - o);
    if ((int64_t)blockIdx.x * SCENE_THREADS >= n) return;
    const int i = blockIdx.x * SCENE_THREADS + tid;
    bool valid = false, far = false;
    if (i < n) {
        const RectPoint r = scene_project(P.raw[o + i], P.calib + b * 24, P.img_hw[b * 2], P.img_hw[b * 2 + 1], P.scope, P.use_scope);
        valid = r.valid;
        far = valid && !(r.z < 40.0f);                 // kitti_rcnn_dataset.py:288: near = depth < 40.0
    }
    const unsigned long long bv = __ballot(valid), bf = __ballot(far);
    __shared__ int wv[SCENE_THREADS / 64], wf[SCENE_THREADS / 64];
    __shared__ int base;
    if (lane == 0) { wv[wave] = (int)__popcll(bv); wf[wave] = (int)__popcll(bf); }
    __syncthreads();
    if (tid == 0) {
        int tv = 0, tf = 0;
        for (int w = 0; w < SCENE_THREADS / 64; w++) { tv += wv[w]; tf += wf[w]; }
        base = tv > 0 ? atomicAdd(P.counters + b * 2, tv) : 0;
        if (tf > 0) atomicAdd(P.counters + b * 2 + 1, tf);
    }
    __syncthreads();
    if (valid) {
        int pos = base + (int)__popcll(bv & ((1ULL << lane) - 1ULL));
        for (int w = 0; w < wave; w++) pos += wv[w];
        const unsigned key = scene_rand(P.seed, 0u, (unsigned)b, (unsigned)i) >> 2;
        P.list[o + pos] = make_uint2(key | (far ? SCENE_FAR : 0u), (unsigned)i);
    }
}

// block-wide: given this thread's histogram bin count c (1024 bins = 1024 threads), find the bin where the running count
// crosses `want` (0-based rank): returns the bin through sel[0] and the rank inside that bin through sel[1]
__device__ __forceinline__ void scene_pick_bin(int c, int want, int* wsum, int* sel) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += wsum[w];
    const int hi = before + incl, lo = hi - c;
    if (c > 0 && lo <= want && want < hi) { sel[0] = tid; sel[1] = want - lo; }
    __syncthreads();
}

__global__ __launch_bounds__(SCENE_THREADS) void scene_sample_kernel(SceneParams P) {
    extern __shared__ u64 keys[];
    __shared__ int hist[1024];
    __shared__ int wsum[SCENE_THREADS / 64];
    __shared__ int sel[2];
    __shared__ unsigned ties[SCENE_MAX_TIES];
    __shared__ int nties, nsel;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t o = P.off[b];
    const uint2* __restrict__ L = P.list + o;
    const int n = P.counters[b * 2], f = P.counters[b * 2 + 1];
    const int np = P.npoints;
    float* oxyz = P.out_xyz + (size_t)b * np * 3;
    float* oint = P.out_int + (size_t)b * np;
    int32_t* osrc = P.out_src + (size_t)b * np;
    if (tid == 0) { P.nvalid[b] = n; nties = 0; nsel = 0; }
    if (n == 0) {
        for (int j = tid; j < np; j += SCENE_THREADS) { oxyz[j * 3] = 0.f; oxyz[j * 3 + 1] = 0.f; oxyz[j * 3 + 2] = 0.f; oint[j] = 0.f; osrc[j] = -1; }
        if (tid == 0) P.status[b] = 2;
        return;
    }
    // what to draw: k candidates with the smallest keys; far points are outside the draw (always kept) when more valid
    // points than npoints exist, inside it otherwise (top-up from ALL valid points, kitti_rcnn_dataset.py:299-303)
    int st = 0, k;
    bool keep_far, keep_all;
    if (n > np) {
        keep_all = false;
        if (f > np) { st = 1; keep_far = false; k = np; }            // the reference's np.random.choice raises (negative size)
        else { keep_far = true; k = np - f; }
    } else {
        keep_all = true; keep_far = false;
        k = np - n;
        if (k > n) { st = 1; k = n; }                                // the reference raises (cannot draw k > n without replacement)
    }
    const int ncand = keep_far ? n - f : n;
    // ---- radix select: T = the k-th smallest 30-bit key among the candidates (rank k-1), need = how many of key == T to take
    unsigned T = 0;
    int need = 0;
    if (k > 0 && k < ncand) {
        unsigned prefix = 0;
        int want = k - 1;
        for (int pass = 0; pass < 3; pass++) {
            const int shift = 20 - 10 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int e = tid; e < n; e += SCENE_THREADS) {
                const unsigned code = L[e].x;
                if (keep_far && (code & SCENE_FAR)) continue;
                const unsigned key = code & (SCENE_FAR - 1u);
                if (pass == 0 || (key >> (shift + 10)) == prefix) atomicAdd(&hist[(key >> shift) & 1023u], 1);
            }
            __syncthreads();
            scene_pick_bin(hist[tid], want, wsum, sel);
            prefix = (prefix << 10) | (unsigned)sel[0];
            want = sel[1];
            __syncthreads();
        }
        T = prefix;
        need = want + 1;
    } else if (k >= ncand) {
        T = SCENE_FAR;                                   // every candidate key is < 2^30: take them all
    }                                                    // k == 0: T = 0, need = 0 -> none
    // ---- ties on key == T: the `need` smallest raw indices
    if (need > 0) {
        for (int e = tid; e < n; e += SCENE_THREADS) {
            const uint2 it = L[e];
            if (keep_far && (it.x & SCENE_FAR)) continue;
            if ((it.x & (SCENE_FAR - 1u)) == T) {
                const int p = atomicAdd(&nties, 1);
                if (p < SCENE_MAX_TIES) ties[p] = it.y;
            }
        }
    }
    __syncthreads();
    const int m = min(nties, SCENE_MAX_TIES);
    // ---- gather the selection into LDS as (shuffle key << 32 | raw index)
    for (int e0 = 0; e0 < n; e0 += SCENE_THREADS) {
        const int e = e0 + tid;
        if (e < n) {
            const uint2 it = L[e];
            const bool isfar = (it.x & SCENE_FAR) != 0u;
            const unsigned key = it.x & (SCENE_FAR - 1u);
            const bool cand = !(keep_far && isfar);
            bool drawn = cand && key < T;
            if (cand && need > 0 && key == T) {
                int rank = 0;
                for (int q = 0; q < m; q++) rank += ties[q] < it.y ? 1 : 0;
                drawn = rank < need;
            }
            if (keep_all || (keep_far && isfar)) {
                const int p = atomicAdd(&nsel, 1);
                keys[lds_phys(p)] = ((u64)scene_rand(P.seed, 1u, (unsigned)b, it.y) << 32) | it.y;
            }
            if (drawn) {
                const int p = atomicAdd(&nsel, 1);
                keys[lds_phys(p)] = ((u64)scene_rand(P.seed, keep_all ? 2u : 1u, (unsigned)b, it.y) << 32) | it.y;
            }
        }
    }
    __syncthreads();
    const int total = nsel;                               // == npoints unless st == 1
    u64 v[16];
    if (tid * 16 < P.NP) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int j = tid * 16 + e;
            v[e] = j < total ? keys[lds_phys(j)] : ~0ULL;
        }
    }
    __syncthreads();
    block_sort16(v, keys, P.NP, tid);
    // ---- rows in shuffled order (a short selection -- status 1 -- repeats cyclically)
    const float* c = P.calib + b * 24;
    for (int j = tid; j < np; j += SCENE_THREADS) {
        const unsigned i = (unsigned)keys[lds_phys(j < total ? j : j % total)];
        const float4 p = P.raw[o + i];
        oxyz[j * 3 + 0] = ((p.x * c[0] + p.y * c[3]) + p.z * c[6]) + c[9];
        oxyz[j * 3 + 1] = ((p.x * c[1] + p.y * c[4]) + p.z * c[7]) + c[10];
        oxyz[j * 3 + 2] = ((p.x * c[2] + p.y * c[5]) + p.z * c[8]) + c[11];
        oint[j] = p.w - 0.5f;
        osrc[j] = (int32_t)i;
    }
    if (tid == 0) P.status[b] = st;
}

PRCNN_API size_t prcnn_scene_workspace_bytes(int64_t total_points, int B) {
    if (total_points < 0 || B < 0) return 0;
    return (size_t)total_points * sizeof(uint2) + (size_t)B * 2 * sizeof(int32_t) + 64;
}

PRCNN_API int prcnn_scene_prepare(const float* raw, const int64_t* offsets, int B, int64_t total_points, int max_points_per_frame,
                                  const float* calib, const int32_t* img_hw, const double* scope, int npoints, uint32_t seed,
                                  float* out_xyz, float* out_intensity, int32_t* out_src, int32_t* nvalid, int32_t* status,
                                  void* workspace, size_t workspace_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && total_points >= 0 && max_points_per_frame >= 0, "prcnn_scene_prepare: bad shape B=%d total=%ld", B, (long)total_points);
    PRCNN_REQUIRE(npoints > 0 && npoints <= 16384, "prcnn_scene_prepare: npoints=%d (1..16384: the shuffle is one LDS-resident sort per frame)", npoints);
    if (B == 0) return PRCNN_OK;
    PRCNN_REQUIRE(offsets && calib && img_hw && out_xyz && out_intensity && out_src && nvalid && status, "prcnn_scene_prepare: null pointer");
    PRCNN_REQUIRE(total_points == 0 || raw, "prcnn_scene_prepare: null raw points");
    PRCNN_REQUIRE(((uintptr_t)raw % 16) == 0, "prcnn_scene_prepare: raw points must be 16-byte aligned");
    PRCNN_REQUIRE(workspace && workspace_bytes >= prcnn_scene_workspace_bytes(total_points, B), "prcnn_scene_prepare: workspace too small");
    PRCNN_REQUIRE((long)max_points_per_frame < (1L << 31) - SCENE_THREADS, "prcnn_scene_prepare: frame too large");
    hipStream_t s = (hipStream_t)stream;
    SceneParams P = {};
    P.raw = reinterpret_cast<const float4*>(raw); P.off = offsets; P.calib = calib; P.img_hw = img_hw;
    P.use_scope = scope != nullptr;
    for (int q = 0; q < 6; q++) P.scope[q] = scope ? scope[q] : 0.0;
    P.B = B; P.npoints = npoints; P.seed = seed;
    int NP = 16;
    while (NP < npoints) NP <<= 1;
    P.NP = NP;
    char* w = static_cast<char*>(workspace);
    P.counters = reinterpret_cast<int32_t*>(w);
    P.list = reinterpret_cast<uint2*>(w + (((size_t)B * 2 * sizeof(int32_t) + 63) / 64) * 64);
    P.out_xyz = out_xyz; P.out_int = out_intensity; P.out_src = out_src; P.nvalid = nvalid; P.status = status;
    if (prcnn_fill_words(P.counters, 0u, (size_t)B * 2, s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_scene_prepare: memset failed");
    if (max_points_per_frame > 0) {
        hipLaunchKernelGGL(scene_flag_kernel, dim3(prcnn_divup(max_points_per_frame, SCENE_THREADS), B), dim3(SCENE_THREADS), 0, s, P);
        PRCNN_LAUNCH_CHECK("prcnn_scene_prepare(flags)");
    }
    static PrcnnLdsLimit attr;
    if (!attr.raise((const void*)scene_sample_kernel, (int)lds_sort_bytes(16384)))
        return prcnn_fail(PRCNN_EHIP, "prcnn_scene_prepare: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(scene_sample_kernel, dim3(B), dim3(SCENE_THREADS), lds_sort_bytes(NP), s, P);
    PRCNN_LAUNCH_CHECK("prcnn_scene_prepare(sample)");
    return PRCNN_OK;
}
