// iou3d.hip -- rotated BEV overlap / IoU matrices and greedy NMS (rotated + axis-aligned) for gfx950.
//
// Replaces iou3d_cuda.{boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu}
// (lib/utils/iou3d/src/iou3d.cpp:31,52,73,123 -> iou3d_kernel.cu:223-387).  The geometry follows the
// reference's clipping algorithm step for step (iou3d_kernel.cu:34-221: edge-edge intersections, contained
// corners, centroid, angular sort, shoelace) under the canonical arithmetic contract shared with
// oracle/prcnn_oracle.c (trig_mode 1), so results are bit-identical to the oracle.
//
// What is different from the reference's execution plan:
//   * per-box work (cos/sin, centre, rotated corners) is done ONCE per box into LDS, not once per pair;
//     the angular sort key is computed once per vertex, not 2x per bubble-sort comparison
//     (the reference evaluates atan2 cnt*(cnt-1) times per pair: iou3d_kernel.cu:104-106,188-196);
//   * the NMS mask kernel only evaluates upper-triangle 64x64 tiles (the reference computes all:
//     iou3d_kernel.cu:258), and the greedy sweep runs ON THE DEVICE (one wave per problem, suppression
//     bitmap in LDS) instead of a synchronous D2H copy of the N x N/64 mask plus a host loop
//     (iou3d.cpp:86-116): no host synchronisation anywhere.
#include "common.h"

struct Pt { float x, y; };
struct RBox {
    float x1, y1, x2, y2;   // raw extents
    float cx, cy;           // centre
    float c, s;             // cos(angle), sin(angle)
    Pt p[5];                // rotated corners, p[4] == p[0]
};

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

__device__ __forceinline__ Pt rotate_around_center(float cx, float cy, float c, float s, float px, float py) {
    float dx = sub(px, cx), dy = sub(py, cy);                                   // iou3d_kernel.cu:98-102
    Pt r;
    r.x = add(add(mul(dx, c), mul(dy, s)), cx);
    r.y = add(add(mul(-dx, s), mul(dy, c)), cy);
    return r;
}

__device__ void make_rbox(const float* __restrict__ b, RBox& r) {
    r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
    r.cx = add(r.x1, r.x2) / 2; r.cy = add(r.y1, r.y2) / 2;
    r.c = (float)cos((double)b[4]);
    r.s = (float)sin((double)b[4]);
    r.p[0] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x1, r.y1);
    r.p[1] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x2, r.y1);
    r.p[2] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x2, r.y2);
    r.p[3] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x1, r.y2);
    r.p[4] = r.p[0];
}

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {                  // iou3d_kernel.cu:38-40
    return sub(mul(sub(p1.x, p0.x), sub(p2.y, p0.y)), mul(sub(p2.x, p0.x), sub(p1.y, p0.y)));
}

__device__ __forceinline__ bool check_rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {  // iou3d_kernel.cu:42-48
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// iou3d_kernel.cu:50-65 with (cos(-a), sin(-a)) = (c, -s)
__device__ __forceinline__ bool check_in_box2d(const RBox& box, Pt p) {
    const float MARGIN = 1e-5f;
    float dx = sub(p.x, box.cx), dy = sub(p.y, box.cy);
    float sn = -box.s;
    float rot_x = add(add(mul(dx, box.c), mul(dy, sn)), box.cx);
    float rot_y = add(add(mul(-dx, sn), mul(dy, box.c)), box.cy);
    return rot_x > sub(box.x1, MARGIN) && rot_x < add(box.x2, MARGIN) && rot_y > sub(box.y1, MARGIN) &&
           rot_y < add(box.y2, MARGIN);
}

// iou3d_kernel.cu:67-96
__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
    const float EPS = 1e-8f;
    if (!check_rect_cross(p0, p1, q0, q1)) return false;
    float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(mul(s1, s2) > 0 && mul(s3, s4) > 0)) return false;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(sub(s5, s1)) > EPS) {
        float den = sub(s5, s1);
        ans.x = sub(mul(s5, q0.x), mul(s1, q1.x)) / den;
        ans.y = sub(mul(s5, q0.y), mul(s1, q1.y)) / den;
    } else {
        float a0 = sub(p0.y, p1.y), b0 = sub(p1.x, p0.x), c0 = sub(mul(p0.x, p1.y), mul(p1.x, p0.y));
        float a1 = sub(q0.y, q1.y), b1 = sub(q1.x, q0.x), c1 = sub(mul(q0.x, q1.y), mul(q1.x, q0.y));
        float D = sub(mul(a0, b1), mul(a1, b0));
        ans.x = sub(mul(b0, c1), mul(b1, c0)) / D;
        ans.y = sub(mul(a1, c0), mul(a0, c1)) / D;
    }
    return true;
}

// canonical ordering key: strictly increasing in atan2(dy,dx) over (-pi, pi]; IEEE +,-,/ only
__device__ __forceinline__ float angle_key(float dx, float dy) {
    float s = add(fabsf(dx), fabsf(dy));
    if (!(s > 0.0f)) return 0.0f;
    float t = dy / s;
    if (dx >= 0.0f) return t;
    return dy >= 0.0f ? sub(2.0f, t) : sub(-2.0f, t);
}

// iou3d_kernel.cu:108-212
__device__ float box_overlap(const RBox& A, const RBox& B) {
    Pt cp[24];
    float key[24];
    float pcx = 0.f, pcy = 0.f;
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            Pt ans;
            if (seg_intersection(A.p[i + 1], A.p[i], B.p[j + 1], B.p[j], ans)) {
                pcx = add(pcx, ans.x); pcy = add(pcy, ans.y);
                cp[cnt++] = ans;
            }
        }
    for (int k = 0; k < 4; k++) {
        if (check_in_box2d(A, B.p[k])) { pcx = add(pcx, B.p[k].x); pcy = add(pcy, B.p[k].y); cp[cnt++] = B.p[k]; }
        if (check_in_box2d(B, A.p[k])) { pcx = add(pcx, A.p[k].x); pcy = add(pcy, A.p[k].y); cp[cnt++] = A.p[k]; }
    }
    if (cnt == 0) return 0.0f;
    pcx = pcx / (float)cnt; pcy = pcy / (float)cnt;
    for (int i = 0; i < cnt; i++) key[i] = angle_key(sub(cp[i].x, pcx), sub(cp[i].y, pcy));
    for (int j = 0; j < cnt - 1; j++)                                            // iou3d_kernel.cu:188-196
        for (int i = 0; i < cnt - j - 1; i++)
            if (key[i] > key[i + 1]) {
                Pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                float tk = key[i]; key[i] = key[i + 1]; key[i + 1] = tk;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; k++) {                                          // iou3d_kernel.cu:206-211
        float ux = sub(cp[k].x, cp[0].x), uy = sub(cp[k].y, cp[0].y);
        float vx = sub(cp[k + 1].x, cp[0].x), vy = sub(cp[k + 1].y, cp[0].y);
        area = add(area, sub(mul(ux, vy), mul(uy, vx)));
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const RBox& a, const RBox& b) {         // iou3d_kernel.cu:214-221
    float sa = mul(sub(a.x2, a.x1), sub(a.y2, a.y1));
    float sb = mul(sub(b.x2, b.x1), sub(b.y2, b.y1));
    float ov = box_overlap(a, b);
    return ov / fmaxf(sub(add(sa, sb), ov), 1e-8f);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {    // iou3d_kernel.cu:295-303
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(sub(right, left), 0.f), height = fmaxf(sub(bottom, top), 0.f);
    float interS = mul(width, height);
    float Sa = mul(sub(a[2], a[0]), sub(a[3], a[1]));
    float Sb = mul(sub(b[2], b[0]), sub(b[3], b[1]));
    return interS / fmaxf(sub(add(Sa, Sb), interS), 1e-8f);
}

// ---- pairwise matrices: 16x16 pair tile per workgroup, boxes prepared once in LDS ---------------
template <bool IOU>
__global__ __launch_bounds__(256) void pair_matrix_kernel(const float* __restrict__ boxes_a, int na,
                                                          const float* __restrict__ boxes_b, int nb,
                                                          float* __restrict__ out) {
    __shared__ RBox sa[16], sb[16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    if (threadIdx.x < 16) { if (a0 + threadIdx.x < na) make_rbox(boxes_a + (size_t)(a0 + threadIdx.x) * 5, sa[threadIdx.x]); }
    else if (threadIdx.x < 32) { int t = threadIdx.x - 16; if (b0 + t < nb) make_rbox(boxes_b + (size_t)(b0 + t) * 5, sb[t]); }
    __syncthreads();
    const int ai = a0 + ty, bi = b0 + tx;
    if (ai >= na || bi >= nb) return;
    out[(size_t)ai * nb + bi] = IOU ? iou_bev(sa[ty], sb[tx]) : box_overlap(sa[ty], sb[tx]);
}

// ---- NMS: upper-triangle suppression mask (64x64 tiles) -----------------------------------------
template <int KIND>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int N, float thresh, int W,
                                                      unsigned long long* __restrict__ mask) {
    const int row_blk = blockIdx.y, col_blk = blockIdx.x;
    if (col_blk < row_blk) return;                       // never read by the sweep
    const int t = threadIdx.x;
    const int row = row_blk * 64 + t, col = col_blk * 64 + t;
    const int col_size = min(64, N - col_blk * 64);
    unsigned long long bits = 0;
    if (KIND == PRCNN_NMS_ROTATED) {
        __shared__ RBox srow[64], scol[64];
        if (row < N) make_rbox(boxes + (size_t)row * 5, srow[t]);
        if (col < N) make_rbox(boxes + (size_t)col * 5, scol[t]);
        __syncthreads();
        if (row < N) {
            int start = (row_blk == col_blk) ? t + 1 : 0;    // iou3d_kernel.cu:281-283
            for (int i = start; i < col_size; i++)
                if (iou_bev(srow[t], scol[i]) > thresh) bits |= 1ULL << i;
        }
    } else {
        __shared__ float scolb[64 * 5];
        if (col < N)
            for (int c = 0; c < 5; c++) scolb[t * 5 + c] = boxes[(size_t)col * 5 + c];
        __syncthreads();
        if (row < N) {
            float rb[5];
            for (int c = 0; c < 5; c++) rb[c] = boxes[(size_t)row * 5 + c];
            int start = (row_blk == col_blk) ? t + 1 : 0;
            for (int i = start; i < col_size; i++)
                if (iou_normal(rb, scolb + i * 5) > thresh) bits |= 1ULL << i;
        }
    }
    if (row < N) mask[(size_t)row * W + col_blk] = bits;
}

// ---- NMS: greedy sweep on the device (iou3d.cpp:100-119), one 256-thread workgroup per problem ------
// Per 64-box block: wave 0 resolves the block serially on the scalar unit (diag word per lane, v_readlane);
// then all 4 waves fold the kept rows into the suppression words of later blocks, one word per thread with 16
// independent loads in flight (the fold is latency-bound: a dependent load per kept row is what made the first
// version slow).  max_keep > 0 ends the sweep as soon as enough boxes are kept.
#define SWEEP_THREADS 256
__global__ __launch_bounds__(SWEEP_THREADS) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int N, int W,
                                                                  int max_keep, int64_t* __restrict__ keep,
                                                                  int32_t* __restrict__ num_keep) {
    extern __shared__ unsigned long long remv[];         // W words: bit set = suppressed; then 2 words of exchange
    unsigned long long* xchg = remv + W;                 // [0] = kept bits of the current block, [1] = running count
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < W; w += SWEEP_THREADS) remv[w] = 0ULL;
    if (tid == 0) xchg[1] = 0ULL;
    __syncthreads();
    for (int blk = 0; blk < W; blk++) {
        if (wave == 0) {
            const int row = blk * 64 + lane;
            unsigned long long diag = row < N ? mask[(size_t)row * W + blk] : 0ULL;
            unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            unsigned long long cur = remv[blk];
            // readfirstlane returns a SIGNED int: go through unsigned or the low word sign-extends into the high one
            cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur >> 32)) << 32) |
                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur);
            int num = (int)xchg[1];
            num = __builtin_amdgcn_readfirstlane(num);
            const int nrows = min(64, N - blk * 64);
            unsigned long long kept = 0ULL;
            for (int t = 0; t < nrows; t++) {            // serial inside the block, registers only
                if (!((cur >> t) & 1ULL)) {
                    kept |= 1ULL << t;
                    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)dlo, t);
                    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)dhi, t);
                    cur |= ((unsigned long long)hi << 32) | lo;
                }
            }
            if ((kept >> lane) & 1ULL) keep[num + __popcll(kept & ((1ULL << lane) - 1ULL))] = row;
            if (lane == 0) { xchg[0] = kept; xchg[1] = (unsigned long long)(num + __popcll(kept)); }
        }
        __syncthreads();
        const unsigned long long kept = xchg[0];
        const int num = (int)xchg[1];
        if (max_keep > 0 && num >= max_keep) break;       // uniform: every thread reads the same LDS word
        // fold the kept rows of this block into the suppression words of later blocks
        for (int w = blk + 1 + tid; w < W; w += SWEEP_THREADS) {
            unsigned long long acc = remv[w];
            const unsigned long long* col = mask + (size_t)blk * 64 * W + w;
#pragma unroll 1
            for (int t0 = 0; t0 < 64; t0 += 16) {
                unsigned long long part[16];
#pragma unroll
                for (int u = 0; u < 16; u++)               // 16 independent loads; `kept` is uniform
                    part[u] = ((kept >> (t0 + u)) & 1ULL) ? col[(size_t)(t0 + u) * W] : 0ULL;
#pragma unroll
                for (int u = 0; u < 16; u++) acc |= part[u];
            }
            remv[w] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) *num_keep = (max_keep > 0 && (int)xchg[1] > max_keep) ? max_keep : (int)xchg[1];
}

static int pair_matrix(const char* op, bool iou, const float* a, int na, const float* b, int nb, float* out, hipStream_t s) {
    if (na < 0 || nb < 0) return prcnn_fail(PRCNN_EINVAL, "%s: bad shape", op);
    if (na == 0 || nb == 0) return PRCNN_OK;
    if (!a || !b || !out) return prcnn_fail(PRCNN_EINVAL, "%s: null pointer", op);
    dim3 grid(prcnn_divup(nb, 16), prcnn_divup(na, 16));
    if (iou) hipLaunchKernelGGL(pair_matrix_kernel<true>, grid, dim3(256), 0, s, a, na, b, nb, out);
    else hipLaunchKernelGGL(pair_matrix_kernel<false>, grid, dim3(256), 0, s, a, na, b, nb, out);
    PRCNN_LAUNCH_CHECK(op);
    return PRCNN_OK;
}

PRCNN_API int prcnn_boxes_overlap_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out,
                                      prcnn_stream_t stream) {
    return pair_matrix("prcnn_boxes_overlap_bev", false, boxes_a, Na, boxes_b, Nb, out, (hipStream_t)stream);
}
PRCNN_API int prcnn_boxes_iou_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out,
                                  prcnn_stream_t stream) {
    return pair_matrix("prcnn_boxes_iou_bev", true, boxes_a, Na, boxes_b, Nb, out, (hipStream_t)stream);
}

PRCNN_API size_t prcnn_nms_workspace_bytes(int N) {
    if (N <= 0) return 0;
    size_t W = (size_t)(N + 63) / 64;
    return (size_t)N * W * sizeof(unsigned long long);
}

PRCNN_API int prcnn_nms(const float* boxes, int N, float thresh, int kind, int max_keep, int64_t* keep, int32_t* num_keep,
                        void* workspace, size_t workspace_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(num_keep, "prcnn_nms: null num_keep");
    PRCNN_REQUIRE(N >= 0 && max_keep >= 0, "prcnn_nms: bad N=%d max_keep=%d", N, max_keep);
    PRCNN_REQUIRE(kind == PRCNN_NMS_ROTATED || kind == PRCNN_NMS_NORMAL, "prcnn_nms: bad kind %d", kind);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_nms: memset failed");
        return PRCNN_OK;
    }
    PRCNN_REQUIRE(boxes && keep && workspace, "prcnn_nms: null pointer");
    PRCNN_REQUIRE(workspace_bytes >= prcnn_nms_workspace_bytes(N), "prcnn_nms: workspace %zu < %zu bytes", workspace_bytes,
                  prcnn_nms_workspace_bytes(N));
    const int W = (N + 63) / 64;
    PRCNN_REQUIRE((size_t)W * 8 <= 60 * 1024, "prcnn_nms: N=%d too large for the LDS suppression bitmap", N);
    unsigned long long* mask = (unsigned long long*)workspace;
    dim3 grid(W, W);
    if (kind == PRCNN_NMS_ROTATED)
        hipLaunchKernelGGL(nms_mask_kernel<PRCNN_NMS_ROTATED>, grid, dim3(64), 0, s, boxes, N, thresh, W, mask);
    else
        hipLaunchKernelGGL(nms_mask_kernel<PRCNN_NMS_NORMAL>, grid, dim3(64), 0, s, boxes, N, thresh, W, mask);
    PRCNN_LAUNCH_CHECK("prcnn_nms(mask)");
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(SWEEP_THREADS), (size_t)(W + 2) * 8, s, mask, N, W, max_keep, keep, num_keep);
    PRCNN_LAUNCH_CHECK("prcnn_nms(sweep)");
    return PRCNN_OK;
}
