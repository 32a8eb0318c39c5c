// kitti_eval.hip -- the two compute kernels of the KITTI object evaluation.
//
// Replaces  tools/kitti_object_eval_python/rotate_iou.py   (numba.cuda rotated-IoU kernel: cannot target ROCm)
//           tools/kitti_object_eval_python/eval.py:87-152   image_box_overlap / bev_box_overlap / d3_box_overlap
//           tools/kitti_object_eval_python/eval.py:155-324  compute_statistics_jit / fused_compute_statistics (numba CPU jit)
// The reference forms dense (sum gt) x (sum dt) overlap matrices over parts of ~75 frames and slices the per-frame
// diagonal blocks out, then runs the greedy matching frame by frame, threshold by threshold on the CPU.  Here
//   kitti_overlap_kernel : one workgroup per frame computes exactly that frame's (n_dt x n_gt) block;
//   kitti_stats_kernel   : one thread per (frame, score threshold) runs the sequential matching -- 3769 frames x 41
//                          thresholds = 154 k independent problems per (class, difficulty, overlap) setting.
// Arithmetic follows oracle/prcnn_oracle.c (kr_* / kitti_stats_frame) operation for operation: fp32 individually
// rounded for the rotated intersection (cos/sin in double, rounded once), float64 elsewhere; integer results exact.
#include "common.h"

#define KR_MAXPTS 24

__device__ __forceinline__ float kmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float kadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float ksub(float a, float b) { return __fsub_rn(a, b); }

__device__ void kr_corners(const float* r, float* c) {                       // rotate_iou.py:203-227
    const float a_cos = (float)cos((double)r[4]), a_sin = (float)sin((double)r[4]);
    const float xd = r[2], yd = r[3];
    const float cx[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2}, cy[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c[2 * i] = kadd(kadd(kmul(a_cos, cx[i]), kmul(a_sin, cy[i])), r[0]);
        c[2 * i + 1] = kadd(kadd(kmul(-a_sin, cx[i]), kmul(a_cos, cy[i])), r[1]);
    }
}
__device__ bool kr_point_in_quad(float px, float py, const float* c) {       // :161-177
    const float ab0 = ksub(c[2], c[0]), ab1 = ksub(c[3], c[1]), ad0 = ksub(c[6], c[0]), ad1 = ksub(c[7], c[1]);
    const float ap0 = ksub(px, c[0]), ap1 = ksub(py, c[1]);
    const float abab = kadd(kmul(ab0, ab0), kmul(ab1, ab1)), abap = kadd(kmul(ab0, ap0), kmul(ab1, ap1));
    const float adad = kadd(kmul(ad0, ad0), kmul(ad1, ad1)), adap = kadd(kmul(ad0, ap0), kmul(ad1, ap1));
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
__device__ bool kr_seg_intersection(const float* p1, const float* p2, int i, int j, float* t) {     // :77-115
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = ksub(B0, A0), BA1 = ksub(B1, A1), DA0 = ksub(D0, A0), CA0 = ksub(C0, A0), DA1 = ksub(D1, A1), CA1 = ksub(C1, A1);
    const bool acd = kmul(DA1, CA0) > kmul(CA1, DA0);
    const bool bcd = kmul(ksub(D1, B1), ksub(C0, B0)) > kmul(ksub(C1, B1), ksub(D0, B0));
    if (acd != bcd) {
        const bool abc = kmul(CA1, BA0) > kmul(BA1, CA0), abd = kmul(DA1, BA0) > kmul(BA1, DA0);
        if (abc != abd) {
            const float DC0 = ksub(D0, C0), DC1 = ksub(D1, C1);
            const float ABBA = ksub(kmul(A0, B1), kmul(B0, A1)), CDDC = ksub(kmul(C0, D1), kmul(D0, C1));
            const float DH = ksub(kmul(BA1, DC0), kmul(BA0, DC1));
            const float Dx = ksub(kmul(ABBA, DC0), kmul(BA0, CDDC)), Dy = ksub(kmul(ABBA, DC1), kmul(BA1, CDDC));
            t[0] = Dx / DH; t[1] = Dy / DH;
            return true;
        }
    }
    return false;
}
__device__ float kr_inter(const float* r1, const float* r2) {                // :230-245
    float c1[8], c2[8], ip[2 * KR_MAXPTS], vs[KR_MAXPTS], t[2];
    kr_corners(r1, c1); kr_corners(r2, c2);
    int n = 0;
    for (int i = 0; i < 4; i++) {
        if (kr_point_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { ip[2 * n] = c1[2 * i]; ip[2 * n + 1] = c1[2 * i + 1]; n++; }
        if (kr_point_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { ip[2 * n] = c2[2 * i]; ip[2 * n + 1] = c2[2 * i + 1]; n++; }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (n < KR_MAXPTS && kr_seg_intersection(c1, c2, i, j, t)) { ip[2 * n] = t[0]; ip[2 * n + 1] = t[1]; n++; }
    if (n > 0) {                                                             // :33-74
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < n; i++) { cx = kadd(cx, ip[2 * i]); cy = kadd(cy, ip[2 * i + 1]); }
        cx = cx / (float)n; cy = cy / (float)n;
        for (int i = 0; i < n; i++) {
            float v0 = ksub(ip[2 * i], cx), v1 = ksub(ip[2 * i + 1], cy);
            const float d = sqrtf(kadd(kmul(v0, v0), kmul(v1, v1)));
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = ksub(-2.0f, v0);
            vs[i] = v0;
        }
        for (int i = 1; i < n; i++)
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1]; ip[2 * j] = ip[2 * j - 2]; ip[2 * j + 1] = ip[2 * j - 1];
                    j--;
                }
                vs[j] = temp; ip[2 * j] = tx; ip[2 * j + 1] = ty;
            }
    }
    float area = 0.f;                                                        // :24-31
    for (int i = 0; i < n - 2; i++) {
        const float *a = ip, *b = ip + 2 * i + 2, *c = ip + 2 * i + 4;
        area = kadd(area, fabsf(ksub(kmul(ksub(a[0], c[0]), ksub(b[1], c[1])), kmul(ksub(a[1], c[1]), ksub(b[0], c[0]))) / 2.0f));
    }
    return area;
}
__device__ float kr_iou_eval(const float* r1, const float* r2, int criterion) {      // :248-260
    const float area1 = kmul(r1[2], r1[3]), area2 = kmul(r2[2], r2[3]), ai = kr_inter(r1, r2);
    if (criterion == -1) return ai / ksub(kadd(area1, area2), ai);
    if (criterion == 0) return ai / area1;
    if (criterion == 1) return ai / area2;
    return ai;
}

__global__ __launch_bounds__(256) void rotate_iou_eval_kernel(const float* __restrict__ boxes, int N, const float* __restrict__ query,
                                                              int K, int criterion, float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)N * K) return;
    const int n = (int)(e / K), k = (int)(e - (long)n * K);
    float rb[5], rq[5];
#pragma unroll
    for (int c = 0; c < 5; c++) { rb[c] = boxes[(size_t)n * 5 + c]; rq[c] = query[(size_t)k * 5 + c]; }
    out[e] = kr_iou_eval(rq, rb, criterion);                                 // kernel argument order, rotate_iou.py:282-284
}

__global__ __launch_bounds__(64) void kitti_overlap_kernel(int metric, const double* __restrict__ dt, const int32_t* __restrict__ dt_off,
                                                           const double* __restrict__ gt, const int32_t* __restrict__ gt_off,
                                                           const int64_t* __restrict__ ov_off, double* __restrict__ out) {
    const int f = blockIdx.x;
    const int d0 = dt_off[f], g0 = gt_off[f];
    const int nd = dt_off[f + 1] - d0, ng = gt_off[f + 1] - g0;
    double* o = out + ov_off[f];
    for (int e = threadIdx.x; e < nd * ng; e += 64) {
        const int n = e / ng, k = e - n * ng;
        double v = 0.0;
        if (metric == 0) {                                                   // eval.py:87-113 image_box_overlap, criterion -1
            const double *b = dt + (size_t)(d0 + n) * 4, *q = gt + (size_t)(g0 + k) * 4;
            const double qarea = (q[2] - q[0]) * (q[3] - q[1]);
            const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
            if (iw > 0) {
                const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
                if (ih > 0) { const double ua = (b[2] - b[0]) * (b[3] - b[1]) + qarea - iw * ih; v = iw * ih / ua; }
            }
        } else {
            const double *b = dt + (size_t)(d0 + n) * 7, *q = gt + (size_t)(g0 + k) * 7;
            const float rb[5] = {(float)b[0], (float)b[2], (float)b[3], (float)b[5], (float)b[6]};
            const float rq[5] = {(float)q[0], (float)q[2], (float)q[3], (float)q[5], (float)q[6]};
            if (metric == 1) v = (double)kr_iou_eval(rq, rb, -1);            // :116-118
            else {                                                           // :121-152 d3_box_overlap
                const float rinc = kr_iou_eval(rq, rb, 2);
                if (rinc > 0) {
                    const double iw = fmin(b[1], q[1]) - fmax(b[1] - b[4], q[1] - q[4]);
                    if (iw > 0) {
                        const double area1 = b[3] * b[4] * b[5], area2 = q[3] * q[4] * q[5];
                        const double inc = iw * (double)rinc;
                        v = (double)(float)(inc / (area1 + area2 - inc));
                    }
                }
            }
        }
        o[e] = v;
    }
}

#define KS_MAX_DET 1024          // detections per frame the per-thread bitmaps cover
struct StatsParams {
    const double* overlaps; const int64_t* ov_off;
    const double* gt_datas; const int32_t* gt_off;
    const double* dt_datas; const int32_t* dt_off;
    const int32_t* ign_gt; const int32_t* ign_det;
    const double* dc; const int32_t* dc_off;
    const double* thresholds;
    double* res; double* matched;
    int F, T, metric, compute_fp, compute_aos;
    double min_overlap;
};
__device__ __forceinline__ bool bit_get(const unsigned long long* m, int j) { return (m[j >> 6] >> (j & 63)) & 1ULL; }
__device__ __forceinline__ void bit_set(unsigned long long* m, int j) { m[j >> 6] |= 1ULL << (j & 63); }

// eval.py:155-268 compute_statistics_jit, one (frame, threshold) per thread
__global__ __launch_bounds__(64) void kitti_stats_kernel(StatsParams P) {
    const long e = (long)blockIdx.x * 64 + threadIdx.x;
    if (e >= (long)P.F * P.T) return;
    const int f = (int)(e / P.T), t = (int)(e - (long)f * P.T);
    const int g0 = P.gt_off[f], d0 = P.dt_off[f], c0 = P.dc_off[f];
    const int ng = P.gt_off[f + 1] - g0, nd = P.dt_off[f + 1] - d0, ndc = P.dc_off[f + 1] - c0;
    const double* ov = P.overlaps + P.ov_off[f];
    const double* gtd = P.gt_datas + (size_t)g0 * 5;
    const double* dtd = P.dt_datas + (size_t)d0 * 6;
    const int32_t* ign_gt = P.ign_gt + g0;
    const int32_t* ign_det = P.ign_det + d0;
    const double thresh = P.thresholds[t], min_overlap = P.min_overlap;
    const bool compute_fp = P.compute_fp != 0, compute_aos = P.compute_aos != 0;
    unsigned long long assigned[KS_MAX_DET / 64], ign_thr[KS_MAX_DET / 64];
#pragma unroll
    for (int w = 0; w < KS_MAX_DET / 64; w++) { assigned[w] = 0ULL; ign_thr[w] = 0ULL; }
    if (compute_fp)
        for (int j = 0; j < nd; j++)
            if (dtd[j * 6 + 5] < thresh) bit_set(ign_thr, j);
    const double NO_DETECTION = -10000000;
    double tp = 0, fp = 0, fn = 0, sim_sum = 0;
    double* matched = (P.matched && t == 0) ? P.matched + g0 : nullptr;
    for (int i = 0; i < ng; i++) {
        if (matched) matched[i] = __longlong_as_double(0x7ff8000000000000LL);
        if (ign_gt[i] == -1) continue;
        int det_idx = -1;
        bool assigned_ignored = false;
        double valid_detection = NO_DETECTION, max_overlap = 0;
        for (int j = 0; j < nd; j++) {
            if (ign_det[j] == -1 || bit_get(assigned, j) || bit_get(ign_thr, j)) continue;
            const double overlap = ov[(size_t)j * ng + i], score = dtd[j * 6 + 5];
            if (!compute_fp && overlap > min_overlap && score > valid_detection) { det_idx = j; valid_detection = score; }
            else if (compute_fp && overlap > min_overlap && (overlap > max_overlap || assigned_ignored) && ign_det[j] == 0) {
                max_overlap = overlap; det_idx = j; valid_detection = 1; assigned_ignored = false;
            } else if (compute_fp && overlap > min_overlap && valid_detection == NO_DETECTION && ign_det[j] == 1) {
                det_idx = j; valid_detection = 1; assigned_ignored = true;
            }
        }
        if (valid_detection == NO_DETECTION && ign_gt[i] == 0) fn += 1;
        else if (valid_detection != NO_DETECTION && (ign_gt[i] == 1 || ign_det[det_idx] == 1)) bit_set(assigned, det_idx);
        else if (valid_detection != NO_DETECTION) {
            tp += 1;
            if (matched) matched[i] = dtd[det_idx * 6 + 5];
            if (compute_aos) sim_sum += (1.0 + cos(gtd[i * 5 + 4] - dtd[det_idx * 6 + 4])) / 2.0;
            bit_set(assigned, det_idx);
        }
    }
    double similarity = 0;
    if (compute_fp) {
        for (int j = 0; j < nd; j++)
            if (!(bit_get(assigned, j) || ign_det[j] == -1 || ign_det[j] == 1 || bit_get(ign_thr, j))) fp += 1;
        int nstuff = 0;
        if (P.metric == 0) {
            const double* dc = P.dc + (size_t)c0 * 4;
            for (int i = 0; i < ndc; i++)
                for (int j = 0; j < nd; j++) {
                    if (bit_get(assigned, j) || ign_det[j] == -1 || ign_det[j] == 1 || bit_get(ign_thr, j)) continue;
                    const double *b = dtd + j * 6, *q = dc + i * 4;          // image_box_overlap(dt, dc, criterion 0)
                    double o = 0;
                    const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
                    if (iw > 0) {
                        const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
                        if (ih > 0) o = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
                    }
                    if (o > min_overlap) { bit_set(assigned, j); nstuff++; }
                }
        }
        fp -= nstuff;
        if (compute_aos) similarity = (tp > 0 || fp > 0) ? sim_sum : -1;
    }
    double* r = P.res + (size_t)e * 4;
    r[0] = tp; r[1] = fp; r[2] = fn; r[3] = similarity;
}

PRCNN_API int prcnn_rotate_iou_eval(const float* boxes, int N, const float* query, int K, int criterion, float* out,
                                    prcnn_stream_t stream) {
    PRCNN_REQUIRE(N >= 0 && K >= 0 && criterion >= -1 && criterion <= 2, "prcnn_rotate_iou_eval: bad arguments N=%d K=%d criterion=%d", N, K, criterion);
    if (N == 0 || K == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes && query && out, "prcnn_rotate_iou_eval: null pointer");
    hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3(prcnn_divup((long)N * K, 256)), dim3(256), 0, (hipStream_t)stream, boxes, N, query, K,
                       criterion, out);
    PRCNN_LAUNCH_CHECK("prcnn_rotate_iou_eval");
    return PRCNN_OK;
}

PRCNN_API int prcnn_kitti_overlaps(int metric, const double* dt, const int32_t* dt_off, const double* gt, const int32_t* gt_off,
                                   const int64_t* ov_off, int F, double* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(metric >= 0 && metric <= 2 && F >= 0, "prcnn_kitti_overlaps: bad metric %d / F %d", metric, F);
    if (F == 0) return PRCNN_OK;
    PRCNN_REQUIRE(dt_off && gt_off && ov_off, "prcnn_kitti_overlaps: null offsets");
    hipLaunchKernelGGL(kitti_overlap_kernel, dim3(F), dim3(64), 0, (hipStream_t)stream, metric, dt, dt_off, gt, gt_off, ov_off, out);
    PRCNN_LAUNCH_CHECK("prcnn_kitti_overlaps");
    return PRCNN_OK;
}

PRCNN_API int prcnn_kitti_statistics(const double* overlaps, const int64_t* ov_off, const double* gt_datas, const int32_t* gt_off,
                                     const double* dt_datas, const int32_t* dt_off, const int32_t* ign_gt, const int32_t* ign_det,
                                     const double* dc, const int32_t* dc_off, int F, int max_det_per_frame, int metric, double min_overlap,
                                     const double* thresholds, int T, int compute_fp, int compute_aos, double* res, double* matched,
                                     prcnn_stream_t stream) {
    PRCNN_REQUIRE(F >= 0 && T >= 0 && metric >= 0 && metric <= 2, "prcnn_kitti_statistics: bad arguments");
    if (F == 0 || T == 0) return PRCNN_OK;
    if (max_det_per_frame > KS_MAX_DET)
        return prcnn_fail(PRCNN_EUNSUPPORTED, "prcnn_kitti_statistics: %d detections in one frame (limit %d)", max_det_per_frame, KS_MAX_DET);
    PRCNN_REQUIRE(ov_off && gt_off && dt_off && dc_off && thresholds && res, "prcnn_kitti_statistics: null pointer");
    StatsParams P;
    P.overlaps = overlaps; P.ov_off = ov_off; P.gt_datas = gt_datas; P.gt_off = gt_off; P.dt_datas = dt_datas; P.dt_off = dt_off;
    P.ign_gt = ign_gt; P.ign_det = ign_det; P.dc = dc; P.dc_off = dc_off; P.thresholds = thresholds; P.res = res; P.matched = matched;
    P.F = F; P.T = T; P.metric = metric; P.compute_fp = compute_fp; P.compute_aos = compute_aos; P.min_overlap = min_overlap;
    hipLaunchKernelGGL(kitti_stats_kernel, dim3(prcnn_divup((long)F * T, 64)), dim3(64), 0, (hipStream_t)stream, P);
    PRCNN_LAUNCH_CHECK("prcnn_kitti_statistics");
    return PRCNN_OK;
}
