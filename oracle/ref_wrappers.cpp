// ref_wrappers.cpp -- TEST INFRASTRUCTURE (oracle/_ref build only).
// extern "C" entry points around the reference's OWN host functions
// (lib/utils/iou3d/src/iou3d.cpp:31,52,73,123; lib/utils/roipool3d/src/roipool3d.cpp:15,48,97,127),
// which are compiled in place from /root/reference against oracle/ref_shim.  Nothing here
// re-implements reference logic: it only wraps raw pointers into the stub at::Tensor.
#include "prcnn_ref_shim.h"
#include <cstdint>

int boxes_overlap_bev_gpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_overlap);
int boxes_iou_bev_gpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou);
int nms_gpu(at::Tensor boxes, at::Tensor keep, float nms_overlap_thresh);
int nms_normal_gpu(at::Tensor boxes, at::Tensor keep, float nms_overlap_thresh);
int roipool3d_gpu(at::Tensor xyz, at::Tensor boxes3d, at::Tensor pts_feature, at::Tensor pooled_features,
                  at::Tensor pooled_empty_flag);
int roipool3d_gpu_slow(at::Tensor xyz, at::Tensor boxes3d, at::Tensor pts_feature, at::Tensor pooled_features,
                       at::Tensor pooled_empty_flag);
int pts_in_boxes3d_cpu(at::Tensor pts_flag, at::Tensor pts, at::Tensor boxes3d);
int roipool3d_cpu(at::Tensor pts, at::Tensor boxes3d, at::Tensor pts_feature, at::Tensor pooled_pts,
                  at::Tensor pooled_features, at::Tensor pooled_empty_flag);

static at::Tensor T(const void* p, std::vector<long> s) { return at::Tensor((void*)p, std::move(s)); }

extern "C" {
int ref_boxes_overlap_bev(const float* a, int na, const float* b, int nb, float* out) {
    return boxes_overlap_bev_gpu(T(a, {na, 5}), T(b, {nb, 5}), T(out, {na, nb}));
}
int ref_boxes_iou_bev(const float* a, int na, const float* b, int nb, float* out) {
    return boxes_iou_bev_gpu(T(a, {na, 5}), T(b, {nb, 5}), T(out, {na, nb}));
}
int ref_nms(const float* boxes, int n, float thr, int64_t* keep) { return nms_gpu(T(boxes, {n, 5}), T(keep, {n}), thr); }
int ref_nms_normal(const float* boxes, int n, float thr, int64_t* keep) {
    return nms_normal_gpu(T(boxes, {n, 5}), T(keep, {n}), thr);
}
int ref_roipool3d_gpu(const float* xyz, const float* boxes, const float* feat, int B, int N, int M, int C, int S,
                      float* out, int* empty, int slow) {
    at::Tensor x = T(xyz, {B, N, 3}), bx = T(boxes, {B, M, 7}), f = T(feat, {B, N, C});
    at::Tensor o = T(out, {B, M, S, 3 + C}), e = T(empty, {B, M});
    return slow ? roipool3d_gpu_slow(x, bx, f, o, e) : roipool3d_gpu(x, bx, f, o, e);
}
int ref_pts_in_boxes3d_cpu(int64_t* flags, const float* pts, const float* boxes, int N, int M) {
    return pts_in_boxes3d_cpu(T(flags, {M, N}), T(pts, {N, 3}), T(boxes, {M, 7}));
}
int ref_roipool3d_cpu(const float* pts, const float* boxes, const float* feat, int N, int M, int C, int S,
                      float* pooled_pts, float* pooled_feat, int64_t* empty) {
    return roipool3d_cpu(T(pts, {N, 3}), T(boxes, {M, 7}), T(feat, {N, C}), T(pooled_pts, {M, S, 3}),
                         T(pooled_feat, {M, S, C}), T(empty, {M}));
}
}
