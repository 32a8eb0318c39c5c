// host_roipool3d.hip -- HOST (CPU) twins of the two dataloader-side entry points of the reference's roipool3d extension.
//
// The reference calls `roipool3d_cuda.pts_in_boxes3d_cpu` / `roipool3d_cuda.roipool3d_cpu`
// (lib/utils/roipool3d/src/roipool3d.cpp:82-195) from `Dataset.__getitem__`, i.e. inside forked DataLoader workers
// (lib/datasets/kitti_rcnn_dataset.py:487,582,625,843,970; tools/generate_gt_database.py:72) where no HIP context may be
// created.  These functions therefore contain no HIP call at all: plain C++ on host pointers, re-entrant, no allocation
// beyond one small per-call box table.  (The file carries the .hip suffix only so that the one build rule compiles it.)
//
// Arithmetic contract = the reference's CPU code, so results are bit-identical to it (checked in tests/test_host_twins.py):
//   cy = float(double(bottom_y) - double(h) / 2);   reject when |x-cx| > 10 or |z-cz| > 10 (float compares) or
//   double(|y-cy|) > double(h)/2;   cos/sin = libm cosf/sinf (the reference's `cos(angle)` on a float argument resolves
//   to the float overload in C++; the DEVICE kernels use double-evaluated trig rounded once instead);
//   x_rot, z_rot in float with individually rounded operations (the library is built with -ffp-contract=off);
//   the four half-extent compares in double, bounds inclusive.
// Organisation differs from the reference's per-(box,point) function call: every box's constants (centre, double
// half-extents, rounded cos/sin) are derived once, the point loop touches nothing else.
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

struct HostBox {
    float cx, cy, cz, cosa, sina;
    double half_h, half_w, half_l;
};

inline HostBox make_box(const float* b) {
    HostBox k;
    k.cx = b[0];
    k.cz = b[2];
    k.half_h = (double)b[3] / 2.0;
    k.half_w = (double)b[4] / 2.0;
    k.half_l = (double)b[5] / 2.0;
    k.cy = (float)((double)b[1] - k.half_h);
    k.cosa = cosf(b[6]);
    k.sina = sinf(b[6]);
    return k;
}

inline bool inside(const HostBox& k, const float* p) {
    const float dx = p[0] - k.cx, dz = p[2] - k.cz;
    if (fabsf(dx) > 10.0f || (double)fabsf(p[1] - k.cy) > k.half_h || fabsf(dz) > 10.0f) return false;
    const float xr = dx * k.cosa + dz * (-k.sina);
    const float zr = dx * k.sina + dz * k.cosa;
    return ((double)xr >= -k.half_l) & ((double)xr <= k.half_l) & ((double)zr >= -k.half_w) & ((double)zr <= k.half_w);
}

}  // namespace

// flags (M,N) int64 = 1 where point n lies in box m, else 0.   [roipool3d.cpp:97-125 pts_in_boxes3d_cpu]
PRCNN_API int prcnn_host_pts_in_boxes3d(const float* pts, const float* boxes3d, int64_t N, int64_t M, int64_t* flags) {
    PRCNN_REQUIRE(N >= 0 && M >= 0, "prcnn_host_pts_in_boxes3d: negative size");
    if (N == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(pts && boxes3d && flags, "prcnn_host_pts_in_boxes3d: null pointer");
    for (int64_t m = 0; m < M; m++) {
        const HostBox k = make_box(boxes3d + m * 7);
        int64_t* row = flags + m * N;
        for (int64_t n = 0; n < N; n++) row[n] = inside(k, pts + n * 3) ? 1 : 0;
    }
    return PRCNN_OK;
}

// Per box: the first <= S in-box points in index order -> pooled_pts (M,S,3), pooled_feat (M,S,C); a box holding cnt < S
// points repeats its rows cyclically (slot j copies slot j % cnt); an empty box sets empty[m] = 1 and leaves its rows
// untouched (the caller zero-initialises, roipool3d_utils.py:77-79).   [roipool3d.cpp:127-195 roipool3d_cpu]
PRCNN_API int prcnn_host_roipool3d(const float* pts, const float* boxes3d, const float* feat, int64_t N, int64_t M, int64_t C,
                                   int64_t S, float* pooled_pts, float* pooled_feat, int64_t* empty) {
    PRCNN_REQUIRE(N >= 0 && M >= 0 && C >= 0 && S >= 0, "prcnn_host_roipool3d: negative size");
    if (M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(boxes3d && empty && (N == 0 || pts) && (S == 0 || pooled_pts) && (S * C == 0 || (pooled_feat && feat)),
                  "prcnn_host_roipool3d: null pointer");
    std::vector<int64_t> hit((size_t)S);
    for (int64_t m = 0; m < M; m++) {
        const HostBox k = make_box(boxes3d + m * 7);
        int64_t cnt = 0;
        // the reference stops scanning at the first in-box point that no longer fits (cnt == S); so does this loop
        for (int64_t n = 0; n < N && cnt < S; n++)
            if (inside(k, pts + n * 3)) hit[(size_t)cnt++] = n;
        empty[m] = cnt == 0;            // (S == 0 marks every box empty, as in the reference)
        if (cnt == 0) continue;
        float* op = pooled_pts + m * S * 3;
        float* of = pooled_feat ? pooled_feat + m * S * C : nullptr;
        for (int64_t j = 0; j < S; j++) {
            const int64_t n = hit[(size_t)(j % cnt)];
            memcpy(op + j * 3, pts + n * 3, 3 * sizeof(float));
            if (C) memcpy(of + j * C, feat + n * C, (size_t)C * sizeof(float));
        }
    }
    return PRCNN_OK;
}
