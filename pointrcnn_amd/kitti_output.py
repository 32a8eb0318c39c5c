"""Detections -> KITTI result files: the data format between the detector's output and the AP evaluator
(tools/eval_rcnn.py:69-94 save_kitti_format, with lib/utils/kitti_utils.py:66-101 boxes3d_to_corners3d and
lib/utils/calibration.py:106-124 corners3d_to_img_boxes underneath).  Host code, whole-array numpy: the reference formats one box
per Python-loop iteration; here a frame's boxes are projected, clipped, filtered and formatted together.

Numerics follow the reference's dtypes so that the printed digits are the same: corners in float32 (the rotation's y row is
[0, 1, 0], so the rotated x / z are two individually rounded products and one sum -- what a float32 matmul over three terms with
an exact zero gives), projection in float64 (float32 corners next to a float64 ones column), the observation angle in float32.
"""
import os

import numpy as np


def box_corners(boxes3d):
    """(N,7) [x, y(bottom), z, h, w, l, ry] -> (N,8,3) float32 corners in the reference's order: the four bottom corners
    (+l/2,+w/2), (+l/2,-w/2), (-l/2,-w/2), (-l/2,+w/2), then the same four at height -h (y points down)"""
    b = np.asarray(boxes3d, np.float32).reshape(-1, 7)
    h, w, l = b[:, 3:4], b[:, 4:5], b[:, 5:6]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float32)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], np.float32)
    top = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.float32)
    xc, zc, yc = (l / np.float32(2)) * sx, (w / np.float32(2)) * sz, -h * top          # (N,8)
    c, s = np.cos(b[:, 6:7]), np.sin(b[:, 6:7])
    xr = xc * c + zc * s                                  # row vector times [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    zr = xc * -s + zc * c
    return np.stack([b[:, 0:1] + xr, b[:, 1:2] + yc, b[:, 2:3] + zr], axis=2).astype(np.float32)


def corners_to_image_boxes(corners3d, P2):
    """(N,8,3) rect-camera corners, (3,4) projection -> (N,4) [x1, y1, x2, y2] float64 image boxes (min / max over the corners)"""
    P = np.asarray(P2, np.float64)
    c = np.asarray(corners3d).astype(np.float64)
    uvw = c @ P[:, :3].T + P[:, 3]
    u, v = uvw[..., 0] / uvw[..., 2], uvw[..., 1] / uvw[..., 2]
    return np.stack([u.min(1), v.min(1), u.max(1), v.max(1)], axis=1)


def observation_angle(boxes3d):
    """alpha = -sign(beta) pi/2 + beta + ry with beta = atan2(z, x), float32 (eval_rcnn.py:86-88)"""
    b = np.asarray(boxes3d, np.float32).reshape(-1, 7)
    beta = np.arctan2(b[:, 2], b[:, 0])
    return -np.sign(beta) * np.pi / 2 + beta + b[:, 6]


def kitti_lines(boxes3d, scores, P2, img_shape, class_name="Car"):
    """the result-file lines of one frame: type, truncation -1, occlusion -1, alpha, image box, h w l, x y z, ry, score -- boxes
    whose clipped image box is wider / taller than 0.8 of the image are dropped (eval_rcnn.py:73-84)"""
    b = np.asarray(boxes3d, np.float32).reshape(-1, 7)
    sc = np.asarray(scores).reshape(-1)
    if b.shape[0] == 0:
        return []
    H, W = int(img_shape[0]), int(img_shape[1])
    ib = corners_to_image_boxes(box_corners(b), P2)
    ib[:, 0::2] = np.clip(ib[:, 0::2], 0, W - 1)
    ib[:, 1::2] = np.clip(ib[:, 1::2], 0, H - 1)
    keep = ((ib[:, 2] - ib[:, 0]) < W * 0.8) & ((ib[:, 3] - ib[:, 1]) < H * 0.8)
    alpha = observation_angle(b)
    cols = np.column_stack([alpha.astype(np.float64), ib, b[:, 3:6].astype(np.float64), b[:, 0:3].astype(np.float64),
                            b[:, 6].astype(np.float64), sc.astype(np.float64)])[keep]
    fmt = class_name + " -1 -1" + " %.4f" * 13
    return [fmt % tuple(r) for r in cols.tolist()]


def save_kitti_format(sample_id, calib, bbox3d, kitti_output_dir, scores, img_shape, class_name="Car"):
    """the reference's signature (eval_rcnn.py:69); calib: anything with a (3,4) P2 (kitti_input.Calibration).  Writes
    <dir>/<sample_id %06d>.txt (an empty file for a frame without detections, as the reference) and returns the lines."""
    lines = kitti_lines(bbox3d, scores, calib.P2, img_shape, class_name)
    with open(os.path.join(kitti_output_dir, "%06d.txt" % int(sample_id)), "w") as f:
        for ln in lines:
            f.write(ln + "\n")
    return lines


def write_detections(pred_boxes3d, raw_scores, keep, num, sample_ids, calibs, img_shapes, kitti_output_dir, class_name="Car"):
    """PointRCNN.detections() output of a batch -- pred_boxes3d (B,M,7), raw_scores (B,M), keep (B,M) kept rows in order (-1
    padded), num (B) -- to one result file per frame (eval_rcnn.py:600-620: boxes / scores selected by the NMS keep list)."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)      # noqa: E731
    pred, raw, keep, num = to_np(pred_boxes3d), to_np(raw_scores), to_np(keep), to_np(num)
    os.makedirs(kitti_output_dir, exist_ok=True)
    out = []
    for b in range(pred.shape[0]):
        k = keep[b, : int(num[b])].astype(np.int64)
        out.append(save_kitti_format(sample_ids[b], calibs[b], pred[b][k], kitti_output_dir, raw[b][k], img_shapes[b], class_name))
    return out


# ---- RPN feature dumps (the on-disk hand-over between `eval_rcnn.py --save_rpn_feature` and `--train_mode rcnn` / `rcnn_offline`) ----
_FEATURE_SUFFIX = {"features": "", "xyz": "_xyz", "seg": "_seg", "intensity": "_intensity", "rawscore": "_rawscore"}


def rpn_feature_files(kitti_features_dir, sample_id):
    """the five .npy paths of a frame, by content (eval_rcnn.py:101-109, kitti_rcnn_dataset.py:139-150)"""
    return {k: os.path.join(kitti_features_dir, "%06d%s.npy" % (int(sample_id), sfx)) for k, sfx in _FEATURE_SUFFIX.items()}


def save_rpn_features(seg_result, rpn_scores_raw, pts_features, backbone_xyz, backbone_features, kitti_features_dir, sample_id):
    """one frame's RPN outputs as the reference dumps them (eval_rcnn.py:97-110): backbone features (N,C), xyz (N,3), segmentation
    mask (N), point intensity = pts_features[:, 0], raw segmentation scores (N); arrays as given (numpy, or tensors moved to host)"""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)      # noqa: E731
    f = rpn_feature_files(kitti_features_dir, sample_id)
    np.save(f["features"], to_np(backbone_features))
    np.save(f["xyz"], to_np(backbone_xyz))
    np.save(f["seg"], to_np(seg_result))
    np.save(f["intensity"], to_np(pts_features)[:, 0])
    np.save(f["rawscore"], to_np(rpn_scores_raw))


def get_rpn_features(rpn_feature_dir, idx, use_seg_score=False):
    """-> xyz (N,3), features (N,C), intensity (N), seg score (N): the mask, or sigmoid(raw score) with cfg.RCNN.USE_SEG_SCORE
    (kitti_rcnn_dataset.py:139-150; the sigmoid in float32 as torch evaluates it there)"""
    f = rpn_feature_files(rpn_feature_dir, idx)
    if use_seg_score:
        import torch
        seg = torch.sigmoid(torch.from_numpy(np.load(f["rawscore"]).reshape(-1))).numpy()
    else:
        seg = np.load(f["seg"]).reshape(-1)
    return np.load(f["xyz"]), np.load(f["features"]), np.load(f["intensity"]).reshape(-1), seg
