#!/usr/bin/env python3
"""Generate tests/golden/residual_ref.npz: adversarial inputs for the decisions that sit within rounding distance of a threshold,
pinned by the reference's OWN native code (oracle/_ref = lib/utils/iou3d/src/*, lib/utils/roipool3d/src/* compiled in place).
Needs /root/reference; the fixture travels to the GPU box.

The reference calls float cosf / sinf / atan2 on the box angle.  Rounds 1-2 of the kernels evaluated cos / sin in double, rounded
once, and ordered polygon vertices without atan2 (oracle trig_mode 1): glibc's cosf differs from the correctly rounded value for
1.3 % of angles, by one ulp, which flipped a handful of the decisions built here.  Since round 3 the kernels restate glibc's
routines operation by operation (csrc/ref_trig.h, oracle trig_mode 2) and reproduce EVERY reference column of this fixture bit
for bit; the trig_mode-1 columns are kept as the record of what the old arithmetic gave.

  nms_*    box pairs bisected to within 1e-5 of the IoU threshold (3 x 150 pairs), and 12 whole NMS problems seeded with ten
           such pairs each: the reference's IoUs and keep sets (and the legacy arithmetic's);
  pib_*    point-in-box: points exactly ON box faces / one and two ulps either side / on the 10 m gate, for boxes at
           ry in {0, +-pi/2, +-pi} (on-grid coordinates: the face points are exact) and at random angles (face points
           rounded to fp32): the reference's flags and pooled tensors (and the legacy arithmetic's).

tests/test_gpu_parity_residuals.py asserts equality of the HIP kernels with the reference columns, and that the legacy columns
still differ from them in exactly the recorded entries (the fixture has not lost its teeth).  Seeds are fixed; re-running
reproduces the file byte for byte.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)
import oracle  # noqa: E402
from util import enlarge, rand_bev  # noqa: E402

NMS_SETS = (("rotated", 0.8), ("rotated", 0.1), ("normal", 0.8))
SPECIAL_RY = (0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi)


def _pad(rows, width, fill=-1):
    out = np.full((len(rows), width), fill, np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def nms_part(cpu, ref):
    from test_oracle_round2 import _near_threshold_pairs
    g = {}
    for kind, thr in NMS_SETS:
        tag = "%s_%s" % (kind, thr)
        pairs = _near_threshold_pairs(cpu, thr, 150, seed=11, kind=kind)
        P = np.stack([np.stack(p) for p in pairs]).astype(np.float32)               # (n, 2, 5): a first (higher score)
        g["nms_pairs_" + tag] = P
        g["nms_ref_keep_" + tag] = np.array([len(ref.nms(p, thr, kind)) for p in P], np.int8)
        g["nms_m1_keep_" + tag] = np.array([len(cpu.nms(p, thr, kind, 1)) for p in P], np.int8)
        if kind == "rotated":
            g["nms_ref_iou_" + tag] = np.array([ref.boxes_iou_bev(p[:1], p[1:])[0, 0] for p in P], np.float32)
            g["nms_m1_iou_" + tag] = np.array([cpu.boxes_iou_bev(p[:1], p[1:], trig_mode=1)[0, 0] for p in P], np.float32)
        print(tag, "pairs", len(P), "decisions differing from the reference:",
              int((g["nms_ref_keep_" + tag] != g["nms_m1_keep_" + tag]).sum()))
    probs, kref, km1 = [], [], []
    for seed in range(12):
        boxes = rand_bev(300, 6.0, seed=100 + seed)
        for j, (a, b) in enumerate(_near_threshold_pairs(cpu, 0.8, 10, seed=200 + seed, kind="rotated")):
            boxes[10 * j], boxes[10 * j + 5] = a, b
        probs.append(boxes)
        kref.append(ref.nms(boxes, 0.8, "rotated"))
        km1.append(cpu.nms(boxes, 0.8, "rotated", 1))
    g["nms_prob_boxes"], g["nms_prob_ref_keep"], g["nms_prob_m1_keep"] = np.stack(probs), _pad(kref, 300), _pad(km1, 300)
    print("whole problems whose keep set differs:", sum(not np.array_equal(a, b) for a, b in zip(kref, km1)), "of 12")
    return g


def _face_points(box, rng):
    """points on / next to the faces of one box [x, y_bottom, z, h, w, l, ry]: local (u along l, v along w) on a grid that
    contains the faces, mapped to world coordinates with the rotation evaluated in double, rounded to fp32, plus the
    neighbours one and two ulps away in x and in z; y on the two horizontal faces, one ulp outside them, and mid-height"""
    cx, yb, cz, h, w, l, ry = [float(v) for v in box]
    cy = yb - h / 2
    us = np.array([-l / 2, -l / 4, 0.0, l / 4, l / 2])
    vs = np.array([-w / 2, 0.0, w / 2])
    c, s = np.cos(float(np.float32(ry))), np.sin(float(np.float32(ry)))
    pts = []
    for u in us:
        for v in vs:
            # inverse of x_rot = dx cos - dz sin, z_rot = dx sin + dz cos
            dx, dz = u * c + v * s, -u * s + v * c
            x0, z0 = np.float32(cx + dx), np.float32(cz + dz)
            for kx in (-2, -1, 0, 1, 2):
                for kz in (-1, 0, 1):
                    x, z = x0, z0
                    for _ in range(abs(kx)):
                        x = np.nextafter(x, np.float32(np.inf if kx > 0 else -np.inf))
                    for _ in range(abs(kz)):
                        z = np.nextafter(z, np.float32(np.inf if kz > 0 else -np.inf))
                    pts.append((x, np.float32(cy), z))
    ys = [np.float32(cy - h / 2), np.float32(cy + h / 2), np.nextafter(np.float32(cy - h / 2), np.float32(-np.inf)),
          np.nextafter(np.float32(cy + h / 2), np.float32(np.inf))]
    for y in ys:
        for u in (-l / 2, 0.0, l / 2):
            dx, dz = u * c, -u * s
            pts.append((np.float32(cx + dx), y, np.float32(cz + dz)))
    return np.array(pts, np.float32)


def pib_scene():
    """(pts (N,3), boxes (M,7), n_special, big_box_index, number of leading points that belong to the special boxes).  Special boxes first: centres on a 1/8 grid, sizes on a 1/4 grid,
    ry in {0, +-pi/2, +-pi} -> at ry = 0 / +-pi the face points are exact fp32 numbers; then 40 random-angle boxes; last a
    box longer than the reference's 10 m gate (l = 30 m) with points beyond the gate but inside the box."""
    rng = np.random.default_rng(5)
    boxes = []
    for i, ry in enumerate(SPECIAL_RY * 2):
        boxes.append([2.5 + 8 * i, 1.75, 20.25 + (i % 3), 1.5, 1.75 - 0.25 * (i % 2), 4.0 + 0.5 * (i % 3), np.float32(ry)])
    n_special = len(boxes)
    for i in range(40):
        boxes.append([rng.uniform(-30, 30), rng.uniform(1.0, 2.0), rng.uniform(5, 60), rng.uniform(1.4, 1.8), rng.uniform(1.5, 1.8),
                      rng.uniform(3.5, 4.5), rng.uniform(-np.pi, np.pi)])
    boxes.append([0.0, 1.75, 100.0, 1.5, 2.0, 30.0, 0.0])
    boxes = np.array(boxes, np.float32)
    pts = [_face_points(b, rng) for b in boxes]
    big = len(boxes) - 1
    # on and around the 10 m gate of the long box (|x - cx| == 10 is NOT rejected: the reference tests `> max_dis`)
    gate = [(10.0, 1.0, 100.0), (np.nextafter(np.float32(10.0), np.float32(20)), 1.0, 100.0), (-10.0, 1.0, 100.5), (12.0, 1.0, 100.0),
            (14.999, 1.0, 99.5), (9.5, 1.0, 100.0)]
    pts.append(np.array(gate, np.float32))
    pts.append(rng.uniform([-35, 0, 0], [35, 2.5, 105], (500, 3)).astype(np.float32))
    return np.concatenate(pts, 0), boxes, n_special, big, len(pts[0]) * n_special


def labels_from_flags(pts, boxes, fg_flags, en_flags):
    """kitti_rcnn_dataset.py:365-394 with the two hull tests replaced by given in-box flags (M,N): boxes applied in order"""
    cls = np.zeros(pts.shape[0], np.int32)
    reg = np.zeros((pts.shape[0], 7), np.float32)
    for k in range(boxes.shape[0]):
        fg, en = fg_flags[k] > 0, en_flags[k] > 0
        cls[fg] = 1
        cls[np.logical_xor(fg, en)] = -1
        center3d = boxes[k][0:3].copy()
        center3d[1] -= boxes[k][3] / 2
        reg[fg, 0:3] = center3d - pts[fg]
        reg[fg, 3:7] = boxes[k][3:7]
    return cls, reg


def pib_part(cpu, ref):
    pts, boxes, n_special, big, n_special_pts = pib_scene()
    g = {"pib_pts": pts, "pib_boxes": boxes, "pib_n_special": np.int64(n_special), "pib_big": np.int64(big),
         "pib_n_special_pts": np.int64(n_special_pts)}
    fr = ref.pts_in_boxes3d_cpu(pts, boxes).astype(np.int8)
    f1 = cpu.pts_in_boxes3d(pts, boxes, trig_mode=1).astype(np.int8)
    assert np.array_equal(cpu.pts_in_boxes3d(pts, boxes, trig_mode=0).astype(np.int8), fr)          # the pin
    g["pib_ref_flags"], g["pib_m1_flags"] = fr, f1
    diff = np.argwhere(fr != f1)
    print("point-in-box: %d boxes x %d points, %d in-box (reference); entries differing: %d (special-angle rows: %d)"
          % (boxes.shape[0], pts.shape[0], int(fr.sum()), len(diff), int((diff[:, 0] < n_special).sum())))
    # the reference's roipool3d_cpu on the same scene (S = 32 < points in most boxes: truncation; feature = point index)
    feat = np.arange(pts.shape[0], dtype=np.float32)[:, None] * np.array([1.0, -0.5], np.float32)[None]
    pp, pf, pe = ref.roipool3d_cpu(pts, boxes, feat, 32)
    g["pib_feat"], g["pib_ref_pooled_pts"], g["pib_ref_pooled_feat"], g["pib_ref_empty"] = feat, pp, pf, pe
    # labels and the GT-augmentation edit derived from the reference's flags (the long box left out of the labels: the label
    # generator has no gate, the reference's pts_in_boxes3d_cpu has)
    lb = boxes[:big]
    en = ref.pts_in_boxes3d_cpu(pts, enlarge(lb, 0.2)).astype(np.int8)
    cls, reg = labels_from_flags(pts, lb, fr[:big], en)
    g["pib_ref_cls"], g["pib_ref_reg"] = cls.astype(np.int8), reg
    tall = boxes.copy()
    tall[:, 3] += 2
    g["pib_ref_removed"] = (ref.pts_in_boxes3d_cpu(pts, tall).max(0) > 0).astype(np.int8)
    return g


def main():
    cpu, ref = oracle.cpu(), oracle.ref()
    if ref is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    g = nms_part(cpu, ref)
    g.update(pib_part(cpu, ref))
    np.savez_compressed(os.path.join(HERE, "residual_ref.npz"), **g)
    print("wrote residual_ref.npz")


if __name__ == "__main__":
    main()
