#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

    iou3d_ref.npz, roipool3d_ref.npz -- outputs of the REFERENCE'S OWN native code (lib/utils/iou3d/src/*,
        lib/utils/roipool3d/src/* compiled for the host by oracle/build_ref.py -> oracle/_ref).  Needs
        /root/reference; the fixtures travel to the GPU box, the reference does not.
    pointnet2_oracle.npz -- outputs of the CPU oracle (oracle/prcnn_oracle.c) for the PointNet++ ops, whose
        reference source is an empty git submodule (parity unpinned: these pin the ORACLE's behaviour across
        refactors, not the reference's).

Seeds are fixed; re-running reproduces the files byte for byte.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from util import enlarge, kitti_cloud, rand_bev, rand_boxes3d, unit_cloud  # noqa: E402


def main():
    cpu, ref = oracle.cpu(), oracle.ref()
    if ref is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    # ---- iou3d: reference code
    a, b = rand_bev(120, 5.0, seed=1), rand_bev(90, 5.0, seed=2)
    nms_boxes = rand_bev(700, 7.0, seed=3)
    out = dict(a=a, b=b, overlap=ref.boxes_overlap_bev(a, b), iou=ref.boxes_iou_bev(a, b), nms_boxes=nms_boxes)
    for kind in ("rotated", "normal"):
        for thr in (0.1, 0.5, 0.8):
            out["keep_%s_%s" % (kind, thr)] = ref.nms(nms_boxes, thr, kind)
    np.savez_compressed(os.path.join(HERE, "iou3d_ref.npz"), **out)
    # ---- roipool3d: reference CPU code
    N, M, C, S = 3000, 24, 8, 64
    xyz = kitti_cloud(1, N, seed=42)
    boxes = enlarge(rand_boxes3d(xyz[0], M, seed=7), 1.0)
    boxes[-1, 0] += 300.0            # an empty box
    feat = np.random.default_rng(9).normal(size=(N, C)).astype(np.float32)
    pp, pf, pe = ref.roipool3d_cpu(xyz[0], boxes, feat, S)
    flags = ref.pts_in_boxes3d_cpu(xyz[0], boxes)
    np.savez_compressed(os.path.join(HERE, "roipool3d_ref.npz"), xyz=xyz, boxes=boxes[None], feat=feat[None], S=S,
                        pooled_pts=pp, pooled_feat=pf, empty=pe, flags=flags.astype(np.uint8))
    # ---- PointNet++ ops: oracle (unpinned upstream)
    pts = unit_cloud(2, 1024, seed=11)
    fidx = cpu.fps(pts, 128)
    new_xyz = np.stack([pts[b][fidx[b]] for b in range(2)])
    bq = cpu.ball_query(0.2, 16, pts, new_xyz)
    d2, i3 = cpu.three_nn(pts[:, :300], new_xyz)
    np.savez_compressed(os.path.join(HERE, "pointnet2_oracle.npz"), xyz=pts, fps_idx=fidx, ball_idx=bq,
                        nn_dist2=d2, nn_idx=i3, nn_w=cpu.three_weights(d2))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
