"""shared by the CPU and GPU KITTI-evaluator tests"""
import os
import sys

import numpy as np

from util import GOLDEN

sys.path.insert(0, GOLDEN)
from make_golden import annos_crc, kitti_eval_inputs  # noqa: E402

MIN_OVERLAPS = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.5, 0.25, 0.25]] * 3)], 0)


def golden():
    return np.load(os.path.join(GOLDEN, "kitti_eval_ref.npz"))


def write_kitti_txt(annos, folder, with_score):
    """annotation dicts -> KITTI label / detection files (the format tools/eval_rcnn.py:69-94 save_kitti_format writes)"""
    os.makedirs(folder, exist_ok=True)
    for i, a in enumerate(annos):
        with open(os.path.join(folder, "%06d.txt" % i), "w") as f:
            for k in range(len(a["name"])):
                l, h, w = a["dimensions"][k]
                vals = [a["truncated"][k], int(a["occluded"][k]), a["alpha"][k], *a["bbox"][k], h, w, l, *a["location"][k], a["rotation_y"][k]]
                line = a["name"][k] + " " + " ".join(repr(float(v)) if not isinstance(v, int) else str(v) for v in vals)
                if with_score:
                    line += " " + repr(float(a["score"][k]))
                f.write(line + "\n")


def check_against_golden(kitti_eval, backend, g):
    gt, dt = kitti_eval_inputs()
    assert annos_crc(gt) == g["crc"][0] and annos_crc(dt) == g["crc"][1], "seeded annotations drifted from the fixture's"
    res, d = kitti_eval.get_official_eval_result(gt, dt, [0, 1, 2], backend=backend)
    assert res == str(g["official_str"])
    assert sorted(d) == list(g["official_keys"])
    np.testing.assert_allclose([d[k] for k in sorted(d)], g["official_vals"], rtol=0, atol=1e-9)
    assert kitti_eval.get_official_eval_result(gt, dt, 0, backend=backend)[0] == str(g["official_car_str"])
    assert kitti_eval.get_coco_eval_result(gt, dt, [0, 1], backend=backend) == str(g["coco_str"])
    for metric in (0, 1, 2):
        _, flat, _ = kitti_eval.calculate_iou_partly(dt, gt, metric, backend)
        np.testing.assert_allclose(flat, g["ov%d" % metric], rtol=0, atol=1e-6)
        r = kitti_eval.eval_class(gt, dt, [0, 1, 2], [0, 1, 2], metric, MIN_OVERLAPS, compute_aos=(metric == 0), backend=backend)
        for k in ("recall", "precision", "orientation"):
            np.testing.assert_allclose(r[k], g["m%d_%s" % (metric, k)], rtol=0, atol=1e-9, equal_nan=True)
