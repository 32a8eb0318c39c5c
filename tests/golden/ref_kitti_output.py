"""The reference's own save_kitti_format (tools/eval_rcnn.py:69-94) run in the build container: the function's source is compiled
on its own (the module imports tensorboardX, tqdm, ... at the top, which this environment lacks), with the reference's kitti_utils
and Calibration underneath.  Used to generate tests/golden/kitti_output_ref.npz and by the live test."""
import ast
import os
import tempfile

import numpy as np

import ref_net


def available():
    return ref_net.available()


def cases(seed, n=40):
    """seeded boxes in front of the camera (a few behind / far to the side: clipped, some dropped), scores, a KITTI-like P2"""
    r = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = r.uniform(-25, 25, n)
    b[:, 1] = r.uniform(0.8, 2.2, n)
    b[:, 2] = r.uniform(2.0, 70, n)
    b[:, 3:6] = r.uniform(0.8, 1.3, (n, 3)) * [1.52563191462, 1.62856739989, 3.88311640418]
    b[:, 6] = r.uniform(-np.pi, np.pi, n)
    b[:4, 2] = r.uniform(1.2, 3.0, 4)                     # very close: huge image boxes (the 0.8 filter)
    b[4:6, 0] = [0.0, -0.0]                               # beta = +-pi/2 exactly
    scores = r.normal(0, 3, n).astype(np.float32)
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]], np.float32)
    return b, scores, P2, (375, 1242)


def reference_lines(boxes, scores, P2, img_shape, sample_id=7):
    ns = ref_net.load()
    src = open(os.path.join(ref_net.REFERENCE, "tools", "eval_rcnn.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "save_kitti_format")
    env = {"np": np, "os": os, "kitti_utils": ns.kitti_utils, "cfg": ns.cfg}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "eval_rcnn.py", "exec"), env)
    import lib.utils.calibration as calibration
    calib = calibration.Calibration.__new__(calibration.Calibration)
    calib.P2 = P2
    with tempfile.TemporaryDirectory() as d:
        env["save_kitti_format"](sample_id, calib, boxes, d, scores, img_shape)
        return open(os.path.join(d, "%06d.txt" % sample_id)).read().splitlines()


def reference_feature_roundtrip(arrays, sample_id=42, use_seg_score=False):
    """the reference's own save_rpn_features (eval_rcnn.py:97-110) writes, its own KittiRCNNDataset.get_rpn_features
    (kitti_rcnn_dataset.py:139-150) reads: -> (sorted file names, the four arrays it returns)"""
    ns = ref_net.load()
    src = open(os.path.join(ref_net.REFERENCE, "tools", "eval_rcnn.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "save_rpn_features")
    env = {"np": np, "os": os}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "eval_rcnn.py", "exec"), env)
    import lib.datasets.kitti_rcnn_dataset as d
    keep = ns.cfg.RCNN.USE_SEG_SCORE
    ns.cfg.RCNN.USE_SEG_SCORE = use_seg_score
    try:
        with tempfile.TemporaryDirectory() as t:
            env["save_rpn_features"](arrays["seg"], arrays["raw"], arrays["pts_features"], arrays["xyz"], arrays["features"], t, sample_id)
            return sorted(os.listdir(t)), d.KittiRCNNDataset.get_rpn_features(t, sample_id)
    finally:
        ns.cfg.RCNN.USE_SEG_SCORE = keep


def make_golden():
    here = os.path.dirname(os.path.abspath(__file__))
    g = {}
    for seed in (1, 2):
        b, s, P2, shape = cases(seed)
        g["lines%d" % seed] = np.array(reference_lines(b, s, P2, shape))
    np.savez_compressed(os.path.join(here, "kitti_output_ref.npz"), **g)
    print({k: len(v) for k, v in g.items()}, g["lines1"][:2])


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    make_golden()
