"""Detection writer (pointrcnn_amd/kitti_output.py: detections -> KITTI result files, the format between the detector and the AP
evaluator) against the lines the REFERENCE'S OWN save_kitti_format (tools/eval_rcnn.py:69-94) prints for the same boxes --
committed fixture (tests/golden/ref_kitti_output.py) and live when /root/reference is present -- and through the evaluator's
parser (written file -> annotation dict -> the boxes back)."""
import os
import sys

import numpy as np
import pytest

from util import GOLDEN

sys.path.insert(0, GOLDEN)
import ref_kitti_output as rko  # noqa: E402
from pointrcnn_amd import kitti_eval, kitti_output  # noqa: E402


class _Calib:
    def __init__(self, P2):
        self.P2 = P2


@pytest.mark.parametrize("seed", [1, 2])
def test_lines_equal_the_references_own_writer(seed):
    g = np.load(os.path.join(GOLDEN, "kitti_output_ref.npz"))
    b, s, P2, shape = rko.cases(seed)
    got = kitti_output.kitti_lines(b, s, P2, shape)
    want = g["lines%d" % seed].tolist()
    assert len(want) < len(b)                              # the fixture exercises the 0.8-of-the-image filter
    assert got == want


@pytest.mark.skipif(not rko.available(), reason="reference checkout absent")
def test_lines_equal_the_reference_live():
    b, s, P2, shape = rko.cases(5, n=64)
    assert kitti_output.kitti_lines(b, s, P2, shape) == rko.reference_lines(b, s, P2, shape)


def test_written_files_parse_back_through_the_evaluators_reader(tmp_path):
    b, s, P2, shape = rko.cases(3)
    pred = np.stack([b, b[::-1].copy()])
    raw = np.stack([s, s[::-1].copy()])
    keep = np.full((2, len(b)), -1, np.int32)
    keep[0, :10] = np.arange(10, 20)
    keep[1, :3] = [5, 1, 7]
    num = np.array([10, 3], np.int32)
    lines = kitti_output.write_detections(pred, raw, keep, num, [11, 12], [_Calib(P2), _Calib(P2)], [shape, shape], str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["000011.txt", "000012.txt"]
    for fi, (f, (bx, sc, k)) in enumerate(zip(("000011.txt", "000012.txt"), ((pred[0], raw[0], keep[0, :10]), (pred[1], raw[1], keep[1, :3])))):
        kept = kitti_output.kitti_lines(bx[k], sc[k], P2, shape)
        assert lines[fi] == kept
        if not kept:
            continue
        anno = kitti_eval.get_label_anno(os.path.join(tmp_path, f))
        assert len(anno["name"]) == len(kept) and (anno["name"] == "Car").all()
        want = np.array([[float(v) for v in ln.split()[8:15]] for ln in kept])          # h w l x y z ry as printed
        got = np.column_stack([anno["dimensions"][:, [1, 2, 0]], anno["location"], anno["rotation_y"]])
        np.testing.assert_allclose(got, want, atol=1e-9)
    # a frame without detections: an empty file, as the reference writes
    assert kitti_output.save_kitti_format(13, _Calib(P2), np.zeros((0, 7), np.float32), str(tmp_path), np.zeros((0,), np.float32), shape) == []
    assert open(os.path.join(tmp_path, "000013.txt")).read() == ""


def _feature_arrays(seed=0, n=300, c=16):
    r = np.random.default_rng(seed)
    return {"seg": (r.random(n) > 0.6).astype(np.float32), "raw": r.normal(0, 2, n).astype(np.float32),
            "pts_features": r.random((n, 1)).astype(np.float32), "xyz": r.normal(0, 10, (n, 3)).astype(np.float32),
            "features": r.normal(size=(n, c)).astype(np.float32)}


@pytest.mark.parametrize("use_seg_score", [False, True])
def test_rpn_feature_dumps_round_trip(tmp_path, use_seg_score):
    """eval_rcnn.py:97-110 file set (names, contents) and kitti_rcnn_dataset.py:139-150 reader"""
    a = _feature_arrays()
    kitti_output.save_rpn_features(a["seg"], a["raw"], a["pts_features"], a["xyz"], a["features"], str(tmp_path), 42)
    assert sorted(os.listdir(tmp_path)) == ["000042.npy", "000042_intensity.npy", "000042_rawscore.npy", "000042_seg.npy", "000042_xyz.npy"]
    xyz, feat, inten, seg = kitti_output.get_rpn_features(str(tmp_path), 42, use_seg_score)
    assert np.array_equal(xyz, a["xyz"]) and np.array_equal(feat, a["features"]) and np.array_equal(inten, a["pts_features"][:, 0])
    want = 1.0 / (1.0 + np.exp(-a["raw"].astype(np.float64))) if use_seg_score else a["seg"]
    np.testing.assert_allclose(seg, want, rtol=2e-7)
    if rko.available():                                    # the reference's own writer + reader on the same arrays
        names, ref = rko.reference_feature_roundtrip(a, 42, use_seg_score)
        assert names == sorted(os.listdir(tmp_path))
        for g, w in zip((xyz, feat, inten, seg), ref):
            assert np.array_equal(g, w)
