"""KITTI evaluator kernels (csrc/kitti_eval.hip) through the C ABI: identical to the CPU oracle and to the golden
vectors from the reference's own Python."""
import numpy as np
import pytest
import torch

from kitti_cases import MIN_OVERLAPS, check_against_golden, golden, kitti_eval_inputs

pytestmark = pytest.mark.gpu


def test_hip_backend_equals_golden_and_oracle(dev):
    import oracle
    from pointrcnn_amd import kitti_eval
    hip, cpu_be = kitti_eval.HipBackend(), oracle.KittiBackend()
    check_against_golden(kitti_eval, hip, golden())
    gt, dt = kitti_eval_inputs()
    for metric in (0, 1, 2):
        a = kitti_eval.calculate_iou_partly(dt, gt, metric, hip)[1]
        b = kitti_eval.calculate_iou_partly(dt, gt, metric, cpu_be)[1]
        assert np.array_equal(a, b), metric                                      # same arithmetic, bit for bit

    class Recorder:                                                              # every statistics() call: HIP == oracle
        def __init__(self):
            self.calls = 0

        def overlaps(self, *a):
            return hip.overlaps(*a)

        def statistics(self, *a):
            r1, m1 = hip.statistics(*a)
            r2, m2 = cpu_be.statistics(*a)
            assert np.array_equal(r1[..., :3], r2[..., :3]) and np.array_equal(m1, m2, equal_nan=True)
            np.testing.assert_allclose(r1[..., 3], r2[..., 3], rtol=1e-13, atol=1e-13)
            self.calls += 1
            return r1, m1

    rec = Recorder()
    kitti_eval.eval_class(gt, dt, [0, 1, 2], [0, 1, 2], 0, MIN_OVERLAPS, compute_aos=True, backend=rec)
    assert rec.calls == 2 * 3 * 3 * 2


def test_rotate_iou_eval_kernel(dev, cpu):
    import oracle
    from pointrcnn_amd import ops
    g = golden()
    be = oracle.KittiBackend()
    b, q = torch.from_numpy(g["riou_boxes"]).to(dev), torch.from_numpy(g["riou_query"]).to(dev)
    for c in (-1, 0, 1, 2):
        got = ops.rotate_iou_eval(b, q, c).cpu().numpy()
        assert np.array_equal(got, be.rotate_iou_eval(g["riou_boxes"], g["riou_query"], c))
        np.testing.assert_allclose(got, g["riou_c%d" % c], rtol=0, atol=2e-6)
    assert ops.rotate_iou_eval(b[:0], q, -1).shape == (0, 30)


def test_full_split_size_runs(dev):
    """3 769 frames (the KITTI val split size) of synthetic annotations: shapes, finiteness, monotone precision envelope"""
    from util import kitti_annos
    from pointrcnn_amd import kitti_eval
    gt = kitti_annos(3769, seed=11)
    dt = kitti_annos(3769, seed=12, with_score=True, gt=gt)
    res, d = kitti_eval.get_official_eval_result(gt, dt, 0)
    assert all(np.isfinite(v) and 0 <= v <= 100 for v in d.values())
    r = kitti_eval.eval_class(gt, dt, [0], [0, 1, 2], 2, np.full((1, 3, 1), 0.7))
    assert (np.diff(r["precision"], axis=-1) <= 1e-12).all()                     # the running max makes it non-increasing
