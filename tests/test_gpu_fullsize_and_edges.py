"""GPU: size-independent properties at the BASELINE metric's full shapes (bs32 x 16384 points, where the CPU oracle
would take minutes), and degenerate / empty inputs through the public op surface."""
import numpy as np
import pytest
import torch

from util import kitti_cloud

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def sa1(dev):
    from pointrcnn_amd import ops
    xyz = T(kitti_cloud(32, 16384), dev)
    fidx = ops.furthest_point_sample(xyz, 4096)
    return xyz, ops.gather_rows(xyz, fidx), fidx


def test_ball_query_full_size_properties(dev, sa1):
    """SA1 of the benchmark (32 x 4096 centroids x 16384 candidates, radii .1/.5): every returned index is inside
    the ball, listed in strictly ascending order up to the hit count, padded with the first hit; counts agree with
    an independent torch count on sampled centroids; centroid itself is always the first hit"""
    from pointrcnn_amd import ops
    xyz, new_xyz, fidx = sa1
    for (r, ns), idx in zip(((0.1, 16), (0.5, 32)), ops.ball_query2(0.1, 16, 0.5, 32, xyz, new_xyz)):
        idx = idx.long()
        pts = torch.gather(xyz.unsqueeze(1).expand(-1, 4096, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))
        d = pts - new_xyz.unsqueeze(2)
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        assert bool((d2 < np.float32(r) * np.float32(r)).all())
        inc = idx[..., 1:] > idx[..., :-1]
        pad = idx[..., 1:] == idx[..., :1]
        assert bool((inc | pad).all())                     # ascending, then first-hit padding
        cnt = 1 + inc.sum(-1)                              # hits = strictly increasing run
        assert bool((inc.cumsum(-1) == torch.arange(1, ns, device=dev)).logical_or(~inc).all())
        # sampled exact counts against a dense torch evaluation
        for b in (0, 17, 31):
            dd = new_xyz[b, :256].unsqueeze(1) - xyz[b].unsqueeze(0)
            full = ((dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2] < np.float32(r) * np.float32(r)).sum(1)
            assert torch.equal(cnt[b, :256], torch.clamp(full, max=ns))
        # the centroid is a point of the cloud: it must be among the hits, and every hit before it has a lower index
        assert bool(((idx == fidx.long().unsqueeze(-1)).any(-1)).all())


def test_three_nn_full_size_properties(dev, sa1):
    """FP0 of the benchmark (32 x 16384 unknown x 4096 known): distances ascending, equal to the canonical formula
    at the returned indices, weights sum to 1, and the nearest known point of a known point is itself (d=0)"""
    from pointrcnn_amd import ops
    xyz, new_xyz, fidx = sa1
    d2, idx, w = ops.three_nn(xyz, new_xyz, want_weight=True)
    assert bool((d2[..., 0] <= d2[..., 1]).all() and (d2[..., 1] <= d2[..., 2]).all())
    nb = torch.gather(new_xyz.unsqueeze(1).expand(-1, 16384, -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
    d = xyz.unsqueeze(2) - nb
    assert torch.equal((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2], d2)
    torch.testing.assert_close(w.sum(-1), torch.ones_like(w[..., 0]), atol=2e-6, rtol=0)
    self_d = torch.gather(d2[..., 0], 1, fidx.long())
    assert bool((self_d == 0).all())


def test_rpn_full_batch_is_batch_invariant(dev):
    """frames are independent (the multi-GPU sharding argument): frame k of a bs32 step equals
    the same frame run alone, bit for bit"""
    from pointrcnn_amd import rpn
    torch.manual_seed(9)
    model = rpn.randomize_bn_stats(rpn.RPN()).to(dev).eval()
    pts = rpn.synthetic_clouds(32, 16384, device=dev)
    with torch.no_grad():
        full = model({"pts_input": pts})
        for k in (0, 13, 31):
            one = model({"pts_input": pts[k:k + 1].contiguous()})
            for key in ("rpn_cls", "rpn_reg"):
                assert torch.equal(full[key][k], one[key][0]), (k, key)


def test_empty_and_degenerate_inputs(dev, cpu):
    from pointrcnn_amd import ops
    z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
    assert ops.furthest_point_sample(z(0, 16, 3), 4).shape == (0, 4)
    assert ops.ball_query(0.1, 4, z(2, 10, 3), z(2, 0, 3)).shape == (2, 0, 4)
    assert ops.gather(z(2, 0, 10), torch.zeros(2, 3, dtype=torch.int32, device=dev)).shape == (2, 0, 3)
    assert ops.group(z(1, 4, 10), torch.zeros(1, 0, 5, dtype=torch.int32, device=dev)).shape == (1, 4, 0, 5)
    d2, idx = ops.three_nn(z(1, 0, 3), z(1, 5, 3))
    assert d2.shape == (1, 0, 3)
    pooled, empty = ops.roipool3d(z(1, 8, 3), z(1, 0, 7), z(1, 8, 2), 4)
    assert pooled.shape == (1, 0, 4, 5) and empty.shape == (1, 0)
    assert ops.boxes_iou_bev(z(0, 5), z(3, 5)).shape == (0, 3)
    # nsample = 1 / npoint = 1 / single candidate
    xyz = T(np.random.default_rng(0).random((2, 50, 3), dtype=np.float32), dev)
    assert np.array_equal(ops.furthest_point_sample(xyz, 1).cpu().numpy(), np.zeros((2, 1), np.int32))
    got = ops.ball_query(0.3, 1, xyz, xyz[:, :7].contiguous()).cpu().numpy()
    assert np.array_equal(got, cpu.ball_query(0.3, 1, xyz.cpu().numpy(), xyz[:, :7].cpu().numpy()))
    one = xyz[:, :1].contiguous()
    d2, idx = ops.three_nn(xyz, one)
    rd2, ridx = cpu.three_nn(xyz.cpu().numpy(), one.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)     # inf padding for m < 3
    # zero-area and zero-size boxes in NMS / IoU: no NaN, identical to the oracle
    boxes = np.array([[0, 0, 0, 0, 0.3], [0, 0, 2, 2, 0.0], [1, 1, 1, 3, 1.0], [0, 0, 2, 2, 0.0]], np.float32)
    iou = ops.boxes_iou_bev(T(boxes, dev), T(boxes, dev)).cpu().numpy()
    assert np.array_equal(iou, cpu.boxes_iou_bev(boxes, boxes)) and np.isfinite(iou).all()
    keep, num = ops.nms_sorted(T(boxes, dev), 0.5)
    assert np.array_equal(keep[: int(num.item())].cpu().numpy(), cpu.nms(boxes, 0.5, "rotated"))
