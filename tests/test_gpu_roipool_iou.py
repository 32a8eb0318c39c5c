"""GPU parity: roipool3d and iou3d through the C ABI vs the CPU oracle (canonical trig, bit-exact),
vs golden fixtures generated from the reference's own code (tests/golden/make_golden.py), and through
the drop-in `roipool3d_cuda` / `iou3d_cuda` modules."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, enlarge, kitti_cloud, rand_bev, rand_boxes3d

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ roipool3d
@pytest.mark.parametrize("B,N,M,C,S", [(2, 4096, 40, 16, 64), (1, 16384, 100, 130, 512), (2, 1000, 7, 0, 32),
                                       (1, 333, 5, 3, 600), (3, 257, 9, 5, 16)])
def test_roipool3d_matches_oracle(dev, cpu, B, N, M, C, S):
    from pointrcnn_amd import ops
    r = np.random.default_rng(N + M)
    xyz = kitti_cloud(B, N, seed=N)
    boxes = np.stack([enlarge(rand_boxes3d(xyz[b], M, seed=b + 1), 1.0) for b in range(B)])
    boxes[:, -1, 0] += 500.0                                  # one box far away from every point: empty
    feat = r.normal(size=(B, N, C)).astype(np.float32)
    want, wempty = cpu.roipool3d(xyz, boxes, feat, S)
    got, gempty = ops.roipool3d(T(xyz, dev), T(boxes, dev), T(feat, dev), S)
    assert np.array_equal(gempty.cpu().numpy(), wempty) and wempty[:, -1].all()
    assert np.array_equal(got.cpu().numpy(), want)            # gathered copies: bit equal


def test_roipool3d_dense_box_truncates_at_S(dev, cpu):
    """more in-box points than S: the FIRST S in index order are kept (roipool3d.cpp:150-160)"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(0)
    N, S = 5000, 128
    xyz = (r.random((1, N, 3)).astype(np.float32) - 0.5) * np.array([3.0, 1.0, 3.0], np.float32)
    box = np.array([[[0, 0.8, 0, 1.6, 4.0, 4.0, 0.3]]], np.float32)
    feat = np.arange(N, dtype=np.float32).reshape(1, N, 1)
    want, _ = cpu.roipool3d(xyz, box, feat, S)
    got, _ = ops.roipool3d(T(xyz, dev), T(box, dev), T(feat, dev), S)
    assert np.array_equal(got.cpu().numpy(), want)
    ids = got.cpu().numpy()[0, 0, :, 3]
    assert (np.diff(ids) > 0).all()                           # strictly ascending point indices, no wrap


def test_roipool3d_binned_selection_is_the_linear_scan(dev, cpu, monkeypatch):
    """prcnn_roipool3d_ws (points bucketed into x-z bins, a box tests the bins under its footprint) == the linear scan == the
    oracle, bit for bit, on inputs built to break a culled search: boxes outside the cloud's extent, boxes covering all of it,
    zero / negative / infinite / NaN boxes, NaN and infinite points, a frame whose points all coincide, N not a multiple of
    32, boxes hugging bin borders, more than S points in a box"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(7)
    B, N, M, C, S = 4, 3001, 48, 5, 64
    xyz = kitti_cloud(B, N, seed=3)
    xyz[1] = xyz[1, :1]                                        # every point of frame 1 at the same place
    xyz[2, ::97] = np.nan
    xyz[2, 5::89, 0] = np.inf
    xyz[2, 7::83, 2] = -np.inf
    boxes = np.stack([enlarge(rand_boxes3d(xyz[0], M, seed=b + 11), 1.0) for b in range(B)])
    boxes[1, :, :3] = xyz[1, 0] + r.normal(size=(M, 3)).astype(np.float32) * 0.3
    boxes[:, 0, 0] += 700.0                                    # far outside
    boxes[:, 1, 3:6] = [3.0, 19.9, 19.9]                       # covers everything the 10 m gate lets through
    boxes[:, 2, 3:6] = 0.0
    boxes[:, 3, 4] = -1.0
    boxes[:, 4, 5] = np.inf
    boxes[:, 5, 6] = np.nan
    boxes[:, 6, 0] = np.nan
    boxes[:, 7, 6] = np.inf
    boxes[:, 8, 4:6] = [40.0, 40.0]
    lo, hi = np.nanmin(np.where(np.isfinite(xyz[0]), xyz[0], np.nan), 0), np.nanmax(np.where(np.isfinite(xyz[0]), xyz[0], np.nan), 0)
    for i in range(9, 20):                                     # centres on bin borders of the 64 x 64 grid, axis-aligned and rotated
        boxes[0, i, 0] = lo[0] + (hi[0] - lo[0]) * (i - 4) / 64.0
        boxes[0, i, 2] = lo[2] + (hi[2] - lo[2]) * (2 * i) / 64.0
        boxes[0, i, 6] = [0.0, np.pi / 2, np.pi, -np.pi / 2, 0.3][i % 5]
    feat = r.normal(size=(B, N, C)).astype(np.float32)
    want, wempty = cpu.roipool3d(xyz, boxes, feat, S)
    a = [t.cpu().numpy() for t in ops.roipool3d(T(xyz, dev), T(boxes, dev), T(feat, dev), S)]
    monkeypatch.setattr(ops, "ROIPOOL_BINS", False)
    b = [t.cpu().numpy() for t in ops.roipool3d(T(xyz, dev), T(boxes, dev), T(feat, dev), S)]
    monkeypatch.setattr(ops, "ROIPOOL_BINS", True)
    assert ops._roipool_work(B, N, M, dev)[1] > 0 and ops._roipool_work(B, N, 2, dev)[1] == 0
    for u, v, w in zip(a, b, (want, wempty)):
        assert np.array_equal(u, v, equal_nan=True) and np.array_equal(u, w, equal_nan=True)
    assert 0 < wempty.sum() < wempty.size and (np.diff(a[0][0, 1, :, 0]) != 0).any()
    # canonical (fused RCNN input) form: same selection
    ex = [T(r.random((B, N)).astype(np.float32), dev), T(r.random((B, N)).astype(np.float32), dev)]
    rois = T(boxes, dev)
    c1 = ops.roipool3d_canonical(T(xyz, dev), T(boxes, dev), rois, ex, T(feat, dev), S, want_distinct=True)
    monkeypatch.setattr(ops, "ROIPOOL_BINS", False)
    c0 = ops.roipool3d_canonical(T(xyz, dev), T(boxes, dev), rois, ex, T(feat, dev), S, want_distinct=True)
    d = c0[3].cpu().numpy().reshape(-1)
    assert np.array_equal(c1[3].cpu().numpy().reshape(-1), d) and np.array_equal(c1[2].cpu().numpy(), c0[2].cpu().numpy())
    assert np.array_equal(c1[0].cpu().numpy(), c0[0].cpu().numpy(), equal_nan=True)
    f1, f0 = c1[1].cpu().numpy().reshape(B * M, S, C), c0[1].cpu().numpy().reshape(B * M, S, C)
    for i in range(B * M):                                     # feature rows of the wrap-copies are not written (distinct)
        assert np.array_equal(f1[i, :d[i]], f0[i, :d[i]], equal_nan=True)


def test_roipool3d_golden_from_reference(dev):
    """fixture produced by the reference's own roipool3d.cpp CPU code (oracle/_ref)"""
    from pointrcnn_amd import ops
    g = np.load(os.path.join(GOLDEN, "roipool3d_ref.npz"))
    got, gempty = ops.roipool3d(T(g["xyz"], dev), T(g["boxes"], dev), T(g["feat"], dev), int(g["S"]))
    got = got.cpu().numpy()
    assert np.array_equal(gempty.cpu().numpy()[0], g["empty"].astype(np.int32))
    assert np.array_equal(got[0, :, :, :3], g["pooled_pts"]) and np.array_equal(got[0, :, :, 3:], g["pooled_feat"])


def test_roipool3d_dropin_module(dev, cpu):
    """roipool3d_cuda.{forward, pts_in_boxes3d_cpu, roipool3d_cpu} with the reference's calling convention
    (roipool3d_utils.py:21-26, 40-41, 65-68)"""
    import roipool3d_cuda
    r = np.random.default_rng(4)
    N, M, C, S = 3000, 12, 6, 48
    xyz = kitti_cloud(1, N, seed=77)
    boxes = enlarge(rand_boxes3d(xyz[0], M, seed=5), 1.0)[None]
    feat = r.normal(size=(1, N, C)).astype(np.float32)
    pooled = torch.cuda.FloatTensor(torch.Size((1, M, S, 3 + C))).zero_()
    empty = torch.cuda.IntTensor(torch.Size((1, M))).zero_()
    assert roipool3d_cuda.forward(T(xyz, dev), T(boxes, dev), T(feat, dev), pooled, empty) == 1
    want, wempty = cpu.roipool3d(xyz, boxes, feat, S)
    assert np.array_equal(pooled.cpu().numpy(), want) and np.array_equal(empty.cpu().numpy(), wempty)
    # forward_slow: the independent second implementation (flag matrix + tensor selection) gives the same bits -- boxes with more
    # than S points, fewer (wrap-duplication), and none (a box moved out of the cloud), two frames
    xyz2 = np.concatenate([xyz, kitti_cloud(1, N, seed=78)])
    boxes2 = np.concatenate([boxes, enlarge(rand_boxes3d(xyz2[1], M, seed=6), 1.0)[None]])
    boxes2[1, 3, 0] += 500.0
    boxes2[0, 5, 3:6] *= 3.0
    feat2 = r.normal(size=(2, N, C)).astype(np.float32)
    outs = []
    for fn in (roipool3d_cuda.forward, roipool3d_cuda.forward_slow):
        p2 = torch.cuda.FloatTensor(torch.Size((2, M, S, 3 + C))).zero_()
        e2 = torch.cuda.IntTensor(torch.Size((2, M))).zero_()
        assert fn(T(xyz2, dev), T(boxes2, dev), T(feat2, dev), p2, e2) == 1
        outs.append((p2, e2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(outs[0][1].sum()) >= 1 and np.array_equal(outs[1][0].cpu().numpy(), cpu.roipool3d(xyz2, boxes2, feat2, S)[0])
    flags = torch.LongTensor(torch.Size((M, N)))
    roipool3d_cuda.pts_in_boxes3d_cpu(flags, torch.from_numpy(xyz[0]), torch.from_numpy(boxes[0]))
    assert np.array_equal(flags.numpy(), cpu.pts_in_boxes3d(xyz[0], boxes[0]))
    pp = torch.zeros(M, S, 3); pf = torch.zeros(M, S, C); pe = torch.zeros(M, dtype=torch.int64)
    roipool3d_cuda.roipool3d_cpu(torch.from_numpy(xyz[0]), torch.from_numpy(boxes[0]), torch.from_numpy(feat[0]), pp, pf, pe)
    assert np.array_equal(pp.numpy(), want[0, :, :, :3]) and np.array_equal(pf.numpy(), want[0, :, :, 3:])
    assert np.array_equal(pe.numpy(), wempty[0])
    with pytest.raises(RuntimeError):
        roipool3d_cuda.forward(torch.from_numpy(xyz), T(boxes, dev), T(feat, dev), pooled, empty)   # CPU input


# ------------------------------------------------------------------ iou3d
@pytest.mark.parametrize("na,nb,spread", [(300, 200, 6.0), (17, 33, 2.0), (1, 1, 1.0), (64, 65, 3.0)])
def test_overlap_and_iou_match_oracle_bitexact(dev, cpu, na, nb, spread):
    from pointrcnn_amd import ops
    a, b = rand_bev(na, spread, seed=na), rand_bev(nb, spread, seed=nb + 100)
    assert np.array_equal(ops.boxes_overlap_bev(T(a, dev), T(b, dev)).cpu().numpy(), cpu.boxes_overlap_bev(a, b))
    assert np.array_equal(ops.boxes_iou_bev(T(a, dev), T(b, dev)).cpu().numpy(), cpu.boxes_iou_bev(a, b))


def test_overlap_degenerate_pairs(dev, cpu):
    """identical boxes, shared edges, containment, axis-aligned (angle 0, pi/2): the clipping corner cases"""
    from pointrcnn_amd import ops
    base = np.array([[0, 0, 2, 4, 0.0], [0, 0, 2, 4, 0.0], [2, 0, 4, 4, 0.0], [0.5, 1, 1.5, 3, 0.0],
                     [0, 0, 2, 4, np.pi / 2], [0, 0, 2, 4, 0.7], [0, 0, 2, 4, 0.7], [10, 10, 12, 14, 0.1],
                     [0, 0, 2, 4, np.pi], [0, 0, 2, 4, -0.7]], np.float32)
    got = ops.boxes_iou_bev(T(base, dev), T(base, dev)).cpu().numpy()
    assert np.array_equal(got, cpu.boxes_iou_bev(base, base))
    np.testing.assert_allclose(np.diag(got), 1.0, atol=1e-5)


@pytest.mark.parametrize("kind", ["rotated", "normal"])
@pytest.mark.parametrize("N,thr", [(1500, 0.1), (1500, 0.5), (1500, 0.8), (65, 0.3), (64, 0.3), (1, 0.5), (130, 0.0)])
def test_nms_keep_matches_oracle(dev, cpu, kind, N, thr):
    from pointrcnn_amd import ops
    boxes = rand_bev(N, 8.0, seed=N)
    keep, num = ops.nms_sorted(T(boxes, dev), thr, rotated=(kind == "rotated"))
    got = keep[: int(num.item())].cpu().numpy()
    assert np.array_equal(got, cpu.nms(boxes, thr, kind))


def test_nms_empty_and_duplicates(dev, cpu):
    from pointrcnn_amd import ops
    keep, num = ops.nms_sorted(torch.zeros((0, 5), device=dev), 0.5)
    assert int(num.item()) == 0
    dup = np.repeat(rand_bev(40, 5.0, seed=2), 3, axis=0)            # every box three times: duplicates suppressed
    for kind in ("rotated", "normal"):
        keep, num = ops.nms_sorted(T(dup, dev), 0.7, rotated=(kind == "rotated"))
        got = keep[: int(num.item())].cpu().numpy()
        assert np.array_equal(got, cpu.nms(dup, 0.7, kind))
        assert (got % 3 == 0).all()


def test_nms_rpn_scale_normal(dev, cpu):
    """the default RPN path: axis-aligned NMS on 6300 score-sorted proposals, thr 0.8 (default.yaml:61,165)"""
    from pointrcnn_amd import ops
    boxes = rand_bev(6300, 30.0, seed=63)
    keep, num = ops.nms_sorted(T(boxes, dev), 0.8, rotated=False)
    got = keep[: int(num.item())].cpu().numpy()
    assert np.array_equal(got, cpu.nms(boxes, 0.8, "normal"))


def test_iou3d_golden_from_reference(dev):
    """fixtures produced by the reference's own iou3d sources run on the host (oracle/_ref): overlaps, IoUs and NMS keep sets
    are bit-identical (the kernels evaluate the reference's libm arithmetic, csrc/ref_trig.h)"""
    from pointrcnn_amd import ops
    g = np.load(os.path.join(GOLDEN, "iou3d_ref.npz"))
    ov = ops.boxes_overlap_bev(T(g["a"], dev), T(g["b"], dev)).cpu().numpy()
    assert np.array_equal(ov, g["overlap"])
    iou = ops.boxes_iou_bev(T(g["a"], dev), T(g["b"], dev)).cpu().numpy()
    assert np.array_equal(iou, g["iou"])
    for kind in ("rotated", "normal"):
        for thr in (0.1, 0.5, 0.8):
            keep, num = ops.nms_sorted(T(g["nms_boxes"], dev), thr, rotated=(kind == "rotated"))
            assert np.array_equal(keep[: int(num.item())].cpu().numpy(), g["keep_%s_%s" % (kind, thr)])


def test_iou3d_dropin_module(dev, cpu):
    """iou3d_cuda.* with the reference's calling convention (iou3d_utils.py:14-16,32-33,64-70)"""
    import iou3d_cuda
    boxes = rand_bev(500, 6.0, seed=11)
    scores = np.random.default_rng(3).random(500).astype(np.float32)
    tb, ts = T(boxes, dev), T(scores, dev)
    ans = torch.cuda.FloatTensor(torch.Size((500, 500))).zero_()
    assert iou3d_cuda.boxes_iou_bev_gpu(tb.contiguous(), tb.contiguous(), ans) == 1
    assert np.array_equal(ans.cpu().numpy(), cpu.boxes_iou_bev(boxes, boxes))
    ans.zero_()
    iou3d_cuda.boxes_overlap_bev_gpu(tb, tb, ans)
    assert np.array_equal(ans.cpu().numpy(), cpu.boxes_overlap_bev(boxes, boxes))
    # iou3d_utils.nms_gpu, verbatim calling sequence
    order = ts.sort(0, descending=True)[1]
    sb = tb[order].contiguous()
    for fn, kind in ((iou3d_cuda.nms_gpu, "rotated"), (iou3d_cuda.nms_normal_gpu, "normal")):
        keep = torch.LongTensor(sb.size(0))
        num_out = fn(sb, keep, 0.4)
        picked = order[keep[:num_out].cuda()].contiguous().cpu().numpy()
        want = order.cpu().numpy()[cpu.nms(sb.cpu().numpy(), 0.4, kind)]
        assert np.array_equal(picked, want)
    with pytest.raises(RuntimeError):
        iou3d_cuda.nms_gpu(torch.from_numpy(boxes), torch.LongTensor(500), 0.4)      # CPU boxes rejected


def test_nms_more_than_256_blocks_and_rotated_at_rpn_scale(dev, cpu):
    """the sweep folds a block's kept rows into the later blocks' suppression words one word per thread, 256 threads: more than 256
    later blocks (N > 16 384) take the kernel's second loop; and the rotated mask kernel's pair lists at the RPN's scale (6 300
    scattered boxes: a dozen listed pairs per 64 x 64 tile, some tiles none, the diagonal tiles their upper triangle only)"""
    from pointrcnn_amd import ops
    boxes = rand_bev(17000, 60.0, seed=170)
    keep, num = ops.nms_sorted(T(boxes, dev), 0.5, rotated=False)
    assert np.array_equal(keep[: int(num.item())].cpu().numpy(), cpu.nms(boxes, 0.5, "normal"))
    boxes = rand_bev(6300, 30.0, seed=64)
    for thr in (0.8, 0.05):
        keep, num = ops.nms_sorted(T(boxes, dev), thr, rotated=True)
        assert np.array_equal(keep[: int(num.item())].cpu().numpy(), cpu.nms(boxes, thr, "rotated"))


def _clustered_bev(n, clusters, spread, seed):
    """n score-ordered boxes drawn around `clusters` centres: what a trained RPN hands the NMS (most boxes suppressed by a few kept ones,
    every 64-box block full of in-block suppressions) -- the dense case of the sweep's resolver"""
    r = np.random.default_rng(seed)
    ctr = r.uniform(0, 40, (clusters, 2))
    own = r.integers(0, clusters, n)
    c = ctr[own] + r.normal(0, spread, (n, 2))
    s = r.uniform(0.8, 1.2, (n, 2)) * [0.8, 2.0]
    return np.concatenate([c - s, c + s, r.uniform(-3.14, 3.14, (n, 1))], 1).astype(np.float32)


@pytest.mark.parametrize("kind", ["rotated", "normal"])
@pytest.mark.parametrize("N,clusters,thr", [(6300, 25, 0.8), (6300, 25, 0.3), (2700, 5, 0.5), (9000, 200, 0.7), (4097, 1, 0.9), (127, 3, 0.5)])
def test_nms_sweep_dense_suppression_and_far_words(dev, cpu, kind, N, clusters, thr):
    """round 6 sweep (resolver wave + folder waves one block behind): heavily overlapping proposals (the resolver walks many rows per
    block, most blocks keep few or none, whole folder chunks have no kept row), more than 129 blocks (the folders' far-word loop), a
    last block with one row, and the max_keep exit on the same boxes"""
    from pointrcnn_amd import ops
    boxes = _clustered_bev(N, clusters, 0.35, seed=N + clusters)
    want = cpu.nms(boxes, thr, kind)
    keep, num = ops.nms_sorted(T(boxes, dev), thr, rotated=(kind == "rotated"))
    assert np.array_equal(keep[: int(num.item())].cpu().numpy(), want)
    for mk in (1, 7, 64, 65):
        keep, num = ops.nms_sorted(T(boxes, dev), thr, rotated=(kind == "rotated"), max_keep=mk)
        n = int(num.item())
        assert n == min(mk, len(want)) and np.array_equal(keep[:n].cpu().numpy(), want[:n])


def test_nms_dropin_one_sync_path_repeated_calls(dev, cpu):
    """iou3d_cuda.nms_*_gpu deliver through one pinned staging buffer per thread and device: back-to-back calls of different sizes (the
    proposal layer's 6300 / 2700 pattern, proposal_layer.py:100-105) must not see each other's entries"""
    import iou3d_cuda
    for rep, (n, kind) in enumerate([(6300, "normal"), (2700, "normal"), (100, "rotated"), (6300, "rotated"), (1, "normal"), (9001, "normal")]):
        boxes = _clustered_bev(n, 40, 0.5, seed=900 + rep)
        keep = torch.LongTensor(n).fill_(-7)
        fn = iou3d_cuda.nms_gpu if kind == "rotated" else iou3d_cuda.nms_normal_gpu
        num = fn(T(boxes, dev), keep, 0.8)
        want = cpu.nms(boxes, 0.8, kind)
        assert num == len(want) and np.array_equal(keep[:num].numpy(), want)
        assert (keep[num:] == -7).all()
    with pytest.raises(RuntimeError):
        iou3d_cuda.nms_gpu(T(boxes, dev), torch.LongTensor(5), 0.8)                  # keep shorter than the boxes
    with pytest.raises(RuntimeError):
        iou3d_cuda.nms_gpu(T(boxes, dev), torch.zeros(9001, dtype=torch.int32), 0.8)  # keep must be int64


def test_nms_max_keep_prefix(dev, cpu):
    """max_keep stops the sweep early: the kept list is the exact prefix of the full result (what
    proposal_layer.py:112 `keep_idx[:post_top_n]` consumes), for both kinds and across block boundaries"""
    from pointrcnn_amd import ops
    boxes = rand_bev(3000, 12.0, seed=77)
    for kind in ("rotated", "normal"):
        full = cpu.nms(boxes, 0.5, kind)
        for mk in (1, 30, 70, 200, 512):
            keep, num = ops.nms_sorted(T(boxes, dev), 0.5, rotated=(kind == "rotated"), max_keep=mk)
            n = int(num.item())
            assert n == min(mk, len(full))
            assert np.array_equal(keep[:n].cpu().numpy(), full[:n])


def test_roipool3d_canonical_equals_oracle_and_reference_python(dev, cpu):
    """fused concat + roipool3d + canonical transform (rcnn_net.py:127-154) against the oracle
    (bit-exact) and the reference's own Python ops run on the reference's pooled tensor (golden, 1e-5)."""
    import os
    from pointrcnn_amd import ops
    from util import GOLDEN
    g, r = np.load(os.path.join(GOLDEN, "canonical_ref.npz")), np.load(os.path.join(GOLDEN, "roipool3d_ref.npz"))
    xyz, boxes, feat, S = r["xyz"], r["boxes"], r["feat"], int(r["S"])
    B, N, C = feat.shape
    M = boxes.shape[1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # features as trailing channels of a wider buffer, outputs into a wider buffer: strides are honoured
    wide = torch.zeros((B, N, C + 5), device=dev)
    wide[..., 2:2 + C] = t(feat)
    out_buf = torch.full((B * M * S, C + 7), -1.0, device=dev)
    pts, fb, empty = ops.roipool3d_canonical(t(xyz), t(boxes), t(g["rois"]), [t(feat[..., 0]), t(feat[..., 1])],
                                             wide[..., 2:2 + C], S, out_feat=(out_buf, 4))
    want = cpu.canonical_transform(cpu.roipool3d(xyz, boxes, feat, S)[0], g["rois"])
    assert np.array_equal(pts[..., 0:3].cpu().numpy().reshape(B, M, S, 3), want[..., 0:3])
    assert np.array_equal(pts[..., 3:5].cpu().numpy().reshape(B, M, S, 2), want[..., 3:5])          # the two scalar channels
    ob = out_buf.cpu().numpy()
    assert np.array_equal(ob[:, 4:4 + C].reshape(B, M, S, C), want[..., 3:]) and (ob[:, :4] == -1).all() and (ob[:, 4 + C:] == -1).all()
    assert np.array_equal(empty.cpu().numpy(), g["empty"])
    np.testing.assert_allclose(pts[..., 0:3].cpu().numpy().reshape(B, M, S, 3), g["pooled_canonical"][..., 0:3], rtol=0, atol=1e-5)
    assert empty.sum() >= 1                                                  # the fixture holds an empty RoI: -centre, rotated
    # rois=None keeps scene coordinates == plain roipool3d
    pts2, fb2, _ = ops.roipool3d_canonical(t(xyz), t(boxes), None, [], t(feat), S)
    plain = cpu.roipool3d(xyz, boxes, feat, S)[0]
    assert np.array_equal(pts2.cpu().numpy().reshape(B, M, S, 3), plain[..., :3])
    assert np.array_equal(fb2.cpu().numpy().reshape(B, M, S, C), plain[..., 3:])
