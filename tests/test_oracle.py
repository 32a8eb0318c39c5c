"""CPU: the oracle (oracle/prcnn_oracle.c) against (1) the committed golden fixtures produced by the reference's
own code, (2) the reference's own sources compiled for the host, live (oracle/_ref), (3) independent slow
numpy / torch restatements ("third opinion") for the PointNet++ ops whose reference source is absent."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, enlarge, kitti_cloud, rand_bev, rand_boxes3d, unit_cloud


# ------------------------------------------------------------------ (1) golden fixtures
def test_golden_iou3d_reference_mode_is_bit_exact(cpu):
    g = np.load(os.path.join(GOLDEN, "iou3d_ref.npz"))
    assert np.array_equal(cpu.boxes_overlap_bev(g["a"], g["b"], 0), g["overlap"])
    assert np.array_equal(cpu.boxes_iou_bev(g["a"], g["b"], 0), g["iou"])
    for kind in ("rotated", "normal"):
        for thr in (0.1, 0.5, 0.8):
            want = g["keep_%s_%s" % (kind, thr)]
            assert np.array_equal(cpu.nms(g["nms_boxes"], thr, kind, 0), want)
            assert np.array_equal(cpu.nms(g["nms_boxes"], thr, kind, 1), want)      # the legacy (round 1-2) kernel arithmetic agrees
            assert np.array_equal(cpu.nms(g["nms_boxes"], thr, kind, 2), want)      # the kernels' arithmetic: the restated libm
    assert np.array_equal(cpu.boxes_overlap_bev(g["a"], g["b"], 2), g["overlap"]) and np.array_equal(cpu.boxes_iou_bev(g["a"], g["b"], 2), g["iou"])


def test_golden_iou3d_canonical_mode_within_tolerance(cpu):
    """trig_mode 1 (the kernels' arithmetic of rounds 1-2; decode / canonical transform still use it) vs the reference's libm"""
    g = np.load(os.path.join(GOLDEN, "iou3d_ref.npz"))
    np.testing.assert_allclose(cpu.boxes_overlap_bev(g["a"], g["b"], 1), g["overlap"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(cpu.boxes_iou_bev(g["a"], g["b"], 1), g["iou"], atol=1e-5, rtol=0)


def test_golden_roipool3d(cpu):
    g = np.load(os.path.join(GOLDEN, "roipool3d_ref.npz"))
    for mode in (0, 1):
        out, empty = cpu.roipool3d(g["xyz"], g["boxes"], g["feat"], int(g["S"]), mode)
        assert np.array_equal(out[0, :, :, :3], g["pooled_pts"]) and np.array_equal(out[0, :, :, 3:], g["pooled_feat"])
        assert np.array_equal(empty[0], g["empty"])
        assert np.array_equal(cpu.pts_in_boxes3d(g["xyz"][0], g["boxes"][0], mode), g["flags"].astype(np.int64))
    assert g["empty"][-1] == 1 and g["empty"][:-1].sum() == 0


def test_golden_pointnet2_oracle_is_stable(cpu):
    g = np.load(os.path.join(GOLDEN, "pointnet2_oracle.npz"))
    xyz = g["xyz"]
    fidx = cpu.fps(xyz, 128)
    assert np.array_equal(fidx, g["fps_idx"])
    new_xyz = np.stack([xyz[b][fidx[b]] for b in range(2)])
    assert np.array_equal(cpu.ball_query(0.2, 16, xyz, new_xyz), g["ball_idx"])
    d2, i3 = cpu.three_nn(xyz[:, :300], new_xyz)
    assert np.array_equal(d2, g["nn_dist2"]) and np.array_equal(i3, g["nn_idx"])
    assert np.array_equal(cpu.three_weights(d2), g["nn_w"])


# ------------------------------------------------------------------ (2) live against the reference's own code
def test_ref_iou3d_live(cpu, ref):
    a, b = rand_bev(150, 4.0, seed=21), rand_bev(110, 4.0, seed=22)
    assert np.array_equal(cpu.boxes_overlap_bev(a, b, 0), ref.boxes_overlap_bev(a, b))
    assert np.array_equal(cpu.boxes_iou_bev(a, b, 0), ref.boxes_iou_bev(a, b))
    boxes = rand_bev(400, 5.0, seed=23)
    for kind in ("rotated", "normal"):
        for thr in (0.05, 0.3, 0.85):
            assert np.array_equal(cpu.nms(boxes, thr, kind, 0), ref.nms(boxes, thr, kind))


def test_ref_nms_mask_layout(cpu, ref):
    """the oracle's mask words reproduce the reference kernel's tile/bit conventions: the sweep over them
    (iou3d.cpp:100-119) gives the reference keep set"""
    boxes = rand_bev(200, 3.0, seed=31)
    mask = cpu.nms_mask(boxes, 0.3, "rotated", 0)
    W = mask.shape[1]
    remv = np.zeros(W, np.uint64)
    keep = []
    for i in range(200):
        if not (int(remv[i // 64]) >> (i % 64)) & 1:
            keep.append(i)
            remv[i // 64:] |= mask[i, i // 64:]
    assert np.array_equal(np.array(keep), ref.nms(boxes, 0.3, "rotated"))


def test_ref_roipool3d_live_cpu_and_emulated_gpu(cpu, ref):
    r = np.random.default_rng(5)
    N, M, C, S = 2500, 20, 5, 40
    xyz = kitti_cloud(2, N, seed=55)
    boxes = np.stack([enlarge(rand_boxes3d(xyz[b], M, seed=b), 1.0) for b in range(2)])
    feat = r.normal(size=(2, N, C)).astype(np.float32)
    want, wempty = cpu.roipool3d(xyz, boxes, feat, S, 0)
    for slow in (False, True):       # roipool3d_kernel.cu:97-194 (fast) and :31-94 (slow), grid-emulated on the host
        got, gempty = ref.roipool3d_gpu(xyz, boxes, feat, S, slow=slow)
        assert np.array_equal(got, want) and np.array_equal(gempty, wempty)
    pp, pf, pe = ref.roipool3d_cpu(xyz[0], boxes[0], feat[0], S)      # roipool3d.cpp:127-195
    assert np.array_equal(want[0, :, :, :3], pp) and np.array_equal(want[0, :, :, 3:], pf)
    assert np.array_equal(wempty[0], pe)
    assert np.array_equal(cpu.pts_in_boxes3d(xyz[0], boxes[0], 0), ref.pts_in_boxes3d_cpu(xyz[0], boxes[0]))


# ------------------------------------------------------------------ (3) third opinion for the unpinned PointNet++ ops
def test_fps_vs_numpy(cpu):
    xyz = unit_cloud(2, 500, seed=1)
    idx = cpu.fps(xyz, 60)
    for b in range(2):
        p = xyz[b]
        t = np.full(500, 1e10, np.float32)
        cur, want = 0, [0]
        for _ in range(59):
            d = p - p[cur]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]     # float32 numpy: individually rounded
            t = np.minimum(t, d2)
            cur = int(np.argmax(t))                                              # first maximum == lowest index
            want.append(cur)
        assert np.array_equal(idx[b], np.array(want))


def test_fps_tie_rule_lowest_index(cpu):
    pts = np.zeros((1, 10, 3), np.float32)
    pts[0, 3:] = [1, 0, 0]            # points 3..9 identical and farthest from point 0
    assert cpu.fps(pts, 3).tolist() == [[0, 3, 0]]     # after 0 and 3, every temp is 0 -> lowest index 0


def test_ball_query_vs_torch(cpu):
    xyz, new_xyz = unit_cloud(2, 400, seed=2), unit_cloud(2, 50, seed=3)
    r, ns = 0.25, 8
    idx = cpu.ball_query(r, ns, xyz, new_xyz)
    a, b = torch.from_numpy(new_xyz), torch.from_numpy(xyz)
    d = a[:, :, None, :] - b[:, None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    hit = d2 < np.float32(r) * np.float32(r)
    for bi in range(2):
        for m in range(50):
            ks = torch.nonzero(hit[bi, m]).flatten().tolist()[:ns]
            want = (ks + [ks[0]] * (ns - len(ks))) if ks else [0] * ns
            assert idx[bi, m].tolist() == want


def test_three_nn_and_interp_vs_torch(cpu):
    unk, kn = unit_cloud(2, 120, seed=4), unit_cloud(2, 40, seed=5)
    d2, idx = cpu.three_nn(unk, kn)
    a, b = torch.from_numpy(unk), torch.from_numpy(kn)
    d = a[:, :, None, :] - b[:, None, :, :]
    full = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    vals, ids = torch.sort(full, dim=2, stable=True)                 # stable: ties keep the earlier index
    assert np.array_equal(d2, vals[:, :, :3].numpy()) and np.array_equal(idx, ids[:, :, :3].numpy().astype(np.int32))
    w = cpu.three_weights(d2)
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-6)
    feat = np.random.default_rng(6).normal(size=(2, 9, 40)).astype(np.float32)
    out = cpu.three_interp(feat, idx, w)
    g = np.take_along_axis(feat[:, :, None, :], idx[:, None, :, :].astype(np.int64), 3)      # (B,C,n,3)
    np.testing.assert_allclose(out, (g * w[:, None]).sum(-1), atol=1e-6)


def test_gather_group_and_grads_vs_numpy(cpu):
    r = np.random.default_rng(7)
    feat = r.normal(size=(2, 4, 30)).astype(np.float32)
    idx = r.integers(0, 30, (2, 11)).astype(np.int32)
    gidx = r.integers(0, 30, (2, 11, 3)).astype(np.int32)
    assert np.array_equal(cpu.gather(feat, idx), np.stack([feat[b][:, idx[b]] for b in range(2)]))
    assert np.array_equal(cpu.group(feat, gidx), np.stack([feat[b][:, gidx[b]] for b in range(2)]))
    go = r.normal(size=(2, 4, 11, 3)).astype(np.float32)
    want = np.zeros((2, 4, 30), np.float32)
    for b in range(2):
        for c in range(4):
            np.add.at(want[b, c], gidx[b].ravel(), go[b, c].ravel())
    np.testing.assert_allclose(cpu.group_grad(go, gidx, 30), want, atol=1e-5)


def test_linear_rows_vs_numpy(cpu):
    r = np.random.default_rng(8)
    a, w, b = r.normal(size=(50, 37)), r.normal(size=(11, 37)), r.normal(size=11)
    want = np.maximum(a.astype(np.float32).astype(np.float64) @ w.astype(np.float32).astype(np.float64).T
                      + b.astype(np.float32), 0)
    np.testing.assert_allclose(cpu.linear_rows(a, w, b, True), want, rtol=1e-6, atol=1e-6)


def test_roipool_wrap_duplication_and_empty(cpu):
    """cnt < S: slot k >= cnt copies slot k % cnt (roipool3d.cpp:176-191); cnt == 0: flag set, zeros"""
    xyz = np.array([[[0, 0, 0], [0.1, 0, 0.1], [50, 0, 50], [-0.1, 0, 0.2]]], np.float32)
    boxes = np.array([[[0, 1, 0, 2, 2, 2, 0.0], [100, 1, 100, 2, 2, 2, 0.0]]], np.float32)
    feat = np.arange(4, dtype=np.float32).reshape(1, 4, 1) + 10
    out, empty = cpu.roipool3d(xyz, boxes, feat, 7)
    assert empty.tolist() == [[0, 1]]
    assert out[0, 0, :, 3].tolist() == [10, 11, 13, 10, 11, 13, 10]
    assert (out[0, 1] == 0).all()


def test_iou_identities(cpu):
    boxes = rand_bev(50, 3.0, seed=9)
    for mode in (0, 1):
        iou = cpu.boxes_iou_bev(boxes, boxes, mode)
        np.testing.assert_allclose(np.diag(iou), 1.0, atol=1e-5)
        np.testing.assert_allclose(iou, iou.T, atol=1e-5)
        assert (iou >= 0).all() and (iou <= 1 + 1e-5).all()
    sq = np.array([[0, 0, 2, 2, 0.0], [1, 1, 3, 3, 0.0], [0, 0, 2, 2, np.pi / 4]], np.float32)
    ov = cpu.boxes_overlap_bev(sq, sq, 1)
    np.testing.assert_allclose(ov[0, 1], 1.0, atol=1e-6)                       # unit square overlap
    np.testing.assert_allclose(ov[0, 2], 8 * (np.sqrt(2) - 1), atol=1e-5)      # regular octagon of two 2x2 squares


def test_arith_modes_of_the_index_operators(cpu):
    """round 6 comparison mode (prcnn_oracle.c sqdist3_mode): arith 0 == the canonical functions; arith 1 == an independent numpy
    restatement of nvcc's contraction fma(dz,dz, fma(dy,dy, dx*dx)) in exact rational arithmetic (each fused step rounds the EXACT
    a*b + c once, to the nearest float32, ties to even); the two arithmetics give different last bits on some distances"""
    from fractions import Fraction
    r = np.random.default_rng(12)
    xyz = r.uniform([-40, -1, 0], [40, 3, 70.4], (2, 1500, 3)).astype(np.float32)
    ctr = xyz[:, ::5].copy()
    assert np.array_equal(cpu.fps_mode(xyz, 300, 0, 0), cpu.fps(xyz, 300))
    assert np.array_equal(cpu.fps_mode(xyz, 300, 1, 0), cpu.fps_upstream(xyz, 300))
    assert np.array_equal(cpu.ball_query_arith(1.0, 16, xyz, ctr, 0), cpu.ball_query(1.0, 16, xyz, ctr))
    d0, i0 = cpu.three_nn_arith(xyz, ctr, 0)
    rd, ri = cpu.three_nn(xyz, ctr)
    assert np.array_equal(d0, rd) and np.array_equal(i0, ri)
    d1, i1 = cpu.three_nn_arith(xyz, ctr, 1)

    def f32(x):
        """the float32 nearest to the exact rational x (ties to even): candidates around the double-rounded guess, compared exactly"""
        g = np.float32(float(x))
        cands = [np.nextafter(g, np.float32(-np.inf)), g, np.nextafter(g, np.float32(np.inf))]
        best = min(cands, key=lambda c: (abs(Fraction(float(c)) - x), int(np.float32(c).view(np.uint32)) & 1))
        return np.float32(best)

    def contracted(u, k):            # fma rounds the EXACT a*b + c once
        dx, dy, dz = (np.float32(u[c] - k[c]) for c in range(3))
        xx = f32(Fraction(float(dx)) * Fraction(float(dx)))
        t = f32(Fraction(float(dy)) * Fraction(float(dy)) + Fraction(float(xx)))
        return f32(Fraction(float(dz)) * Fraction(float(dz)) + Fraction(float(t)))

    for b in range(2):
        for i in range(0, 1500, 37):
            for s in range(3):
                assert contracted(xyz[b, i], ctr[b, i1[b, i, s]]) == d1[b, i, s], (b, i, s)
    same = (i0 == i1).all(2)
    assert (d0[same] != d1[same]).any(), "no distance differs between the arithmetics"
    assert (np.abs(d0[same].astype(np.float64) - d1[same]) <= 2 * np.spacing(np.maximum(d0[same], d1[same]))).all()
