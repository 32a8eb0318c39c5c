"""RCNN-stage training targets (SURVEY 8(a) a14's call site, lib/rpn/proposal_target_layer.py): device RoI sampling + noise
augmentation.

Pin: tests/golden/proposal_target_ref.npz holds what the REFERENCE'S OWN ProposalTargetLayer.sample_rois_for_rcnn returns on
seeded scenes when its random calls are answered from the counter-based table the kernels draw from
(tests/golden/ref_proposal_target.py): the oracle in reference arithmetic (trig_mode 0) reproduces it bit for bit -- all four
sampling cases, the accept / retry loop, both noise methods -- and, live, so does the reference's code when it is present.
GPU: the HIP kernel equals the oracle AND the reference fixture bit for bit (the kernels evaluate the reference's host libm
arithmetic, csrc/ref_trig.h)."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from util import GOLDEN

sys.path.insert(0, GOLDEN)
import ref_proposal_target as rpt  # noqa: E402

CASES = ((1, "multiple"), (2, "multiple"), (3, "single"))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "proposal_target_ref.npz"))


def test_oracle_reproduces_the_references_own_sampler_bit_for_bit(g, cpu):
    for seed, method in CASES:
        roi, gt = rpt.scenes(seed)
        tag = "s%d_" % seed
        assert np.uint32(zlib.crc32(gt.tobytes(), zlib.crc32(roi.tobytes()))) == g[tag + "crc"], "seeded inputs changed"
        o = cpu.proposal_target_sample(roi, gt, seed=40 + seed, aug_method=method, trig_mode=0)
        assert np.array_equal(o["rois"], g[tag + "rois"]) and np.array_equal(o["gt_of_rois"], g[tag + "gt_of_rois"])
        assert np.array_equal(o["roi_iou"], g[tag + "roi_iou"])
        k = int((gt[0].sum(1) != 0).sum())
        assert np.array_equal(cpu.boxes_iou3d(roi[0], gt[0, :k], trig_mode=0), g[tag + "iou0"])       # iou3d_utils.boxes_iou3d_gpu
        # the fixture exercises every branch: fg + bg, fg only (with replacement), bg only
        c = o["counts"]
        assert (c[0, 0] > 32 and c[0, 1] > 0 and c[0, 2] > 0 and c[0, 3] == 32) and (c[1, 1] + c[1, 2] == 0 and c[1, 3] == 64) and c[2, 0] == 0
        assert (o["status"] == 0).all()


@pytest.mark.skipif(not rpt.available(), reason="reference checkout absent")
def test_reference_sampler_live(cpu, ref):
    roi, gt = rpt.scenes(5, B=2)
    o = cpu.proposal_target_sample(roi, gt, seed=9, trig_mode=0)
    rr, rg, ri, iou0 = rpt.run_reference(roi, gt, 9, o)
    assert np.array_equal(rr, o["rois"]) and np.array_equal(rg, o["gt_of_rois"]) and np.array_equal(ri, o["roi_iou"])


def test_sampler_invariants_and_edge_cases(cpu):
    """what the reference guarantees about a sample, checked on the oracle's output in the kernels' arithmetic"""
    roi, gt = rpt.scenes(7)
    o = cpu.proposal_target_sample(roi, gt, seed=1)
    fg_t = 0.55
    for b in range(roi.shape[0]):
        src, mo = o["src"][b], o["max_overlaps"][b]
        fs = int(o["counts"][b, 3])
        assert (src >= 0).all() and (mo[src[:fs]] >= fg_t).all() and (mo[src[fs:]] < 0.45).all()
        if 0 < fs < 64:
            assert len(set(src[:fs].tolist())) == fs                       # foreground: without replacement
        nh = int((64 - fs) * 0.8) if o["counts"][b, 1] > 0 and o["counts"][b, 2] > 0 else (64 - fs if o["counts"][b, 1] > 0 else 0)
        assert (mo[src[fs:fs + nh]] >= 0.05).all() and (mo[src[fs + nh:]] < 0.05).all()
        assert np.array_equal(o["gt_of_rois"][b], gt[b][o["gt_assignment"][b][src]])
        # a slot's box is the source RoI itself (IoU = its max overlap) or a noisy copy whose reported IoU is its 3-D IoU with the gt
        same = (o["rois"][b] == roi[b][src]).all(1)
        assert np.array_equal(o["roi_iou"][b][same], mo[src][same])
        for t in np.flatnonzero(~same):
            assert o["roi_iou"][b, t] == cpu.boxes_iou3d(o["rois"][b, t:t + 1], o["gt_of_rois"][b, t:t + 1])[0, 0]
        assert (np.abs(o["rois"][b][:, :3] - roi[b][src][:, :3]) <= 1.0 + 1e-6).all()      # 'multiple': shifts within +-1 m
    # no ground truth at all -> status 2, zero outputs; neither fg nor bg candidates -> status 1 (the reference raises)
    z = cpu.proposal_target_sample(roi[:1], np.zeros((1, 4, 7), np.float32), seed=1)
    assert z["status"][0] == 2 and not z["rois"].any()
    shift = next(t for t in np.arange(0.05, 3.0, 0.02)                     # a copy of the box at IoU ~0.5: in neither candidate set
                 if 0.47 <= cpu.boxes_iou3d(gt[0, :1] + np.array([t, 0, 0, 0, 0, 0, 0], np.float32), gt[0, :1])[0, 0] <= 0.53)
    mid = np.tile(gt[0, :1], (16, 1))[None].copy()
    mid[0, :, 0] += np.float32(shift)
    m = cpu.proposal_target_sample(mid, gt[:1, :1], seed=1)
    assert m["status"][0] == 1 and 0.45 <= m["max_overlaps"].min() and m["max_overlaps"].max() < 0.55
    # different seeds draw differently, the same seed identically
    a, b2 = cpu.proposal_target_sample(roi, gt, seed=2), cpu.proposal_target_sample(roi, gt, seed=2)
    assert np.array_equal(a["rois"], b2["rois"]) and not np.array_equal(a["src"], o["src"])


def test_mirror_data_augmentation_equals_reference_fixture(g):
    """ProposalTargetLayer.data_augmentation (:302-363) is torch-only: the mirror on CPU with the same generator state must
    return what the reference's own method returned (fixture), including its `- 0.5 / 0.5` precedence quirk"""
    from pointrcnn_amd.proposal_target_layer import ProposalTargetLayer
    pts, rois, gts = rpt.aug_inputs()
    torch.manual_seed(7)
    p2, r2, g2 = ProposalTargetLayer().data_augmentation(pts.clone(), rois.clone(), gts.clone())
    np.testing.assert_allclose(p2.numpy(), g["aug_pts"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r2.numpy(), g["aug_rois"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(g2.numpy(), g["aug_gt"], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_hip_sampler_equals_oracle_and_reference_fixture(g, cpu, dev):
    from pointrcnn_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)        # noqa: E731
    for seed, method in CASES:
        roi, gt = rpt.scenes(seed)
        got = ops.proposal_target_sample(t(roi), t(gt), aug_method=method, seed=40 + seed)
        want = cpu.proposal_target_sample(roi, gt, seed=40 + seed, aug_method=method)
        for k in want:
            assert np.array_equal(got[k].cpu().numpy(), want[k]), k
        tag = "s%d_" % seed
        # and the reference's own output, bit for bit (the kernels evaluate the reference's libm arithmetic, csrc/ref_trig.h)
        assert np.array_equal(got["rois"].cpu().numpy(), g[tag + "rois"]) and np.array_equal(got["roi_iou"].cpu().numpy(), g[tag + "roi_iou"])
        assert np.array_equal(got["gt_of_rois"].cpu().numpy(), g[tag + "gt_of_rois"])
    # ragged padding, 8-column ground truth (the docstring's [.., cls] layout), other sizes
    roi, gt = rpt.scenes(9, B=5, M=300, G=20)
    gt8 = np.concatenate([gt, (gt[..., :1] != 0).astype(np.float32)], -1)
    got = ops.proposal_target_sample(t(roi), t(gt8), roi_per_image=48, aug_times=3, seed=77)
    want = cpu.proposal_target_sample(roi, gt8, roi_per_image=48, aug_times=3, seed=77)
    for k in want:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    k = 6
    assert np.array_equal(ops.boxes_iou3d(t(roi[0]), t(gt[0, :k])).cpu().numpy(), cpu.boxes_iou3d(roi[0], gt[0, :k]))


@pytest.mark.gpu
def test_proposal_target_layer_mirror_forward(cpu, dev):
    """the whole layer on the device: output dictionary of the reference (keys, shapes, dtypes), labels consistent with the IoUs,
    canonical transform = centre subtraction + rotation by the RoI angle"""
    from pointrcnn_amd.proposal_target_layer import ProposalTargetConfig, ProposalTargetLayer
    roi, gt = rpt.scenes(4, B=2)
    B, N, C = 2, 4096, 16
    r = np.random.default_rng(0)
    xyz = r.uniform([-30, 0, 5], [30, 2, 65], (B, N, 3)).astype(np.float32)
    for b in range(B):                                                     # a few hundred points inside the first ground-truth box
        xyz[b, :400] = gt[b, 0, :3] + r.uniform(-0.7, 0.7, (400, 3)).astype(np.float32) * np.array([1.5, 0.5, 0.6], np.float32) - [0, 0.7, 0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)        # noqa: E731
    inp = {"roi_boxes3d": t(roi), "gt_boxes3d": t(gt), "rpn_xyz": t(xyz), "rpn_features": t(r.normal(size=(B, N, C)).astype(np.float32)),
           "seg_mask": t((r.random((B, N)) > 0.5).astype(np.float32)), "pts_depth": t(np.linalg.norm(xyz, axis=2).astype(np.float32))}
    cfg = type("C", (ProposalTargetConfig,), {"AUG_DATA": False})
    layer = ProposalTargetLayer(cfg, seed=3)
    out = layer(inp)
    R, S = cfg.ROI_PER_IMAGE, cfg.NUM_POINTS
    assert out["sampled_pts"].shape == (B * R, S, 3) and out["pts_feature"].shape == (B * R, S, C + 2)
    assert out["cls_label"].dtype == torch.int64 and out["reg_valid_mask"].dtype == torch.int64
    iou = out["gt_iou"].cpu().numpy()
    cls, reg = out["cls_label"].cpu().numpy(), out["reg_valid_mask"].cpu().numpy()
    assert ((cls == 1) <= (iou > 0.6)).all() and ((reg == 1) <= (iou > 0.55)).all() and (cls[(iou > 0.45) & (iou < 0.6)] == -1).all()
    # canonical frame: the RoI itself sits at the origin, the ground truth relative to it
    want = cpu.proposal_target_sample(roi, gt, seed=3)
    rois = out["roi_boxes3d"].cpu().numpy()
    assert np.array_equal(rois, want["rois"].reshape(-1, 7))
    gtc = out["gt_of_rois"].cpu().numpy()
    d = want["gt_of_rois"].reshape(-1, 7)[:, :3] - rois[:, :3]
    ry = rois[:, 6] % (2 * np.pi)
    c, s = np.cos(ry), np.sin(ry)
    np.testing.assert_allclose(gtc[:, 0], d[:, 0] * c - d[:, 2] * s, atol=1e-4)
    np.testing.assert_allclose(gtc[:, 2], d[:, 0] * s + d[:, 2] * c, atol=1e-4)
    # with augmentation on: same keys, finite
    out2 = ProposalTargetLayer(ProposalTargetConfig, seed=3)(inp)
    assert set(out2) == set(out) and all(torch.isfinite(v.float()).all() for v in out2.values())


def _hard_slots(o, b):
    fs = int(o["counts"][b, 3])
    mo = o["max_overlaps"][b][o["src"][b][fs:]]
    return fs, int((mo >= 0.05).sum())


def test_slot_arithmetic_is_python_double_arithmetic(cpu):
    """round-3 advisor finding: int(bg_rois_per_this_image * HARD_BG_RATIO) (proposal_target_layer.py:159) is Python double arithmetic --
    10 * 0.7 = 7, whereas 10 * 0.7f = 6.9999998 -> 6.  The ratios therefore cross the boundary as doubles."""
    roi, gt = rpt.scenes(1)
    o = cpu.proposal_target_sample(roi, gt, roi_per_image=20, cfgv=(0.55, 0.6, 0.45, 0.05, 0.5, 0.7), seed=3)
    c = o["counts"][0]
    assert c[0] >= 10 and c[1] >= 7 and c[2] >= 3, "frame 0 of this scene has all three candidate kinds"
    assert _hard_slots(o, 0) == (10, 7)
    o = cpu.proposal_target_sample(roi, gt, roi_per_image=20, cfgv=(0.55, 0.6, 0.45, 0.05, 0.35, 0.7), seed=3)     # np.round(7.0) = 7 foreground slots
    fs, nh = _hard_slots(o, 0)
    assert fs == 7 and nh == int(13 * 0.7)


@pytest.mark.gpu
def test_hip_sampler_slot_arithmetic_and_large_roi_lists(cpu, dev):
    from pointrcnn_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)        # noqa: E731
    roi, gt = rpt.scenes(1)
    got = ops.proposal_target_sample(t(roi), t(gt), roi_per_image=20, fg_ratio=0.5, hard_bg_ratio=0.7, seed=3)
    want = cpu.proposal_target_sample(roi, gt, roi_per_image=20, cfgv=(0.55, 0.6, 0.45, 0.05, 0.5, 0.7), seed=3)
    for k in want:
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert _hard_slots(want, 0) == (10, 7)
    # the documented cap: 8192 RoIs per frame = 128 KB of dynamic LDS (the limit is raised for the kernel), and a size in between
    for M in (5000, 8192):
        roi, gt = rpt.scenes(11, B=2, M=M, G=12)
        got = ops.proposal_target_sample(t(roi), t(gt), seed=5)
        want = cpu.proposal_target_sample(roi, gt, seed=5)
        for k in want:
            assert np.array_equal(got[k].cpu().numpy(), want[k]), (M, k)
