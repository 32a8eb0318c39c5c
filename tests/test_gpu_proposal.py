"""Proposal stage on the GPU (decode_kernel / sort_split_kernel / greedy_nms_kernel / assemble_kernel) through the
C ABI: bit-exact against the CPU oracle and the golden vectors made by the reference's own Python."""
import numpy as np
import pytest
import torch

from proposal_cases import case_inputs, golden, proposal_cases
from rcnn_bev import bev
from util import ANCHOR, rand_boxes3d, rpn_like_scene

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", sorted(proposal_cases()))
def test_proposal_layer_equals_golden_and_oracle(dev, cpu, name):
    from pointrcnn_amd import ops
    g = golden()
    xyz, sc, reg, kw = case_inputs(name)
    B, N = sc.shape
    boxes = ops.decode_bbox_target(_t(xyz.reshape(-1, 3), dev), _t(reg.reshape(-1, 76), dev), 3.0, 0.5, 12, ANCHOR,
                                   get_xz_fine=True, y_to_bottom=True).view(B, N, 7)
    ob = cpu.decode_bbox_target(xyz.reshape(-1, 3), reg.reshape(-1, 76), 3.0, 0.5, 12, ANCHOR, get_xz_fine=True,
                                y_to_bottom=True).reshape(B, N, 7)
    assert np.array_equal(boxes.cpu().numpy(), ob)
    rois, scores, cnt = ops.proposal_layer(_t(sc, dev), boxes, kw["pre"], kw["post"], kw["thresh"],
                                           rotated=kw["kind"] == "rotated", ranges=kw["ranges"])
    o_rois, o_scores, o_cnt = cpu.proposal_layer(sc, ob, kw["pre"], kw["post"], kw["thresh"], kw["kind"], kw["ranges"])
    assert np.array_equal(cnt.cpu().numpy(), o_cnt)
    assert np.array_equal(rois.cpu().numpy(), o_rois) and np.array_equal(scores.cpu().numpy(), o_scores)
    assert np.array_equal(rois.cpu().numpy(), g[name + "_rois"]) and np.array_equal(scores.cpu().numpy(), g[name + "_scores"])


def test_decode_variants_equal_oracle_and_golden(dev, cpu):
    from pointrcnn_amd import ops
    g = golden()
    xyz, sc, reg = rpn_like_scene(1, 4096, seed=6)
    out = ops.decode_bbox_target(_t(xyz[0], dev), _t(reg[0], dev), 3.0, 0.5, 12, ANCHOR, get_xz_fine=True)
    assert np.array_equal(out.cpu().numpy(), g["dec_rpn"])
    out = ops.decode_bbox_target(_t(xyz[0], dev), _t(reg[0][:, 24:], dev), 3.0, 0.5, 12, ANCHOR, get_xz_fine=False)
    assert np.array_equal(out.cpu().numpy(), g["dec_rpn_coarse"])
    for key, regk, ybin in (("dec_rcnn", "dec_reg46", False), ("dec_rcnn_ybin", "dec_reg53", True)):
        out = ops.decode_bbox_target(_t(g["dec_rois7"], dev), _t(g[regk], dev), 1.5, 0.5, 9, ANCHOR, get_xz_fine=True,
                                     get_y_by_bin=ybin, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True).cpu().numpy()
        o = cpu.decode_bbox_target(g["dec_rois7"], g[regk], 1.5, 0.5, 9, ANCHOR, get_xz_fine=True, get_y_by_bin=ybin,
                                   loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True, trig_mode=1)
        assert np.array_equal(out, o)                                   # canonical trig: bit-exact
        np.testing.assert_allclose(out, g[key], rtol=0, atol=1e-5)      # reference Python (torch cos/sin + bmm)
    # odd row counts, NaN / tie logits: first maximum wins, NaN is maximal (torch.argmax)
    r = np.random.default_rng(3)
    reg = r.integers(-2, 3, (777, 76)).astype(np.float32)              # many exact ties
    reg[5, 3] = np.nan
    reg[6, 20] = np.nan
    pts = r.normal(size=(777, 3)).astype(np.float32)
    out = ops.decode_bbox_target(_t(pts, dev), _t(reg, dev), 3.0, 0.5, 12, ANCHOR).cpu().numpy()
    assert np.array_equal(out, cpu.decode_bbox_target(pts, reg, 3.0, 0.5, 12, ANCHOR), equal_nan=True)


def test_decode_and_proposal_argument_errors(dev):
    from pointrcnn_amd import ops
    from pointrcnn_amd._cabi import PointOpsError
    z = torch.zeros
    with pytest.raises(PointOpsError):
        ops.decode_bbox_target(z((4, 3), device=dev), z((4, 75), device=dev), 3.0, 0.5, 12, ANCHOR)
    with pytest.raises(PointOpsError):
        ops.decode_bbox_target(z((4, 5), device=dev), z((4, 76), device=dev), 3.0, 0.5, 12, ANCHOR)
    with pytest.raises(RuntimeError):
        ops.proposal_layer(z((1, 8)), z((1, 8, 7)), (6, 2), (2, 1), 0.8)          # CPU tensors: no CPU path
    # empty problems are fine
    rois, scores, cnt = ops.proposal_layer(z((2, 0), device=dev), z((2, 0, 7), device=dev), (6300, 2700), (70, 30), 0.8)
    assert rois.shape == (2, 100, 7) and float(rois.abs().sum()) == 0 and cnt.tolist() == [0, 0]


def test_score_ties_nan_and_small_frames(dev, cpu):
    from pointrcnn_amd import ops
    r = np.random.default_rng(11)
    for N in (1, 15, 16, 17, 100, 1000, 1025):
        boxes = rand_boxes3d(np.stack([r.uniform(-30, 30, N), np.ones(N), r.uniform(1, 75, N)], 1), N, seed=N, jitter=0.3)[None]
        sc = r.integers(-3, 4, (1, N)).astype(np.float32)              # heavy ties -> order by row index
        if N > 20:
            sc[0, 7] = np.nan
            sc[0, 11] = -0.0
        rois, scores, cnt = ops.proposal_layer(_t(sc, dev), _t(boxes, dev), (6300, 2700), (70, 30), 0.8, rotated=True)
        o = cpu.proposal_layer(sc, boxes, (6300, 2700), (70, 30), 0.8, "rotated")
        assert np.array_equal(rois.cpu().numpy(), o[0]) and np.array_equal(cnt.cpu().numpy(), o[2]), N
        assert np.array_equal(scores.cpu().numpy(), o[1], equal_nan=True), N


@pytest.mark.parametrize("thresh", [0.8, 0.3, -1.0])
def test_rotated_prefiltered_kernel_equals_chunk_kernel_and_oracle(dev, cpu, monkeypatch, thresh):
    """greedy_nms_rot_kernel (round 5: a batch of 256 candidates is tested against the kept list before the serial chunk step) keeps
    exactly what greedy_nms_kernel<ROTATED> (PRCNN_NMS_PREFILTER=0) and the oracle keep, on the scene it was written for: 24 cars,
    40 % of the points voting for them in tight clusters, so that ~2 500 candidates are scanned before 70 survive; a low threshold
    (few survivors per batch, several batches per kept box) and a negative one (every pair suppresses: the circle test is off)."""
    from pointrcnn_amd import ops
    xyz, sc, reg = rpn_like_scene(3, 16384, seed=21)
    B, N = sc.shape
    boxes = ops.decode_bbox_target(_t(xyz.reshape(-1, 3), dev), _t(reg.reshape(-1, 76), dev), 3.0, 0.5, 12, ANCHOR,
                                   get_xz_fine=True, y_to_bottom=True).view(B, N, 7)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PRCNN_NMS_PREFILTER", flag)
        rois, scores, cnt = ops.proposal_layer(_t(sc, dev), boxes, (6300, 2700), (70, 30), thresh, rotated=True)
        got[flag] = (rois.cpu().numpy(), scores.cpu().numpy(), cnt.cpu().numpy())
    for a, b in zip(got["1"], got["0"]):
        assert np.array_equal(a, b)
    o = cpu.proposal_layer(sc, boxes.cpu().numpy(), (6300, 2700), (70, 30), thresh, "rotated")
    for a, b in zip(got["1"], o):
        assert np.array_equal(a, b)
    # the detection select's shape (eval_rcnn.py:600-614): 100 boxes per frame, every survivor kept
    keep = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PRCNN_NMS_PREFILTER", flag)
        k, n = ops.nms_batched(boxes[:, :300].contiguous(), _t(sc[:, :300], dev), None, 0.1, True)
        keep[flag] = (k.cpu().numpy(), n.cpu().numpy())
    assert np.array_equal(keep["1"][0], keep["0"][0]) and np.array_equal(keep["1"][1], keep["0"][1])


@pytest.mark.parametrize("thresh", [0.8, 0.3, -1.0])
def test_normal_prefiltered_kernel_equals_chunk_kernel_and_oracle(dev, cpu, monkeypatch, thresh):
    """greedy_nms_pre_kernel (round 6: growing batches of candidates are tested against the kept list before the serial chunk step, on the
    default axis-aligned path) keeps exactly what greedy_nms_kernel<NORMAL> (PRCNN_NMS_PREFILTER=0) and the oracle keep, on the 24-car
    scene, at the RPN's threshold, a low one and a negative one (every pair suppresses: one box survives per range)."""
    from pointrcnn_amd import ops
    xyz, sc, reg = rpn_like_scene(3, 16384, seed=22)
    B, N = sc.shape
    boxes = ops.decode_bbox_target(_t(xyz.reshape(-1, 3), dev), _t(reg.reshape(-1, 76), dev), 3.0, 0.5, 12, ANCHOR,
                                   get_xz_fine=True, y_to_bottom=True).view(B, N, 7)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PRCNN_NMS_PREFILTER", flag)
        rois, scores, cnt = ops.proposal_layer(_t(sc, dev), boxes, (6300, 2700), (70, 30), thresh, rotated=False)
        got[flag] = (rois.cpu().numpy(), scores.cpu().numpy(), cnt.cpu().numpy())
    for a, b in zip(got["1"], got["0"]):
        assert np.array_equal(a, b)
    o = cpu.proposal_layer(sc, boxes.cpu().numpy(), (6300, 2700), (70, 30), thresh, "normal")
    for a, b in zip(got["1"], o):
        assert np.array_equal(a, b)
    # small problems (fewer candidates than the first batch; more kept than one chunk) through nms_batched
    keep = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PRCNN_NMS_PREFILTER", flag)
        k, n = ops.nms_batched(boxes[:, :300].contiguous(), _t(sc[:, :300], dev), None, 0.1, False)
        k2, n2 = ops.nms_batched(boxes[:, :40].contiguous(), _t(sc[:, :40], dev), None, 0.5, False)
        keep[flag] = (k.cpu().numpy(), n.cpu().numpy(), k2.cpu().numpy(), n2.cpu().numpy())
    for a, b in zip(keep["1"], keep["0"]):
        assert np.array_equal(a, b)


def test_nms_batched_equals_oracle_and_sorted_nms(dev, cpu):
    """tools/eval_rcnn.py:600-614: score-threshold select + rotated NMS 0.1 on the refined boxes, whole batch at once"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(5)
    B, M = 6, 100
    ctr = np.stack([r.uniform(-10, 10, 30), np.ones(30), r.uniform(5, 30, 30)], 1)
    boxes = np.stack([rand_boxes3d(ctr, M, seed=40 + b, jitter=0.4) for b in range(B)])
    sc = r.normal(size=(B, M)).astype(np.float32)
    valid = 1 / (1 + np.exp(-sc)) > 0.3
    valid[3] = False                                                     # a frame with nothing above threshold
    for kind, thr, v in (("rotated", 0.1, valid), ("normal", 0.5, None), ("rotated", 0.8, None)):
        keep, num = ops.nms_batched(_t(boxes, dev), _t(sc, dev), None if v is None else _t(v, dev), thr, kind == "rotated")
        ok, on = cpu.nms_batched(boxes, sc, v, thr, kind)
        assert np.array_equal(num.cpu().numpy(), on) and np.array_equal(keep.cpu().numpy(), ok)
        for b in range(B):                                               # same answer as the pre-sorted single-frame NMS
            order = cpu.argsort_desc(sc[b])
            sel = order if v is None else order[v[b][order]]
            assert np.array_equal(sel[cpu.nms(bev(boxes[b][sel]), thr, kind)], ok[b, :on[b]])
    keep, num = ops.nms_batched(_t(boxes, dev), _t(sc, dev), None, 0.1, True, max_keep=5)
    ok, on = cpu.nms_batched(boxes, sc, None, 0.1, "rotated", 5)
    assert keep.shape == (B, 5) and np.array_equal(keep.cpu().numpy(), ok) and np.array_equal(num.cpu().numpy(), on)


def test_full_batch_properties(dev):
    """BASELINE config 3 size (bs32 x 16384): per-frame results do not depend on the batch, survivors are ordered,
    inside their distance area, mutually below the NMS threshold, and the padding is zero."""
    from pointrcnn_amd import ops
    xyz, sc, reg = rpn_like_scene(32, 16384, seed=9)
    B, N = sc.shape
    boxes = ops.decode_bbox_target(_t(xyz.reshape(-1, 3), dev), _t(reg.reshape(-1, 76), dev), 3.0, 0.5, 12, ANCHOR,
                                   y_to_bottom=True).view(B, N, 7)
    sct = _t(sc, dev)
    for rotated in (False, True):
        rois, scores, cnt = ops.proposal_layer(sct, boxes, (6300, 2700), (70, 30), 0.8, rotated=rotated)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
        r2, s2, c2 = ops.proposal_layer(sct[perm].contiguous(), boxes[perm].contiguous(), (6300, 2700), (70, 30), 0.8, rotated=rotated)
        assert torch.equal(r2, rois[perm]) and torch.equal(s2, scores[perm]) and torch.equal(c2, cnt[perm])
        r1, s1, c1 = ops.proposal_layer(sct[5:6].contiguous(), boxes[5:6].contiguous(), (6300, 2700), (70, 30), 0.8, rotated=rotated)
        assert torch.equal(r1[0], rois[5]) and torch.equal(s1[0], scores[5])
        rois_c, scores_c, cnt_c = rois.cpu().numpy(), scores.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            n = cnt_c[b]
            assert (rois_c[b, n:] == 0).all() and (scores_c[b, n:] == 0).all()
            z = rois_c[b, :n, 2]
            n1 = int((z <= 40).sum())
            assert n1 <= 70 and n - n1 <= 30 and (z[:n1] > 0).all() and (z[n1:] > 40).all() and (z[n1:] <= 80).all()
            assert (np.diff(scores_c[b, :n1]) <= 0).all() and (np.diff(scores_c[b, n1:n]) <= 0).all()
        # survivors of one area are pairwise below the threshold
        bv = torch.from_numpy(bev(rois_c[0, :cnt_c[0]])).to(dev)
        n1 = int((rois_c[0, :cnt_c[0], 2] <= 40).sum())
        if rotated:
            iou = ops.boxes_iou_bev(bv[:n1].contiguous(), bv[:n1].contiguous()).cpu().numpy()
            assert (np.triu(iou, 1) <= 0.8).all()


def test_point_rcnn_end_to_end_config3(dev, cpu):
    """BASELINE config 3: RPN -> proposal layer -> roipool3d -> RCNN -> decode + rotated NMS, random-init weights.
    The proposal / detection glue on the device must equal the oracle run on the same head outputs."""
    from pointrcnn_amd.point_rcnn import PointRCNN
    from pointrcnn_amd.rpn import randomize_bn_stats, synthetic_clouds
    torch.manual_seed(0)
    for nms_type in ("normal", "rotate"):
        model = randomize_bn_stats(PointRCNN(mode="TEST")).to(dev).eval()
        model.rpn.proposal_layer.cfg = type("Cfg", (model.rpn.proposal_layer.cfg,), {"NMS_TYPE": nms_type})
        pts = synthetic_clouds(2, 16384, device=dev)
        with torch.no_grad():
            out = model({"pts_input": pts})
            pred, raw, keep, num = model.detections(out, score_thresh=0.3)
        assert out["rois"].shape == (2, 100, 7) and out["roi_scores_raw"].shape == (2, 100)
        assert out["rcnn_cls"].shape == (200, 1) and out["rcnn_reg"].shape == (200, 46)
        sc = out["rpn_cls"][:, :, 0].cpu().numpy()
        reg = out["rpn_reg"].cpu().numpy()
        xyz = out["backbone_xyz"].cpu().numpy()
        boxes = cpu.decode_bbox_target(xyz.reshape(-1, 3), reg.reshape(-1, 76), 3.0, 0.5, 12, ANCHOR, y_to_bottom=True)
        o_rois, o_scores, _ = cpu.proposal_layer(sc, boxes.reshape(2, -1, 7), (6300, 2700), (70, 30), 0.8,
                                                 "rotated" if nms_type == "rotate" else "normal")
        assert np.array_equal(out["rois"].cpu().numpy(), o_rois) and np.array_equal(out["roi_scores_raw"].cpu().numpy(), o_scores)
        o_pred = cpu.decode_bbox_target(o_rois.reshape(-1, 7), out["rcnn_reg"].cpu().numpy(), 1.5, 0.5, 9, ANCHOR,
                                        get_xz_fine=True, get_ry_fine=True).reshape(2, 100, 7)
        assert np.array_equal(pred.cpu().numpy(), o_pred)
        raw_c = raw.cpu().numpy()
        valid = torch.sigmoid(raw).cpu().numpy() > 0.3
        ok, on = cpu.nms_batched(o_pred, raw_c, valid, 0.1, "rotated")
        assert np.array_equal(keep.cpu().numpy(), ok) and np.array_equal(num.cpu().numpy(), on)


@pytest.mark.parametrize("N", [16385, 20000, 40000, 65536])
def test_frames_larger_than_the_lds_sort(dev, cpu, N):
    """more than 16384 rows per frame (BASELINE config 5: 65 536 points): the bitonic network runs in 16384-key chunks with
    the long strides through HBM -- same proposals as the oracle, including score ties and the borrow rule"""
    from pointrcnn_amd import ops
    xyz, sc, reg = rpn_like_scene(2, N, seed=N % 97, z_max=70.4 if N != 40000 else 38.0)      # 40000: far area empty
    sc[0, ::7] = sc[0, 3]                                                                        # many exact ties
    boxes = cpu.decode_bbox_target(xyz.reshape(-1, 3), reg.reshape(-1, 76), 3.0, 0.5, 12, ANCHOR, y_to_bottom=True).reshape(2, N, 7)
    for kind in ("normal", "rotated"):
        rois, scores, cnt = ops.proposal_layer(_t(sc, dev), _t(boxes, dev), (6300, 2700), (70, 30), 0.8, rotated=kind == "rotated")
        o = cpu.proposal_layer(sc, boxes, (6300, 2700), (70, 30), 0.8, kind)
        assert np.array_equal(rois.cpu().numpy(), o[0]) and np.array_equal(scores.cpu().numpy(), o[1]) and \
            np.array_equal(cnt.cpu().numpy(), o[2]), (N, kind)
    if N == 20000:
        valid = sc > 1.0
        keep, num = ops.nms_batched(_t(boxes, dev), _t(sc, dev), _t(valid, dev), 0.5, rotated=False, max_keep=200)
        ok, on = cpu.nms_batched(boxes, sc, valid, 0.5, "normal", 200)
        assert np.array_equal(keep.cpu().numpy(), ok) and np.array_equal(num.cpu().numpy(), on)
