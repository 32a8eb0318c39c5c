"""Proposal stage: the CPU oracle against golden vectors produced by the REFERENCE'S OWN Python
(lib/utils/bbox_transform.py, lib/rpn/proposal_layer.py; tests/golden/make_golden.py) and, when the reference checkout
is present (build container only), against that Python run live."""
import numpy as np
import pytest

from proposal_cases import case_inputs, crc, golden, proposal_cases, split_top_n
from util import ANCHOR, rand_boxes3d, rpn_like_scene


def test_top_n_split_matches_the_reference_arithmetic():
    assert split_top_n(9000) == (6300, 2700) and split_top_n(100) == (70, 30) and split_top_n(512) == (358, 154)
    assert split_top_n(300) == (210, 90)


@pytest.mark.parametrize("name", sorted(proposal_cases()))
@pytest.mark.parametrize("trig_mode", [0, 1])
def test_proposal_layer_equals_reference_python(cpu, name, trig_mode):
    g = golden()
    xyz, sc, reg, kw = case_inputs(name)
    assert crc(xyz, sc, reg) == g[name + "_crc"], "seeded inputs drifted from the ones the fixture was made with"
    B, N = sc.shape
    boxes = cpu.decode_bbox_target(xyz.reshape(-1, 3), reg.reshape(-1, 76), 3.0, 0.5, 12, ANCHOR, get_xz_fine=True,
                                   y_to_bottom=True).reshape(B, N, 7)
    rois, scores, cnt = cpu.proposal_layer(sc, boxes, kw["pre"], kw["post"], kw["thresh"], kw["kind"], kw["ranges"],
                                           trig_mode=trig_mode)
    assert np.array_equal(rois, g[name + "_rois"])          # bit-exact, both trig contracts
    assert np.array_equal(scores, g[name + "_scores"])
    assert np.array_equal(cnt, (np.abs(g[name + "_rois"]).sum(-1) > 0).sum(1))


def test_decode_bbox_target_equals_reference_python(cpu):
    g = golden()
    xyz, sc, reg = rpn_like_scene(1, 4096, seed=6)
    assert crc(xyz, reg) == g["dec_crc"]
    out = cpu.decode_bbox_target(xyz[0], reg[0], 3.0, 0.5, 12, ANCHOR, get_xz_fine=True)
    assert np.array_equal(out, g["dec_rpn"])                 # roi = xyz: no trig involved, bit-exact
    out = cpu.decode_bbox_target(xyz[0], reg[0][:, 24:], 3.0, 0.5, 12, ANCHOR, get_xz_fine=False)
    assert np.array_equal(out, g["dec_rpn_coarse"])
    # roi = boxes: rotated back by roi ry with torch's cos/sin + a 2x2 bmm -> tolerance (1e-5, north_star)
    for key, regk, ybin in (("dec_rcnn", "dec_reg46", False), ("dec_rcnn_ybin", "dec_reg53", True)):
        for tm in (0, 1):
            out = cpu.decode_bbox_target(g["dec_rois7"], g[regk], 1.5, 0.5, 9, ANCHOR, get_xz_fine=True, get_y_by_bin=ybin,
                                         loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True, trig_mode=tm)
            np.testing.assert_allclose(out, g[key], rtol=0, atol=1e-5)
            assert np.array_equal(out[:, [1, 3, 4, 5]], g[key][:, [1, 3, 4, 5]])      # y, h, w, l untouched by the rotation


def test_decode_rejects_wrong_channel_count(cpu):
    with pytest.raises(ValueError):
        cpu.decode_bbox_target(np.zeros((4, 3), np.float32), np.zeros((4, 75), np.float32), 3.0, 0.5, 12, ANCHOR)


def test_argsort_is_descending_nan_first_ties_by_index(cpu):
    s = np.array([1.0, np.nan, 3.0, 1.0, -0.0, 0.0, np.inf, -np.inf, 3.0, np.nan], np.float32)
    assert cpu.argsort_desc(s).tolist() == [1, 9, 6, 2, 8, 0, 3, 4, 5, 7]


def test_proposal_edge_cases(cpu):
    r = np.random.default_rng(0)
    boxes = rand_boxes3d(np.stack([r.uniform(-30, 30, 500), np.ones(500), r.uniform(45, 70, 500)], 1), 500, seed=1)[None]
    sc = r.normal(size=(1, 500)).astype(np.float32)
    # nothing in area 1: the reference asserts (proposal_layer.py:90); the oracle returns area 2's survivors only
    rois, scores, cnt = cpu.proposal_layer(sc, boxes, (6300, 2700), (70, 30), 0.8, "normal")
    assert cnt[0] <= 30 and (rois[0, cnt[0]:] == 0).all() and (rois[0, :cnt[0], 2] > 40).all()
    assert (np.diff(scores[0, :cnt[0]]) <= 0).all()
    # threshold so low that everything overlapping is suppressed; score_based
    rois, scores, cnt = cpu.proposal_layer(sc, boxes, (500, 0), (100, 0), 0.0, "rotated", None)
    keep, num = cpu.nms_batched(boxes, sc, None, 0.0, "rotated", 100)
    assert num[0] == cnt[0] and np.array_equal(boxes[0][keep[0, :num[0]]], rois[0, :cnt[0]])
    # valid mask: only the selected rows compete
    valid = (sc > 0.3)
    keep, num = cpu.nms_batched(boxes, sc, valid, 0.1, "rotated")
    k = keep[0, :num[0]]
    assert valid[0][k].all() and (np.diff(sc[0][k]) <= 0).all() and (keep[0, num[0]:] == -1).all()
    order = cpu.argsort_desc(sc[0])
    sel = order[valid[0][order]]
    from rcnn_bev import bev
    assert np.array_equal(sel[cpu.nms(bev(boxes[0][sel]), 0.1, "rotated")], k)
