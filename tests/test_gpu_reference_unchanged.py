"""Row g1: the reference's OWN, UNCHANGED model code running on the HIP kernels ON THE GPU.

`north_star`: "exposed through the same pointnet2_lib / roipool3d / iou3d Python op surface so lib/net and tools/eval_rcnn.py
call it unchanged".  Here the reference tree (read-only; $PRCNN_REFERENCE, /root/reference, or the archive staged under
oracle/_ref by oracle/stage_reference.py for the GPU box -- never a committed copy) is put on sys.path exactly as
tools/_init_path.py:1-4 does, `lib/net/point_rcnn.py` is imported as it is, `PointRCNN(num_classes=2, use_xyz=True, mode='TEST')`
(tools/eval_rcnn.py:882) is built on pointrcnn_amd/dropin, moved to cuda:0 with the reference's own `.cuda()` and run.  Nothing is
patched: every `import iou3d_cuda` / `roipool3d_cuda` / `pointnet2_lib...` of the reference resolves to the drop-in modules, every
operator underneath is a C-ABI call into libprcnn_pointops.so.

Checked against
  * tests/golden/net_ref.npz -- the same reference code run in the build container on the CPU oracle (1e-5 * scale; identical
    RoI selection and segmentation mask);
  * the mirror pointrcnn_amd.point_rcnn.PointRCNN on the same parameters and clouds: backbone features BIT-IDENTICAL (same
    modules, same kernels), everything behind them within 1e-5 * scale (the mirror fuses the heads into one chain kernel, the
    proposal stage into four launches and the canonical transform into roipool3d).
"""
import os
import sys

import numpy as np
import pytest
import torch

from util import GOLDEN

sys.path.insert(0, GOLDEN)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def refnet(tmp_path_factory):
    from oracle import stage_reference
    import ref_net
    where = stage_reference.locate(tmp_path_factory.mktemp("reference"))
    if where is None:
        pytest.skip("no reference tree: neither $PRCNN_REFERENCE / /root/reference nor oracle/_ref/reference_py.tar.gz "
                    "(python -m oracle.stage_reference in the build container) exists on this box")
    ref_net.set_reference(where)
    ref_net.load()
    print("\n[g1] reference tree: %s" % where)
    return ref_net


def _tol(ref, k=1e-5):
    return k * max(1.0, float(np.abs(ref).max()))


def test_reference_modules_come_from_the_reference_tree_and_ops_from_the_dropin(dev, refnet):
    """what runs above the operators is the reference's file, what runs below is this package"""
    import inspect
    import pointrcnn_amd
    ns = refnet.load()
    src = inspect.getsourcefile(ns.PointRCNN)
    assert os.path.realpath(src).startswith(os.path.realpath(refnet.REFERENCE)), src
    import lib.net.rpn as ref_rpn
    import lib.net.rcnn_net as ref_rcnn
    import lib.rpn.proposal_layer as ref_pl
    for m in (ref_rpn, ref_rcnn, ref_pl, ns.iou3d_utils, ns.roipool3d_utils):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(refnet.REFERENCE)), m.__file__
    import iou3d_cuda
    import roipool3d_cuda
    import pointnet2_lib.pointnet2.pointnet2_modules as pm
    for m in (iou3d_cuda, roipool3d_cuda, pm):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(pointrcnn_amd.DROPIN_DIR)), m.__file__
    assert ns.iou3d_utils.iou3d_cuda is iou3d_cuda and ns.roipool3d_utils.roipool3d_cuda is roipool3d_cuda


def test_unchanged_reference_point_rcnn_on_hip_kernels(dev, refnet):
    from make_golden import NET_CASE, crc
    from pointrcnn_amd import rpn
    from pointrcnn_amd.point_rcnn import PointRCNN as Mirror
    import cpu_ops
    g = np.load(os.path.join(GOLDEN, "net_ref.npz"))
    c = NET_CASE
    model = refnet.build_reference_model("TEST", seed=c["wseed"])       # the reference's class, parameters by state-dict name
    sd = model.state_dict()
    assert sorted(sd) == g["keys"].tolist()
    model = model.cuda().eval()                                          # tools/eval_rcnn.py:883 `model.cuda()`
    clouds = rpn.synthetic_clouds(c["B"], c["N"], seed0=c["seed0"])
    assert crc(clouds.numpy()) == g["crc_in"]
    inputs = clouds.cuda(non_blocking=True).float()                      # eval_rcnn.py:477
    with torch.no_grad():
        out = model({"pts_input": inputs})
    assert all(out[k].is_cuda for k in ("rpn_cls", "rpn_reg", "rois", "rcnn_cls", "rcnn_reg"))

    # (1) against the same code on the CPU oracle (committed fixture)
    cls = out["rpn_cls"][:, :, 0].cpu().numpy()
    assert np.abs(cls - g["rpn_cls"]).max() <= _tol(g["rpn_cls"])
    reg = out["rpn_reg"].cpu().numpy()
    assert np.abs(reg[:, ::8] - g["rpn_reg_s8"]).max() <= _tol(g["rpn_reg_s8"])
    assert np.allclose(np.abs(reg.astype(np.float64)).sum((1, 2)), g["rpn_reg_abs_sum"], rtol=1e-5)
    feat = out["backbone_features"].cpu().numpy()
    assert np.abs(feat[:, :, ::16] - g["feat_s16"]).max() <= _tol(g["feat_s16"])
    assert np.array_equal(out["seg_result"].cpu().numpy(), g["seg_result"])
    rois = out["rois"].cpu().numpy()
    assert rois.shape == g["rois"].shape
    assert np.abs(out["roi_scores_raw"].cpu().numpy() - g["roi_scores_raw"]).max() <= _tol(g["roi_scores_raw"])
    assert np.abs(rois - g["rois"]).max() <= 2e-4
    assert np.abs(out["rcnn_cls"].cpu().numpy() - g["rcnn_cls"]).max() <= _tol(g["rcnn_cls"], 5e-5)
    assert np.abs(out["rcnn_reg"].cpu().numpy() - g["rcnn_reg"]).max() <= _tol(g["rcnn_reg"], 5e-5)

    # (2) against the mirror on the same parameters
    mirror = cpu_ops.fill_params_by_name(Mirror(mode="TEST"), c["wseed"]).to(dev).eval()
    assert sorted(mirror.state_dict()) == sorted(sd)
    for k in sd:
        assert torch.equal(mirror.state_dict()[k].cpu(), sd[k].cpu()), k
    with torch.no_grad():
        mo = mirror({"pts_input": inputs})
    assert torch.equal(mo["backbone_xyz"], out["backbone_xyz"])
    assert torch.equal(mo["backbone_features"], out["backbone_features"]), "same modules, same kernels: must be the same bits"
    for k in ("rpn_cls", "rpn_reg", "roi_scores_raw", "rcnn_cls", "rcnn_reg"):
        a, b = mo[k].float().cpu().numpy(), out[k].float().cpu().numpy()
        assert np.abs(a - b).max() <= _tol(b, 5e-5 if k.startswith("rcnn") else 1e-5), (k, float(np.abs(a - b).max()))
    assert torch.equal(mo["seg_result"], out["seg_result"])
    assert np.abs(mo["rois"].cpu().numpy() - rois).max() <= 2e-4


@pytest.mark.parametrize("nms_type", ["normal", "rotate"])
def test_unchanged_reference_proposal_layer_and_iou3d_utils_on_hip_kernels(dev, refnet, nms_type):
    """lib/rpn/proposal_layer.py:15-141 + lib/utils/iou3d/iou3d_utils.py:56-87 + lib/utils/bbox_transform.py, unchanged, on
    cuda tensors through `iou3d_cuda` == the batched device proposal stage (csrc/proposal.hip) and the committed fixture that the
    same reference code produced in the build container with the NMS calls routed to the reference's own compiled sources"""
    from proposal_cases import case_inputs, golden
    from pointrcnn_amd.proposal_layer import ProposalConfig, ProposalLayer
    ns = refnet.load()
    import lib.rpn.proposal_layer as ref_pl
    case = "test_normal" if nms_type == "normal" else "test_rotate"
    xyz, scores, reg, _ = case_inputs(case)
    cfg = ns.cfg
    keep_type = cfg.RPN.NMS_TYPE
    cfg.RPN.NMS_TYPE = nms_type
    try:
        layer = ref_pl.ProposalLayer(mode="TEST")
        with torch.no_grad():
            rois, raw = layer(torch.from_numpy(scores).cuda(), torch.from_numpy(reg).cuda(), torch.from_numpy(xyz).cuda())
    finally:
        cfg.RPN.NMS_TYPE = keep_type
    g = golden()
    # the same RoIs in the same order (scores are copied through: equal bits <=> identical selection); box coordinates to the
    # rounding of torch's device cos / sin in the reference's decode against torch's host ones in the fixture
    assert np.array_equal(raw.cpu().numpy(), g[case + "_scores"])
    assert np.abs(rois.cpu().numpy() - g[case + "_rois"]).max() <= 1e-4
    mine = ProposalLayer("TEST", cfg=type("Cfg", (ProposalConfig,), {"NMS_TYPE": nms_type}))
    with torch.no_grad():
        r2, s2 = mine(torch.from_numpy(scores).to(dev), torch.from_numpy(reg).to(dev), torch.from_numpy(xyz).to(dev))
    assert torch.equal(s2, raw)
    assert float((r2 - rois).abs().max()) <= 1e-4


def test_unchanged_reference_training_step_on_hip_kernels(dev, refnet):
    """lib/net/train_functions.py model_fn (:12-52) + get_rpn_loss on the reference's PointRCNN(mode='TRAIN'), unchanged, one
    forward/backward on cuda:0 -- the drop-in modules take their fused TRAINING path (hand-written forward / dgrad / wgrad
    kernels) underneath -- against train_ref.npz (the same step on the CPU oracle)"""
    from make_golden import TRAIN_CASE, crc, train_batch
    g = np.load(os.path.join(GOLDEN, "train_ref.npz"))
    c = TRAIN_CASE
    pts, gt, cls, reg = train_batch(c)
    assert crc(pts, gt, cls, reg) == g["step_crc"]
    ns = refnet.load()
    model = refnet.build_reference_model("TRAIN", seed=c["wseed"], rpn_only=True).cuda()
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    data = {"pts_rect": pts, "pts_features": np.zeros((c["B"], c["N"], 1), np.float32), "pts_input": pts, "gt_boxes3d": gt,
            "rpn_cls_label": cls, "rpn_reg_label": reg}
    ret = ns.train_functions.model_joint_fn_decorator()(model, data)
    ret.loss.backward()
    want = float(g["step_loss"])
    assert abs(float(ret.loss.item()) - want) <= 1e-5 * max(1.0, abs(want)), (float(ret.loss.item()), want)
    params = dict(model.named_parameters())
    names = g["step_names"].tolist()
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(names)
    rel = {n: abs(float(params[n].grad.double().norm()) - float(g["step_gnorm"][i])) / float(g["step_gnorm"][i]) for i, n in enumerate(names)}
    heads = [n for n in names if "rpn_cls_layer" in n or "rpn_reg_layer" in n]
    print("\n[reference training step on the HIP kernels] gradient-norm deviation: heads max %.2e, all max %.2e, median %.2e" % (max(rel[n] for n in heads), max(rel.values()), float(np.median(list(rel.values())))))
    # measured (round 5, the training kernels are bit-repeatable): heads 1.1e-5, median over all parameters 3.0e-5, largest 1.7e-2 -- parameters
    # whose gradient passes through max-pools: the CPU route of the golden run and the GPU route may resolve near-ties of a pooled maximum to
    # different rows (both valid sub-gradients), which moves single norms by up to that much and leaves the median where it is
    assert max(rel[n] for n in heads) <= 5e-5, [(n, rel[n]) for n in heads]
    assert max(rel.values()) <= 3e-2 and float(np.median(list(rel.values()))) <= 2e-4, max(rel.items(), key=lambda kv: kv[1])
