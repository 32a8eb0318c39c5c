"""RCNN-stage training (`train_rcnn.py --train_mode rcnn`, lib/net/rcnn_net.py with cfg.RCNN.USE_BN = False) on the hand-written
training stacks: layers WITHOUT normalisation (Conv with bias -> ReLU), the GroupAll level, the (B, C, N, 1) SharedMLPs, the heads
on rows -- each against the same modules run module by module through torch (PRCNN_TRAIN_FUSED off), and the whole training step
of the PointRCNN mirror (fixed RPN, device ProposalTargetLayer) fused vs composed.  Tolerances as tests/test_gpu_train_mlp.py."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from test_gpu_train_mlp import _close, _close_pooled, _compare_modules

pytestmark = pytest.mark.gpu


def _pm():
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm, pytorch_utils as pt
    return pm, pt


def _both(pm, fused, ref, run):
    a = run(fused)
    pm.TRAIN_FUSED = False
    try:
        b = run(ref)
    finally:
        pm.TRAIN_FUSED = True
    return a, b


@pytest.mark.parametrize("chans", [[5, 128, 128], [256, 128], [128, 64, 32]])
def test_shared_mlp_without_batchnorm_on_point_rows(dev, chans):
    """pt_utils.SharedMLP(bn=False) on a (B, C, N, 1) tensor (rcnn_net.py xyz_up_layer / merge_down_layer)"""
    pm, pt = _pm()
    from pointrcnn_amd import train_mlp
    torch.manual_seed(len(chans) + chans[0])
    fused = pt.SharedMLP(chans, bn=False).to(dev).train()
    with torch.no_grad():
        for m in fused.modules():
            if isinstance(m, nn.Conv2d):
                m.bias.normal_(0, 0.3)
    ref = copy.deepcopy(fused)
    assert train_mlp.stack_ok(fused.layers())
    x = torch.randn(3, chans[0], 700, 1, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = _both(pm, fused, ref, lambda m: m(xa if m is fused else xb))
    _close(ya, yb, 1e-5, "output")
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    _close(xa.grad, xb.grad, 1e-4, "input gradient")
    _compare_modules(fused, ref, pooled=False)


@pytest.mark.parametrize("npoint", [48, None])
def test_sa_module_without_batchnorm_and_group_all(dev, npoint):
    """PointnetSAModule(bn=False) as the RCNN stage builds it (radius ball or GroupAll, nsample 64 -> here 16 / all)"""
    pm, pt = _pm()
    torch.manual_seed(7)
    kw = dict(npoint=npoint, radius=0.4 if npoint else None, nsample=16 if npoint else None, mlp=[32, 32, 64], use_xyz=True, bn=False)
    fused = pm.PointnetSAModule(**kw).to(dev).train()
    ref = copy.deepcopy(fused)
    B, N = 4, 200 if npoint else 60
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = torch.randn(B, 32, N, generator=g).to(dev)
    fa, fb = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    assert fused._train_ok(xyz, fa)
    (nxa, oa), (nxb, ob) = _both(pm, fused, ref, lambda m: m(xyz, fa if m is fused else fb))
    assert (nxa is None and nxb is None) or torch.equal(nxa, nxb)
    _close(oa, ob, 1e-5, "pooled features")
    go = torch.randn(ob.shape, generator=g).to(dev)
    oa.backward(go)
    ob.backward(go)
    _close_pooled(fa.grad, fb.grad, "feature gradient")
    _compare_modules(fused, ref, pooled=True)


def _rcnn_batch(dev, B=2, N=16384):
    from pointrcnn_amd import rpn
    pts = rpn.synthetic_clouds(B, N, seed0=300).to(dev)
    g = torch.Generator().manual_seed(5)
    pick = torch.randint(0, N, (B, 10), generator=g).to(dev)
    ctr = torch.gather(pts, 1, pick[..., None].expand(-1, -1, 3))
    hwl = torch.tensor([1.56, 1.6, 3.9], device=dev)
    ry = (torch.rand((B, 10, 1), generator=g) * 6.283 - 3.1416).to(dev)
    gt = torch.cat([ctr[..., 0:1], ctr[..., 1:2] + hwl[0] / 2, ctr[..., 2:3], hwl.expand(B, 10, 3), ry], 2).contiguous()
    return {"pts_input": pts, "gt_boxes3d": gt}


def test_rcnn_training_step_fused_equals_composed(dev):
    """PointRCNN mirror, mode TRAIN: fixed RPN (eval, no gradient) -> proposals -> device ProposalTargetLayer -> RCNNNet -> the
    reference's RCNN loss -> backward.  Hand-written stacks vs the composed torch path on the same sampled RoIs: loss 1e-5,
    every rcnn_net parameter gradient within the pooled-stack bar; the RPN takes no gradient"""
    pm, pt = _pm()
    from pointrcnn_amd import point_rcnn, rpn, train_functions as tf
    torch.manual_seed(11)
    model = point_rcnn.PointRCNN(mode="TRAIN").to(dev)
    rpn.randomize_bn_stats(model.rpn, seed=3)
    ref = copy.deepcopy(model)
    batch = _rcnn_batch(dev)

    def run(m):
        m.train()
        m.rcnn_net.proposal_target_layer.seed = 21
        torch.manual_seed(77)                                  # the elementwise RoI augmentation draws from torch's generator
        out = m(batch)
        loss = tf.get_rcnn_loss(out, m.rcnn_cfg)
        loss.backward()
        return out, loss

    (oa, la), (ob, lb) = _both(pm, model, ref, run)
    assert torch.equal(oa["roi_boxes3d"], ob["roi_boxes3d"]) and torch.equal(oa["cls_label"], ob["cls_label"])
    assert oa["rcnn_cls"].shape == (2 * 64, 1) and oa["rcnn_reg"].shape[0] == 2 * 64
    assert abs(la.item() - lb.item()) <= 1e-5 * max(1.0, abs(lb.item())), (la.item(), lb.item())
    pa, pb = dict(model.rcnn_net.named_parameters()), dict(ref.rcnn_net.named_parameters())
    worst = 0.0
    for n in pa:
        assert pa[n].grad is not None and pb[n].grad is not None, n
        d = (pa[n].grad - pb[n].grad).double().norm().item() / max(1e-12, pb[n].grad.double().norm().item())
        worst = max(worst, d)
    assert worst <= 5e-3, worst
    assert all(p.grad is None for p in model.rpn.parameters())
    # the trainer: a few steps reduce the loss on the same batch, no host synchronisation inside
    tr = tf.RCNNTrainer(model)
    l0 = float(tr.step(batch).item())
    for _ in range(5):
        l = tr.step(batch)
    assert np.isfinite(float(l.item())) and np.isfinite(l0)
