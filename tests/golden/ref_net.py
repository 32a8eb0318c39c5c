"""Run the REFERENCE's own, unchanged network code (lib/net/point_rcnn.py, rpn.py, pointnet2_msg.py, rcnn_net.py,
lib/rpn/proposal_layer.py, lib/utils/{bbox_transform,kitti_utils,iou3d/iou3d_utils,roipool3d/roipool3d_utils}.py,
lib/net/train_functions.py + lib/utils/loss_utils.py) on the drop-in op surface -- in the BUILD CONTAINER, on CPU.

`import iou3d_cuda`, `import roipool3d_cuda` and `from pointnet2_lib.pointnet2 import ...` inside the reference resolve to
pointrcnn_amd/dropin (pointrcnn_amd.install(), exactly the route INTEGRATION.md describes); because this container has no
GPU, the operator entry points underneath are routed to the CPU oracle (tests/cpu_ops.py) and `.cuda()` is made a no-op.
What executes above the operators is the reference's Python, byte for byte.

Used to GENERATE tests/golden/net_ref.npz and train_loss_ref.npz (make_golden.py) -- /root/reference does not exist on the
GPU box, the fixtures travel -- and by the container-side tests that compare the mirrors (pointrcnn_amd/{rpn,rcnn,point_rcnn,
train_functions}.py) with the reference live.
"""
import os
import sys

import numpy as np
import torch

REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)


def available():
    return os.path.isdir(os.path.join(REFERENCE, "lib", "net"))


def set_reference(path):
    """point the loader at another copy of the reference tree (the GPU box: the archive staged by oracle/stage_reference.py,
    unpacked into a temporary directory); must be called before load()"""
    global REFERENCE
    assert _loaded is None or path == REFERENCE, "reference modules already imported from %s" % REFERENCE
    REFERENCE = path


def _host_only():
    """in the build container (no GPU) `.cuda()` has to be a no-op; on the MI355X nothing is patched: the reference's code runs
    as it is"""
    import contextlib
    import cpu_ops
    return contextlib.nullcontext() if torch.cuda.is_available() else cpu_ops.cuda_is_cpu()


_loaded = None


def load(cfg_file="tools/cfgs/default.yaml"):
    """-> module namespace with cfg, PointRCNN, train_functions, kitti_utils (the reference's own objects)"""
    global _loaded
    if _loaded is not None:
        return _loaded
    import pointrcnn_amd
    pointrcnn_amd.install()
    # tools/_init_path.py:1-4 puts the repo root, lib/datasets and lib/net on sys.path
    for p in (os.path.join(TESTS, "compat"), os.path.join(REFERENCE, "lib", "net"), os.path.join(REFERENCE, "lib", "datasets"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)      # lib/config.py:187 predates PyYAML 6
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REFERENCE, cfg_file))
    yaml.load = _load
    import types
    ns = types.SimpleNamespace(cfg=cfg)
    sys.dont_write_bytecode = True            # the reference checkout is read-only
    with _host_only():
        from lib.net.point_rcnn import PointRCNN
        import lib.net.train_functions as train_functions
        import lib.utils.kitti_utils as kitti_utils
        import lib.utils.iou3d.iou3d_utils as iou3d_utils
        import lib.utils.roipool3d.roipool3d_utils as roipool3d_utils
    ns.PointRCNN, ns.train_functions, ns.kitti_utils = PointRCNN, train_functions, kitti_utils
    ns.iou3d_utils, ns.roipool3d_utils = iou3d_utils, roipool3d_utils
    _loaded = ns
    return ns


def build_reference_model(mode="TEST", seed=5, rpn_only=False):
    """the reference's PointRCNN(num_classes=2, use_xyz=True, mode=mode) with parameters that depend only on their names"""
    import cpu_ops
    ns = load()
    ns.cfg.RPN.ENABLED, ns.cfg.RCNN.ENABLED = True, not rpn_only
    ns.cfg.RPN.FIXED = False
    with _host_only():
        model = ns.PointRCNN(num_classes=2, use_xyz=True, mode=mode)
    cpu_ops.fill_params_by_name(model, seed)
    return model


def run_reference(model, pts_input):
    """forward of the reference model on CPU through the oracle-backed drop-in operators"""
    import cpu_ops
    model.eval()
    with cpu_ops.oracle_ops(), cpu_ops.cuda_is_cpu(), torch.no_grad():
        return model({"pts_input": pts_input})


def train_loss_case(seed, B=2, N=2048, nfg=300):
    """seeded (rpn_cls, rpn_reg, rpn_cls_label, rpn_reg_label) for the loss fixtures: plausible regression targets for the
    foreground points, a band of ignored points"""
    r = np.random.default_rng(seed)
    rpn_cls = r.normal(0, 2, (B, N, 1)).astype(np.float32)
    rpn_reg = r.normal(0, 1, (B, N, 76)).astype(np.float32)
    label = np.zeros((B, N), np.int32)
    reg = np.zeros((B, N, 7), np.float32)
    for b in range(B):
        fg = r.choice(N, nfg, replace=False)
        label[b, fg[: nfg - 40]] = 1
        label[b, fg[nfg - 40:]] = -1
        k = fg[: nfg - 40]
        reg[b, k, 0:3] = r.uniform(-3.5, 3.5, (len(k), 3)) * [1, 0.3, 1]
        reg[b, k, 3:6] = r.uniform(0.8, 1.2, (len(k), 3)) * [1.52563191462, 1.62856739989, 3.88311640418]
        reg[b, k, 6] = r.uniform(-2 * np.pi, 2 * np.pi, len(k))
    return rpn_cls, rpn_reg, label, reg


def reference_rpn_loss(rpn_cls, rpn_reg, label, reg_label, loss_cls="SigmoidFocalLoss"):
    """the reference's own model_fn (train_functions.py:12-52 -> get_rpn_loss :54-127) on given network outputs:
    -> loss value, tb_dict, d loss / d rpn_cls, d loss / d rpn_reg"""
    import cpu_ops
    ns = load()
    cfg = ns.cfg
    cfg.RPN.ENABLED, cfg.RCNN.ENABLED, cfg.RPN.FIXED = True, False, False
    cfg.RPN.LOSS_CLS = loss_cls
    import lib.utils.loss_utils as loss_utils
    cls_t = torch.from_numpy(rpn_cls).requires_grad_(True)
    reg_t = torch.from_numpy(rpn_reg).requires_grad_(True)

    class FakeRPN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rpn_cls_loss_func = (loss_utils.DiceLoss(ignore_target=-1) if loss_cls == "DiceLoss" else
                                      loss_utils.SigmoidFocalClassificationLoss(alpha=cfg.RPN.FOCAL_ALPHA[0], gamma=cfg.RPN.FOCAL_GAMMA))

    class FakeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rpn = FakeRPN()

        def forward(self, input_data):
            return {"rpn_cls": cls_t, "rpn_reg": reg_t}

    B, N = label.shape
    data = {"pts_rect": np.zeros((B, N, 3), np.float32), "pts_features": np.zeros((B, N, 1), np.float32),
            "pts_input": np.zeros((B, N, 3), np.float32), "gt_boxes3d": np.zeros((B, 1, 7), np.float32),
            "rpn_cls_label": label, "rpn_reg_label": reg_label}
    with cpu_ops.cuda_is_cpu():
        model_fn = ns.train_functions.model_joint_fn_decorator()
        ret = model_fn(FakeModel(), data)
        ret.loss.backward()
    cfg.RPN.LOSS_CLS = "SigmoidFocalLoss"
    gz = lambda t: np.zeros(tuple(t.shape), np.float32) if t.grad is None else t.grad.numpy()      # noqa: E731 (no fg: reg unused)
    return float(ret.loss.item()), dict(ret.tb_dict), gz(cls_t), gz(reg_t)


def rcnn_loss_case(seed, R=128, nfg=40, nign=16):
    """seeded RCNN-stage network outputs and ProposalTargetLayer-style targets: (rcnn_cls (R,1), rcnn_reg (R,46), cls_label (R) in
    {-1,0,1}, reg_valid_mask (R), roi_boxes3d (R,7), gt_of_rois (R,7) in the RoI's canonical frame)"""
    r = np.random.default_rng(seed)
    rcnn_cls = r.normal(0, 2, (R, 1)).astype(np.float32)
    rcnn_reg = r.normal(0, 1, (R, 46)).astype(np.float32)
    cls_label = np.zeros((R,), np.int64)
    reg_valid = np.zeros((R,), np.int64)
    order = r.permutation(R)
    fg, ign = order[:nfg], order[nfg:nfg + nign]
    cls_label[fg] = 1
    cls_label[ign] = -1
    reg_valid[fg] = 1
    reg_valid[ign[: nign // 2]] = 1                      # IoU in (0.55, 0.6): regressed but ignored by the classifier
    rois = np.zeros((R, 7), np.float32)
    rois[:, 3:6] = r.uniform(0.8, 1.2, (R, 3)) * [1.52563191462, 1.62856739989, 3.88311640418]
    rois[:, 6] = r.uniform(-np.pi, np.pi, R)
    gt = np.zeros((R, 7), np.float32)
    gt[:, 0:3] = r.uniform(-1.4, 1.4, (R, 3)) * [1, 0.3, 1]
    gt[:, 3:6] = r.uniform(0.8, 1.2, (R, 3)) * [1.52563191462, 1.62856739989, 3.88311640418]
    gt[:, 6] = r.uniform(-np.pi / 3, np.pi / 3, R)
    return rcnn_cls, rcnn_reg, cls_label, reg_valid, rois, gt


def reference_rcnn_loss(rcnn_cls, rcnn_reg, cls_label, reg_valid, rois, gt, loss_cls="BinaryCrossEntropy"):
    """the reference's own get_rcnn_loss (train_functions.py:122-214, through model_fn with cfg.RPN.ENABLED = False) on given
    network outputs and targets: -> loss value, tb_dict, d loss / d rcnn_cls, d loss / d rcnn_reg"""
    import cpu_ops
    ns = load()
    cfg = ns.cfg
    keep = (cfg.RPN.ENABLED, cfg.RCNN.ENABLED, cfg.RCNN.LOSS_CLS)
    cfg.RPN.ENABLED, cfg.RCNN.ENABLED, cfg.RCNN.LOSS_CLS = False, True, loss_cls
    import lib.utils.loss_utils as loss_utils
    cls_t = torch.from_numpy(rcnn_cls).requires_grad_(True)
    reg_t = torch.from_numpy(rcnn_reg).requires_grad_(True)

    class FakeRCNN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cls_loss_func = (loss_utils.SigmoidFocalClassificationLoss(alpha=cfg.RCNN.FOCAL_ALPHA[0], gamma=cfg.RCNN.FOCAL_GAMMA)
                                  if loss_cls == "SigmoidFocalLoss" else torch.nn.functional.binary_cross_entropy)

    class FakeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rcnn_net = FakeRCNN()

        def forward(self, input_data):
            return {"rcnn_cls": cls_t, "rcnn_reg": reg_t, "cls_label": torch.from_numpy(cls_label), "reg_valid_mask": torch.from_numpy(reg_valid),
                    "roi_boxes3d": torch.from_numpy(rois), "gt_of_rois": torch.from_numpy(gt),
                    "pts_input": torch.zeros((rcnn_cls.shape[0], 4, 3))}

    # torch >= 1.? rejects binary_cross_entropy targets outside [0, 1]; the reference (torch 1.0) passes its ignore label -1 and
    # masks those entries afterwards (train_functions.py:159-161).  Clamping the target changes only the masked entries.
    F = torch.nn.functional
    orig_bce = F.binary_cross_entropy
    F.binary_cross_entropy = lambda inp, target, *a, **kw: orig_bce(inp, target.clamp(0, 1), *a, **kw)
    try:
        with cpu_ops.cuda_is_cpu():
            ret = ns.train_functions.model_joint_fn_decorator()(FakeModel(), {"pts_input": np.zeros((1, 4, 3), np.float32)})
            ret.loss.backward()
    finally:
        F.binary_cross_entropy = orig_bce
        cfg.RPN.ENABLED, cfg.RCNN.ENABLED, cfg.RCNN.LOSS_CLS = keep
    gz = lambda t: np.zeros(tuple(t.shape), np.float32) if t.grad is None else t.grad.numpy()      # noqa: E731
    return float(ret.loss.item()), dict(ret.tb_dict), gz(cls_t), gz(reg_t)
