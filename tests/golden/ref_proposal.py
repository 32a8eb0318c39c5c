"""Import the REFERENCE's own proposal stage (lib/utils/bbox_transform.py, lib/rpn/proposal_layer.py) on CPU.

Only used to GENERATE golden fixtures (make_golden.py) and by tests that run in the build container: it needs
/root/reference, which does not exist on the GPU box.  The two compiled-extension calls inside
lib/utils/iou3d/iou3d_utils.nms_{gpu,normal_gpu} are routed to oracle/_ref (the reference's iou3d sources compiled
for the host); everything else is the reference's Python, unmodified.
"""
import os
import sys
import types

import numpy as np
import torch

REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REFERENCE, "lib", "rpn"))


class _StayPut(torch.Tensor):
    """decode_bbox_target does anchor_size.to(roi.get_device()); get_device() is -1 for CPU tensors"""

    def to(self, *a, **k):
        return self


def load(cfg_file="tools/cfgs/default.yaml"):
    """-> (cfg, decode_bbox_target, make_proposal_layer(mode))"""
    import oracle
    ref = oracle.ref()
    if ref is None:
        raise RuntimeError("oracle/_ref is not built")
    for p in (os.path.join(os.path.dirname(HERE), "compat"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.modules.setdefault("iou3d_cuda", types.ModuleType("iou3d_cuda"))
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)      # lib/config.py:187 predates PyYAML 6
    import lib.utils.iou3d.iou3d_utils as iou3d_utils
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REFERENCE, cfg_file))
    yaml.load = _load

    def nms_via_ref(kind):
        def f(boxes, scores, thresh):                                          # iou3d_utils.py:56-87
            order = scores.sort(0, descending=True)[1]
            keep = ref.nms(boxes[order].numpy(), thresh, kind)
            return order[torch.from_numpy(keep)].contiguous()
        return f

    iou3d_utils.nms_gpu, iou3d_utils.nms_normal_gpu = nms_via_ref("rotated"), nms_via_ref("normal")
    from lib.utils.bbox_transform import decode_bbox_target
    from lib.rpn.proposal_layer import ProposalLayer

    def make_proposal_layer(mode):
        pl = ProposalLayer.__new__(ProposalLayer)                              # __init__ calls .cuda()
        torch.nn.Module.__init__(pl)
        pl.mode = mode
        pl.MEAN_SIZE = torch.from_numpy(cfg.CLS_MEAN_SIZE[0]).as_subclass(_StayPut)
        return pl

    def anchor():
        return torch.from_numpy(np.asarray(cfg.CLS_MEAN_SIZE[0], np.float32)).as_subclass(_StayPut)

    return cfg, decode_bbox_target, make_proposal_layer, anchor
