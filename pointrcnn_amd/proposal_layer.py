"""Host-side mirror of the reference's proposal stage on the batched device kernels (csrc/proposal.hip).

Mirrors  lib/utils/bbox_transform.py:24-121  decode_bbox_target   (same name, arguments, return)
         lib/rpn/proposal_layer.py:9-141     ProposalLayer        (same constructor / forward contract)
The reference loops over frames in Python and synchronises the host several times per frame (boolean-mask indexing,
two blocking NMS calls); here one forward is 4 kernel launches for the whole batch and never touches the host, so it
can be captured in a hipGraph together with the RPN.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

CLS_MEAN_SIZE = np.array([[1.52563191462, 1.62856739989, 3.88311640418]], dtype=np.float32)   # default.yaml:19 (Car)


class ProposalConfig:
    """The cfg entries ProposalLayer reads (tools/cfgs/default.yaml:31-34,61,156-166)."""
    LOC_XZ_FINE = True
    LOC_SCOPE = 3.0
    LOC_BIN_SIZE = 0.5
    NUM_HEAD_BIN = 12
    NMS_TYPE = "normal"
    TRAIN = dict(RPN_PRE_NMS_TOP_N=9000, RPN_POST_NMS_TOP_N=512, RPN_NMS_THRESH=0.85, RPN_DISTANCE_BASED_PROPOSE=True)
    TEST = dict(RPN_PRE_NMS_TOP_N=9000, RPN_POST_NMS_TOP_N=100, RPN_NMS_THRESH=0.8, RPN_DISTANCE_BASED_PROPOSE=True)


def _anchor3(anchor_size):
    a = anchor_size.detach().cpu().numpy() if isinstance(anchor_size, torch.Tensor) else np.asarray(anchor_size)
    a = a.reshape(-1)
    if a.size != 3:
        raise ValueError("anchor_size must hold 3 values (h, w, l), got shape %s" % (tuple(np.shape(anchor_size)),))
    return [float(v) for v in a]


def decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size, get_xz_fine=True,
                       get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=False):
    """lib/utils/bbox_transform.py:24-121 -- (N,3|7), (N,C) -> (N,7) [x,y,z,h,w,l,ry] in one kernel.
    Like the reference, raises when C does not match the bin layout (it asserts, :106)."""
    return ops.decode_bbox_target(roi_box3d.contiguous(), pred_reg.contiguous(), loc_scope, loc_bin_size, num_head_bin,
                                  _anchor3(anchor_size), get_xz_fine, get_y_by_bin, loc_y_scope, loc_y_bin_size, get_ry_fine)


class ProposalLayer(nn.Module):
    def __init__(self, mode="TRAIN", cfg=ProposalConfig, mean_size=CLS_MEAN_SIZE):
        super().__init__()
        self.mode = mode
        self.cfg = cfg
        self.MEAN_SIZE = _anchor3(mean_size[0])               # host floats: the kernel takes the anchor by value

    def forward(self, rpn_scores, rpn_reg, xyz):
        """rpn_scores (B,N) raw logits, rpn_reg (B,N,C), xyz (B,N,3) -> ret_bbox3d (B,M,7), ret_scores (B,M)
        (proposal_layer.py:15-56; M = RPN_POST_NMS_TOP_N, zero padded)"""
        cfg, m = self.cfg, getattr(self.cfg, self.mode)
        B, N = rpn_scores.shape
        proposals = ops.decode_bbox_target(xyz.reshape(-1, 3), rpn_reg.reshape(-1, rpn_reg.shape[-1]), cfg.LOC_SCOPE,
                                           cfg.LOC_BIN_SIZE, cfg.NUM_HEAD_BIN, self.MEAN_SIZE, cfg.LOC_XZ_FINE, False,
                                           y_to_bottom=True).view(B, N, 7)      # :23-33 incl. y -> bottom centre
        pre, post = m["RPN_PRE_NMS_TOP_N"], m["RPN_POST_NMS_TOP_N"]
        if m["RPN_DISTANCE_BASED_PROPOSE"]:                   # :58-117
            if cfg.NMS_TYPE not in ("rotate", "normal"):
                raise NotImplementedError(cfg.NMS_TYPE)
            pre1, post1 = int(pre * 0.7), int(post * 0.7)     # :66,68
            rois, scores, _ = ops.proposal_layer(rpn_scores.contiguous(), proposals, (pre1, pre - pre1), (post1, post - post1),
                                                 m["RPN_NMS_THRESH"], rotated=cfg.NMS_TYPE == "rotate", ranges=(0.0, 40.0, 80.0))
        else:                                                 # score_based_proposal always uses nms_gpu (:135)
            rois, scores, _ = ops.proposal_layer(rpn_scores.contiguous(), proposals, (pre, 0), (post, 0), m["RPN_NMS_THRESH"],
                                                 rotated=True, ranges=None)
        return rois, scores
