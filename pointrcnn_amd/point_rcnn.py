"""Host-side mirror of lib/net/point_rcnn.py:8-71 (PointRCNN: RPN -> proposal layer -> RCNN) at inference, plus the
detection post-processing of tools/eval_rcnn.py:505-614 as batched device code.

`PointRCNN.forward` returns the reference's output dictionary (same keys / shapes).  `detections` replaces the
per-frame Python loop of eval_one_epoch_joint (decode_bbox_target with the RCNN bins, score threshold, rotated NMS
0.1) by two launches for the whole batch; its outputs stay on the device, padded, with per-frame counts.
"""
import torch
import torch.nn as nn

from . import ops
from .proposal_layer import CLS_MEAN_SIZE, ProposalConfig, ProposalLayer, _anchor3
from .rcnn import RCNNConfig, RCNNNet
from .rpn import RPN, RPNConfig


class PointRCNN(nn.Module):
    def __init__(self, num_classes=2, use_xyz=True, mode="TEST", rpn_cfg=RPNConfig, rcnn_cfg=RCNNConfig,
                 proposal_cfg=ProposalConfig, rpn_score_thresh=0.3):
        super().__init__()
        self.rpn = RPN(use_xyz=use_xyz, cfg=rpn_cfg)
        self.rpn.proposal_layer = ProposalLayer(mode=mode, cfg=proposal_cfg)      # lib/net/rpn.py:48
        self.rcnn_net = RCNNNet(num_classes=num_classes, input_channels=128, use_xyz=use_xyz, cfg=rcnn_cfg)
        self.rpn_score_thresh = rpn_score_thresh                                  # cfg.RPN.SCORE_THRESH
        self.rcnn_cfg = rcnn_cfg

    def forward(self, input_data):
        output = {}
        with torch.no_grad():                                                     # point_rcnn.py:30-52 (RPN.FIXED)
            if self.training:
                self.rpn.eval()                                                   # :31-32: a fixed RPN stays in eval mode
            rpn_output = self.rpn(input_data)
            output.update(rpn_output)
            rpn_cls, rpn_reg = rpn_output["rpn_cls"], rpn_output["rpn_reg"]
            backbone_xyz, backbone_features = rpn_output["backbone_xyz"], rpn_output["backbone_features"]
            rpn_scores_raw = rpn_cls[:, :, 0]
            rpn_scores_norm = torch.sigmoid(rpn_scores_raw)
            seg_mask = (rpn_scores_norm > self.rpn_score_thresh).float()
            pts_depth = torch.norm(backbone_xyz, p=2, dim=2)
            rois, roi_scores_raw = self.rpn.proposal_layer(rpn_scores_raw, rpn_reg, backbone_xyz)
            output["rois"], output["roi_scores_raw"], output["seg_result"] = rois, roi_scores_raw, seg_mask
        rcnn_input_info = {"rpn_xyz": backbone_xyz, "rpn_features": backbone_features.permute((0, 2, 1)),
                           "seg_mask": seg_mask, "roi_boxes3d": rois, "pts_depth": pts_depth}
        if self.training:
            rcnn_input_info["gt_boxes3d"] = input_data["gt_boxes3d"]              # point_rcnn.py:59-60
        output.update(self.rcnn_net(rcnn_input_info))
        return output

    def detections(self, ret_dict, score_thresh=0.3, nms_thresh=None):
        """tools/eval_rcnn.py:505-524,600-614 for the whole batch, on the device:
        -> pred_boxes3d (B,M,7), raw_scores (B,M), keep (B,M) int32 rows in kept order (-1 padded), num (B) int32"""
        cfg = self.rcnn_cfg
        rois = ret_dict["rois"]
        B, M = rois.shape[:2]
        rcnn_reg = ret_dict["rcnn_reg"].view(B * M, -1)
        pred = ops.decode_bbox_target(rois.view(-1, 7), rcnn_reg, cfg.LOC_SCOPE, cfg.LOC_BIN_SIZE, cfg.NUM_HEAD_BIN,
                                      _anchor3(CLS_MEAN_SIZE[0]), True, cfg.LOC_Y_BY_BIN, 0.5, 0.25, True).view(B, M, 7)
        raw = ret_dict["rcnn_cls"].view(B, M)
        valid = torch.sigmoid(raw) > score_thresh
        keep, num = ops.nms_batched(pred, raw.contiguous(), valid, cfg.NMS_THRESH if nms_thresh is None else nms_thresh, True)
        return pred, raw, keep, num
