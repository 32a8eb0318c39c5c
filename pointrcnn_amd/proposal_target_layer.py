"""RCNN-stage training targets -- host-side mirror of lib/rpn/proposal_target_layer.py (ProposalTargetLayer), the glue
`train_rcnn.py --train_mode rcnn` runs between the RPN's proposals and the RCNN network.

Same entry point and dictionaries as the reference: forward(input_dict) with roi_boxes3d (B,M,7), gt_boxes3d (B,G,7),
rpn_xyz (B,N,3), rpn_features (B,N,C), seg_mask (B,N), pts_depth (B,N) [, rpn_intensity] -> sampled_pts (B*R,512,3),
pts_feature (B*R,512,C'), cls_label, reg_valid_mask, gt_of_rois, gt_iou, roi_boxes3d.

What runs where:
  * RoI sampling and noise augmentation (sample_rois_for_rcnn, sample_bg_inds, aug_roi_by_noise_torch, random_aug_box3d,
    :75-300): ONE kernel launch for the batch (csrc/proposal_target.hip).  The reference spends ~700 blocking device round
    trips per frame here (a Python while-loop of 1 x 1 IoU launches per sampled RoI).
  * point pooling: prcnn_roipool3d (the reference's roipool3d_gpu).
  * the per-RoI rotation / scale / flip augmentation (:302-363), the canonical transform (:45-57) and the labels (:59-68) are a
    few dozen elementwise torch kernels on (B, R)-sized tensors and one pass over the pooled points; restated here without the
    reference's per-frame Python loops (its loop recomputes the same angles every iteration: the result after the last frame is
    what the vectorised form computes once).
Randomness: the sampler draws from a counter-based table (seed argument / `seed` attribute, advanced every call); the
elementwise augmentation draws from torch's generator as the reference does.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .rcnn import enlarge_box3d, rotate_pc_along_y_torch


class ProposalTargetConfig:
    """tools/cfgs/default.yaml: AUG_DATA, AUG_ROT_RANGE and the RCNN section's sampling parameters"""
    AUG_DATA = True
    AUG_ROT_RANGE = 18
    USE_INTENSITY = False
    USE_DEPTH = True
    POOL_EXTRA_WIDTH = 1.0
    NUM_POINTS = 512
    ROI_PER_IMAGE = 64
    FG_RATIO = 0.5
    HARD_BG_RATIO = 0.8
    REG_FG_THRESH = 0.55
    CLS_FG_THRESH = 0.6
    CLS_BG_THRESH = 0.45
    CLS_BG_THRESH_LO = 0.05
    ROI_FG_AUG_TIMES = 10
    REG_AUG_METHOD = "multiple"


def _alpha(boxes):
    """observation angle from (x, z, ry): -sign(beta) pi/2 + beta + ry (proposal_target_layer.py:314-320)"""
    beta = torch.atan2(boxes[:, :, 2], boxes[:, :, 0])
    return -torch.sign(beta) * math.pi / 2 + beta + boxes[:, :, 6]


class ProposalTargetLayer(nn.Module):
    def __init__(self, cfg=ProposalTargetConfig, seed=None):
        """seed None (default): every call keys the sampler's counter-based random table with a fresh draw from torch's global
        (host) generator -- as the reference's sampler, which draws from the numpy / torch global RNGs, it follows
        torch.manual_seed, differs between DDP ranks seeded differently and resumes with the generator state; an int: a
        deterministic counter (seed, seed + 1, ...) for tests.  (No registered buffer: the module must keep the reference's
        state-dict keys.)"""
        super().__init__()
        self.cfg, self.seed = cfg, None if seed is None else int(seed)

    def forward(self, input_dict):
        cfg = self.cfg
        roi_boxes3d, gt_boxes3d = input_dict["roi_boxes3d"], input_dict["gt_boxes3d"]
        batch_rois, batch_gt_of_rois, batch_roi_iou = self.sample_rois_for_rcnn(roi_boxes3d, gt_boxes3d)
        rpn_xyz, rpn_features = input_dict["rpn_xyz"], input_dict["rpn_features"]
        extra = [input_dict["rpn_intensity"].unsqueeze(dim=2), input_dict["seg_mask"].unsqueeze(dim=2)] if cfg.USE_INTENSITY \
            else [input_dict["seg_mask"].unsqueeze(dim=2)]
        if cfg.USE_DEPTH:
            extra.append((input_dict["pts_depth"] / 70.0 - 0.5).unsqueeze(dim=2))
        pts_feature = torch.cat(extra + [rpn_features], dim=2).contiguous()
        pooled, pooled_empty_flag = ops.roipool3d(rpn_xyz.contiguous(), enlarge_box3d(batch_rois, cfg.POOL_EXTRA_WIDTH).contiguous(),
                                                  pts_feature, cfg.NUM_POINTS)
        sampled_pts, sampled_features = pooled[:, :, :, 0:3], pooled[:, :, :, 3:]
        if cfg.AUG_DATA:
            sampled_pts, batch_rois, batch_gt_of_rois = self.data_augmentation(sampled_pts, batch_rois, batch_gt_of_rois)
        # canonical transformation (:45-57)
        B, R = batch_rois.shape[0], batch_rois.shape[1]
        roi_ry = batch_rois[:, :, 6] % (2 * math.pi)
        roi_center = batch_rois[:, :, 0:3]
        sampled_pts = sampled_pts - roi_center.unsqueeze(dim=2)
        batch_gt_of_rois = batch_gt_of_rois.clone()
        batch_gt_of_rois[:, :, 0:3] = batch_gt_of_rois[:, :, 0:3] - roi_center
        batch_gt_of_rois[:, :, 6] = batch_gt_of_rois[:, :, 6] - roi_ry
        sampled_pts = rotate_pc_along_y_torch(sampled_pts.reshape(B * R, cfg.NUM_POINTS, 3).clone(), batch_rois[:, :, 6].reshape(-1))
        batch_gt_of_rois = rotate_pc_along_y_torch(batch_gt_of_rois.reshape(B * R, 1, 7), roi_ry.reshape(-1)).reshape(B, R, 7)
        # labels (:59-68)
        valid_mask = pooled_empty_flag == 0
        reg_valid_mask = ((batch_roi_iou > cfg.REG_FG_THRESH) & valid_mask).long()
        batch_cls_label = (batch_roi_iou > cfg.CLS_FG_THRESH).long()
        invalid_mask = (batch_roi_iou > cfg.CLS_BG_THRESH) & (batch_roi_iou < cfg.CLS_FG_THRESH)
        batch_cls_label[valid_mask == 0] = -1
        batch_cls_label[invalid_mask > 0] = -1
        return {"sampled_pts": sampled_pts.view(-1, cfg.NUM_POINTS, 3),
                "pts_feature": sampled_features.reshape(-1, cfg.NUM_POINTS, sampled_features.shape[3]),
                "cls_label": batch_cls_label.view(-1), "reg_valid_mask": reg_valid_mask.view(-1),
                "gt_of_rois": batch_gt_of_rois.reshape(-1, 7), "gt_iou": batch_roi_iou.view(-1), "roi_boxes3d": batch_rois.reshape(-1, 7)}

    def sample_rois_for_rcnn(self, roi_boxes3d, gt_boxes3d, seed=None):
        """(B,M,7), (B,G,7) -> batch_rois (B,R,7), batch_gt_of_rois (B,R,7), batch_roi_iou (B,R); `self.last` keeps the sampler's
        other outputs (source RoI of every slot, candidate counts, status) on the device"""
        cfg = self.cfg
        if seed is None and self.seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())          # host generator: no device round trip
        elif seed is None:
            seed, self.seed = self.seed, self.seed + 1
        o = ops.proposal_target_sample(roi_boxes3d.contiguous(), gt_boxes3d.contiguous(), cfg.ROI_PER_IMAGE,
                                       (cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH, cfg.CLS_BG_THRESH, cfg.CLS_BG_THRESH_LO), cfg.FG_RATIO,
                                       cfg.HARD_BG_RATIO, cfg.ROI_FG_AUG_TIMES, cfg.REG_AUG_METHOD, seed)
        self.last = o
        return o["rois"], o["gt_of_rois"], o["roi_iou"]

    def data_augmentation(self, pts, rois, gt_of_rois):
        """(B,R,512,3), (B,R,7), (B,R,7) -> the same, rotated / scaled / flipped per RoI (:302-363)"""
        cfg = self.cfg
        B, R = pts.shape[0], pts.shape[1]
        dev = pts.device
        angles = (torch.rand((B, R), device=dev) - 0.5 / 0.5) * (math.pi / cfg.AUG_ROT_RANGE)      # (sic: the reference's precedence)
        gt_alpha, roi_alpha = _alpha(gt_of_rois), _alpha(rois)
        flat = angles.reshape(-1)
        pts = rotate_pc_along_y_torch(pts.reshape(B * R, -1, 3).clone(), flat).reshape(B, R, -1, 3)
        gt_of_rois = rotate_pc_along_y_torch(gt_of_rois.reshape(B * R, 1, 7).clone(), flat).reshape(B, R, 7)
        rois = rotate_pc_along_y_torch(rois.reshape(B * R, 1, 7).clone(), flat).reshape(B, R, 7)
        beta = torch.atan2(gt_of_rois[:, :, 2], gt_of_rois[:, :, 0])
        gt_of_rois[:, :, 6] = torch.sign(beta) * math.pi / 2 + gt_alpha - beta
        beta = torch.atan2(rois[:, :, 2], rois[:, :, 0])
        rois[:, :, 6] = torch.sign(beta) * math.pi / 2 + roi_alpha - beta
        scales = 1 + ((torch.rand((B, R), device=dev) - 0.5) / 0.5) * 0.05
        pts = pts * scales.unsqueeze(dim=2).unsqueeze(dim=3)
        gt_of_rois[:, :, 0:6] = gt_of_rois[:, :, 0:6] * scales.unsqueeze(dim=2)
        rois[:, :, 0:6] = rois[:, :, 0:6] * scales.unsqueeze(dim=2)
        flip = torch.sign(torch.rand((B, R), device=dev) - 0.5)
        pts[:, :, :, 0] = pts[:, :, :, 0] * flip.unsqueeze(dim=2)
        for boxes in (gt_of_rois, rois):
            boxes[:, :, 0] = boxes[:, :, 0] * flip
            ry = boxes[:, :, 6]
            boxes[:, :, 6] = (flip == 1).float() * ry + (flip == -1).float() * (torch.sign(ry) * math.pi - ry)
        return pts, rois, gt_of_rois
