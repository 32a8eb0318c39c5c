"""Host-side mirror of the reference's RCNN inference stage, built from the drop-in modules.

Mirrors the eval branch of lib/net/rcnn_net.py:115-190 (RCNNNet.forward with ROI_SAMPLE_JIT) with the shapes of
tools/cfgs/default.yaml:67-126: point features [seg mask, depth, 128 RPN features] -> roipool3d (M RoIs x 512 pts)
-> canonical transform -> xyz_up_layer / merge_down_layer SharedMLPs -> 3 PointnetSAModule (npoint 128 / 32 /
GroupAll, nsample 64) -> cls / reg Conv1d heads.  Attribute names follow the reference (`SA_modules`,
`xyz_up_layer`, `merge_down_layer`, `cls_layer`, `reg_layer`) so `rcnn_net.*` checkpoint entries load.

The two pieces of torch glue it needs are restated from lib/utils/kitti_utils.py (enlarge_box3d :150-160,
rotate_pc_along_y_torch :45-63, boxes3d_to_bev_torch :134-147); they are elementwise torch, outside the hot path
(SURVEY.md 8(a) a16); the inference path does not use them (prcnn_roipool3d_canonical builds the stage's inputs in one
kernel).  RoIs come from pointrcnn_amd/proposal_layer.py (see pointrcnn_amd/point_rcnn.py for the whole two-stage graph).
"""
import os

import torch
import torch.nn as nn

import pointrcnn_amd

pointrcnn_amd.install()
from pointnet2_lib.pointnet2.pointnet2_modules import PointnetSAModule  # noqa: E402
import pointnet2_lib.pointnet2.pytorch_utils as pt_utils  # noqa: E402
from pointnet2_lib.pointnet2 import pointnet2_modules as pn2_modules  # noqa: E402
from . import ops  # noqa: E402

# RoI duplicate elimination in the fused inference path (see _forward_fused); 0 = A/B switch, same bits
ROI_DEDUP = os.environ.get("PRCNN_ROI_DEDUP", "1") != "0"


class RCNNConfig:
    """tools/cfgs/default.yaml:67-126 (RCNN section)"""
    USE_BN = False
    USE_MASK = True
    USE_DEPTH = True
    USE_INTENSITY = False
    POOL_EXTRA_WIDTH = 1.0
    NUM_POINTS = 512
    XYZ_UP_LAYER = [128, 128]
    SA_NPOINTS = [128, 32, -1]
    SA_RADIUS = [0.2, 0.4, 100]
    SA_NSAMPLE = [64, 64, 64]
    SA_MLPS = [[128, 128, 128], [128, 128, 256], [256, 256, 512]]
    CLS_FC = [256, 256]
    REG_FC = [256, 256]
    DP_RATIO = 0.0
    LOC_SCOPE = 1.5
    LOC_BIN_SIZE = 0.5
    NUM_HEAD_BIN = 9
    LOC_Y_BY_BIN = False
    LOC_Y_SCOPE = 0.5
    LOC_Y_BIN_SIZE = 0.25
    SIZE_RES_ON_ROI = False
    NMS_THRESH = 0.1
    # training (tools/cfgs/default.yaml:111-126 + AUG_DATA / AUG_ROT_RANGE of the root)
    LOSS_CLS = "BinaryCrossEntropy"
    FOCAL_ALPHA = [0.25, 0.75]
    FOCAL_GAMMA = 2.0
    ROI_SAMPLE_JIT = True
    ROI_PER_IMAGE = 64
    FG_RATIO = 0.5
    HARD_BG_RATIO = 0.8
    REG_FG_THRESH = 0.55
    CLS_FG_THRESH = 0.6
    CLS_BG_THRESH = 0.45
    CLS_BG_THRESH_LO = 0.05
    ROI_FG_AUG_TIMES = 10
    REG_AUG_METHOD = "multiple"
    AUG_DATA = True
    AUG_ROT_RANGE = 18


def enlarge_box3d(boxes3d, extra_width):
    """lib/utils/kitti_utils.py:150-160"""
    large = boxes3d.clone()
    large[..., 3:6] += extra_width * 2
    large[..., 1] += extra_width
    return large


def rotate_pc_along_y_torch(pc, rot_angle):
    """lib/utils/kitti_utils.py:45-63: pc (N,S,3+C) rotated in the x-z plane by rot_angle (N)"""
    cosa, sina = torch.cos(rot_angle).view(-1, 1), torch.sin(rot_angle).view(-1, 1)
    R = torch.cat((torch.cat([cosa, -sina], 1).unsqueeze(1), torch.cat([sina, cosa], 1).unsqueeze(1)), 1)
    pc[:, :, [0, 2]] = torch.matmul(pc[:, :, [0, 2]], R.permute(0, 2, 1))
    return pc


def boxes3d_to_bev_torch(boxes3d):
    """lib/utils/kitti_utils.py:134-147: (N,7) [x,y,z,h,w,l,ry] -> (N,5) [x1,y1,x2,y2,ry]"""
    bev = boxes3d.new_empty((boxes3d.shape[0], 5))
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    bev[:, 0], bev[:, 1] = cu - half_l, cv - half_w
    bev[:, 2], bev[:, 3] = cu + half_l, cv + half_w
    bev[:, 4] = boxes3d[:, 6]
    return bev


def roipool3d_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512):
    """lib/utils/roipool3d/roipool3d_utils.py:7-28 (same name / arguments / returns)"""
    B = pts.shape[0]
    pooled_boxes3d = enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(B, -1, 7)
    return ops.roipool3d(pts.contiguous(), pooled_boxes3d.contiguous(), pts_feature.contiguous(), sampled_pt_num)


def nms_gpu(boxes_bev, scores, thresh):
    """lib/utils/iou3d/iou3d_utils.py:56-70 without the host round trip: kept indices (device) in score order"""
    order = scores.sort(0, descending=True)[1]
    keep, num = ops.nms_sorted(boxes_bev[order].contiguous(), thresh, rotated=True)
    return order[keep[: int(num.item())]].contiguous()


class RCNNNet(nn.Module):
    def __init__(self, num_classes=2, input_channels=128, use_xyz=True, cfg=RCNNConfig):
        super().__init__()
        self.cfg = cfg
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels
        self.rcnn_input_channel = 3 + int(cfg.USE_INTENSITY) + int(cfg.USE_MASK) + int(cfg.USE_DEPTH)
        self.xyz_up_layer = pt_utils.SharedMLP([self.rcnn_input_channel] + cfg.XYZ_UP_LAYER, bn=cfg.USE_BN)
        c_out = cfg.XYZ_UP_LAYER[-1]
        self.merge_down_layer = pt_utils.SharedMLP([c_out * 2, c_out], bn=cfg.USE_BN)
        for k in range(len(cfg.SA_NPOINTS)):
            mlps = [channel_in] + cfg.SA_MLPS[k]
            npoint = cfg.SA_NPOINTS[k] if cfg.SA_NPOINTS[k] != -1 else None
            self.SA_modules.append(PointnetSAModule(npoint=npoint, radius=cfg.SA_RADIUS[k], nsample=cfg.SA_NSAMPLE[k],
                                                    mlp=mlps, use_xyz=use_xyz, bn=cfg.USE_BN))
            channel_in = mlps[-1]
        cls_channel = 1 if num_classes == 2 else num_classes

        def head(fc, out_ch):
            layers, pre = [], channel_in
            for k in range(len(fc)):
                layers.append(pt_utils.Conv1d(pre, fc[k], bn=cfg.USE_BN))
                pre = fc[k]
            layers.append(pt_utils.Conv1d(pre, out_ch, activation=None))
            if cfg.DP_RATIO >= 0:
                layers.insert(1, nn.Dropout(cfg.DP_RATIO))
            return nn.Sequential(*layers)

        from .proposal_target_layer import ProposalTargetLayer
        self.proposal_target_layer = ProposalTargetLayer(cfg)                    # rcnn_net.py:78 (ROI_SAMPLE_JIT, training only)
        self.cls_layer = head(cfg.CLS_FC, cls_channel)
        per_loc_bin_num = int(cfg.LOC_SCOPE / cfg.LOC_BIN_SIZE) * 2
        reg_channel = per_loc_bin_num * 4 + cfg.NUM_HEAD_BIN * 2 + 3 + 1
        self.reg_channel = reg_channel
        self.reg_layer = head(cfg.REG_FC, reg_channel)
        for m in self.modules():                       # rcnn_net.py:86-104 (xavier)
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def pool_rois(self, input_data):
        """rcnn_net.py:127-154: build per-point features, pool RoIs, canonical transform -> (B*M, 512, 133)"""
        cfg = self.cfg
        rpn_xyz, rpn_features = input_data["rpn_xyz"], input_data["rpn_features"]
        batch_rois = input_data["roi_boxes3d"]
        extra = [input_data["seg_mask"].unsqueeze(2)]
        if cfg.USE_DEPTH:
            extra.append((input_data["pts_depth"] / 70.0 - 0.5).unsqueeze(2))
        pts_feature = torch.cat(extra + [rpn_features], dim=2)
        pooled, empty = roipool3d_gpu(rpn_xyz, pts_feature, batch_rois, cfg.POOL_EXTRA_WIDTH, cfg.NUM_POINTS)
        B = batch_rois.shape[0]
        pooled[:, :, :, 0:3] -= batch_rois[:, :, 0:3].unsqueeze(2)
        for k in range(B):
            pooled[k, :, :, 0:3] = rotate_pc_along_y_torch(pooled[k, :, :, 0:3], batch_rois[k, :, 6])
        return pooled.view(-1, pooled.shape[2], pooled.shape[3]), empty

    def _fused_ok(self, input_data):
        x = input_data["rpn_xyz"]
        return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and not self.cfg.USE_INTENSITY \
            and self.cfg.USE_MASK and self.xyz_up_layer.fusable() and self.merge_down_layer.fusable()

    def _forward_fused(self, input_data):
        """Inference fast path of rcnn_net.py:127-190.  One kernel builds the stage's inputs (prcnn_roipool3d_canonical:
        per-point channels read where they are, canonical transform applied while gathering, pooled RPN features written
        straight into the second half of merge_down_layer's input rows), so the (B,N,130) concat, the (B,M,512,133)
        pooled tensor, the per-frame rotate loop and the (B*M,256,512) concat of the reference never exist."""
        cfg = self.cfg
        rpn_xyz, rois = input_data["rpn_xyz"].contiguous(), input_data["roi_boxes3d"].contiguous()
        B, M = rois.shape[:2]
        S = cfg.NUM_POINTS
        extras = [input_data["seg_mask"]]
        if cfg.USE_DEPTH:
            extras.append(input_data["pts_depth"] / 70.0 - 0.5)
        feat_cl = pt_utils._rows_view(input_data["rpn_features"])
        c_up, C = cfg.XYZ_UP_LAYER[-1], feat_cl.shape[-1]
        merged_in = torch.empty((B * M * S, c_up + C), dtype=torch.float32, device=rpn_xyz.device)
        pool_boxes = enlarge_box3d(rois.view(-1, 7), cfg.POOL_EXTRA_WIDTH).view(B, M, 7)
        # RoI duplicate elimination: an RoI holding fewer than S points is padded by roipool3d with copies of its first rows.
        # The per-point layers skip the copies (seg), the first SA level never gathers them (valid_n) -- same outputs.
        sa0 = self.SA_modules[0]
        dedup = ROI_DEDUP and pn2_modules.GROUP_DEDUP and S % 128 == 0 and sa0.npoint is not None and \
            all(g.nsample in pn2_modules._POOL_FUSED for g in sa0.groupers)
        if dedup:
            pts, _, empty, distinct = ops.roipool3d_canonical(rpn_xyz, pool_boxes, rois, extras, feat_cl, S, out_feat=(merged_in, c_up),
                                                              want_distinct=True)
            seg = (distinct.view(-1), S)
        else:
            pts, _, empty = ops.roipool3d_canonical(rpn_xyz, pool_boxes, rois, extras, feat_cl, S, out_feat=(merged_in, c_up))
            seg = None
        up = [m.packed() for m in self.xyz_up_layer.layers()]
        if len(up) > 1 and ops.chain_supported(0, up, 0):
            ops.mlp_chain_rows(pts, up, out=(merged_in, 0), seg=seg)
        else:
            x = pts
            for li, lin in enumerate(up):
                x = ops.mlp_rows(x, lin, out=(merged_in, 0) if li == len(up) - 1 else None, seg=seg)
        x = merged_in
        for m in self.merge_down_layer.layers():
            x = ops.mlp_rows(x, m.packed(), seg=seg)
        xyz0 = pts[..., 0:3]
        if dedup:
            xyz0 = xyz0.contiguous()
            xyz0._prcnn_valid_n = seg[0]
        l_xyz, l_features = [xyz0], [x.view(B * M, S, -1).transpose(1, 2)]
        for i in range(len(self.SA_modules)):
            li_xyz, li_features = self.SA_modules[i](l_xyz[i], l_features[i])
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        rcnn_cls = pt_utils.fused_sequential(self.cls_layer, l_features[-1]).transpose(1, 2).contiguous().squeeze(1)
        rcnn_reg = pt_utils.fused_sequential(self.reg_layer, l_features[-1]).transpose(1, 2).contiguous().squeeze(1)
        return {"rcnn_cls": rcnn_cls, "rcnn_reg": rcnn_reg, "pooled_empty_flag": empty}

    def forward(self, input_data):
        target_dict = None
        if self.training and self.cfg.ROI_SAMPLE_JIT:
            # rcnn_net.py:120-126: RoI sampling, noise augmentation, pooling, canonical transform, labels (device sampler:
            # proposal_target_layer.py) -- no gradient; the network below trains on its (B * 64, 512, 3 + C') rows
            with torch.no_grad():
                target_dict = self.proposal_target_layer(input_data)
            pts_input = torch.cat((target_dict["sampled_pts"], target_dict["pts_feature"]), dim=2)
            target_dict["pts_input"] = pts_input
            empty = None
        elif self._fused_ok(input_data):
            return self._forward_fused(input_data)
        else:
            pts_input, empty = self.pool_rois(input_data)
        xyz = pts_input[..., 0:3].contiguous()
        c = self.rcnn_input_channel
        xyz_input = pts_input[..., 0:c].transpose(1, 2).unsqueeze(3)
        xyz_feature = self.xyz_up_layer(xyz_input)
        rpn_feature = pts_input[..., c:].transpose(1, 2).unsqueeze(3)
        merged = self.merge_down_layer(torch.cat((xyz_feature, rpn_feature), dim=1))
        l_xyz, l_features = [xyz], [merged.squeeze(3)]
        for i in range(len(self.SA_modules)):
            li_xyz, li_features = self.SA_modules[i](l_xyz[i], l_features[i])
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        rcnn_cls = pt_utils.fused_sequential(self.cls_layer, l_features[-1]).transpose(1, 2).contiguous().squeeze(1)
        rcnn_reg = pt_utils.fused_sequential(self.reg_layer, l_features[-1]).transpose(1, 2).contiguous().squeeze(1)
        ret = {"rcnn_cls": rcnn_cls, "rcnn_reg": rcnn_reg, "pooled_empty_flag": empty, "pts_input": pts_input}
        if target_dict is not None:
            ret.update(target_dict)                                              # rcnn_net.py:186-188
        return ret
