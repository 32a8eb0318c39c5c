"""`pointnet2_lib.pointnet2.pointnet2_utils` -- the PointNet++ operator surface the reference's modules are
built from [UPSTREAM sshaoshuai/Pointnet2.PyTorch, absent from the reference tree; semantics per SURVEY.md
Appendix A.1-A.6].  Each operator is an autograd Function whose forward/backward are single calls into
libprcnn_pointops.so (pointrcnn_amd.ops); index-valued ops are non-differentiable.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from pointrcnn_amd import ops


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices; starts at index 0, ties -> lowest index"""
        idx = ops.furthest_point_sample(xyz.contiguous(), npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint)"""
        idx = idx.contiguous()
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return ops.gather(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.gather_grad(grad_out.contiguous(), idx, ctx.N), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> dist (B,n,3) [sqrt of squared distance], idx (B,n,3) int32"""
        dist2, idx = ops.three_nn(unknown.contiguous(), known.contiguous())
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n)"""
        idx, weight = idx.contiguous(), weight.contiguous()
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[2]
        return ops.three_interpolate(features.contiguous(), idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return ops.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)"""
        idx = idx.contiguous()
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return ops.group(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.group_grad(grad_out.contiguous(), idx, ctx.N), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32"""
        idx = ops.ball_query(radius, nsample, xyz.contiguous(), new_xyz.contiguous())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query + grouping; output channel order is [dxyz(3), features(C)]."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)                  # (B,3,npoint,nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)                    # (B,3,1,N)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        return grouped_xyz
