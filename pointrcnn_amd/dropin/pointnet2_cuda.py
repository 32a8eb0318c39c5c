"""`pointnet2_cuda` -- Python stand-in for upstream's pybind module of the same name [UPSTREAM, not in the
reference tree; SURVEY.md 8(b)]: same function names and argument order, caller-allocated outputs, every
call forwarded to libprcnn_pointops.so on torch's current stream.  No computation happens in Python."""
import torch

from pointrcnn_amd import _cabi
from pointrcnn_amd.ops import _chk, _p, _stream

_I32 = torch.int32


def furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, idx):
    _chk(xyz, "xyz"); _chk(idx, "idx", _I32)
    _cabi.check(_cabi.lib().prcnn_fps(_p(xyz), B, N, npoint, _p(temp), _p(idx), _stream()), "prcnn_fps")
    return 1


def gather_points_wrapper(B, C, N, npoint, features, idx, out):
    _chk(features, "features"); _chk(idx, "idx", _I32); _chk(out, "out")
    _cabi.check(_cabi.lib().prcnn_gather(_p(features), _p(idx), B, C, N, npoint, _p(out), _stream()), "prcnn_gather")
    return 1


def gather_points_grad_wrapper(B, C, N, npoint, grad_out, idx, grad_features):
    _chk(grad_out, "grad_out"); _chk(idx, "idx", _I32); _chk(grad_features, "grad_features")
    _cabi.check(_cabi.lib().prcnn_gather_grad(_p(grad_out), _p(idx), B, C, N, npoint, _p(grad_features), _stream()),
                "prcnn_gather_grad")
    return 1


def ball_query_wrapper(B, N, M, radius, nsample, new_xyz, xyz, idx):
    _chk(new_xyz, "new_xyz"); _chk(xyz, "xyz"); _chk(idx, "idx", _I32)
    _cabi.check(_cabi.lib().prcnn_ball_query(_p(xyz), _p(new_xyz), B, N, M, float(radius), nsample, _p(idx), _stream()),
                "prcnn_ball_query")
    return 1


def group_points_wrapper(B, C, N, npoint, nsample, features, idx, out):
    _chk(features, "features"); _chk(idx, "idx", _I32); _chk(out, "out")
    _cabi.check(_cabi.lib().prcnn_group(_p(features), _p(idx), B, C, N, npoint, nsample, _p(out), _stream()),
                "prcnn_group")
    return 1


def group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out, idx, grad_features):
    _chk(grad_out, "grad_out"); _chk(idx, "idx", _I32); _chk(grad_features, "grad_features")
    _cabi.check(_cabi.lib().prcnn_group_grad(_p(grad_out), _p(idx), B, C, N, npoint, nsample, _p(grad_features),
                                             _stream()), "prcnn_group_grad")
    return 1


def three_nn_wrapper(B, n, m, unknown, known, dist2, idx):
    _chk(unknown, "unknown"); _chk(known, "known"); _chk(dist2, "dist2"); _chk(idx, "idx", _I32)
    _cabi.check(_cabi.lib().prcnn_three_nn(_p(unknown), _p(known), B, n, m, _p(dist2), _p(idx), None, _stream()),
                "prcnn_three_nn")
    return 1


def three_interpolate_wrapper(B, C, m, n, features, idx, weight, out):
    _chk(features, "features"); _chk(idx, "idx", _I32); _chk(weight, "weight"); _chk(out, "out")
    _cabi.check(_cabi.lib().prcnn_three_interp(_p(features), _p(idx), _p(weight), B, C, m, n, _p(out), _stream()),
                "prcnn_three_interp")
    return 1


def three_interpolate_grad_wrapper(B, C, n, m, grad_out, idx, weight, grad_features):
    _chk(grad_out, "grad_out"); _chk(idx, "idx", _I32); _chk(weight, "weight"); _chk(grad_features, "grad_features")
    _cabi.check(_cabi.lib().prcnn_three_interp_grad(_p(grad_out), _p(idx), _p(weight), B, C, n, m, _p(grad_features),
                                                    None, _stream()), "prcnn_three_interp_grad")     # no scratch in this API shape: direct kernel
    return 1
