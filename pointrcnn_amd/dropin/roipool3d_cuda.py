"""`roipool3d_cuda` -- Python stand-in for the reference's pybind module
(lib/utils/roipool3d/src/roipool3d.cpp:198-203): forward / forward_slow / pts_in_boxes3d_cpu / roipool3d_cpu
with the reference's argument order and caller-allocated outputs.  All computation is in
libprcnn_pointops.so; the two *_cpu entry points (whose contract is CPU tensors in, CPU tensors out --
they serve the dataloader, kitti_rcnn_dataset.py:487,843) stage through the device and run the same HIP
kernels: there is no CPU implementation in the product."""
import torch

from pointrcnn_amd import _cabi
from pointrcnn_amd.ops import _p, _stream


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDAtensor " % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous " % name)


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    for t, n in ((xyz, "xyz"), (boxes3d, "boxes3d"), (pts_feature, "pts_feature"),
                 (pooled_features, "pooled_features"), (pooled_empty_flag, "pooled_empty_flag")):
        _check_input(t, n)
    B, N = xyz.shape[0], xyz.shape[1]
    M, C, S = boxes3d.shape[1], pts_feature.shape[2], pooled_features.shape[2]
    if pooled_empty_flag.dtype != torch.int32 or pooled_features.dtype != torch.float32:
        raise RuntimeError("pooled_features must be float32 and pooled_empty_flag int32")
    _cabi.check(_cabi.lib().prcnn_roipool3d(_p(xyz), _p(boxes3d), _p(pts_feature), B, N, M, C, S, _p(pooled_features),
                                            _p(pooled_empty_flag), _stream()), "prcnn_roipool3d")
    return 1


forward_slow = forward   # same contract; the reference's one-thread-per-box variant has no separate meaning here


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("roipool3d_cuda: no HIP device; the HIP kernels are the only implementation")
    return torch.device("cuda")


def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    """pts_flag (M,N) int64 CPU out, pts (N,3), boxes3d (M,7) CPU in   [roipool3d.cpp:97-125]"""
    for t, n in ((pts_flag, "pts_flag"), (pts, "pts"), (boxes3d, "boxes3d")):
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous " % n)
    d = _dev()
    p, b = pts.float().to(d), boxes3d.float().to(d)
    N, M = p.shape[0], b.shape[0]
    flags = torch.empty((M, N), dtype=torch.int32, device=d)
    _cabi.check(_cabi.lib().prcnn_pts_in_boxes3d(_p(p), _p(b), N, M, _p(flags), _stream()), "prcnn_pts_in_boxes3d")
    pts_flag.copy_(flags.to(torch.int64).cpu())
    return 1


def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    """pts (N,3), boxes3d (M,7), pts_feature (N,C) -> pooled_pts (M,S,3), pooled_features (M,S,C),
    pooled_empty_flag (M) int64   [roipool3d.cpp:127-195]"""
    for t, n in ((pts, "pts"), (boxes3d, "boxes3d"), (pts_feature, "pts_feature"), (pooled_pts, "pooled_pts"),
                 (pooled_features, "pooled_features"), (pooled_empty_flag, "pooled_empty_flag")):
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous " % n)
    d = _dev()
    p, b, f = pts.float().to(d), boxes3d.float().to(d), pts_feature.float().to(d)
    N, M, C, S = p.shape[0], b.shape[0], f.shape[1], pooled_pts.shape[1]
    out = torch.empty((1, M, S, 3 + C), dtype=torch.float32, device=d)
    empty = torch.empty((1, M), dtype=torch.int32, device=d)
    _cabi.check(_cabi.lib().prcnn_roipool3d(_p(p), _p(b), _p(f), 1, N, M, C, S, _p(out), _p(empty), _stream()),
                "prcnn_roipool3d")
    out = out[0].cpu()
    pooled_pts.copy_(out[:, :, :3])
    pooled_features.copy_(out[:, :, 3:])
    pooled_empty_flag.copy_(empty[0].to(torch.int64).cpu())
    return 1
