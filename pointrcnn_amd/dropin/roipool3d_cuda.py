"""`roipool3d_cuda` -- Python stand-in for the reference's pybind module
(lib/utils/roipool3d/src/roipool3d.cpp:198-203): forward / forward_slow / pts_in_boxes3d_cpu / roipool3d_cpu
with the reference's argument order and caller-allocated outputs.  All computation is in
libprcnn_pointops.so: `forward` is the HIP kernel; the two *_cpu entry points (CPU tensors in, CPU tensors out -- the
reference's dataloader calls them inside forked worker processes, kitti_rcnn_dataset.py:487,582,625,843,970, where no
HIP context may exist) are the library's prcnn_host_* functions: plain host code, no device, no HIP call, bit-identical
to the reference's roipool3d.cpp:82-195."""
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


def forward_slow(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    """The reference keeps a second, one-thread-per-box implementation of the same computation under this name (roipool3d.cpp:15-46
    -> roipool3d_kernel.cu:30-95 roipool3dLauncher_slow).  Here it is a second implementation too, sharing nothing with the fused
    kernel but the point-in-box test: per frame the (M, N) flag matrix of prcnn_pts_in_boxes3d, then the reference's selection
    rule spelled with tensor operations -- the first S in-box points in index order (a stable descending sort of the flags), slot
    k >= count takes slot k mod count, empty boxes are flagged and zero-filled.  Same results as `forward`, bit for bit
    (tests/test_gpu_roipool_iou.py); an independent path to check the fused kernel against, not a fast one."""
    from pointrcnn_amd import ops
    for t, n in ((xyz, "xyz"), (boxes3d, "boxes3d"), (pts_feature, "pts_feature"),
                 (pooled_features, "pooled_features"), (pooled_empty_flag, "pooled_empty_flag")):
        _check_input(t, n)
    B, N = xyz.shape[0], xyz.shape[1]
    M, S = boxes3d.shape[1], pooled_features.shape[2]
    slots = torch.arange(S, device=xyz.device)
    for b in range(B):
        flags = ops.pts_in_boxes3d(xyz[b], boxes3d[b])                                   # (M, N) int32
        cnt = flags.sum(1)
        first = torch.sort(flags, dim=1, descending=True, stable=True).indices[:, :S]   # in-box points first, ascending index
        if first.shape[1] < S:                                                          # N < S
            first = torch.cat([first, first.new_zeros((M, S - first.shape[1]))], 1)
        take = torch.minimum(cnt.clamp(min=1), cnt.new_tensor(S)).long()
        src = torch.gather(first, 1, (slots[None, :] % take[:, None]))                  # wrap-duplicate (roipool3d.cpp:171-191)
        rows = torch.cat([xyz[b][src], pts_feature[b][src]], 2)                         # (M, S, 3 + C)
        empty = cnt == 0
        rows[empty] = 0
        pooled_features[b] = rows
        pooled_empty_flag[b] = empty.to(pooled_empty_flag.dtype)
    return 1


def _host(t, name, dtype):
    if t.is_cuda:
        raise RuntimeError("%s must be a CPU tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous " % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    return t.data_ptr()


def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    """pts_flag (M,N) int64 CPU out, pts (N,3), boxes3d (M,7) CPU in   [roipool3d.cpp:97-125]"""
    N, M = pts.shape[0], boxes3d.shape[0]
    if tuple(pts_flag.shape) != (M, N):
        raise RuntimeError("pts_flag must be (%d, %d)" % (M, N))
    _cabi.check(_cabi.lib().prcnn_host_pts_in_boxes3d(_host(pts, "pts", torch.float32), _host(boxes3d, "boxes3d", torch.float32),
                                                      N, M, _host(pts_flag, "pts_flag", torch.int64)), "prcnn_host_pts_in_boxes3d")
    return 1


def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    """pts (N,3), boxes3d (M,7), pts_feature (N,C) -> pooled_pts (M,S,3), pooled_features (M,S,C),
    pooled_empty_flag (M) int64   [roipool3d.cpp:127-195]"""
    N, M, C, S = pts.shape[0], boxes3d.shape[0], pts_feature.shape[1], pooled_pts.shape[1]
    if tuple(pooled_pts.shape) != (M, S, 3) or tuple(pooled_features.shape) != (M, S, C) or pooled_empty_flag.numel() != M:
        raise RuntimeError("pooled_pts / pooled_features / pooled_empty_flag must be (M,S,3) / (M,S,C) / (M)")
    f32 = torch.float32
    _cabi.check(_cabi.lib().prcnn_host_roipool3d(_host(pts, "pts", f32), _host(boxes3d, "boxes3d", f32),
                                                 _host(pts_feature, "pts_feature", f32), N, M, C, S,
                                                 _host(pooled_pts, "pooled_pts", f32), _host(pooled_features, "pooled_features", f32),
                                                 _host(pooled_empty_flag, "pooled_empty_flag", torch.int64)), "prcnn_host_roipool3d")
    return 1
