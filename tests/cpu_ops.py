"""TEST INFRASTRUCTURE: run the drop-in modules on CPU tensors by routing the tensor-level host layer
(pointrcnn_amd.ops) and the two pybind stand-ins (iou3d_cuda, roipool3d_cuda.forward) to the CPU oracle.

The product has no CPU implementation of the device operators (pointrcnn_amd.ops rejects CPU tensors).  For three
kinds of test the *graph code* above the operators has to run where there is no GPU:
  * the reference's own unchanged lib/net/*.py driven through the drop-in modules, to generate golden outputs in the
    build container (tests/golden/ref_net.py) -- /root/reference does not exist on the GPU box;
  * the DistributedDataParallel wiring of the RPN training step over gloo, world size 2 (tests/test_train_ddp_gloo.py);
  * gradient plumbing of the autograd Functions (composed path).
`oracle_ops()` is a context manager that patches the operator entry points with oracle-backed versions for CPU
tensors and restores them afterwards.  Nothing here is imported by the product."""
import contextlib

import numpy as np
import torch


def _np(t):
    return t.detach().cpu().numpy()


def _t(a, like=None):
    return torch.from_numpy(np.ascontiguousarray(a))


def _make(cpu):
    def furthest_point_sample(xyz, npoint, order="canonical"):
        assert order == "canonical"
        return _t(cpu.fps(_np(xyz), npoint))

    def gather(features, idx):
        return _t(cpu.gather(_np(features), _np(idx)))

    def gather_grad(grad_out, idx, N):
        return _t(cpu.gather_grad(_np(grad_out), _np(idx), N))

    def ball_query(radius, nsample, xyz, new_xyz):
        return _t(cpu.ball_query(radius, nsample, _np(xyz), _np(new_xyz)))

    def group(features, idx):
        return _t(cpu.group(_np(features), _np(idx)))

    def group_grad(grad_out, idx, N):
        return _t(cpu.group_grad(_np(grad_out), _np(idx), N))

    def three_nn(unknown, known, want_weight=False):
        d2, idx = cpu.three_nn(_np(unknown), _np(known))
        if want_weight:
            return _t(d2), _t(idx), _t(cpu.three_weights(d2))
        return _t(d2), _t(idx)

    def three_interpolate(features, idx, weight):
        return _t(cpu.three_interp(_np(features), _np(idx), _np(weight)))

    def three_interpolate_grad(grad_out, idx, weight, m):
        return _t(cpu.three_interp_grad(_np(grad_out), _np(idx), _np(weight), m))

    def pts_in_boxes3d(pts, boxes3d):
        return _t(cpu.pts_in_boxes3d(_np(pts), _np(boxes3d)).astype(np.int32))

    return {k: v for k, v in locals().items() if callable(v) and not k.startswith("_") and k != "cpu"}


@contextlib.contextmanager
def oracle_ops(trig_mode=2):
    """patch pointrcnn_amd.ops / iou3d_cuda / roipool3d_cuda.forward with oracle-backed CPU versions"""
    import oracle
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointrcnn_amd import ops
    import iou3d_cuda
    import roipool3d_cuda
    cpu = oracle.cpu()
    saved = []

    def patch(mod, name, fn):
        saved.append((mod, name, getattr(mod, name)))
        setattr(mod, name, fn)

    for name, fn in _make(cpu).items():
        patch(ops, name, fn)

    def rp_forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
        out, empty = cpu.roipool3d(_np(xyz), _np(boxes3d), _np(pts_feature), pooled_features.shape[2], trig_mode)
        pooled_features.copy_(_t(out))
        pooled_empty_flag.copy_(_t(empty))
        return 1

    # iou3d: the ENTRY POINTS of the drop-in module run as they are (argument checks, the (N, 1, 5) acceptance, the copy into the
    # caller's CPU `keep`); what is routed to the oracle is the tensor-level layer underneath them, and the is_cuda test of CHECK_INPUT
    def nms_sorted(boxes_sorted, thresh, rotated=True, max_keep=0):
        assert boxes_sorted.dim() == 2 and boxes_sorted.shape[1] == 5, tuple(boxes_sorted.shape)
        k = cpu.nms(_np(boxes_sorted), thresh, "rotated" if rotated else "normal", trig_mode)
        keep = torch.zeros((max(boxes_sorted.shape[0], 1),), dtype=torch.int64)
        keep[: len(k)] = _t(k)
        return keep, torch.tensor([len(k)], dtype=torch.int32)

    def nms_sorted_to_host(boxes_sorted, thresh, rotated, keep_cpu):
        keep, num = nms_sorted(boxes_sorted, thresh, rotated)
        n = int(num[0])
        keep_cpu[:n].copy_(keep[:n])
        return n

    def overlap(boxes_a, boxes_b, out=None):
        r = _t(cpu.boxes_overlap_bev(_np(boxes_a), _np(boxes_b), trig_mode))
        return r if out is None else out.copy_(r)

    def iou(boxes_a, boxes_b, out=None):
        r = _t(cpu.boxes_iou_bev(_np(boxes_a), _np(boxes_b), trig_mode))
        return r if out is None else out.copy_(r)

    def check_input(t, name):
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous " % name)

    patch(roipool3d_cuda, "forward", rp_forward)
    patch(ops, "nms_sorted", nms_sorted)
    patch(ops, "nms_sorted_to_host", nms_sorted_to_host)
    patch(ops, "boxes_overlap_bev", overlap)
    patch(ops, "boxes_iou_bev", iou)
    patch(iou3d_cuda, "_check_input", check_input)
    try:
        yield cpu
    finally:
        for mod, name, fn in reversed(saved):
            setattr(mod, name, fn)


@contextlib.contextmanager
def cuda_is_cpu():
    """The reference moves tensors with `.cuda()` and allocates with torch.cuda.FloatTensor; on a box without a GPU make
    both stay on the CPU (golden generation in the build container only)."""
    saved_cuda, saved_mod_cuda, saved_gd = torch.Tensor.cuda, torch.nn.Module.cuda, torch.Tensor.get_device
    names = ("FloatTensor", "IntTensor", "LongTensor", "ByteTensor")
    saved_ft = {n: getattr(torch.cuda, n, None) for n in names}
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: "cpu"      # bbox_transform.py:40 does anchor_size.to(roi.get_device()); -1 is not a device
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for n in names:
        setattr(torch.cuda, n, getattr(torch, n))
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.Tensor.get_device = saved_cuda, saved_mod_cuda, saved_gd
        for n, v in saved_ft.items():
            if v is not None:
                setattr(torch.cuda, n, v)


def fill_params_by_name(model, seed=0):
    """Deterministic parameters / BatchNorm statistics that depend only on (seed, state-dict key, shape): two models with
    the same state-dict keys get identical values whatever order they were constructed in.  Weights ~ N(0, gain/sqrt(fan_in)),
    biases small, BN running stats / affine terms non-trivial."""
    import zlib
    sd = model.state_dict()
    with torch.no_grad():
        for name in sorted(sd):
            t = sd[name]
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.1
            elif t.dim() >= 2:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (1.4 / fan_in ** 0.5)
            elif name.endswith("weight"):                     # BN scale
                v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
            else:                                             # biases
                v = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(v.to(t.dtype))
    return model
