"""`iou3d_cuda` -- Python stand-in for the reference's pybind module (lib/utils/iou3d/src/iou3d.cpp:174-179):
same four functions, same argument order and ownership (caller allocates; `keep` is a CPU int64 tensor for
the nms functions, iou3d_utils.py:68,85), argument errors raise RuntimeError (AT_CHECK, iou3d.cpp:7-9).
All computation is in libprcnn_pointops.so.  The NMS sweep runs on the device and writes the kept indices and their count
into a pinned, device-mapped staging buffer; the ONE host synchronisation per call is the wait for the stream that the API
shape forces -- the count is the return value -- (pointrcnn_amd.ops.nms_sorted is the sync-free form)."""
import torch

from pointrcnn_amd import ops


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDAtensor " % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous " % name)


def _rows5(boxes, name):
    """The pybind functions take `boxes.size(0)` rows of five contiguous floats from the data pointer, whatever the tensor's other
    dimensions (iou3d.cpp:80-81,40-41): tools/eval_rcnn.py:617-619 hands nms_gpu an (N, 1, 5) tensor -- scores of shape (N, 1) index
    the boxes in iou3d_utils.py:64-66.  Same acceptance here: any contiguous tensor of N x 5 elements is N rows."""
    n = boxes.shape[0] if boxes.dim() else 0
    if boxes.dim() == 2 and boxes.shape[1] == 5:
        return boxes
    if boxes.numel() != n * 5:
        raise RuntimeError("%s must hold size(0) x 5 floats [x1, y1, x2, y2, ry], got shape %s" % (name, tuple(boxes.shape)))
    return boxes.view(n, 5)


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_overlap, "ans_overlap")):
        _check_input(t, n)
    ops.boxes_overlap_bev(_rows5(boxes_a, "boxes_a"), _rows5(boxes_b, "boxes_b"), out=ans_overlap)
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        _check_input(t, n)
    ops.boxes_iou_bev(_rows5(boxes_a, "boxes_a"), _rows5(boxes_b, "boxes_b"), out=ans_iou)
    return 1


def _nms(boxes, keep, thresh, rotated):
    _check_input(boxes, "boxes")
    if not keep.is_contiguous():
        raise RuntimeError("keep must be contiguous ")
    if keep.is_cuda or keep.dtype != torch.int64:
        raise RuntimeError("keep must be a CPU int64 tensor (iou3d_utils.py:68,85)")
    boxes = _rows5(boxes, "boxes")
    if keep.numel() < boxes.shape[0]:
        raise RuntimeError("keep holds %d entries for %d boxes" % (keep.numel(), boxes.shape[0]))
    return ops.nms_sorted_to_host(boxes, thresh, rotated, keep)


def nms_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, True)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms(boxes, keep, nms_overlap_thresh, False)
