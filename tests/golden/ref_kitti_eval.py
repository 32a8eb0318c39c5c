"""Import the REFERENCE'S OWN KITTI evaluator (tools/kitti_object_eval_python/{eval,rotate_iou,kitti_common}.py) without
numba: golden-vector generation only (needs /root/reference; build container only).

numba is not installed here and numba.cuda cannot target ROCm, so the decorators are stubbed to identity -- the
reference's functions then run as the plain Python they are written in -- and the one CUDA kernel launch
(rotate_iou_gpu_eval) is replaced by a loop that calls the reference's own device function devRotateIoUEval for every
pair with the kernel's argument order (rotate_iou.py:282-284).  numba's float32/float64 type inference is not
reproduced by numpy scalars, so these vectors pin the evaluator's LOGIC exactly and its IoU arithmetic to ~1e-6.
"""
import os
import sys
import types

import numpy as np

REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE, "tools", "kitti_object_eval_python"))


def _passthrough(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def load():
    """-> (eval module, kitti_common module)"""
    numba = types.ModuleType("numba")
    numba.jit = _passthrough
    numba.float32, numba.int32, numba.prange = np.float32, np.int32, range
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _passthrough

    class _Arr:
        @staticmethod
        def array(shape, dtype):
            return np.zeros(shape, dtype)

    cuda.local, cuda.shared = _Arr, _Arr
    numba.cuda = cuda
    sys.modules.setdefault("numba", numba)
    sys.modules.setdefault("numba.cuda", cuda)
    sys.modules.setdefault("fire", types.ModuleType("fire"))
    skimage = types.ModuleType("skimage")
    skimage.io = types.ModuleType("skimage.io")
    sys.modules.setdefault("skimage", skimage)
    sys.modules.setdefault("skimage.io", skimage.io)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import tools.kitti_object_eval_python.rotate_iou as riou
    import tools.kitti_object_eval_python.eval as ev
    import tools.kitti_object_eval_python.kitti_common as kc

    def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):     # rotate_iou.py:287-329
        boxes = boxes.astype(np.float32)
        query_boxes = query_boxes.astype(np.float32)
        N, K = boxes.shape[0], query_boxes.shape[0]
        iou = np.zeros((N, K), dtype=np.float32)
        for n in range(N):
            for k in range(K):
                iou[n, k] = riou.devRotateIoUEval(query_boxes[k], boxes[n], criterion)     # kernel :282-284
        return iou

    ev.rotate_iou_gpu_eval = rotate_iou_gpu_eval
    return ev, kc
