"""Stand-in for `numba` (absent in this image; numba.cuda cannot target ROCm at all).  TEST INFRASTRUCTURE / environment shim: put
tests/compat on PYTHONPATH and the reference's UNCHANGED tools/kitti_object_eval_python/{eval,rotate_iou}.py import and run -- the
`@numba.jit` functions as the plain Python they are written in, the one CUDA kernel through the launch emulator in numba/cuda.py.
numba's float32 type inference is not reproduced by numpy scalars: the evaluator's LOGIC is exact, its IoU arithmetic ~1e-6."""
import numpy as np

from . import cuda  # noqa: F401

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64
prange = range


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit
