"""pointrcnn_amd -- MI355X (gfx950) implementation of PointRCNN's point-ops hot path.

    csrc/       hand-written HIP kernels + the C ABI (include/prcnn_pointops.h)
    _cabi.py    ctypes binding of libprcnn_pointops.so (the drop-in boundary)
    ops.py      tensor-level host layer (allocation, checks, stream plumbing)
    dropin/     the reference's own Python op surface: pointnet2_lib.pointnet2.{pointnet2_utils,
                pointnet2_modules,pytorch_utils}, pointnet2_cuda, iou3d_cuda, roipool3d_cuda
    rpn.py      host-side mirror of the reference's RPN inference graph (lib/net/rpn.py + pointnet2_msg.py)
                built from the drop-in modules; what bench.py times

`install()` puts dropin/ on sys.path so that the reference's unchanged `lib/` and `tools/` resolve
`import iou3d_cuda`, `import roipool3d_cuda` and `from pointnet2_lib.pointnet2 import ...` to this package.
"""
import os
import sys

__version__ = "0.1.0"

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def install():
    """Make the drop-in modules importable under the names the reference uses."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    return DROPIN_DIR
