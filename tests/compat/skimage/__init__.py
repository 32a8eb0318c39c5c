"""`from skimage import io` of the reference's kitti_common.py:8 (only its unused image readers need it).  Environment shim."""
from . import io  # noqa: F401
