import numpy as np


def bev(boxes3d):
    """lib/utils/kitti_utils.py:134-147 boxes3d_to_bev_torch in numpy fp32"""
    b = np.asarray(boxes3d, np.float32)
    hl, hw = b[:, 5] / np.float32(2), b[:, 4] / np.float32(2)
    return np.stack([b[:, 0] - hl, b[:, 2] - hw, b[:, 0] + hl, b[:, 2] + hw, b[:, 6]], 1).astype(np.float32)
