"""seeded synthetic inputs shared by the tests (shapes follow SURVEY.md 8(d) configs)"""
import numpy as np

GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def unit_cloud(B, N, seed=0):
    return np.random.default_rng(seed).random((B, N, 3), dtype=np.float32)


def kitti_cloud(B, N, seed=100):
    out = np.empty((B, N, 3), np.float32)
    for b in range(B):
        r = np.random.default_rng(seed + b)
        out[b] = np.stack([r.uniform(-40, 40, N), r.uniform(-1, 3, N), r.uniform(0, 70.4, N)], 1)
    return out


def rand_bev(n, spread=8.0, seed=0):
    r = np.random.default_rng(seed)
    cx, cy = r.uniform(-spread, spread, n), r.uniform(0, 2 * spread, n)
    w, l, a = r.uniform(1.2, 2.2, n), r.uniform(3.0, 5.0, n), r.uniform(-np.pi, np.pi, n)
    return np.stack([cx - w / 2, cy - l / 2, cx + w / 2, cy + l / 2, a], 1).astype(np.float32)


def rand_boxes3d(pts, M, seed=0, jitter=0.5):
    """(M,7) [x,y(bottom),z,h,w,l,ry] boxes centred near random points of pts (N,3)"""
    r = np.random.default_rng(seed)
    ctr = pts[r.integers(0, pts.shape[0], M)] + r.normal(0, jitter, (M, 3))
    h, w, l = r.uniform(1.4, 1.8, (M, 1)), r.uniform(1.5, 1.8, (M, 1)), r.uniform(3.5, 4.5, (M, 1))
    ry = r.uniform(-np.pi, np.pi, (M, 1))
    return np.concatenate([ctr[:, :1], ctr[:, 1:2] + h / 2, ctr[:, 2:3], h, w, l, ry], 1).astype(np.float32)


def enlarge(boxes, e):
    """lib/utils/kitti_utils.py:150-160 enlarge_box3d"""
    b = boxes.copy()
    b[..., 3:6] += 2 * e
    b[..., 1] += e
    return b


def mlp_tol(ref):
    """absolute tolerance for the fp32-MFMA MLP against the double-accumulated oracle (1e-5 relative to scale)"""
    return 1e-5 * max(1.0, float(np.abs(ref).max()))
