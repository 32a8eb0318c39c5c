"""RPN input builder: the CPU oracle against the reference's own Python (golden fixture made by
tests/golden/ref_scene.py from lib/utils/calibration.py + lib/datasets/kitti_rcnn_dataset.py), the re-specified draw's
statistical properties, and the host-side mirror (calibration parsing, batch packing)."""
import os

import numpy as np
import pytest

import oracle
from util import GOLDEN, KITTI_CALIB_TXT, scene_invariants, synthetic_scan

FRAMES = [dict(n=30000, seed=11, fov=0.5, far=0.12, hw=(375, 1242)),
          dict(n=9000, seed=12, fov=0.35, far=0.2, hw=(370, 1224)),
          dict(n=20000, seed=13, fov=0.6, far=0.05, hw=(376, 1241))]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "scene_ref.npz"))


def _calib24(gold, k):
    return np.concatenate([gold["f%d_M" % k].reshape(-1), gold["f%d_P2" % k].reshape(-1)]).astype(np.float32)


def _near_boundary(rect, img, depth, hw, scope, tol):
    """points whose validity hinges on the last bits of the fp32 arithmetic"""
    H, W = hw
    d = np.minimum.reduce([np.abs(img[:, 0]), np.abs(img[:, 0] - W), np.abs(img[:, 1]), np.abs(img[:, 1] - H), np.abs(depth),
                           np.abs(rect[:, 0] - scope[0]), np.abs(rect[:, 0] - scope[1]), np.abs(rect[:, 1] - scope[2]),
                           np.abs(rect[:, 1] - scope[3]), np.abs(rect[:, 2] - scope[4]), np.abs(rect[:, 2] - scope[5])])
    return d < tol


@pytest.mark.parametrize("k", [0, 1, 2])
def test_projection_and_valid_flag_match_the_reference(gold, k):
    fr = FRAMES[k]
    scan = synthetic_scan(fr["n"], fr["seed"], fr["fov"], fr["far"])
    scope = gold["area_scope"]
    rect, img, depth, flag = oracle.scene_project(scan, _calib24(gold, k), fr["hw"][0], fr["hw"][1], scope)
    s = int(gold["stride"])
    # the reference multiplies through a BLAS sgemm (unspecified summation order / FMA): last-ulp differences only
    np.testing.assert_allclose(rect[::s], gold["f%d_rect" % k], rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(depth[::s], gold["f%d_depth" % k], rtol=2e-6, atol=2e-5)
    front = np.abs(rect[::s, 2]) > 0.5                     # image coordinates blow up next to the camera plane
    np.testing.assert_allclose(img[::s][front], gold["f%d_img" % k][front], rtol=2e-5, atol=2e-3)
    ref_flag = np.unpackbits(gold["f%d_flag" % k])[:fr["n"]].astype(bool)
    diff = flag != ref_flag
    assert diff.sum() <= 3                                   # and only where validity hinges on the last bits
    assert _near_boundary(rect, img, depth, fr["hw"], scope, 1e-3)[diff].all()
    assert abs(int(flag.sum()) - int(gold["f%d_nvalid" % k])) <= 3


@pytest.mark.parametrize("k", [0, 1, 2])
def test_sample_invariants_hold_for_reference_and_oracle(gold, k):
    """the reference's sample comes from numpy's global stream, ours from the counter-based contract: both must satisfy
    everything kitti_rcnn_dataset.py:285-306 guarantees (far points kept, no replacement, top-up of short frames)"""
    fr = FRAMES[k]
    npoints = int(gold["npoints"])
    scan = synthetic_scan(fr["n"], fr["seed"], fr["fov"], fr["far"])
    ref_flag = np.unpackbits(gold["f%d_flag" % k])[:fr["n"]].astype(bool)
    calib = _calib24(gold, k)
    rect, _, _, flag = oracle.scene_project(scan, calib, fr["hw"][0], fr["hw"][1], gold["area_scope"])
    # reference sample against the reference's own valid set (rows compared through the oracle's rect of the same points:
    # identical up to the sgemm ulp, so match rows by nearest valid point instead of bytes)
    ref_rows = gold["f%d_sample_rect" % k]
    vr = rect[ref_flag]
    from scipy.spatial import cKDTree
    dist, nn = cKDTree(vr).query(ref_rows)
    assert dist.max() < 1e-4
    mult = np.bincount(nn, minlength=vr.shape[0])
    far = ~(vr[:, 2] < 40.0)
    if vr.shape[0] > npoints:
        assert mult.max() == 1 and (mult[far] == 1).all()
    else:
        assert mult.min() >= 1 and mult.max() <= 2
    np.testing.assert_allclose(gold["f%d_sample_feat" % k][:, 0], (scan[ref_flag][nn, 3] - np.float32(0.5)), atol=0)
    # oracle sample, bytes exact against its own valid set
    off = np.array([0, fr["n"]], np.int64)
    xyz, inten, src, nvalid, status = oracle.scene_prepare(scan, off, calib[None], np.array([fr["hw"]], np.int32), gold["area_scope"],
                                                           npoints, seed=5)
    assert status[0] == 0 and nvalid[0] == flag.sum()
    scene_invariants(xyz[0], inten[0], rect[flag], scan[flag, 3] - np.float32(0.5), npoints)
    assert np.array_equal(xyz[0], rect[src[0]]) and np.array_equal(inten[0], scan[src[0], 3] - np.float32(0.5))


def test_draw_is_deterministic_seeded_and_uniform():
    scan = synthetic_scan(6000, seed=3, fov_frac=0.8, far_frac=0.1)
    calib = _calib_from_txt()
    off = np.array([0, 6000], np.int64)
    hw = np.array([[375, 1242]], np.int32)
    scope = [-40, 40, -1, 3, 0, 70.4]
    _, _, _, flag = oracle.scene_project(scan, calib, 375, 1242, scope)
    near = flag & True
    a = oracle.scene_prepare(scan, off, calib[None], hw, scope, 1024, seed=1)
    b = oracle.scene_prepare(scan, off, calib[None], hw, scope, 1024, seed=1)
    c = oracle.scene_prepare(scan, off, calib[None], hw, scope, 1024, seed=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[2], c[2])
    # inclusion frequency of every near valid point over many seeds ~ Binomial(T, k / n_near); positions ~ uniform
    rect, _, _, _ = oracle.scene_project(scan, calib, 375, 1242, scope)
    is_far = flag & ~(rect[:, 2] < 40.0)
    n_near, n_far = int((flag & ~is_far).sum()), int(is_far.sum())
    T = 300
    hits = np.zeros(6000, np.int64)
    pos_sum = np.zeros(6000, np.float64)
    for s in range(T):
        src = oracle.scene_prepare(scan, off, calib[None], hw, scope, 1024, seed=100 + s)[2][0]
        hits[src] += 1
        pos_sum[src] += np.arange(1024)
    assert (hits[is_far] == T).all() and (hits[~flag] == 0).all()
    p = (1024 - n_far) / n_near
    z = (hits[flag & ~is_far] - T * p) / np.sqrt(T * p * (1 - p))
    assert abs(z.mean()) < 0.15 and 0.85 < z.std() < 1.15 and np.abs(z).max() < 5.5
    mean_pos = pos_sum[is_far] / T                         # far points are in every sample: their mean slot ~ 511.5
    assert abs(mean_pos.mean() - 511.5) < 6 and np.abs(mean_pos - 511.5).max() < 5 * 295.6 / np.sqrt(T)
    assert near.any()


def _calib_from_txt(tmpdir=None):
    from pointrcnn_amd import kitti_input
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "000000.txt")
        with open(path, "w") as f:
            f.write(KITTI_CALIB_TXT)
        return kitti_input.Calibration(path).packed()


def test_host_mirror_parses_calibration_like_the_reference(gold):
    packed = _calib_from_txt()
    assert packed.dtype == np.float32 and packed.shape == (24,)
    assert np.array_equal(packed[:12].reshape(4, 3), gold["f0_M"]) and np.array_equal(packed[12:].reshape(3, 4), gold["f0_P2"])


def test_edge_cases_and_reference_domain():
    """empty scans, no valid point, frames the reference itself rejects (ValueError in np.random.choice): status codes"""
    calib = _calib_from_txt()
    scope = [-40, 40, -1, 3, 0, 70.4]
    behind = synthetic_scan(500, seed=1, fov_frac=0.0)
    behind[:, 0] = -np.abs(behind[:, 0]) - 1.0                 # everything behind the camera
    many_far = synthetic_scan(4000, seed=2, fov_frac=0.9, far_frac=0.9)
    few = synthetic_scan(300, seed=4, fov_frac=0.5, far_frac=0.1)
    ok = synthetic_scan(3000, seed=5, fov_frac=0.7, far_frac=0.1)
    scans = [behind, np.zeros((0, 4), np.float32), many_far, few, ok]
    off = np.concatenate([[0], np.cumsum([s.shape[0] for s in scans])]).astype(np.int64)
    raw = np.concatenate(scans)
    B = len(scans)
    xyz, inten, src, nvalid, status = oracle.scene_prepare(raw, off, np.tile(calib, (B, 1)), np.tile([[375, 1242]], (B, 1)), scope, 1024, 9)
    assert list(status) == [2, 2, 1, 1, 0] and nvalid[0] == 0 and nvalid[1] == 0
    assert (src[0] == -1).all() and (xyz[1] == 0).all()
    assert nvalid[3] * 2 < 1024 and set(src[3]) == set(np.nonzero(oracle.scene_project(few, calib, 375, 1242, scope)[3])[0])
    assert len(set(src[2])) == 1024                              # too many far points: a plain draw of npoints
    assert len(set(src[4])) == min(1024, nvalid[4])
