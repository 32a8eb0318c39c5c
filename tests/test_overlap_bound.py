"""iou3d_geom.h: cannot_exceed -- the overlap bound that lets a rotated-NMS threshold decision skip the polygon clip (round 6).

The bound is restated here in numpy float32, operation for operation, and held against the ORACLE's overlaps (the reference's clip
restated in C, pinned to the reference's own sources compiled for the host): it must never decide a pair the clip would decide the
other way, and it must never fall below the clip's overlap by more than rounding.  The GPU tests (tests/test_gpu_proposal.py,
test_gpu_roipool_iou.py, test_gpu_parity_residuals.py) hold the kernels that USE it to the oracle's keep lists."""
import numpy as np
import pytest

f = np.float32


def _rbox(b):
    x1, y1, x2, y2, ang = [b[:, i].astype(f) for i in range(5)]
    return dict(hx=(x2 - x1) * f(0.5), hy=(y2 - y1) * f(0.5), cx=(x1 + x2) / f(2), cy=(y1 + y2) / f(2), c=np.cos(ang).astype(f), s=np.sin(ang).astype(f))


def _pao(h, d, e):          # padded_axis_overlap
    return np.maximum(np.minimum(h, d + e) - np.maximum(-h, d - e) + f(2e-3), f(0))


def overlap_bound(A, B):
    """-> (u, Sa, Sb) for all pairs: u >= overlap of A[i] and B[j] (cannot_exceed's u)"""
    a = {k: v[:, None] for k, v in _rbox(A).items()}
    b = {k: v[None, :] for k, v in _rbox(B).items()}
    cd = np.abs(a["c"] * b["c"] + a["s"] * b["s"])
    sd = np.abs(b["s"] * a["c"] - b["c"] * a["s"])
    dx, dy = b["cx"] - a["cx"], b["cy"] - a["cy"]
    u1 = _pao(a["hx"], dx * a["c"] - dy * a["s"], b["hx"] * cd + b["hy"] * sd) * _pao(a["hy"], dx * a["s"] + dy * a["c"], b["hx"] * sd + b["hy"] * cd)
    u2 = _pao(b["hx"], dy * b["s"] - dx * b["c"], a["hx"] * cd + a["hy"] * sd) * _pao(b["hy"], -dx * b["s"] - dy * b["c"], a["hx"] * sd + a["hy"] * cd)
    sa, sb = f(4) * a["hx"] * a["hy"], f(4) * b["hx"] * b["hy"]
    return np.minimum(np.minimum(u1, u2), np.minimum(sa, sb)), sa, sb


def _boxes(rng, n, spread, clusters, shift=(0.0, 0.0)):
    ctr = rng.uniform([-spread, 0], [spread, 2 * spread], (n, 2))
    l = rng.uniform(3.2, 4.6, n) * rng.choice([1, 1, 1, 0.3], n)
    w = rng.uniform(1.4, 1.9, n) * rng.choice([1, 1, 1, 2.0], n)
    ang = rng.uniform(-np.pi, np.pi, n)
    if clusters:          # eight near-copies of every box, as an RPN emits around one object (incl. heading flips)
        k = n // 8
        base = np.concatenate([ctr[:k], l[:k, None], w[:k, None], ang[:k, None]], 1)
        rep = np.repeat(base, 8, 0) + rng.normal(0, 1, (8 * k, 5)) * np.array([0.3, 0.3, 0.2, 0.1, 0.08])
        rep[::5, 4] += np.pi
        ctr, l, w, ang = rep[:, :2], np.abs(rep[:, 2]) + 0.1, np.abs(rep[:, 3]) + 0.1, rep[:, 4]
    out = np.stack([ctr[:, 0] - l / 2, ctr[:, 1] - w / 2, ctr[:, 0] + l / 2, ctr[:, 1] + w / 2, ang], 1).astype(f)
    out[:, [0, 2]] += f(shift[0]); out[:, [1, 3]] += f(shift[1])
    return out


@pytest.mark.parametrize("name,n,spread,clusters,shift", [("scattered", 400, 20, False, (0, 0)), ("dense", 400, 4, False, (0, 0)),
                                                        ("clusters", 400, 15, True, (0, 0)), ("far from the origin", 400, 4, False, (70, 35))])
def test_bound_never_decides_against_the_clip(name, n, spread, clusters, shift):
    import oracle
    cpu = oracle.cpu()
    rng = np.random.default_rng(7)
    B = _boxes(rng, n, spread, clusters, shift)
    ov, iou = cpu.boxes_overlap_bev(B, B), cpu.boxes_iou_bev(B, B)
    u, sa, sb = overlap_bound(B, B)
    assert (ov <= u * f(1.0001) + f(1e-4)).all(), "the bound fell below the clip's overlap by more than rounding"
    decided = 0
    for thr in (0.85, 0.8, 0.7, 0.3, 0.1):
        skip = u < f(0.9) * f(thr) * (sa + sb - u)              # cannot_exceed's decision
        assert not (skip & (iou > thr)).any(), (name, thr)
        assert iou[skip].max(initial=0.0) < 0.9 * thr + 1e-3       # what is skipped is below the threshold with the margin the comment states
        decided = max(decided, (skip & (ov > 0)).sum() / max(1, (ov > 0).sum()))
    assert decided > 0.5                                        # and it does decide most overlapping pairs (else it is dead code)
