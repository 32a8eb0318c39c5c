"""Comparison mode of the index operators (round 6): selectable tie order and squared-distance arithmetic behind the C ABI
(prcnn_fps_mode, prcnn_ball_query_arith, prcnn_three_nn_arith).

The PointNet++ operators are PARITY UNPINNED (the upstream source is absent: /root/reference/.gitmodules:1-4); worse, the real
upstream build is nvcc with default FMA contraction, the contract chosen here is no-FMA.  "upstream" arithmetic restates the
contracted form  fma(dz,dz, fma(dy,dy, dx*dx)).  Here: the mode kernels equal the oracle bit for bit in BOTH arithmetics; canonical
mode equals the fast kernels; and the disagreement counter used for DESIGN.md section 2 (tools/arith_disagreement.py) agrees with
a count made from the oracle's outputs."""
import os
import sys

import numpy as np
import pytest
import torch

from util import kitti_cloud

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("arith", ["canonical", "upstream"])
@pytest.mark.parametrize("order", ["canonical", "upstream"])
@pytest.mark.parametrize("B,N,npoint", [(2, 4096, 1024), (1, 16384, 600), (2, 1000, 256), (2, 100, 37), (1, 1, 1)])
def test_fps_mode_matches_oracle(dev, cpu, B, N, npoint, order, arith):
    from pointrcnn_amd import ops
    xyz = kitti_cloud(B, N, seed=N + 3)
    if N >= 1000:
        xyz[:, N // 2:] = xyz[:, : N - N // 2]               # duplicates: the tie orders differ here
    got = ops.furthest_point_sample_mode(T(xyz, dev), npoint, order=order, arith=arith).cpu().numpy()
    want = cpu.fps_mode(xyz, npoint, order == "upstream", arith == "upstream")
    assert np.array_equal(got, want)
    if order == "canonical" and arith == "canonical":
        assert np.array_equal(got, ops.furthest_point_sample(T(xyz, dev), npoint).cpu().numpy())
        assert np.array_equal(got, cpu.fps(xyz, npoint))


@pytest.mark.parametrize("arith", ["canonical", "upstream"])
def test_ball_query_and_three_nn_arith_match_oracle(dev, cpu, arith):
    from pointrcnn_amd import ops
    xyz = kitti_cloud(2, 4096, seed=9)
    ctr = xyz[:, ::4].copy()
    for r, ns in ((0.5, 16), (2.0, 32), (1e-4, 8)):
        got = ops.ball_query_arith(r, ns, T(xyz, dev), T(ctr, dev), arith=arith).cpu().numpy()
        assert np.array_equal(got, cpu.ball_query_arith(r, ns, xyz, ctr, arith == "upstream"))
        if arith == "canonical":
            assert np.array_equal(got, cpu.ball_query(r, ns, xyz, ctr))
            assert np.array_equal(got, ops.ball_query(r, ns, T(xyz, dev), T(ctr, dev)).cpu().numpy())
    d2, idx = ops.three_nn_arith(T(xyz, dev), T(ctr, dev), arith=arith)
    wd, wi = cpu.three_nn_arith(xyz, ctr, arith == "upstream")
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    if arith == "canonical":
        rd, ri = cpu.three_nn(xyz, ctr)
        assert np.array_equal(wi, ri) and np.array_equal(wd, rd)


def test_the_two_arithmetics_really_differ_and_only_in_the_last_bit(dev, cpu):
    """the contracted form is not the canonical one under another name: on random clouds some squared distances differ -- by one unit in
    the last place at most for the nearest neighbours (both are within half an ulp of the two-rounding / one-rounding results)"""
    from pointrcnn_amd import ops
    xyz = kitti_cloud(1, 8192, seed=21)
    ctr = xyz[:, ::8].copy()
    a, ia = ops.three_nn_arith(T(xyz, dev), T(ctr, dev), arith="canonical")
    b, ib = ops.three_nn_arith(T(xyz, dev), T(ctr, dev), arith="upstream")
    a, b = a.cpu().numpy(), b.cpu().numpy()
    same_idx = (ia == ib).all(dim=2).cpu().numpy()
    diff = a[same_idx] != b[same_idx]
    assert diff.any(), "no distance differs: the upstream mode is not contracting"
    ulp = np.spacing(np.maximum(a[same_idx], b[same_idx]))
    assert (np.abs(a[same_idx].astype(np.float64) - b[same_idx]) <= 2 * ulp).all()


def test_arith_mode_argument_errors(dev):
    from pointrcnn_amd import _cabi
    x = torch.zeros(1, 8, 3, device=dev)
    o = torch.zeros(1, 4, dtype=torch.int32, device=dev)
    t = torch.zeros(1, 8, device=dev)
    L = _cabi.lib()
    s = torch.cuda.current_stream().cuda_stream
    assert L.prcnn_fps_mode(x.data_ptr(), 1, 8, 4, 0, 7, t.data_ptr(), o.data_ptr(), s) == -1        # unknown arithmetic
    assert L.prcnn_fps_mode(x.data_ptr(), 1, 8, 4, 5, 0, t.data_ptr(), o.data_ptr(), s) == -1        # unknown order
    assert L.prcnn_fps_mode(x.data_ptr(), 1, 8, 4, 0, 1, None, o.data_ptr(), s) == -1                # tmp is required
    assert L.prcnn_ball_query_arith(x.data_ptr(), x.data_ptr(), 1, 8, 8, 1.0, 4, 2, o.data_ptr(), s) == -1


def test_disagreement_counter_agrees_with_the_oracle(dev, cpu):
    """tools/arith_disagreement.py (the numbers DESIGN.md section 2 quotes) on a small cloud: its counts == counts made from the
    oracle's outputs under the two arithmetics"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import arith_disagreement as ad
    xyz = kitti_cloud(2, 4096, seed=33)
    rep = ad.compare(T(xyz, dev), npoint=1024, radii=((0.5, 16), (1.0, 32)))
    fa, fb = cpu.fps_mode(xyz, 1024, 0, 0), cpu.fps_mode(xyz, 1024, 0, 1)
    assert rep["fps"]["frames_with_a_different_index"] == int((fa != fb).any(1).sum())
    assert rep["fps"]["indices_different"] == int((fa != fb).sum())
    ctr = np.take_along_axis(xyz, fa[..., None].astype(np.int64), 1)
    for (r, ns), row in zip(((0.5, 16), (1.0, 32)), rep["ball_query"]):
        qa, qb = cpu.ball_query_arith(r, ns, xyz, ctr, 0), cpu.ball_query_arith(r, ns, xyz, ctr, 1)
        assert row["rows_different"] == int((qa != qb).any(2).sum())
    (da, ia), (db, ib) = cpu.three_nn_arith(xyz, ctr, 0), cpu.three_nn_arith(xyz, ctr, 1)
    assert rep["three_nn"]["triples_different"] == int((ia != ib).any(2).sum())
