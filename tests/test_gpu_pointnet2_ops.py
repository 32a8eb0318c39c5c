"""GPU parity: PointNet++ index/gather operators through the C ABI vs the CPU oracle.
Indices are BIT-EXACT (north star); gathered values are exact copies; grads within 1e-5."""
import numpy as np
import pytest
import torch

from util import kitti_cloud, unit_cloud

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,npoint", [(1, 4096, 512), (3, 64, 64), (2, 100, 37), (2, 512, 128), (3, 1000, 256),
                                        (2, 1024, 256), (2, 2048, 300), (1, 8192, 512), (2, 16384, 1024),
                                        (1, 20000, 200), (1, 1, 1), (2, 130, 130)])
def test_fps_matches_oracle(dev, cpu, B, N, npoint):
    from pointrcnn_amd import ops
    xyz = unit_cloud(B, N, seed=N + npoint)
    got = ops.furthest_point_sample(T(xyz, dev), npoint).cpu().numpy()
    assert np.array_equal(got, cpu.fps(xyz, npoint))


def test_fps_config1_and_full_sa1(dev, cpu):
    """BASELINE config 1 (1x4096, npoint 512) and one frame of the real SA1 problem (16384 -> 4096)"""
    from pointrcnn_amd import ops
    xyz = kitti_cloud(1, 16384)
    got = ops.furthest_point_sample(T(xyz, dev), 4096).cpu().numpy()
    assert np.array_equal(got, cpu.fps(xyz, 4096))


def test_fps_ties_duplicates(dev, cpu):
    """duplicated / identical points: arg-max ties must resolve to the LOWEST index"""
    from pointrcnn_amd import ops
    base = unit_cloud(1, 300, seed=5)
    dup = np.concatenate([base, base, base[:, :100]], 1)             # every point 2-3 times
    same = np.ones((2, 257, 3), np.float32) * 0.25                    # all identical
    grid = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(4), indexing="ij"), -1)
    grid = grid.reshape(1, -1, 3).astype(np.float32)                  # lattice: many exact distance ties
    for xyz, npnt in ((dup, 200), (same, 50), (grid, 512)):
        got = ops.furthest_point_sample(T(xyz, dev), npnt).cpu().numpy()
        assert np.array_equal(got, cpu.fps(xyz, npnt))


def test_fps_full_size_properties(dev, cpu):
    """bs32 x 16384 -> 4096 (the benchmark shape): size-independent properties + two frames vs the oracle"""
    from pointrcnn_amd import ops
    xyz = kitti_cloud(32, 16384)
    idx = ops.furthest_point_sample(T(xyz, dev), 4096).cpu().numpy()
    assert idx.shape == (32, 4096) and idx.min() >= 0 and idx.max() < 16384
    assert (idx[:, 0] == 0).all()
    for b in range(32):
        assert len(np.unique(idx[b])) == 4096          # distinct points => a sample is never repeated
    want = cpu.fps(xyz[[0, 31]], 4096)
    assert np.array_equal(idx[[0, 31]], want)


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("B,N,M,r,ns", [(1, 4096, 512, 0.2, 32), (2, 1000, 77, 0.15, 16), (2, 3000, 300, 0.05, 64),
                                        (1, 2049, 257, 0.3, 5), (2, 64, 64, 0.5, 16)])
def test_ball_query_matches_oracle(dev, cpu, B, N, M, r, ns):
    from pointrcnn_amd import ops
    xyz = unit_cloud(B, N, seed=N)
    new_xyz = xyz[:, cpu.fps(xyz, M)[0]] if M <= N else unit_cloud(B, M, seed=1)
    got = ops.ball_query(r, ns, T(xyz, dev), T(new_xyz, dev)).cpu().numpy()
    assert np.array_equal(got, cpu.ball_query(r, ns, xyz, new_xyz))


def test_ball_query_edge_cases(dev, cpu):
    from pointrcnn_amd import ops
    xyz = unit_cloud(2, 500, seed=3)
    far = unit_cloud(2, 40, seed=4) + 10.0                 # centroids with no neighbour at all -> zeros
    got = ops.ball_query(0.2, 8, T(xyz, dev), T(far, dev)).cpu().numpy()
    assert np.array_equal(got, cpu.ball_query(0.2, 8, xyz, far)) and (got == 0).all()
    tiny = ops.ball_query(1e-6, 8, T(xyz, dev), T(xyz[:, :100].copy(), dev)).cpu().numpy()   # only self in range
    assert np.array_equal(tiny, cpu.ball_query(1e-6, 8, xyz, xyz[:, :100]))
    assert (tiny == np.arange(100)[None, :, None]).all()
    same = np.zeros((1, 300, 3), np.float32)                # all identical: first nsample indices
    g = ops.ball_query(0.1, 16, T(same, dev), T(same[:, :10].copy(), dev)).cpu().numpy()
    assert np.array_equal(g, cpu.ball_query(0.1, 16, same, same[:, :10]))


def test_ball_query2_equals_two_single_queries(dev, cpu):
    from pointrcnn_amd import ops
    xyz = kitti_cloud(2, 16384)
    new_xyz = xyz[:, ::4][:, :1024].copy()
    ia, ib = ops.ball_query2(0.5, 16, 1.0, 32, T(xyz, dev), T(new_xyz, dev))
    assert np.array_equal(ia.cpu().numpy(), cpu.ball_query(0.5, 16, xyz, new_xyz))
    assert np.array_equal(ib.cpu().numpy(), cpu.ball_query(1.0, 32, xyz, new_xyz))


# ------------------------------------------------------------------ three_nn
@pytest.mark.parametrize("B,n,m", [(2, 256, 64), (1, 1024, 256), (2, 1000, 333), (1, 4096, 2500), (2, 50, 2), (1, 7, 1)])
def test_three_nn_matches_oracle(dev, cpu, B, n, m):
    from pointrcnn_amd import ops
    unk, kn = unit_cloud(B, n, seed=n), unit_cloud(B, m, seed=m + 1)
    d2, idx, w = ops.three_nn(T(unk, dev), T(kn, dev), want_weight=True)
    rd2, ridx = cpu.three_nn(unk, kn)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)                       # same rounded ops -> bit equal
    if m >= 3:
        assert np.array_equal(w.cpu().numpy(), cpu.three_weights(rd2))


def test_three_nn_ties_keep_earlier_index(dev, cpu):
    from pointrcnn_amd import ops
    kn = np.tile(unit_cloud(1, 50, seed=9), (1, 3, 1))                 # every known point three times
    unk = unit_cloud(1, 200, seed=10)
    _, idx = ops.three_nn(T(unk, dev), T(kn, dev))
    assert np.array_equal(idx.cpu().numpy(), cpu.three_nn(unk, kn)[1])


# ------------------------------------------------------------------ gather / group / interpolate (+ grads)
def test_gather_group_interp_forward_exact(dev, cpu):
    from pointrcnn_amd import ops
    r = np.random.default_rng(0)
    B, C, N, M, ns = 2, 19, 700, 130, 12
    feat = r.normal(size=(B, C, N)).astype(np.float32)
    idx = r.integers(0, N, (B, M)).astype(np.int32)
    gidx = r.integers(0, N, (B, M, ns)).astype(np.int32)
    assert np.array_equal(ops.gather(T(feat, dev), T(idx, dev)).cpu().numpy(), cpu.gather(feat, idx))
    assert np.array_equal(ops.group(T(feat, dev), T(gidx, dev)).cpu().numpy(), cpu.group(feat, gidx))
    cl = np.ascontiguousarray(feat.transpose(0, 2, 1))
    assert np.array_equal(ops.gather_rows(T(cl, dev), T(idx, dev)).cpu().numpy(),
                          cpu.gather(feat, idx).transpose(0, 2, 1))
    i3 = r.integers(0, N, (B, M, 3)).astype(np.int32)
    w3 = r.random((B, M, 3)).astype(np.float32)
    got = ops.three_interpolate(T(feat, dev), T(i3, dev), T(w3, dev)).cpu().numpy()
    assert np.array_equal(got, cpu.three_interp(feat, i3, w3))        # same op order, no FMA -> bit equal


@pytest.mark.parametrize("B,C,m,n", [(4, 64, 1024, 4096), (2, 37, 513, 3000), (8, 256, 4096, 16384), (3, 36, 6000, 12001)])
def test_three_interpolate_lds_staged_kernel_is_bit_identical(dev, cpu, monkeypatch, B, C, m, n):
    """n >= 2 m with enough (frame, channel group) workgroups: the source rows are staged in LDS (point-major
    three_interp_pm_kernel<8 | 4> when the channel count divides, channel-major three_interp_lds_kernel otherwise /
    PRCNN_INTERP_LAYOUT=rows); bit-equal to the oracle and to the direct kernel (PRCNN_INTERP_DIRECT=1), odd m / ragged
    channel groups included"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(m + n)
    feat = r.normal(size=(B, C, m)).astype(np.float32)
    i3 = r.integers(0, m, (B, n, 3)).astype(np.int32)
    w3 = r.random((B, n, 3)).astype(np.float32)
    monkeypatch.delenv("PRCNN_INTERP_DIRECT", raising=False)
    got = ops.three_interpolate(T(feat, dev), T(i3, dev), T(w3, dev))
    monkeypatch.setenv("PRCNN_INTERP_DIRECT", "1")
    direct = ops.three_interpolate(T(feat, dev), T(i3, dev), T(w3, dev))
    monkeypatch.delenv("PRCNN_INTERP_DIRECT", raising=False)
    monkeypatch.setenv("PRCNN_INTERP_LAYOUT", "rows")
    rows = ops.three_interpolate(T(feat, dev), T(i3, dev), T(w3, dev))
    monkeypatch.delenv("PRCNN_INTERP_LAYOUT", raising=False)
    assert torch.equal(got, direct) and torch.equal(rows, direct)
    if B * C * n <= 4_000_000:
        assert np.array_equal(got.cpu().numpy(), cpu.three_interp(feat, i3, w3))


def test_gather_group_interp_backward(dev, cpu):
    from pointrcnn_amd import ops
    r = np.random.default_rng(1)
    B, C, N, M, ns = 2, 7, 300, 90, 6
    idx = r.integers(0, N, (B, M)).astype(np.int32)
    gidx = r.integers(0, N, (B, M, ns)).astype(np.int32)
    go = r.normal(size=(B, C, M)).astype(np.float32)
    ggo = r.normal(size=(B, C, M, ns)).astype(np.float32)
    np.testing.assert_allclose(ops.gather_grad(T(go, dev), T(idx, dev), N).cpu().numpy(), cpu.gather_grad(go, idx, N),
                               atol=1e-5)
    np.testing.assert_allclose(ops.group_grad(T(ggo, dev), T(gidx, dev), N).cpu().numpy(),
                               cpu.group_grad(ggo, gidx, N), atol=1e-5)
    i3 = r.integers(0, N, (B, M, 3)).astype(np.int32)
    w3 = r.random((B, M, 3)).astype(np.float32)
    np.testing.assert_allclose(ops.three_interpolate_grad(T(go, dev), T(i3, dev), T(w3, dev), N).cpu().numpy(),
                               cpu.three_interp_grad(go, i3, w3, N), atol=1e-5)


def test_autograd_functions_backward(dev, cpu):
    """the drop-in autograd Functions route gradients through the HIP backward kernels"""
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointnet2_lib.pointnet2 import pointnet2_utils as pu
    r = np.random.default_rng(2)
    feat = torch.tensor(r.normal(size=(2, 5, 100)).astype(np.float32), device=dev, requires_grad=True)
    gidx = T(r.integers(0, 100, (2, 30, 4)).astype(np.int32), dev)
    out = pu.grouping_operation(feat, gidx)
    out.sum().backward()
    want = cpu.group_grad(np.ones((2, 5, 30, 4), np.float32), gidx.cpu().numpy(), 100)
    np.testing.assert_allclose(feat.grad.cpu().numpy(), want, atol=1e-5)


def test_op_argument_errors(dev):
    from pointrcnn_amd import ops, _cabi
    with pytest.raises(RuntimeError):
        ops.furthest_point_sample(torch.zeros(1, 10, 3), 4)                       # CPU tensor: no fallback
    with pytest.raises(_cabi.PointOpsError):
        ops.furthest_point_sample(torch.zeros(1, 10, 3, device=dev), 11)          # npoint > N


@pytest.mark.parametrize("pruned", [True, "one-level", "batch", False])
@pytest.mark.parametrize("B,N,npoint", [(2, 16384, 2048), (2, 12000, 1500), (2, 4096, 1024), (1, 8192, 700), (2, 5000, 333), (1, 2049, 64)])
def test_fps_pruned_and_plain_variants_are_bit_identical(dev, cpu, monkeypatch, B, N, npoint, pruned):
    """the FPS kernels for 2048 < N <= 16384 -- the spatially pruned ones (Morton pre-sort + exact bounding-box skip, csrc/fps.hip:
    two-level for N > 8192, the default, and one-level, PRCNN_FPS_SLOTS=0) and the plain register-resident one
    (PRCNN_FPS_PRUNED=0) -- return exactly the oracle's indices, on distinct points, duplicated points (original-index
    tie-break) and lattices"""
    from pointrcnn_amd import ops
    old = ops.FPS_PRUNED
    ops.FPS_PRUNED = bool(pruned)
    if pruned == "one-level":
        monkeypatch.setenv("PRCNN_FPS_SLOTS", "0")
    if pruned == "batch":                                             # two samples per exchange where the second is provable (opt-in)
        monkeypatch.setenv("PRCNN_FPS_BATCH", "1")
    try:
        clouds = [kitti_cloud(B, N, seed=N)]
        dup = clouds[0].copy()
        dup[:, N // 2:] = dup[:, : N - N // 2]                      # every point of the first half twice
        clouds.append(dup)
        g = np.stack(np.meshgrid(np.arange(32), np.arange(8), np.arange(32), indexing="ij"), -1).reshape(1, -1, 3)
        clouds.append(np.tile(g.astype(np.float32)[:, :N], (B, 1, 1)) if g.shape[1] >= N else None)
        for xyz in clouds:
            if xyz is None:
                continue
            got = ops.furthest_point_sample(T(xyz, dev), npoint).cpu().numpy()
            assert np.array_equal(got, cpu.fps(xyz, npoint))
    finally:
        ops.FPS_PRUNED = old


@pytest.mark.parametrize("busy", [False, True, "batch"])
def test_fps_slot_masks_on_ties_in_every_slot_alone_and_under_co_resident_waves(dev, cpu, busy, monkeypatch):
    """round-5 advisor finding: the hand-scheduled slot-mask blocks (csrc/fps.hip WaveMaxEq / wave_max_eq2_16) read a compare's
    SGPR pair from a v_addc the hazard recognizer cannot see.  Directed at them: clouds on which the maximum is held by SEVERAL slots of
    one lane in every sample -- each point 16 / 8 / 2 times (equal Morton codes: the copies are neighbours in the sorted order, so they
    sit in the slots of one lane or of adjacent lanes, 0/1/8/9 included) and a 32 x 16 x 32 lattice -- on the 16-slot kernel (16384 -> 4096), the 4-slot kernel
    (4096 -> 1024) and the single-wave kernels, once on an idle chip and once while matrix products from another stream share the CUs
    (the VALU co-issue conditions differ).  Indices must equal the oracle's (ties -> lowest original index)."""
    from pointrcnn_amd import ops
    if busy == "batch":                      # the same clouds through fps_batch_kernel (ties fail its proof: one sample per round)
        monkeypatch.setenv("PRCNN_FPS_BATCH", "1")
        busy = False
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    stop = [False]

    def spin():
        with torch.cuda.stream(side):
            for _ in range(40):
                torch.mm(a, a)

    rng = np.random.default_rng(5)
    cases = []
    for N, npoint, copies in ((16384, 4096, 16), (16384, 1024, 8), (16384, 2048, 2), (4096, 1024, 4), (4096, 700, 2), (1024, 256, 16), (256, 64, 4)):
        base = rng.uniform([-40, -1, 0], [40, 3, 70.4], (2, N // copies, 3)).astype(np.float32)
        xyz = np.repeat(base, copies, axis=1)
        perm = rng.permutation(N)
        cases.append((xyz[:, perm], npoint))
    g = np.stack(np.meshgrid(np.arange(32), np.arange(16), np.arange(32), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    cases.append((np.tile(g, (2, 1, 1)), 4096))
    cases.append((np.tile(g[:, :4096] * 0.5, (2, 1, 1)), 1024))
    for xyz, npoint in cases:
        want = cpu.fps(xyz, npoint)
        if busy:
            spin()
        got = ops.furthest_point_sample(T(xyz, dev), npoint).cpu().numpy()
        assert np.array_equal(got, want), (xyz.shape, npoint, int((got != want).sum()))
    torch.cuda.synchronize()
    del stop


def test_backward_kernels_at_training_shapes(dev, cpu):
    """group_grad: the padding-aware kernel (nsample 16 / 32 / 64, one lane per group) on real ball-query index tensors (first
    hit repeated), on arbitrary indices and on ragged M / C; three_interpolate_grad: the channels-last accumulator path
    (n, m, C not multiples of 64).  Scatter-adds in another order: 1e-5 of the result's scale."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(9)
    B, N = 2, 3000
    xyz = r.random((B, N, 3), dtype=np.float32) * np.array([8, 2, 8], np.float32)
    for ns, M, C, radius in ((16, 333, 37, 0.25), (32, 256, 96, 0.4), (64, 130, 5, 0.6)):
        new_xyz = np.ascontiguousarray(xyz[:, :M])
        bq = cpu.ball_query(radius, ns, xyz, new_xyz)                       # padded with the first hit
        rnd = r.integers(0, N, (B, M, ns)).astype(np.int32)                 # arbitrary indices: no padding structure at all
        same = np.repeat(r.integers(0, N, (B, M, 1)).astype(np.int32), ns, 2)     # every entry the same point
        for gidx in (bq, rnd, same):
            ggo = r.normal(size=(B, C, M, ns)).astype(np.float32)
            want = cpu.group_grad(ggo, gidx, N)
            got = ops.group_grad(T(ggo, dev), T(gidx, dev), N).cpu().numpy()
            assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    for n, m, C in ((1000, 257, 70), (4096, 1024, 128), (65, 64, 3)):
        i3 = r.integers(0, m, (B, n, 3)).astype(np.int32)
        w3 = r.random((B, n, 3)).astype(np.float32)
        go = r.normal(size=(B, C, n)).astype(np.float32)
        want = cpu.three_interp_grad(go, i3, w3, m)
        got = ops.three_interpolate_grad(T(go, dev), T(i3, dev), T(w3, dev), m).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("B,C,N,M,ns", [(2, 96, 4096, 1024, 32), (3, 16, 1000, 260, 16), (1, 8, 64, 64, 4), (2, 24, 4096, 512, 64)])
def test_grouping_lds_staged_kernel_is_exact(dev, cpu, monkeypatch, B, C, N, M, ns):
    """round 6: grouping_operation with the source rows staged point-major in LDS (C a multiple of 8, at least four outputs per source
    point) == the plain gather kernel (PRCNN_GATHER_DIRECT=1) == the oracle, element for element (pure copies); output slices per
    workgroup when B x C / 8 alone would not fill the chip"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(B * 1000 + C)
    feat = r.normal(size=(B, C, N)).astype(np.float32)
    idx = r.integers(0, N, size=(B, M, ns)).astype(np.int32)
    idx[:, 0] = N - 1
    idx[:, -1] = 0
    want = cpu.group(feat, idx)
    got = ops.group(T(feat, dev), T(idx, dev)).cpu().numpy()
    assert np.array_equal(got, want)
    monkeypatch.setenv("PRCNN_GATHER_DIRECT", "1")
    assert np.array_equal(ops.group(T(feat, dev), T(idx, dev)).cpu().numpy(), want)
