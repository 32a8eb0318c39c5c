"""Grid-accelerated neighbour search (csrc/grid.hip) must reproduce the full scans bit for bit -- checked against the
scan kernels and the CPU oracle on uniform, LiDAR-like, clustered and degenerate clouds."""
import numpy as np
import pytest
import torch

from util import kitti_cloud

pytestmark = pytest.mark.gpu


def _clouds():
    r = np.random.default_rng(0)
    N = 4096
    out = {"kitti": kitti_cloud(2, N, seed=5)}
    rng, ang = r.uniform(2, 70, (2, N)) ** 0.5 * 8.4, r.uniform(-0.7, 0.7, (2, N))        # dense near the sensor
    out["lidar"] = np.stack([rng * np.sin(ang), r.normal(1.2, 0.3, (2, N)), rng * np.cos(ang)], 2).astype(np.float32)
    ctr = r.uniform(-20, 20, (2, 8, 3))
    out["clustered"] = (ctr[:, r.integers(0, 8, N)][np.arange(2)[:, None], np.arange(N)[None, :] * 0 + np.arange(N)[None, :] % 8]
                        + r.normal(0, 0.15, (2, N, 3))).astype(np.float32)
    out["identical"] = np.tile(np.array([[[3.0, 1.0, 20.0]]], np.float32), (2, N, 1))
    line = np.zeros((2, N, 3), np.float32)
    line[..., 0] = np.linspace(-40, 40, N)
    out["line_x"] = line                                                                    # zero extent in z
    dup = kitti_cloud(2, N, seed=6)
    dup[:, N // 2:] = dup[:, :N // 2]                                                       # exact duplicates: index ties
    out["duplicates"] = dup
    bad = kitti_cloud(2, N, seed=7)
    bad[0, 5] = np.nan
    bad[0, 17, 0] = np.inf
    bad[1, 100, 2] = -np.inf
    out["nonfinite"] = bad
    return out


@pytest.mark.parametrize("name", ["kitti", "lidar", "clustered", "identical", "line_x", "duplicates", "nonfinite"])
def test_grid_ball_query_and_three_nn_equal_scan_and_oracle(dev, cpu, name):
    from pointrcnn_amd import ops
    pts = _clouds()[name]
    B, N, _ = pts.shape
    x = torch.from_numpy(pts).to(dev)
    M = 512
    q_np = np.ascontiguousarray(pts[:, ::N // M][:, :M])
    q = torch.from_numpy(q_np).to(dev)
    lib = ops._cabi.lib()
    for cells, (ra, nsa, rb, nsb) in ((128, (0.1, 16, 0.5, 32)), (64, (0.5, 16, 1.0, 32)), (128, (2.0, 8, 4.0, 64))):
        g = ops.Grid(x, max(ra, rb), cells)
        ga, gb = ops.ball_query_grid(g, q, ra, nsa, rb, nsb)
        sa = torch.empty_like(ga)
        sb = torch.empty_like(gb)
        ops._cabi.check(lib.prcnn_ball_query2(x.data_ptr(), q.data_ptr(), B, N, M, ra, nsa, sa.data_ptr(), rb, nsb, sb.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "scan")
        assert torch.equal(ga, sa) and torch.equal(gb, sb), (name, ra, rb)
        assert np.array_equal(ga.cpu().numpy(), cpu.ball_query(ra, nsa, pts, q_np))
        single = ops.ball_query_grid(g, q, rb, nsb)
        assert torch.equal(single, sb)
    # three_nn: queries = another cloud overlapping / partly outside the known points' bounding box
    unk_np = (kitti_cloud(B, 3000, seed=11) * np.array([1.3, 1.0, 1.2], np.float32) - np.array([5, 0, 8], np.float32)).astype(np.float32)
    if name == "nonfinite":
        unk_np[0, 3] = np.nan
        unk_np[1, 9, 0] = np.inf
    unk = torch.from_numpy(unk_np).to(dev)
    g = ops.Grid(x, 0.0, 64 if name != "lidar" else 128)
    d2 = torch.empty((B, 3000, 3), device=dev)
    i3 = torch.empty((B, 3000, 3), dtype=torch.int32, device=dev)
    w3 = torch.empty((B, 3000, 3), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ops._cabi.check(lib.prcnn_three_nn_grid(g.buf.data_ptr(), unk.data_ptr(), B, 3000, N, d2.data_ptr(), i3.data_ptr(), w3.data_ptr(), st), "grid")
    sd, si, sw = torch.empty_like(d2), torch.empty_like(i3), torch.empty_like(w3)
    ops._cabi.check(lib.prcnn_three_nn(unk.data_ptr(), x.data_ptr(), B, 3000, N, sd.data_ptr(), si.data_ptr(), sw.data_ptr(), st), "scan")
    assert torch.equal(i3, si) and torch.equal(d2.view(torch.int32), sd.view(torch.int32)), name
    assert torch.equal(w3.view(torch.int32), sw.view(torch.int32))
    od, oi = cpu.three_nn(unk_np, pts)
    assert np.array_equal(i3.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od, equal_nan=True)


def test_grid_is_used_above_the_threshold_and_matches_full_size(dev, cpu):
    from pointrcnn_amd import ops
    assert ops.GRID_MIN_POINTS == 2048
    pts = kitti_cloud(2, 16384, seed=21)
    x = torch.from_numpy(pts).to(dev)
    new = ops.gather_rows(x, ops.furthest_point_sample(x, 4096))
    ia, ib = ops.ball_query2(0.1, 16, 0.5, 32, x, new)                   # grid path (N = 16384)
    lib, st = ops._cabi.lib(), torch.cuda.current_stream().cuda_stream
    sa, sb = torch.empty_like(ia), torch.empty_like(ib)
    ops._cabi.check(lib.prcnn_ball_query2(x.data_ptr(), new.data_ptr(), 2, 16384, 4096, 0.1, 16, sa.data_ptr(), 0.5, 32, sb.data_ptr(), st), "scan")
    assert torch.equal(ia, sa) and torch.equal(ib, sb)
    d2, i3, w3 = ops.three_nn(x, new, want_weight=True)                  # grid path (m = 4096)
    sd, si, sw = torch.empty_like(d2), torch.empty_like(i3), torch.empty_like(w3)
    ops._cabi.check(lib.prcnn_three_nn(x.data_ptr(), new.data_ptr(), 2, 16384, 4096, sd.data_ptr(), si.data_ptr(), sw.data_ptr(), st), "scan")
    assert torch.equal(i3, si) and torch.equal(d2, sd) and torch.equal(w3, sw)


def test_dense_frames_fall_back_to_the_scan_with_identical_results(dev, cpu):
    """prcnn_grid_build estimates the candidates a ball visits from the cell occupancy; frames above N/32 are answered by the
    index-order scan launched behind the grid kernel (prcnn_ball_query2_grid with xyz).  A batch mixing a sparse KITTI-like
    frame, a saturated 1.6 m cube (every ball full) and a frame on the threshold's other side must come out identical to the
    scan / the oracle whichever kernel took which frame -- and identical to the grid kernel alone (xyz=None)."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(3)
    N, M = 16384, 4096
    sparse = kitti_cloud(1, N, seed=21)[0]
    cube = (r.random((N, 3), dtype=np.float32) * 1.6 + np.array([-0.8, 0.2, 20.0], np.float32)).astype(np.float32)
    medium = (r.random((N, 3), dtype=np.float32) * np.array([12, 3, 12], np.float32)).astype(np.float32)
    pts = np.stack([sparse, cube, medium, cube[::-1].copy()])
    x = torch.from_numpy(pts).to(dev)
    q_np = np.ascontiguousarray(pts[:, ::N // M][:, :M])
    q = torch.from_numpy(q_np).to(dev)
    ra, nsa, rb, nsb = 0.1, 16, 0.5, 32
    g = ops.Grid(x, rb, 128)
    ga, gb = ops.ball_query_grid(g, q, ra, nsa, rb, nsb, xyz=x)                 # grid + scan fallback
    ha, hb = ops.ball_query_grid(g, q, ra, nsa, rb, nsb)                        # grid kernel on every frame
    assert torch.equal(ga, ha) and torch.equal(gb, hb)
    for f in range(4):
        assert np.array_equal(ga[f].cpu().numpy(), cpu.ball_query(ra, nsa, pts[f:f + 1], q_np[f:f + 1])[0]), f
        assert np.array_equal(gb[f].cpu().numpy(), cpu.ball_query(rb, nsb, pts[f:f + 1], q_np[f:f + 1])[0]), f
    # the density estimate itself: the cube is flagged, the KITTI-like frame is not
    fb = ops._cabi.lib().prcnn_grid_bytes(1, N)
    cand = [float(g.buf[f * fb: f * fb + 32].view(torch.float32)[5].item()) for f in range(4)]
    assert cand[0] < N / 32 < cand[1] and cand[3] > N / 32, cand
    one = ops.ball_query_grid(g, q, rb, nsb, xyz=x)                              # single radius through the same path
    assert torch.equal(one, gb)
    assert torch.equal(ops.ball_query2(ra, nsa, rb, nsb, x, q)[1], gb)           # what the modules call


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1024, 256, 67])
def test_grid_three_nn_on_small_known_sets_equals_scan(dev, cpu, m):
    """ops.three_nn takes the grid from 1024 known points on (round 2: FP1's 4096 x 1024 search, 88 -> 47 us incl. the grid
    build): sparse grids (well under one point per cell, many empty rings) must still give the scan's bits -- incl. duplicates"""
    from pointrcnn_amd import ops
    lib = ops._cabi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for seed, dup in ((3, False), (4, True)):
        known_np = kitti_cloud(3, m, seed=seed)
        if dup:
            known_np[:, m // 2:] = known_np[:, :m - m // 2]
        unk_np = kitti_cloud(3, 4096, seed=seed + 10)
        known, unk = torch.from_numpy(known_np).to(dev), torch.from_numpy(unk_np).to(dev)
        g = ops.Grid(known, 0.0)
        d2, i3, w3 = (torch.empty((3, 4096, 3), device=dev), torch.empty((3, 4096, 3), dtype=torch.int32, device=dev),
                      torch.empty((3, 4096, 3), device=dev))
        sd, si, sw = torch.empty_like(d2), torch.empty_like(i3), torch.empty_like(w3)
        ops._cabi.check(lib.prcnn_three_nn_grid(g.buf.data_ptr(), unk.data_ptr(), 3, 4096, m, d2.data_ptr(), i3.data_ptr(), w3.data_ptr(), st), "grid")
        ops._cabi.check(lib.prcnn_three_nn(unk.data_ptr(), known.data_ptr(), 3, 4096, m, sd.data_ptr(), si.data_ptr(), sw.data_ptr(), st), "scan")
        assert torch.equal(i3, si) and torch.equal(d2, sd) and torch.equal(w3, sw)
        od, oi = cpu.three_nn(unk_np, known_np)
        assert np.array_equal(i3.cpu().numpy(), oi)
