"""How often do the two squared-distance arithmetics disagree?  

The library's contract is no-FMA (three rounded products summed left to right); the real upstream build is nvcc with default FMA
contraction: fma(dz,dz, fma(dy,dy, dx*dx)).  Both are available behind the C ABI (prcnn_fps_mode / prcnn_ball_query_arith /
prcnn_three_nn_arith, held to the oracle bit for bit: tests/test_gpu_arith_modes.py).  This script runs the RPN's index operators
under both on the bench clouds (uniform, LiDAR-like, and a cloud with every point duplicated) and counts

  FPS         frames whose sample lists differ at all, the first differing position, how many of the npoint indices differ, and
              how many sampled POINTS differ as a set (one flipped near-tie re-orders everything after it in the list, but the
              set of samples changes far less);
  ball_query  query rows (one centroid, one radius) whose nsample indices differ, on the SAME centroids;
  three_nn    unknown points whose 3 neighbour indices differ, on the same known set.

    python tools/arith_disagreement.py [--batch 32] [--out profiles/r06_arith_disagreement.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compare(xyz, npoint=4096, radii=((0.1, 16), (0.5, 32)), known_stride=4):
    """xyz (B,N,3) device tensor -> counts of disagreement between the canonical and the upstream arithmetic"""
    from pointrcnn_amd import ops
    B, N, _ = xyz.shape
    fa = ops.furthest_point_sample_mode(xyz, npoint, arith="canonical")
    fb = ops.furthest_point_sample_mode(xyz, npoint, arith="upstream")
    ne = fa != fb
    first = [int(torch.nonzero(ne[b])[0]) if bool(ne[b].any()) else None for b in range(B)]
    set_diff = 0
    for b in range(B):
        sa, sb = set(fa[b].tolist()), set(fb[b].tolist())
        set_diff += len(sa - sb)
    rep = {"frames": B, "points_per_frame": N, "npoint": npoint,
           "fps": {"frames_with_a_different_index": int(ne.any(1).sum()), "indices_different": int(ne.sum()), "indices_total": B * npoint,
                   "first_difference_at": first, "sampled_points_different_as_a_set": set_diff}}
    ctr = ops.gather_rows(xyz, fa)
    rep["ball_query"] = []
    for r, ns in radii:
        qa = ops.ball_query_arith(r, ns, xyz, ctr, arith="canonical")
        qb = ops.ball_query_arith(r, ns, xyz, ctr, arith="upstream")
        rows = (qa != qb).any(2)
        rep["ball_query"].append({"radius": r, "nsample": ns, "rows_different": int(rows.sum()), "rows_total": int(rows.numel())})
    da, ia = ops.three_nn_arith(xyz, ctr, arith="canonical")
    db, ib = ops.three_nn_arith(xyz, ctr, arith="upstream")
    tri = (ia != ib).any(2)
    rep["three_nn"] = {"triples_different": int(tri.sum()), "triples_total": int(tri.numel()),
                       "squared_distances_different": int((da != db).sum()), "squared_distances_total": int(da.numel())}
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from pointrcnn_amd import rpn
    dev = torch.device("cuda:0")
    out = {"what": "canonical (no-FMA) vs upstream (nvcc-contracted: fma(dz,dz, fma(dy,dy, dx*dx))) squared-distance arithmetic, RPN level 0 "
                   "(16384 -> 4096, radii 0.1 / 0.5, three_nn 16384 -> 4096), %d frames per cloud kind" % args.batch, "clouds": {}}
    kinds = {"uniform": rpn.synthetic_clouds, "lidar": rpn.lidar_like_clouds}
    for name, fn in kinds.items():
        out["clouds"][name] = compare(fn(args.batch, 16384, seed0=100).to(dev))
    dup = rpn.synthetic_clouds(args.batch, 16384, seed0=100)
    dup[:, 8192:] = dup[:, :8192]
    out["clouds"]["uniform, every point twice"] = compare(dup.to(dev))
    # deeper levels see fewer, sparser points: level 1 (4096 -> 1024, radii 0.5 / 1.0)
    lvl1 = rpn.synthetic_clouds(args.batch, 16384, seed0=100).to(dev)
    from pointrcnn_amd import ops
    lvl1 = ops.gather_rows(lvl1, ops.furthest_point_sample(lvl1, 4096))
    out["clouds"]["uniform, level 1 (4096 -> 1024)"] = compare(lvl1, npoint=1024, radii=((0.5, 16), (1.0, 32)))
    s = json.dumps(out, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
