"""dev tool: fps_batch_kernel (two samples per exchange, PRCNN_FPS_BATCH) against fps_slot_kernel (PRCNN_FPS_BATCH=0): same indices,
time of 16 384 -> 4 096 at bs32 on the uniform and the LiDAR-like clouds.
    python tools/fps_batch_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops, rpn

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    for _ in range(4):          # (a launch right after a host-side pause runs at the idle clocks: warm up past it)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, gen in (("uniform", rpn.synthetic_clouds), ("lidar", rpn.lidar_like_clouds), ("saturated", rpn.saturated_clouds)):
    for N, npoint in ((16384, 4096), (12000, 3000)):
        xyz = gen(32, 16384, device=dev)[:, :N].contiguous()
        res = {}
        for mode in ("0", "1"):
            os.environ["PRCNN_FPS_BATCH"] = mode
            res[mode] = (ops.furthest_point_sample(xyz, npoint), timeit(lambda: ops.furthest_point_sample(xyz, npoint)))
        same = bool((res["0"][0] == res["1"][0]).all())
        print("%-9s %5d -> %4d  slot %.1f us  batch %.1f us  identical %s" % (name, N, npoint, res["0"][1], res["1"][1], same), flush=True)
        if not same:
            d = (res["0"][0] != res["1"][0])
            print("   first mismatch per frame:", [int(r.nonzero()[0]) if r.any() else -1 for r in d][:8])
