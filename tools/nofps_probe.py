"""dev experiment: the default bench with FPS (or another op family) replaced by a cached result -- how much of the
saturated step does that family cost?   python tools/nofps_probe.py [bench args]"""
import os
import sys
sys.path.insert(0, ".")
import bench          # noqa: E402  (sets GPU_MAX_HW_QUEUES before torch)
import torch          # noqa: E402
from pointrcnn_amd import ops          # noqa: E402

real = ops.furthest_point_sample
cache = {}


def cached_fps(xyz, npoint):
    key = (tuple(xyz.shape), npoint)
    if key not in cache:
        cache[key] = real(xyz, npoint).clone()
    return cache[key]


if os.environ.get("NOFPS", "1") == "1":
    ops.furthest_point_sample = cached_fps
bench.main()
