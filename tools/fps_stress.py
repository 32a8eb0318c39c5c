"""dev tool: the FPS kernels against the C oracle on many clouds at full size (the survival shortcut of round 6 and the opt-in batch
kernel take data-dependent paths: a candidate that survives an update, a second sample proved or not) -- uniform, LiDAR-like, saturated,
duplicated and lattice clouds, 16 384 -> 4 096, 12 000 -> 3 000, 4 096 -> 1 024, 8 192 -> 2 048.
    python tools/fps_stress.py [frames per kind, default 24]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import oracle
from pointrcnn_amd import ops, rpn

cpu = oracle.cpu()
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(11)
bad = 0
for kind, gen in (("uniform", rpn.synthetic_clouds), ("lidar", rpn.lidar_like_clouds), ("saturated", rpn.saturated_clouds)):
    base = gen(F, 16384, seed0=1000).numpy()
    dup = base.copy(); dup[:, 8192:] = dup[:, :8192]                 # every point twice
    quant = np.round(base * 4) / 4                                      # 25 cm lattice: many exact ties
    for tag, cl in ((kind, base), (kind + " dup", dup[: F // 4]), (kind + " lattice", quant[: F // 4].astype(np.float32))):
        for N, npoint in ((16384, 4096), (12000, 3000), (8192, 2048), (4096, 1024)):
            xyz = np.ascontiguousarray(cl[:, :N])
            want = cpu.fps(xyz, npoint)
            for batch in ("0", "1"):
                os.environ["PRCNN_FPS_BATCH"] = batch
                got = ops.furthest_point_sample(torch.from_numpy(xyz).to(dev), npoint).cpu().numpy()
                ok = np.array_equal(got, want)
                bad += 0 if ok else 1
                print("%-18s %5d -> %4d  batch=%s  frames %3d  %s" % (tag, N, npoint, batch, xyz.shape[0], "identical" if ok else "MISMATCH in %d frames" % int((got != want).any(1).sum())), flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
