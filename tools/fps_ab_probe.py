#!/usr/bin/env python3
"""dev probe: furthest point sampling under an environment switch (read once per process): time per launch on the benchmark
clouds and the sample sets, saved for comparison with another setting's run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops, rpn

def main():
    dev, tag, out = "cuda", sys.argv[1], {}
    for B, N, M in ((32, 16384, 4096), (32, 4096, 1024), (1, 16384, 4096), (32, 8192, 2048)):
        pts = rpn.synthetic_clouds(B, N, device=dev)
        idx = ops.furthest_point_sample(pts, M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.furthest_point_sample(pts, M)
        e1.record()
        torch.cuda.synchronize()
        print("%s B=%d %d -> %d: %.1f us" % (tag, B, N, M, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
        out[(B, N, M)] = idx.cpu()
    torch.save(out, "gpurun_out/fps_idx_%s.pt" % tag)

if __name__ == "__main__":
    main()
