"""roipool3d A/B: binned vs linear point selection at BASELINE configs 3 and 5 (run on the GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from pointrcnn_amd import ops, rpn
from pointrcnn_amd.opbench import timeit


def rois_for(x, M, seed, dev):
    gg = torch.Generator().manual_seed(seed)
    pick = torch.randint(0, x.shape[1], (x.shape[0], M), generator=gg).to(dev)
    ctr = torch.gather(x, 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    sz = torch.tensor([1.6 + 2, 1.7 + 2, 4.0 + 2], device=dev).expand(x.shape[0], M, 3)
    ry = (torch.rand(x.shape[0], M, 1, generator=gg).to(dev) - 0.5) * 6.28
    return torch.cat([ctr[..., 0:1], ctr[..., 1:2] + 1.8, ctr[..., 2:3], sz, ry], 2).contiguous()


def main():
    dev = torch.device("cuda:0")
    for name, B, N, M in (("config3", 32, 16384, 100), ("config5", 8, 65536, 512)):
        x = rpn.synthetic_clouds(B, N, seed0=500, device=dev)
        pf = torch.randn(B, N, 130, device=dev)
        r = rois_for(x, M, 2, dev)
        out = {}
        for bins in (True, False):
            ops.ROIPOOL_BINS = bins
            out["bins" if bins else "linear"] = round(timeit(lambda: ops.roipool3d(x, r, pf, 512), 10, 2) * 1e6, 1)
        a = ops.roipool3d(x, r, pf, 512)
        ops.ROIPOOL_BINS = True
        b = ops.roipool3d(x, r, pf, 512)
        out["equal"] = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
        out["avg_pts_in_box"] = float((a[0][..., 3:].abs().sum(-1) > 0).float().sum(-1).mean())
        r8 = r[:, :8].contiguous()
        out["m8_bins_us"] = round(timeit(lambda: ops.roipool3d(x, r8, pf, 512), 10, 2) * 1e6, 1)      # ~ the bin build alone
        comp = B * (M * 512 * 133 * 4 + N * 133 * 4)
        out["frac_of_6p3_bins"] = round(comp / (out["bins"] * 1e-6) / 6.3e12, 3)
        print(name, json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
