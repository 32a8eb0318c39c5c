"""dev probe: the two-layer stack kernel on the SA3 / SA4 flat-list shapes (run once per ablation library, tools/build_ablation.py)"""
import os
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
line = [os.path.basename(os.environ.get("PRCNN_POINTOPS_LIB", "product"))]
for rows, nsrc, K0, N0, N1 in ((8192, 32768, 128, 196, 256), (2048, 8192, 256, 256, 512), (2048, 8192, 256, 384, 512), (16384, 32768, 128, 196, 256)):
    xyz = torch.randn(1, nsrc, 3, device=dev)
    # the host-side bound is 4x the live rows, as in the padding-free lists of the real graph (device-side count)
    ctr = torch.randn(1, 4 * rows, 3, device=dev)
    idx = torch.randint(0, nsrc, (1, 4 * rows, 1), device=dev, dtype=torch.int32)
    gd = torch.tensor([rows], device=dev, dtype=torch.int32)
    feat = torch.randn(1, nsrc, K0, device=dev)
    layers = [ops.PackedLinear(torch.randn(N0, K0, device=dev) * 0.05, torch.randn(N0, device=dev), relu=True),
              ops.PackedLinear(torch.randn(N1, N0, device=dev) * 0.05, torch.randn(N1, device=dev), relu=True)]
    act = (torch.randn(K0, 3, device=dev), torch.randn(K0, device=dev))
    out = torch.empty(4 * rows, N1, device=dev)
    for _ in range(3):
        ops.mlp_chain_group(xyz, ctr, idx, feat, layers, out=(out, 0), act=act, groups_dev=gd)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.mlp_chain_group(xyz, ctr, idx, feat, layers, out=(out, 0), act=act, groups_dev=gd)
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    # cold: the way the launch runs inside a step (once, after other kernels swept the caches) -- eager, event-timed, flushed
    junk = torch.empty(192 << 20, device=dev)
    cold = []
    for _ in range(8):
        junk.add_(1.0)
        s.record()
        ops.mlp_chain_group(xyz, ctr, idx, feat, layers, out=(out, 0), act=act, groups_dev=gd)
        e.record(); torch.cuda.synchronize()
        cold.append(s.elapsed_time(e) * 1e3)
    line.append("%dx%d-%d-%d warm %5.1f us (%5.1f TF) cold %5.1f us" % (rows, K0, N0, N1, us, 2.0 * rows * (K0 * N0 + N0 * N1) / us / 1e6,
                                                                         sorted(cold)[len(cold) // 2]))
print(" | ".join(line))
