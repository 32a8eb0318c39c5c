#!/usr/bin/env python3
"""dev probe: the split-bf16 plain-row layer (prcnn_mlp_rows_split, 3 / 6 terms) next to the fp32-MFMA layer on the plain GEMM
shapes of the RPN graph -- time per launch and error against a float64 product.  Needs a GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops

SHAPES = [(32768, 512, 512), (131072, 256, 256), (8192, 512, 512), (131072, 96, 128), (2048, 1024, 512), (32768, 512, 256),
          (131072, 256, 128), (1000, 64, 96)]


def run(x, lin, terms, iters=20):
    ops.MLP_SPLIT_TERMS = terms
    y = ops.mlp_rows(x, lin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.mlp_rows(x, lin)
    e1.record()
    torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    for rows, K, N in SHAPES:
        x = (torch.randn((rows, K), generator=g) * torch.exp(torch.randn((rows, 1), generator=g))).to(dev)
        w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev)
        b = torch.randn((N,), generator=g).to(dev)
        lin = ops.PackedLinear(w, b, relu=True)
        ref = torch.relu(x[:4096].double() @ w.double().T + b.double())
        scale = (x[:4096].double().abs() @ w.double().abs().T + b.double().abs())
        line = "%7d x %4d -> %3d:" % (rows, K, N)
        for terms in (0, 6, 3):
            y, us = run(x, lin, terms)
            err = ((y[:4096].double() - ref).abs() / scale).max().item()
            line += "  %s %7.1f us (%5.1f TF) err/scale %.2e" % ({0: "f32", 6: "bf16x6", 3: "bf16x3"}[terms], us, 2.0 * rows * K * N / us * 1e-6, err)
        print(line, flush=True)
    # the heads: two-layer chains on 524288 rows
    x = torch.randn((524288, 128), generator=g).to(dev)
    for n1 in (76, 1):
        layers = [ops.PackedLinear((torch.randn((128, 128), generator=g) * 0.1).to(dev), torch.randn((128,), generator=g).to(dev), relu=True),
                  ops.PackedLinear((torch.randn((n1, 128), generator=g) * 0.1).to(dev), torch.randn((n1,), generator=g).to(dev), relu=False)]
        line = "chain 524288 x 128 -> 128 -> %2d:" % n1
        ref = None
        for terms in (0, 6, 3):
            ops.MLP_SPLIT_TERMS = terms
            y = ops.mlp_chain_rows(x, layers)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.mlp_chain_rows(x, layers)
            e1.record()
            torch.cuda.synchronize()
            ref = y if ref is None else ref
            line += "  %s %7.1f us  diff %.2e" % ({0: "f32", 6: "bf16x6", 3: "bf16x3"}[terms], e0.elapsed_time(e1) / 10 * 1e3,
                                                 ((y - ref).abs().max() / ref.abs().max()).item())
        print(line, flush=True)
    # hoisted FP0: interpolated rows through one 128 -> 128 layer
    B, n, m = 32, 16384, 4096
    y = torch.randn((B, m, 128), generator=g).to(dev)
    idx3 = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32).to(dev)
    idx3 = (torch.arange(n, device=dev).view(1, n, 1) // 4 + idx3 % 8).clamp_(max=m - 1).int().contiguous()      # neighbours near n/4: L2-local as in the graph
    w3 = torch.rand((B, n, 3), generator=g).to(dev)
    w3 = (w3 / w3.sum(-1, keepdim=True)).contiguous()
    b0 = torch.randn((128,), generator=g).to(dev)
    l1 = ops.PackedLinear((torch.randn((128, 128), generator=g) * 0.1).to(dev), torch.randn((128,), generator=g).to(dev), relu=True)
    line, ref = "FP0 chain 524288 x interp(128) -> 128:", None
    for terms in (0, 6, 3):
        ops.MLP_SPLIT_TERMS = terms
        yy = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0)
        e1.record()
        torch.cuda.synchronize()
        ref = yy if ref is None else ref
        line += "  %s %7.1f us  diff %.2e" % ({0: "f32", 6: "bf16x6", 3: "bf16x3"}[terms], e0.elapsed_time(e1) / 10 * 1e3,
                                             ((yy - ref).abs().max() / ref.abs().max()).item())
    print(line, flush=True)
    ops.MLP_SPLIT_TERMS = 0


if __name__ == "__main__":
    main()
