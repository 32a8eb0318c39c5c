"""dev probe: prcnn_mlp_rows on a few plain GEMM shapes (run once per ablation library, see tools/build_ablation.py)"""
import os
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
fill = os.environ.get("ABL_FILL", "randn")          # randn | zeros | ones: operand data changes the power draw, hence the clock
line = [os.path.basename(os.environ.get("PRCNN_POINTOPS_LIB", "product")) + " fill=" + fill]
for rows, K, N in ((32768, 512, 512), (131072, 256, 256), (8192, 512, 512), (524288, 128, 128)):
    mk = {"randn": torch.randn, "zeros": torch.zeros, "ones": torch.ones}[fill]
    x = mk(rows, K, device=dev)
    lin = ops.PackedLinear(mk(N, K, device=dev) * 0.05, torch.randn(N, device=dev), relu=True)
    out = torch.empty(rows, N, device=dev)
    for _ in range(3):
        ops.mlp_rows(x, lin, out=(out, 0))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        ops.mlp_rows(x, lin, out=(out, 0))
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    line.append("%dx%dx%d %6.1f us %5.1f TF" % (rows, K, N, us, 2.0 * rows * K * N / us / 1e6))
print(" | ".join(line))
