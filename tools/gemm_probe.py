"""dev probe: plain prcnn_mlp_rows GEMM shapes of the RPN graph, us and TFLOP/s (used with ablation builds of the library)"""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for rows, K, N in ((32768, 512, 512), (131072, 256, 256), (131072, 256, 128), (8192, 512, 512), (524288, 128, 128)):
    x = torch.randn(rows, K, device=dev)
    lin = ops.PackedLinear(torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev), relu=True)
    out = torch.empty(rows, N, device=dev)
    ops.mlp_rows(x, lin, out=(out, 0)); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.mlp_rows(x, lin, out=(out, 0))
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print("rows %7d K %4d N %4d  %7.1f us  %6.1f TF/s" % (rows, K, N, us, 2.0 * rows * K * N / us / 1e6))
