"""dev probe: the plain prcnn_mlp_rows GEMM shapes of the RPN graph (bs32, after hoisting), us and TFLOP/s, next to the
library sgemm torch dispatches to (rocBLAS / hipBLASLt) with its bias + ReLU as separate elementwise kernels -- a yardstick
for what a tuned fp32 GEMM reaches on this part at these sizes (used with ablation builds of the library)."""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = ((2048, 1024, 512, "FP3 Y"), (8192, 512, 512, "FP3 skip / layer2, SA4 Z, FP2 Y"), (32768, 256, 512, "FP2 skip"),
          (32768, 512, 512, "FP2 layer2"), (32768, 512, 256, "FP1 Y"), (32768, 256, 256, "SA3 Z"), (131072, 96, 256, "FP1 skip"),
          (131072, 256, 256, "FP1 layer2"), (131072, 256, 128, "FP0 Y"), (131072, 96, 128, "SA2 Z"), (524288, 128, 128, "FP0 layer2 / heads"))


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for rows, K, N, what in SHAPES:
    x = torch.randn(rows, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    lin = ops.PackedLinear(w, b, relu=True)
    out = torch.empty(rows, N, device=dev)
    us = timeit(lambda: ops.mlp_rows(x, lin, out=(out, 0)))
    wt = w.t().contiguous()
    us_mm = timeit(lambda: torch.mm(x, wt, out=out))
    us_lin = timeit(lambda: torch.relu_(torch.addmm(b, x, wt)))
    fl = 2.0 * rows * K * N
    print("rows %7d K %4d N %4d  ours %7.1f us %6.1f TF/s | torch.mm %7.1f us %6.1f TF/s | addmm+relu %7.1f us  (%s)"
          % (rows, K, N, us, fl / us / 1e6, us_mm, fl / us_mm / 1e6, us_lin, what))
