"""dev probe: segment-prefix live rows -- time prcnn_mlp_rows / chain_rows with and without seg on RCNN-sized inputs"""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
R, S = 3200, 512
x = torch.randn(R * S, 256, device=dev)
pts = torch.randn(R * S, 5, device=dev)
cnt = torch.randint(20, 90, (R,), device=dev, dtype=torch.int32)
lin = ops.PackedLinear(torch.randn(128, 256, device=dev) * 0.1, torch.randn(128, device=dev), relu=True)
up = [ops.PackedLinear(torch.randn(128, 5, device=dev) * 0.1, torch.randn(128, device=dev), relu=True),
      ops.PackedLinear(torch.randn(128, 128, device=dev) * 0.1, torch.randn(128, device=dev), relu=True)]


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for seg in (None, (cnt, S)):
    print("seg" if seg else "all", "mlp_rows %.1f us" % t(lambda: ops.mlp_rows(x, lin, seg=seg)),
          "chain_rows %.1f us" % t(lambda: ops.mlp_chain_rows(pts, up, seg=seg)))
a = ops.mlp_rows(x, lin)
b = ops.mlp_rows(x, lin, seg=(cnt, S))
live = (torch.arange(S, device=dev)[None, :] < cnt[:, None].long()).view(-1)
print("live rows equal:", torch.equal(a[live], b[live]), "live frac %.3f" % live.float().mean().item())
