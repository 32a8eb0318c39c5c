import sys, torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(2048, 1024, device=dev)
lin = ops.PackedLinear(torch.randn(512, 1024, device=dev) * 0.05, torch.randn(512, device=dev), relu=True)
out = torch.empty(2048, 512, device=dev)
for _ in range(3): ops.mlp_rows(x, lin, out=(out, 0))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): ops.mlp_rows(x, lin, out=(out, 0))
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print("warm %.1f us" % (s.elapsed_time(e) / 20 * 1e3))
junk = torch.empty(192 << 20, device=dev); cold = []
for _ in range(8):
    junk.add_(1.0); s.record(); ops.mlp_rows(x, lin, out=(out, 0)); e.record(); torch.cuda.synchronize(); cold.append(s.elapsed_time(e) * 1e3)
print("cold %.1f us" % sorted(cold)[4])
