"""dev tool: FP0-shaped hoisted interp layer, chain kernel vs tiled layer kernel"""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops, rpn
from pointrcnn_amd.opbench import timeit
dev = torch.device("cuda:0")
B, n, m = 32, 16384, 4096
xyz = rpn.synthetic_clouds(B, n, device=dev)
known = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, m))
d2, idx3, w3 = ops.three_nn(xyz, known, want_weight=True)
y = torch.randn(B, m, 128, device=dev)
w = torch.randn(128, 128, device=dev) * 0.1
b = torch.randn(128, device=dev)
l1 = ops.PackedLinear(w, b, relu=True)
b0 = torch.randn(128, device=dev)
for name, fn in (("chain", lambda: ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0)),
                 ("tiled", lambda: ops.mlp_interp(y, idx3, w3, None, l1, act_bias=b0))):
    t = timeit(fn, 10, 2)
    print("%s: %.1f us  %.1f TF/s" % (name, t * 1e6, 2.0 * B * n * 128 * 128 / t / 1e12))
a, c = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0), ops.mlp_interp(y, idx3, w3, None, l1, act_bias=b0)
print("max diff", float((a - c).abs().max()))
