"""dev probe: is the hoisted-FP addY layer / the hoisted FP0 chain bound by the VOLUME of its gathers?  Same launches with the
three neighbour indices of every row drawn from (a) the real 3-NN search, (b) 64 rows per frame, (c) one row."""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops, rpn
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3


B = 32
for N, m, C2, Cs, Nout in ((16384, 4096, 128, 0, 128), (4096, 1024, 256, 96, 256), (1024, 256, 512, 256, 512)):
    xyz = rpn.synthetic_clouds(B, 16384, seed0=100, device=dev)[:, :N].contiguous()
    known = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, m))
    z = torch.randn(B, m, C2, device=dev)
    lin = ops.PackedLinear(torch.randn(Nout, C2 if Cs == 0 else Cs, device=dev) * 0.05, torch.randn(Nout, device=dev), relu=True)
    bias0 = torch.randn(C2, device=dev)
    _, idx_real, w3 = ops.three_nn(xyz, known, want_weight=True)
    res = []
    for name, idx3 in (("real", idx_real), ("64 rows", (idx_real % 64).contiguous()), ("1 row", torch.zeros_like(idx_real))):
        if Cs == 0:
            us = timed(lambda: ops.mlp_chain_interp(z, idx3, w3, None, [lin], act_bias=bias0))
        else:
            skip = torch.randn(B, N, Cs, device=dev)
            us = timed(lambda: ops.mlp_rows_addinterp(skip, lin, z, idx3, w3))
        res.append("%s %.1f us" % (name, us))
    print("n=%d m=%d C2=%d skip=%d -> %d: " % (N, m, C2, Cs, Nout) + " | ".join(res))
