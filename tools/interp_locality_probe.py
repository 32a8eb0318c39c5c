"""dev probe: does the FP interpolation gather care about the ORDER the fine rows are processed in?  Times the FP0 chain and the
FP1 add-interp layer on a cloud in its given (random) point order and on the same cloud Morton-sorted (same rows, same work)."""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops, rpn
dev = torch.device("cuda:0")
torch.manual_seed(0)


def morton_order(x):                                  # (B,N,3) -> (B,N) argsort by interleaved 10-bit x/z cell
    lo, hi = x.amin(1, keepdim=True), x.amax(1, keepdim=True)
    q = ((x - lo) / (hi - lo + 1e-9) * 1023).long()
    key = torch.zeros(x.shape[:2], dtype=torch.long, device=x.device)
    for b in range(10):
        key |= ((q[..., 0] >> b) & 1) << (2 * b + 1)
        key |= ((q[..., 2] >> b) & 1) << (2 * b)
    return key.argsort(1)


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
for N, m, C2, Cs, Nout in ((16384, 4096, 128, 0, 128), (4096, 1024, 256, 96, 256)):
    xyz = rpn.synthetic_clouds(B, N, seed0=100, device=dev)
    known = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, m))
    z = torch.randn(B, m, C2, device=dev)
    lin = ops.PackedLinear(torch.randn(Nout, C2 if Cs == 0 else Cs, device=dev) * 0.05, torch.randn(Nout, device=dev), relu=True)
    bias0 = torch.randn(C2, device=dev)
    res = []
    for name, unk in (("given order", xyz), ("morton order", torch.gather(xyz, 1, morton_order(xyz).unsqueeze(-1).expand(-1, -1, 3)).contiguous())):
        _, idx3, w3 = ops.three_nn(unk, known, want_weight=True)
        if Cs == 0:
            us = timed(lambda: ops.mlp_chain_interp(z, idx3, w3, None, [lin], act_bias=bias0))
        else:
            skip = torch.randn(B, N, Cs, device=dev)
            us = timed(lambda: ops.mlp_rows_addinterp(skip, lin, z, idx3, w3))
        res.append("%s %.1f us" % (name, us))
    print("n=%d m=%d C2=%d skip=%d -> %d: " % (N, m, C2, Cs, Nout) + " | ".join(res))
