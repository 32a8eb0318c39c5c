"""dev tool: where the cycles of mlp_chain_c_kernel go (library built with -DMLP_TIMING):
    python tools/build_variant.py mlp.hip mlptiming -DMLP_TIMING
    PRCNN_POINTOPS_LIB=pointrcnn_amd/lib/libprcnn_mlptiming.so python tools/mlp_timing.py
Runs the three launches of the RPN graph that use the kernel (hoisted FP0 128->128 on interpolated rows, the two heads) at the bench shapes
and prints wave 0's cycles per workgroup by phase (mean over the workgroups)."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import _cabi, ops

dev = torch.device("cuda:0")
L = _cabi.lib()
L.prcnn_debug_mlp_timing.restype = ctypes.c_int
L.prcnn_debug_mlp_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["prologue(meta+ptrs+stage0)", "finish+lds-write (waits rows)", "lds-read+split+mfma", "stage store", "barrier", "layer1 stage0", "layer1 mfma", "epilogue"]


def report(tag):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.prcnn_debug_mlp_timing(buf, 1)
    t = np.array(list(buf), dtype=np.float64)
    n = max(t[8], 1)
    print("%s: %d workgroups, cycles per workgroup (wave 0): total %.0f" % (tag, n, t[:8].sum() / n))
    for k in range(8):
        if t[k]:
            print("    %-32s %8.0f" % (names[k], t[k] / n))


B, n, m = 32, 16384, 4096
g = torch.Generator().manual_seed(0)
y = torch.randn(B, m, 128, generator=g).to(dev)
unk = torch.rand(B, n, 3, generator=g).to(dev) * 70
kn = unk[:, ::4].contiguous()
_, idx3, w3 = ops.three_nn(unk, kn, want_weight=True)
w = (torch.randn(128, 128, generator=g) * 0.1).to(dev)
b = torch.randn(128, generator=g).to(dev)
lin = ops.PackedLinear(w, b, relu=True)
bias0 = torch.randn(128, generator=g).to(dev)
L.prcnn_debug_mlp_timing(None, 1)
for _ in range(3):
    out = ops.mlp_chain_interp(y, idx3, w3, None, [lin], act_bias=bias0)
report("FP0 chain_interp 128->128, 524288 rows (3 launches)")
x = torch.randn(B * n, 128, generator=g).to(dev)
l1 = ops.PackedLinear((torch.randn(76, 128, generator=g) * 0.1).to(dev), torch.randn(76, generator=g).to(dev), relu=False)
l1c = ops.PackedLinear((torch.randn(1, 128, generator=g) * 0.1).to(dev), torch.randn(1, generator=g).to(dev), relu=False)
for _ in range(3):
    ops.mlp_chain_rows(x, [lin, l1])
report("head 128->128->76 (3 launches)")
for _ in range(3):
    ops.mlp_chain_rows(x, [lin, l1c])
report("head 128->128->1 (3 launches)")
