"""dev tool: where the cycles of nms_sweep_kernel go (library built with -DSWEEP_TIMING, tools/build_variant.py):
    python tools/build_variant.py iou3d.hip sweeptiming -DSWEEP_TIMING
    PRCNN_POINTOPS_LIB=pointrcnn_amd/lib/libprcnn_sweeptiming.so python tools/nms_timing.py [N]
prints, per 64-box block, the resolver's and folder wave 1's cycles by phase."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import _cabi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
c = torch.rand(N, 2, generator=g) * torch.tensor([80.0, 70.0])
s = torch.rand(N, 2, generator=g) * torch.tensor([0.5, 1.5]) + torch.tensor([0.8, 1.7])
bev = torch.cat([c - s, c + s, (torch.rand(N, 1, generator=g) - 0.5) * 6.28], 1).to(dev)
L = _cabi.lib()
wsb = L.prcnn_nms_workspace_bytes(N)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
keep = torch.empty((N,), dtype=torch.int64, device=dev)
num = torch.empty((1,), dtype=torch.int32, device=dev)
for kind in (1, 0):
    for _ in range(3):
        _cabi.check(L.prcnn_nms(bev.data_ptr(), N, 0.8, kind, 0, keep.data_ptr(), num.data_ptr(), ws.data_ptr(), wsb,
                                torch.cuda.current_stream().cuda_stream), "prcnn_nms")
    torch.cuda.synchronize()
    t = ws[:128].view(torch.int64).cpu().numpy().astype(np.float64)
    W = (N + 63) // 64
    print("kind %s N %d W %d kept %d" % ("normal" if kind else "rotated", N, W, int(num.item())))
    print("  resolver cycles/block: remv-read %.0f  resolve %.0f  bookkeeping %.0f  barrier %.0f  carry+prefetch %.0f   total %.0f"
          % tuple(list(t[:5] / W) + [t[:5].sum() / W]))
    print("  folder-1 cycles/block: consume %.0f  prefetch-issue %.0f  barrier %.0f  stop-read %.0f   total %.0f"
          % tuple(list(t[8:12] / W) + [t[8:12].sum() / W]))
