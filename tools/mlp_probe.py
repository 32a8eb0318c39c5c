"""dev tool: per-launch table of the MLP kernels in one RPN step (HIP events per call): python tools/mlp_probe.py [rpn|rcnn]"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from pointrcnn_amd import _cabi, rpn

which = sys.argv[1] if len(sys.argv) > 1 else "rpn"
dev = torch.device("cuda:0")
torch.manual_seed(0)
_cabi.lib()
if which == "rpn":
    model = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    data = {"pts_input": rpn.synthetic_clouds(32, 16384, device=dev)}
    import os
    if os.environ.get("MORTON"):       # experiment: input cloud pre-sorted along a z-order curve (x-z cells)
        def spread(v):
            v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
            return v
        pts = data["pts_input"]
        qx = ((pts[..., 0] + 40) / 80 * 1023).long().clamp(0, 1023); qz = (pts[..., 2] / 70.4 * 1023).long().clamp(0, 1023)
        key = spread(qx) | (spread(qz) << 1)
        order = key.argsort(dim=1)
        data = {"pts_input": torch.gather(pts, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()}
    fn = lambda: model(data)
else:
    from pointrcnn_amd.point_rcnn import PointRCNN
    model = rpn.randomize_bn_stats(PointRCNN(mode="TEST")).to(dev).eval()
    data = {"pts_input": rpn.synthetic_clouds(32, 16384, device=dev)}
    fn = lambda: model(data)
with torch.no_grad():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    from pointrcnn_amd import ops
    prof = bench.EventProfiler(_cabi._lib)
    real = _cabi._lib
    _cabi._lib, ops._split_log = prof, prof.splits
    try:
        fn()
    finally:
        _cabi._lib, ops._split_log = real, None
torch.cuda.synchronize()
counts = {}
for sp in prof.splits:
    for k, v in enumerate(sp.counts.cpu().tolist()):
        counts[sp.counts.data_ptr() + 4 * k] = v
tot = 0.0
for name, s, e, (per_row, rows, ptr, unit) in prof.records:
    ms = s.elapsed_time(e)
    tot += ms
    live = min(rows, counts[ptr] * unit) if ptr else rows
    fl = per_row * live
    if name.startswith("prcnn_mlp") or ms > 0.05:
        print("%-28s %8.1f us  rows %8d / %8d  %6.0f flop/row %7.2f GFLOP  %6.1f TF/s" % (name[6:], ms * 1e3, live, rows, per_row, fl / 1e9, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
print("total %.2f ms" % tot)
