"""dev probe: what the two-stage graph's padding-free lists hold against their worst-case sizes, and the peak memory of one eager batch
(the worst-case-sized RoI-stage buffers are what caps the number of in-flight batches: DESIGN.md section 8)."""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops, rpn
from pointrcnn_amd.point_rcnn import PointRCNN

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
torch.manual_seed(1234)
model = rpn.randomize_bn_stats(PointRCNN(mode="TEST"), seed=7).to(dev).eval()
clouds = {"uniform": rpn.synthetic_clouds, "lidar": rpn.lidar_like_clouds, "saturated": rpn.saturated_clouds}[kind](32, 16384, seed0=100).to(dev)
with torch.no_grad():
    for _ in range(2):
        out = model({"pts_input": clouds})
        model.detections(out)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ops._split_log = []
    out = model({"pts_input": clouds})
    model.detections(out)
    torch.cuda.synchronize()
    print("clouds %s: peak memory of one batch above the resident state: %.2f GB" % (kind, (torch.cuda.max_memory_allocated() - base) / 2 ** 30))
    for sp in ops._split_log:
        if getattr(sp, 'slice_count', False):
            continue
        c = sp.counts.cpu().tolist()
        print("  GroupSplit G=%7d ns=%2d flat cap %9d rows: flat rows %9d (%.3f), dense groups %7d (%.3f of G), sparse groups %7d"
              % (sp.G, sp.ns, sp.max_rows, c[0], c[0] / sp.max_rows, c[1], c[1] / sp.G, c[2]))
    ops._split_log = None

# where the big allocations come from: every torch.empty / zeros above 256 MB with its caller
import traceback
_empty = torch.empty
big = []


def logged_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel() * t.element_size() >= 256 * 2 ** 20:
        fr = traceback.extract_stack(limit=4)[:-1]
        big.append((t.numel() * t.element_size() / 2 ** 30, tuple(t.shape), " <- ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in reversed(fr))))
    return t


torch.empty = logged_empty
with torch.no_grad():
    out = model({"pts_input": clouds})
    model.detections(out)
torch.cuda.synchronize()
torch.empty = _empty
for gb, shape, where in big:
    print("  %.2f GB %s  %s" % (gb, shape, where))
