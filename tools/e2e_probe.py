"""dev tool: time the full two-stage eval graph (BASELINE config 3) and its pieces: python tools/e2e_probe.py [B]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import _cabi, rpn
from pointrcnn_amd.point_rcnn import PointRCNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = rpn.randomize_bn_stats(PointRCNN(mode="TEST")).to(dev).eval()
pts = rpn.synthetic_clouds(B, 16384, device=dev)


def run():
    with torch.no_grad():
        out = model({"pts_input": pts})
        return model.detections(out)


for _ in range(2):
    run()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("full PointRCNN eval graph B=%d: %.1f ms/batch  %.0f frames/s" % (B, dt * 1e3, B / dt))

# stage split with events
def timed(fn, n=5):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n):
        r = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r

with torch.no_grad():
    t_rpn, ro = timed(lambda: model.rpn({"pts_input": pts}))
    sc = ro["rpn_cls"][:, :, 0]
    t_prop, (rois, rs) = timed(lambda: model.rpn.proposal_layer(sc, ro["rpn_reg"], ro["backbone_xyz"]))
    info = {"rpn_xyz": ro["backbone_xyz"], "rpn_features": ro["backbone_features"].permute(0, 2, 1),
            "seg_mask": (torch.sigmoid(sc) > 0.3).float(), "roi_boxes3d": rois, "pts_depth": torch.norm(ro["backbone_xyz"], p=2, dim=2)}
    t_pool, _ = timed(lambda: model.rcnn_net.pool_rois(info))
    t_rcnn, rc = timed(lambda: model.rcnn_net(info))
    out = dict(ro); out.update(rois=rois); out.update(rc)
    t_det, _ = timed(lambda: model.detections(out))
print("rpn %.2f ms | proposal %.2f | rcnn %.2f (of which roi pooling + canonical transform %.2f) | detections %.2f" % (t_rpn, t_prop, t_rcnn, t_pool, t_det))
