"""dev tool: the batched proposal stage (prefiltered greedy NMS, axis-aligned and rotated; the overlap bound that skips polygon clips)
against the C oracle on many driving-like scenes (tests/util.py: rpn_like_scene -- cars with tight clusters of votes), several
thresholds and object counts.
    python tools/nms_stress.py [scenes per setting, default 8]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle
from pointrcnn_amd import ops
from util import ANCHOR, rpn_like_scene

cpu = oracle.cpu()
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for seed, (nobj, fg) in enumerate(((24, 0.4), (6, 0.6), (60, 0.3), (24, 0.05))):
    xyz, sc, reg = rpn_like_scene(F, 16384, seed=100 + seed, nobj=nobj, fg_frac=fg)
    boxes = ops.decode_bbox_target(torch.from_numpy(xyz.reshape(-1, 3)).to(dev), torch.from_numpy(reg.reshape(-1, 76)).to(dev), 3.0, 0.5, 12, ANCHOR,
                                   get_xz_fine=True, y_to_bottom=True).view(F, 16384, 7)
    bnp = boxes.cpu().numpy()
    for kind in ("normal", "rotated"):
        for thr in (0.85, 0.8, 0.7, 0.5):
            for pre, post in (((6300, 2700), (70, 30)), ((6300, 2700), (210, 90))):
                rois, scores, cnt = ops.proposal_layer(torch.from_numpy(sc).to(dev), boxes, pre, post, thr, rotated=kind == "rotated")
                o = cpu.proposal_layer(sc, bnp, pre, post, thr, kind)
                ok = np.array_equal(rois.cpu().numpy(), o[0]) and np.array_equal(cnt.cpu().numpy(), o[2]) and np.array_equal(scores.cpu().numpy(), o[1], equal_nan=True)
                bad += 0 if ok else 1
                print("objects %2d fg %.2f  %-7s thr %.2f  post %s  frames %d  %s" % (nobj, fg, kind, thr, post, F, "identical" if ok else "MISMATCH"), flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
