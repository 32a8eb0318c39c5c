"""dev probe: where does the wall time of one RPN training step go?  host enqueue time vs GPU time per phase, with and
without synchronisation between phases (bench.py --workload train reports 43 ms per step while the GPU is busy 29 ms)."""
import os
import sys
import time
import torch
sys.path[:0] = ["."]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pointrcnn_amd import ops, rpn, train_functions as tf

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = tf.init_rpn_head_weights(rpn.randomize_bn_stats(rpn.RPN(), seed=7)).to(dev)
tr = tf.RPNTrainer(model, ddp=False)
B, N = 16, 16384
g = torch.Generator().manual_seed(1)
batches = []
for s in range(2):
    pts = rpn.synthetic_clouds(B, N, seed0=100 + s * B).to(dev)
    pick = torch.randint(0, N, (B, 12), generator=g).to(dev)
    ctr = torch.gather(pts, 1, pick[..., None].expand(-1, -1, 3))
    hwl = torch.tensor([1.56, 1.6, 3.9], device=dev) * 2.0
    ry = (torch.rand((B, 12, 1), generator=g) * 6.283 - 3.1416).to(dev)
    gt = torch.cat([ctr[..., 0:1], ctr[..., 1:2] + hwl[0] / 2, ctr[..., 2:3], hwl.expand(B, 12, 3), ry], 2).contiguous()
    cls, reg = ops.rpn_labels(pts, gt)
    batches.append({"pts_input": pts, "rpn_cls_label": cls.long(), "rpn_reg_label": reg})
for k in range(4):
    tr.step(batches[k % 2])
torch.cuda.synchronize()


def phases(sync):
    tr.model.train()
    t = [time.perf_counter()]
    tr.optimizer.zero_grad(set_to_none=True)
    out = tr.model({"pts_input": batches[0]["pts_input"]})
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    loss = tf.get_rpn_loss(out["rpn_cls"], out["rpn_reg"], batches[0]["rpn_cls_label"], batches[0]["rpn_reg_label"], tr.cfg, tr.cls_loss_func)
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    loss.backward()
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(model.parameters(), tr.cfg.GRAD_NORM_CLIP)
    tr.optimizer.step()
    if sync: torch.cuda.synchronize()
    t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]


for sync in (True, False, True, False):
    torch.cuda.synchronize()
    p = phases(sync)
    torch.cuda.synchronize()
    print("sync between phases" if sync else "host enqueue only  ", "forward %.2f  loss %.2f  backward %.2f  clip+opt %.2f ms" % tuple(p), flush=True)
for n in (1, 5, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        tr.step(batches[k % 2])
    torch.cuda.synchronize()
    print("%d steps back to back: %.2f ms per step" % (n, 1e3 * (time.perf_counter() - t0) / n), flush=True)
