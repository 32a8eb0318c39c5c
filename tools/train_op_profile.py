"""dev probe: torch.profiler view of the RPN training step -- which aten ops are the torch-side kernels, host time per step."""
import os
import sys
import time
import torch
sys.path[:0] = ["."]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from pointrcnn_amd import ops, rpn, train_functions as tf
from torch.profiler import profile, ProfilerActivity

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
for k in range(8):
    tr.step(batches[k % 2], next_batch=batches[(k + 1) % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(10):
    tr.step(batches[k % 2], next_batch=batches[(k + 1) % 2])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("10 steps: host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for k in range(3):
        tr.step(batches[k % 2], next_batch=batches[(k + 1) % 2])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50, max_shapes_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
