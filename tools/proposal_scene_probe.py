"""dev tool: the batched proposal stage on the 24-car scene of opbench (40 % of the points vote for 24 cars), bs32, normal and rotated NMS,
ten calls each -- run under rocprofv3 --kernel-trace --stats for the per-kernel split (profiles/r06_proposal_layer_kernels.txt).
    python tools/proposal_scene_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from pointrcnn_amd import ops

dev = torch.device("cuda:0")
B, N = 32, 16384
gg = torch.Generator().manual_seed(3)
nfg = int(N * 0.4)
obj = torch.rand(B, 24, 7, generator=gg) * torch.tensor([70., .4, 62., .3, .3, 1., 6.28]) + torch.tensor([-35., .8, 4., 1.4, 1.5, 3.4, -3.14])
own = torch.randint(0, 24, (B, nfg), generator=gg)
fgb = torch.gather(obj, 1, own.unsqueeze(-1).expand(-1, -1, 7)) + torch.randn(B, nfg, 7, generator=gg) * torch.tensor([.08, .03, .12, .03, .03, .06, .03])
bgb = torch.rand(B, N - nfg, 7, generator=gg) * torch.tensor([80., 4., 70., 1., .8, 2., 6.28]) + torch.tensor([-40., -1., .2, 1., 1.2, 3., -3.14])
boxes3d = torch.cat([fgb, bgb], 1).to(dev).contiguous()
scores = torch.cat([torch.randn(B, nfg, generator=gg) + 2.5, torch.randn(B, N - nfg, generator=gg) - 3.0], 1).to(dev)
for rot in (False, True):
    for i in range(10):
        ops.proposal_layer(scores, boxes3d, (6300, 2700), (70, 30), 0.8, rotated=rot)
torch.cuda.synchronize()
