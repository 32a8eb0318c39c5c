"""Host-side mirror of the reference's RPN inference graph, built from the drop-in modules.

Mirrors lib/net/pointnet2_msg.py:12-70 (Pointnet2MSG: 4 SA-MSG + 4 FP) and lib/net/rpn.py:12-82 (RPN: backbone
+ classification / regression Conv1d heads) with the shapes of tools/cfgs/default.yaml:23-64, using the SAME
attribute names (`backbone_net.SA_modules`, `FP_modules`, `rpn_cls_layer`, `rpn_reg_layer`) so that the `rpn.*`
entries of a reference checkpoint load with `load_state_dict`.  Only the inference subset is mirrored (no losses); the
proposal layer lives in pointrcnn_amd/proposal_layer.py and is attached by pointrcnn_amd/point_rcnn.py.

This is what bench.py times; the reference's own unchanged lib/net/rpn.py builds the same graph through
`pointrcnn_amd.install()`.
"""
import os
import threading

import torch
import torch.nn as nn

import pointrcnn_amd

pointrcnn_amd.install()
from pointnet2_lib.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG  # noqa: E402
import pointnet2_lib.pointnet2.pytorch_utils as pt_utils  # noqa: E402


from . import ops  # noqa: E402

# sampling chain of all SA levels ahead of the set-abstraction work, on a side stream (see Pointnet2MSG._sample_ahead):
# "auto" = in the training step only.  Measured on MI355X: training step 51.0 -> 48.8 ms; one inference batch in flight
# 8.82 -> 8.52 ms per bs32 step (PRCNN_FPS_AHEAD=1); but 20 captured batches in flight 12.4 k -> 8.9 k frames/s -- every
# graph's fork asks for a second hardware queue, and more than ~20 busy queues collide (bench.py, stream-count sweep).
FPS_AHEAD = os.environ.get("PRCNN_FPS_AHEAD", "auto")
_SIDE_STREAMS = {}
_SIDE_LOCK = threading.Lock()


class RPNConfig:
    """tools/cfgs/default.yaml:23-64 (RPN section), the values that fix every shape on the path."""
    USE_INTENSITY = False
    USE_BN = True
    NUM_POINTS = 16384
    SA_NPOINTS = [4096, 1024, 256, 64]
    SA_RADIUS = [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]]
    SA_NSAMPLE = [[16, 32], [16, 32], [16, 32], [16, 32]]
    SA_MLPS = [[[16, 16, 32], [32, 32, 64]],
               [[64, 64, 128], [64, 96, 128]],
               [[128, 196, 256], [128, 196, 256]],
               [[256, 256, 512], [256, 384, 512]]]
    FP_MLPS = [[128, 128], [256, 256], [512, 512], [512, 512]]
    CLS_FC = [128]
    REG_FC = [128]
    DP_RATIO = 0.5
    LOC_XZ_FINE = True
    LOC_SCOPE = 3.0
    LOC_BIN_SIZE = 0.5
    NUM_HEAD_BIN = 12


class Pointnet2MSG(nn.Module):
    def __init__(self, input_channels=0, use_xyz=True, cfg=RPNConfig):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels
        skip_channel_list = [input_channels]
        for k in range(len(cfg.SA_NPOINTS)):
            mlps = [list(m) for m in cfg.SA_MLPS[k]]
            channel_out = 0
            for idx in range(len(mlps)):
                mlps[idx] = [channel_in] + mlps[idx]
                channel_out += mlps[idx][-1]
            self.SA_modules.append(PointnetSAModuleMSG(npoint=cfg.SA_NPOINTS[k], radii=cfg.SA_RADIUS[k],
                                                       nsamples=cfg.SA_NSAMPLE[k], mlps=mlps, use_xyz=use_xyz,
                                                       bn=cfg.USE_BN))
            skip_channel_list.append(channel_out)
            channel_in = channel_out
        self._prefetched = None
        self.FP_modules = nn.ModuleList()
        for k in range(len(cfg.FP_MLPS)):
            pre_channel = cfg.FP_MLPS[k + 1][-1] if k + 1 < len(cfg.FP_MLPS) else channel_out
            self.FP_modules.append(PointnetFPModule(mlp=[pre_channel + skip_channel_list[k]] + cfg.FP_MLPS[k]))

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def prefetch_samples(self, pointcloud, ready=None):
        """Draw the sample sets of a FUTURE forward pass now, on the side stream, underneath whatever the current stream is
        doing (the rest of this training step).  The furthest-point-sampling chain depends on the input coordinates only and is
        a serial 4.5 ms chain on one workgroup per frame: drawn inside the step it is the first thing every level waits for;
        drawn one step ahead -- the next batch is known to any prefetching data loader -- it costs 16 of 256 CUs for a few
        milliseconds.  The next forward() on the SAME tensor object (identity, shape, version) picks the result up; any other
        input drops it.
        pointcloud: the (B, N, 3+C) tensor the next forward will be called with.  Its producer must be visible to the side
        stream: pass `ready` (an event recorded after the copy / kernel that wrote it, on whatever stream that was -- an
        asynchronous H2D copy on a loader stream, typically); with ready=None the tensor must already be MATERIALISED, i.e.
        every write to it was enqueued on the current stream before the work the side stream is allowed to overtake -- then an
        event recorded on the current stream right here orders the side stream behind those writes and nothing else.
        The xyz slice of a (B, N, 3+C) cloud with C > 0 is a copy kernel: it is enqueued on the SIDE stream, after that event
        (round-3 advisor finding: on the current stream it would have queued behind this step's forward while the side stream
        read its output)."""
        cur = torch.cuda.current_stream(pointcloud.device)
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(cur)
        with torch.no_grad():
            side = self._side_stream(pointcloud.device, cur)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                xyz, _ = self._break_up_pc(pointcloud)
            side, ahead = self._sample_ahead(xyz, wait_current=False)
        # the key holds the tensor ITSELF: an identity test at pick-up cannot be fooled by a new tensor landing at a dropped
        # batch's address (same pointer, shape, version 0)
        self._prefetched = (pointcloud, self._pc_key(pointcloud), side, ahead, xyz)

    @staticmethod
    def _pc_key(pc):
        return (pc.data_ptr(), tuple(pc.shape), tuple(pc.stride()), pc._version)

    @staticmethod
    def _side_stream(device, cur):
        key = (device.index, cur.cuda_stream)          # the default stream's handle is 0 on EVERY device
        with _SIDE_LOCK:                               # DataParallel-style callers run one thread per device
            side = _SIDE_STREAMS.get(key)
            if side is None:
                side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return side

    def _sample_ahead(self, xyz, wait_current=True):
        """The furthest-point-sampling chain of ALL levels on a side stream.  Level k's sample set depends on level k-1's
        and on nothing else (no features), and FPS is a serial chain on one workgroup per frame (4.8 ms at level 0, 5.7 ms
        for the four levels, 32 of 256 CUs): with a single batch in flight -- latency mode, the training step -- the other
        levels' samples can be drawn underneath the set-abstraction work of the levels before them.
        -> per level (new_xyz, ready-event); the tensors are held by the caller until the end of the forward pass."""
        cur = torch.cuda.current_stream(xyz.device)
        side = self._side_stream(xyz.device, cur)
        if wait_current:
            side.wait_stream(cur)
        ahead, x = [], xyz
        with torch.cuda.stream(side):
            for m in self.SA_modules:
                idx = ops.furthest_point_sample(x, m.npoint)
                x = ops.gather_rows(x, idx)
                ev = torch.cuda.Event()
                ev.record(side)
                ahead.append((x, ev, idx))
        return side, ahead

    def forward(self, pointcloud):
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        want = FPS_AHEAD == "1" or (FPS_AHEAD == "auto" and torch.is_grad_enabled())
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is pointcloud and pre[1] == self._pc_key(pointcloud):
            side, ahead = pre[2], pre[3]                 # drawn during the previous step (prefetch_samples)
            if pre[4].data_ptr() != xyz.data_ptr():      # C > 0: the side stream sliced its own xyz copy; keep it alive for this stream
                pre[4].record_stream(torch.cuda.current_stream(xyz.device))
        else:
            side, ahead = self._sample_ahead(xyz) if (want and xyz.is_cuda) else (None, None)
        if ahead is not None:
            # allocated on the side stream, consumed (and kept for backward) on this one: the allocator must not hand the blocks
            # back to the side stream while kernels of this stream still read them
            cur = torch.cuda.current_stream(xyz.device)
            for t_xyz, _, t_idx in ahead:
                t_xyz.record_stream(cur)
                t_idx.record_stream(cur)
        for i in range(len(self.SA_modules)):
            if ahead is not None:
                torch.cuda.current_stream().wait_event(ahead[i][1])
                li_xyz, li_features = self.SA_modules[i](l_xyz[i], l_features[i], ahead[i][0])
            else:
                li_xyz, li_features = self.SA_modules[i](l_xyz[i], l_features[i])
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)      # (joins the fork: required inside a graph capture)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        return l_xyz[0], l_features[0]


class RPN(nn.Module):
    """Inference subset of lib/net/rpn.py:12-82."""

    def __init__(self, use_xyz=True, cfg=RPNConfig):
        super().__init__()
        self.backbone_net = Pointnet2MSG(input_channels=int(cfg.USE_INTENSITY), use_xyz=use_xyz, cfg=cfg)
        cls_layers = []
        pre_channel = cfg.FP_MLPS[0][-1]
        for k in range(len(cfg.CLS_FC)):
            cls_layers.append(pt_utils.Conv1d(pre_channel, cfg.CLS_FC[k], bn=cfg.USE_BN))
            pre_channel = cfg.CLS_FC[k]
        cls_layers.append(pt_utils.Conv1d(pre_channel, 1, activation=None))
        if cfg.DP_RATIO >= 0:
            cls_layers.insert(1, nn.Dropout(cfg.DP_RATIO))
        self.rpn_cls_layer = nn.Sequential(*cls_layers)

        per_loc_bin_num = int(cfg.LOC_SCOPE / cfg.LOC_BIN_SIZE) * 2
        if cfg.LOC_XZ_FINE:
            reg_channel = per_loc_bin_num * 4 + cfg.NUM_HEAD_BIN * 2 + 3
        else:
            reg_channel = per_loc_bin_num * 2 + cfg.NUM_HEAD_BIN * 2 + 3
        reg_channel += 1
        self.reg_channel = reg_channel
        reg_layers = []
        pre_channel = cfg.FP_MLPS[0][-1]
        for k in range(len(cfg.REG_FC)):
            reg_layers.append(pt_utils.Conv1d(pre_channel, cfg.REG_FC[k], bn=cfg.USE_BN))
            pre_channel = cfg.REG_FC[k]
        reg_layers.append(pt_utils.Conv1d(pre_channel, reg_channel, activation=None))
        if cfg.DP_RATIO >= 0:
            reg_layers.insert(1, nn.Dropout(cfg.DP_RATIO))
        self.rpn_reg_layer = nn.Sequential(*reg_layers)

    def forward(self, input_data):
        pts_input = input_data["pts_input"]
        backbone_xyz, backbone_features = self.backbone_net(pts_input)                     # (B,N,3), (B,C,N)
        # fused_sequential == self.rpn_*_layer(x) (it falls back to exactly that), but runs conv+bn+relu -> conv as
        # one register-chain kernel at inference; the reference's own rpn.py calls the Sequential directly and gets
        # one fused kernel per layer instead
        rpn_cls = pt_utils.fused_sequential(self.rpn_cls_layer, backbone_features).transpose(1, 2).contiguous()  # (B,N,1)
        rpn_reg = pt_utils.fused_sequential(self.rpn_reg_layer, backbone_features).transpose(1, 2).contiguous()  # (B,N,reg)
        return {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg, "backbone_xyz": backbone_xyz,
                "backbone_features": backbone_features}


def randomize_bn_stats(module, seed=0):
    """Give every BatchNorm non-trivial running statistics / affine terms (seeded): a random-init network with
    identity BN would not exercise the folded-BN path."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return module


def synthetic_clouds(batch, npoints=16384, seed0=100, device="cpu"):
    """Per frame x~U(-40,40), y~U(-1,3), z~U(0,70.4) (PC_AREA_SCOPE, tools/cfgs/default.yaml:18), seed 100+frame."""
    out = torch.empty((batch, npoints, 3), dtype=torch.float32)
    lo = torch.tensor([-40.0, -1.0, 0.0])
    hi = torch.tensor([40.0, 3.0, 70.4])
    for f in range(batch):
        g = torch.Generator().manual_seed(seed0 + f)
        out[f] = torch.rand((npoints, 3), generator=g) * (hi - lo) + lo
    return out.to(device)


def lidar_like_clouds(batch, npoints=16384, seed0=100, device="cpu"):
    """A driving-scene-like distribution inside the same PC_AREA_SCOPE (SURVEY.md 8(d), config 2 note): point density falls
    off with range as a spinning LiDAR's does (range ~ sqrt-uniform, 90-degree frontal fan), 85 % of the points on a ground
    band around y = 1.6 m, the rest on 30 car-sized clusters.  Dense near the sensor, sparse far away -- the regime that
    exercises ball-query saturation, crowded grid cells and the FPS pruning differently from the uniform clouds."""
    out = torch.empty((batch, npoints, 3), dtype=torch.float32)
    for f in range(batch):
        g = torch.Generator().manual_seed(seed0 + f)
        n_obj = int(npoints * 0.15)
        n_gnd = npoints - n_obj
        rng = 2.0 + 68.0 * torch.rand(n_gnd, generator=g) ** 1.5
        ang = (torch.rand(n_gnd, generator=g) - 0.5) * 1.5708
        gnd = torch.stack([rng * torch.sin(ang), 1.6 + 0.05 * torch.randn(n_gnd, generator=g), rng * torch.cos(ang)], 1)
        ctr_r = 4.0 + 55.0 * torch.rand(30, generator=g)
        ctr_a = (torch.rand(30, generator=g) - 0.5) * 1.4
        ctr = torch.stack([ctr_r * torch.sin(ctr_a), torch.full((30,), 0.9), ctr_r * torch.cos(ctr_a)], 1)
        own = torch.randint(0, 30, (n_obj,), generator=g)
        obj = ctr[own] + torch.randn(n_obj, 3, generator=g) * torch.tensor([0.8, 0.4, 1.6])
        pts = torch.cat([gnd, obj])[torch.randperm(npoints, generator=g)]
        pts[:, 0].clamp_(-40.0, 40.0); pts[:, 1].clamp_(-1.0, 3.0); pts[:, 2].clamp_(0.0, 70.4)
        out[f] = pts
    return out.to(device)


def saturated_clouds(batch, npoints=16384, seed0=100, device="cpu"):
    """16384 points uniform in a 1.6 m cube inside PC_AREA_SCOPE (3 900 points per cubic metre): every ball of every SA level
    holds at least nsample points, so ball_query never pads -- the cloud model of BASELINE config 1 at the RPN's size, and the
    worst case for padding-free grouping (nothing to remove, the split is pure overhead)."""
    out = torch.empty((batch, npoints, 3), dtype=torch.float32)
    lo = torch.tensor([-0.8, 0.2, 20.0])
    for f in range(batch):
        g = torch.Generator().manual_seed(seed0 + f)
        out[f] = torch.rand((npoints, 3), generator=g) * 1.6 + lo
    return out.to(device)


def rpn_flops_per_frame(cfg=RPNConfig):
    """Algorithmic MLP FLOPs per frame (2*MACs) of the RPN inference graph -- SURVEY.md 8(d): 14.95 GFLOP."""
    macs = 0
    cin = int(cfg.USE_INTENSITY)
    skip = [cin]
    for k, npoint in enumerate(cfg.SA_NPOINTS):
        cout = 0
        for ns, spec in zip(cfg.SA_NSAMPLE[k], cfg.SA_MLPS[k]):
            chans = [cin + 3] + list(spec)
            macs += npoint * ns * sum(a * b for a, b in zip(chans[:-1], chans[1:]))
            cout += spec[-1]
        skip.append(cout)
        cin = cout
    n_at = [cfg.NUM_POINTS] + list(cfg.SA_NPOINTS)
    for k in range(len(cfg.FP_MLPS)):
        pre = cfg.FP_MLPS[k + 1][-1] if k + 1 < len(cfg.FP_MLPS) else cin
        chans = [pre + skip[k]] + list(cfg.FP_MLPS[k])
        macs += n_at[k] * sum(a * b for a, b in zip(chans[:-1], chans[1:]))
    c = cfg.FP_MLPS[0][-1]
    reg = int(cfg.LOC_SCOPE / cfg.LOC_BIN_SIZE) * 2 * (4 if cfg.LOC_XZ_FINE else 2) + cfg.NUM_HEAD_BIN * 2 + 3 + 1
    macs += cfg.NUM_POINTS * (c * cfg.CLS_FC[0] + cfg.CLS_FC[0] * 1 + c * cfg.REG_FC[0] + cfg.REG_FC[0] * reg)
    return 2 * macs
