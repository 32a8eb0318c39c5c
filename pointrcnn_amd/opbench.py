"""Per-operator roofline report on one MI355X:  python -m pointrcnn_amd.opbench [--quick]

For every operator on the hot path, at the shapes of tools/cfgs/default.yaml (per-GPU batch 32, 16 384 pts) and the
BASELINE config-5 dense case (65 536 pts, 512 RoIs, batch 8), prints one JSON line with the average launch duration
(HIP events on the launch stream, 20 iterations after 3 warm-ups) and
    gather-class ops (gather / group / three_interpolate / roipool3d): two byte counts --
        compulsory  = what HBM must move at least once: every input tensor ONCE + the output (a gather re-reads its source from
                      L2 / MALL, which the 256 MB Infinity Cache serves);  this is the roofline number: GB/s and its fraction of
                      the 8 TB/s HBM3E peak and of the 6.3 TB/s a streaming copy achieves on this part;
        algorithmic = SURVEY 8(d)'s count (index + one read per gathered element + write): the rate the consumer SEES.  It may
                      exceed what HBM can deliver -- then the re-reads were cache hits, and the line says so
                      (`served_from_cache`) instead of reporting a fraction above 1 (the round-2 file listed 0.92 of 8 TB/s for
                      grouping_operation: 7.4 TB/s of algorithmic bytes, 1.3 TB/s of HBM traffic);
      PMC bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction) are merged into the committed
      file by profiles/join_op_traffic.py when those passes were run;
    search ops (fps / ball_query / three_nn / nms): distance (pair) evaluations per second;
    fused MLP: algorithmic FLOP/s vs 157.3 TFLOP/s dense fp32 MFMA.
Algorithmic work per unit follows SURVEY.md 8(d).
"""
import argparse
import json

import torch

from . import ops, rpn

HBM_PEAK_GBS = 8000.0
HBM_COPY_GBS = 6300.0          # measured float4 copy ceiling (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def emit(name, shape, sec, **kw):
    d = {"op": name, "shape": shape, "avg_launch_us": round(sec * 1e6, 1)}
    if "bytes" in kw:
        comp = kw.get("compulsory", kw["bytes"])
        gbs, agbs = comp / sec / 1e9, kw["bytes"] / sec / 1e9
        d.update(bound="hbm", compulsory_MB=round(comp / 1e6, 1), achieved_GBps=round(gbs, 1),
                 frac_of_8TBps=round(gbs / HBM_PEAK_GBS, 3), frac_of_6p3TBps_copy_ceiling=round(gbs / HBM_COPY_GBS, 3),
                 algorithmic_MB=round(kw["bytes"] / 1e6, 1), algorithmic_GBps=round(agbs, 1),
                 served_from_cache=bool(agbs > HBM_COPY_GBS))
    if "kernels" in kw:            # [name substring, grid size in threads (0 = any)]: what profiles/join_op_traffic.py sums PMC bytes over
        d["kernels"] = [[k, int(g)] for k, g in kw["kernels"]]
    if "pairs" in kw:
        d.update(bound="valu", pair_evals=kw["pairs"], achieved_Gpairs_per_s=round(kw["pairs"] / sec / 1e9, 1))
    if "flops" in kw:
        tf = kw["flops"] / sec / 1e12
        d.update(bound="mfma", achieved_TFLOPs=round(tf, 2), frac_of_peak=round(tf / MFMA_F32_PEAK_TF, 3))
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = 8 if args.quick else 32
    N = 16384
    xyz = rpn.synthetic_clouds(B, N, device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)

    # ---- what plain streams reach on THIS box (1 GiB buffers): the ceilings the gather ops are read against, next to the 8 TB/s headline.
    # A pure write stream (roipool3d's 1 GB of pooled rows written once) and a read + write copy are different ceilings.
    buf_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    buf_b = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    t_fill = timeit(lambda: buf_a.fill_(1.0), 10, 2)
    t_copy = timeit(lambda: buf_b.copy_(buf_a), 10, 2)
    print(json.dumps({"op": "stream ceilings", "shape": "1 GiB fp32", "write_only_GBps": round(buf_a.numel() * 4 / t_fill / 1e9, 1),
                      "copy_read_plus_write_GBps": round(2 * buf_a.numel() * 4 / t_copy / 1e9, 1),
                      "note": "torch fill_ / copy_ on this box; an op that writes W bytes and reads R cannot finish before W / write_only (R << W) or "
                              "(R + W) / copy (R ~ W)"}), flush=True)
    del buf_a, buf_b
    torch.cuda.empty_cache()

    # ---- search ops
    emit("fps", "B%d %d->4096" % (B, N), timeit(lambda: ops.furthest_point_sample(xyz, 4096), 5, 1), pairs=B * N * 4096)
    new_xyz = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, 4096))
    emit("ball_query2", "B%d N%d M4096 r(.1,.5) ns(16,32)" % (B, N),
         timeit(lambda: ops.ball_query2(0.1, 16, 0.5, 32, xyz, new_xyz)), pairs=B * N * 4096)
    emit("three_nn", "B%d n%d m4096" % (B, N), timeit(lambda: ops.three_nn(xyz, new_xyz)), pairs=B * N * 4096)

    # ---- gather-class ops in the op-surface (B,C,N) layout
    feat = torch.randn(B, 96, 4096, device=dev)
    xyz1 = new_xyz
    nx2 = ops.gather_rows(xyz1, ops.furthest_point_sample(xyz1, 1024))
    idx = ops.ball_query(1.0, 32, xyz1, nx2)
    emit("grouping_operation", "B%d C96 N4096 M1024 ns32" % B, timeit(lambda: ops.group(feat, idx)),
         bytes=B * (1024 * 32 * 4 + 2 * 96 * 1024 * 32 * 4), compulsory=B * (1024 * 32 * 4 + 96 * 4096 * 4 + 96 * 1024 * 32 * 4),
         kernels=[("gather_pm_kernel", 0)])
    fidx = ops.furthest_point_sample(xyz1, 1024)
    emit("gather_operation", "B%d C96 N4096 M1024" % B, timeit(lambda: ops.gather(feat, fidx)),
         bytes=B * (1024 * 4 + 2 * 96 * 1024 * 4), compulsory=B * (1024 * 4 + 2 * 96 * 1024 * 4), kernels=[("gather_kernel", B * 1024)])
    d2, i3, w3 = ops.three_nn(xyz, xyz1, want_weight=True)
    kf = torch.randn(B, 256, 4096, device=dev)
    emit("three_interpolate", "B%d C256 m4096 n%d" % (B, N), timeit(lambda: ops.three_interpolate(kf, i3, w3)),
         bytes=B * (N * 24 + 3 * 256 * N * 4 + 256 * N * 4), compulsory=B * (N * 24 + 256 * 4096 * 4 + 256 * N * 4),
         kernels=[("three_interp", 0)])

    # ---- roipool3d: config 3 (M=100, C=130, S=512) and config 5 dense (65536 pts, 512 RoIs, batch 8)
    def rois_for(x, M, seed):
        gg = torch.Generator().manual_seed(seed)
        pick = torch.randint(0, x.shape[1], (x.shape[0], M), generator=gg).to(dev)
        ctr = torch.gather(x, 1, pick.unsqueeze(-1).expand(-1, -1, 3))
        sz = torch.tensor([1.6 + 2, 1.7 + 2, 4.0 + 2], device=dev).expand(x.shape[0], M, 3)
        ry = (torch.rand(x.shape[0], M, 1, generator=gg).to(dev) - 0.5) * 6.28
        return torch.cat([ctr[..., 0:1], ctr[..., 1:2] + 1.8, ctr[..., 2:3], sz, ry], 2).contiguous()

    pf = torch.randn(B, N, 130, device=dev)
    rois = rois_for(xyz, 100, 1)
    emit("roipool3d", "B%d N%d M100 C130 S512 (config 3)" % (B, N), timeit(lambda: ops.roipool3d(xyz, rois, pf, 512)),
         bytes=B * 2 * 100 * 512 * 133 * 4, compulsory=B * (100 * 512 * 133 * 4 + N * 133 * 4),
         kernels=[("roipool3d_kernel", B * 100 * 256)] + [("rp_bins_%s_kernel" % k, 0) for k in ("extent", "count", "scan", "fill")])
    if not args.quick:
        Bd, Nd, Md = 8, 65536, 512
        xd = rpn.synthetic_clouds(Bd, Nd, seed0=500, device=dev)
        pfd = torch.randn(Bd, Nd, 130, device=dev)
        rd = rois_for(xd, Md, 2)
        emit("roipool3d", "B%d N%d M%d C130 S512 (config 5 dense)" % (Bd, Nd, Md),
             timeit(lambda: ops.roipool3d(xd, rd, pfd, 512), 5, 1), bytes=Bd * 2 * Md * 512 * 133 * 4,
             compulsory=Bd * (Md * 512 * 133 * 4 + Nd * 133 * 4),
             kernels=[("roipool3d_kernel", Bd * Md * 256)] + [("rp_bins_%s_kernel" % k, 0) for k in ("extent", "count", "scan", "fill")])
        del xd, pfd, rd

    # ---- GT-augmentation scene edit (kitti_rcnn_dataset.py:484-507): 15 accepted objects per scene, ~4000 pasted points
    ab = rois_for(xyz, 15, 3)
    npts, nint = torch.randn(B, 4000, 3, device=dev), torch.rand(B, 4000, device=dev)
    inten = torch.rand(B, N, device=dev)
    emit("gt_aug_edit", "B%d N%d K15 P4000" % (B, N), timeit(lambda: ops.gt_aug_edit(xyz, inten, ab, npts, nint)),
         bytes=B * (2 * N * 16 + 2 * 4000 * 16))

    # ---- NMS (default RPN path: normal, 6300 boxes, thr 0.8) and rotated
    c = torch.rand(6300, 2, generator=g) * torch.tensor([80.0, 70.0])
    s = torch.rand(6300, 2, generator=g) * torch.tensor([0.5, 1.5]) + torch.tensor([0.8, 1.7])
    bev = torch.cat([c - s, c + s, (torch.rand(6300, 1, generator=g) - 0.5) * 6.28], 1).to(dev)
    emit("nms_normal", "N6300 thr0.8", timeit(lambda: ops.nms_sorted(bev, 0.8, rotated=False)), pairs=6300 * 6299 // 2)
    emit("nms_rotated", "N6300 thr0.8", timeit(lambda: ops.nms_sorted(bev, 0.8, rotated=True), 5, 1), pairs=6300 * 6299 // 2)

    # ---- proposal stage, whole batch per call: decode 76 regression channels -> boxes, then
    # score sort + distance split + NMS + top-k.  Scene: 40 % of the points vote for one of 24 cars (tight clusters
    # of overlapping high-score boxes), the rest is clutter -- the regime where NMS has to reject most candidates.
    reg = torch.randn(B * N, 76, device=dev)
    anchor = (1.52563191462, 1.62856739989, 3.88311640418)
    emit("decode_bbox_target", "B%d N%d C76" % (B, N),
         timeit(lambda: ops.decode_bbox_target(xyz.view(-1, 3), reg, 3.0, 0.5, 12, anchor, y_to_bottom=True)),
         bytes=B * N * (76 + 3 + 7) * 4, kernels=[("decode_kernel", 0)])
    gg = torch.Generator().manual_seed(3)
    nfg = int(N * 0.4)
    obj = torch.rand(B, 24, 7, generator=gg) * torch.tensor([70., .4, 62., .3, .3, 1., 6.28]) + torch.tensor([-35., .8, 4., 1.4, 1.5, 3.4, -3.14])
    own = torch.randint(0, 24, (B, nfg), generator=gg)
    fgb = torch.gather(obj, 1, own.unsqueeze(-1).expand(-1, -1, 7)) + torch.randn(B, nfg, 7, generator=gg) * torch.tensor([.08, .03, .12, .03, .03, .06, .03])
    bgb = torch.rand(B, N - nfg, 7, generator=gg) * torch.tensor([80., 4., 70., 1., .8, 2., 6.28]) + torch.tensor([-40., -1., .2, 1., 1.2, 3., -3.14])
    boxes3d = torch.cat([fgb, bgb], 1).to(dev).contiguous()
    scores = torch.cat([torch.randn(B, nfg, generator=gg) + 2.5, torch.randn(B, N - nfg, generator=gg) - 3.0], 1).to(dev)
    for rot in (False, True):
        emit("proposal_layer(%s nms 0.8, 6300/2700 -> 70/30)" % ("rotated" if rot else "normal"), "B%d N%d" % (B, N),
             timeit(lambda: ops.proposal_layer(scores, boxes3d, (6300, 2700), (70, 30), 0.8, rotated=rot)), pairs=B * N)
    rb = boxes3d[:, :100].contiguous()
    rs = scores[:, :100].contiguous()
    emit("nms_batched(rotated 0.1)", "B%d M100" % B, timeit(lambda: ops.nms_batched(rb, rs, None, 0.1, True)), pairs=B * 100)

    # ---- fused MLP: the whole RPN graph's MLP FLOPs are reported by bench.py; here two representative stacks
    model = rpn.randomize_bn_stats(rpn.RPN()).to(dev).eval()
    sa2 = model.backbone_net.SA_modules[1]
    f1 = torch.randn(B, 96, 4096, device=dev)
    with torch.no_grad():
        sec = timeit(lambda: sa2(xyz1, f1))
    macs = 1024 * (16 * (99 * 64 + 64 * 64 + 64 * 128) + 32 * (99 * 64 + 64 * 96 + 96 * 128))
    emit("SA2 module (fps+ball_query+fused MLP chains)", "B%d 4096->1024" % B, sec, flops=2.0 * B * macs)


if __name__ == "__main__":
    main()
