"""GPU parity at module level: the drop-in PointNet++ modules' fused inference path vs (a) the oracle
pipeline op by op and (b) the composed torch path of the same module, and the full RPN graph."""
import numpy as np
import pytest
import torch

from util import assert_same_result, kitti_cloud, mlp_tol, unit_cloud

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def oracle_sa(cpu, xyz, feat, npoint, radius, ns, ws):
    """SA module restated with oracle ops: fps -> gather -> ball_query -> group -> MLP(relu) -> max"""
    fidx = cpu.fps(xyz, npoint)
    new_xyz = cpu.gather(xyz.transpose(0, 2, 1), fidx).transpose(0, 2, 1)
    idx = cpu.ball_query(radius, ns, xyz, new_xyz)
    gx = cpu.group(xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    g = gx if feat is None else np.concatenate([gx, cpu.group(feat, idx)], 1)
    B, K, M, _ = g.shape
    rows = g.transpose(0, 2, 3, 1).reshape(-1, K)
    for w in ws:
        rows = cpu.linear_rows(rows, w, None, True)
    return fidx, new_xyz, idx, rows.reshape(B, M, ns, -1).max(2).transpose(0, 2, 1)


def test_config1_single_sa_layer(dev, cpu):
    """BASELINE.json configs[0]: single SA layer (npoint=512, r=0.2, nsample=32, C=3) on 1x4096 random points;
    FPS / ball-query indices bit-exact, output within 1e-5"""
    from pointnet2_lib.pointnet2.pointnet2_modules import PointnetSAModule
    from pointrcnn_amd import ops
    xyz = torch.rand(1, 4096, 3, generator=torch.Generator().manual_seed(0)).numpy()
    feat = torch.randn(1, 3, 4096, generator=torch.Generator().manual_seed(1)).numpy()
    mod = PointnetSAModule(npoint=512, radius=0.2, nsample=32, mlp=[3, 32, 32, 64], use_xyz=True, bn=False).to(dev).eval()
    g = torch.Generator().manual_seed(2)
    ws = []
    for layer in mod.mlps[0].layers():
        w = torch.randn(layer.conv.weight.shape, generator=g) * 0.1
        with torch.no_grad():
            layer.conv.weight.copy_(w)
            layer.conv.bias.zero_()
        ws.append(w.reshape(w.shape[0], -1).numpy())
    fidx, new_xyz, idx, want = oracle_sa(cpu, xyz, feat, 512, 0.2, 32, ws)
    txyz = T(xyz, dev)
    assert np.array_equal(ops.furthest_point_sample(txyz, 512).cpu().numpy(), fidx)
    assert np.array_equal(ops.ball_query(0.2, 32, txyz, T(new_xyz, dev)).cpu().numpy(), idx)
    with torch.no_grad():
        nx, nf = mod(txyz, T(feat, dev))
    assert np.array_equal(nx.cpu().numpy(), new_xyz)
    assert nf.shape == (1, 64, 512)
    np.testing.assert_allclose(nf.cpu().numpy(), want, atol=mlp_tol(want), rtol=0)


def test_sa_msg_fused_equals_composed(dev):
    """same module, same weights: fused inference path (no_grad) == composed autograd path (MIOpen conv/BN)"""
    from pointnet2_lib.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    from pointrcnn_amd.rpn import randomize_bn_stats
    torch.manual_seed(0)
    mod = PointnetSAModuleMSG(npoint=256, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[24, 32, 48], [24, 40, 64]],
                              use_xyz=True, bn=True)
    randomize_bn_stats(mod).to(dev).eval()
    xyz = T(unit_cloud(2, 2000, seed=3), dev)
    feat = torch.randn(2, 24, 2000, device=dev)
    with torch.no_grad():
        nx_f, nf_f = mod(xyz, feat)
    with torch.enable_grad():
        nx_c, nf_c = mod(xyz, feat.clone().requires_grad_(True))
    assert torch.equal(nx_f, nx_c)
    assert nf_f.shape == nf_c.shape == (2, 112, 256)
    np.testing.assert_allclose(nf_f.cpu().numpy(), nf_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)


def test_sa_group_all_and_odd_nsample(dev):
    """npoint=None (GroupAll, rcnn_net.py:31) and an nsample outside {16,32,64} (generic pooling kernel)"""
    from pointnet2_lib.pointnet2.pointnet2_modules import PointnetSAModule
    torch.manual_seed(1)
    xyz = T(unit_cloud(3, 32, seed=5), dev)
    feat = torch.randn(3, 20, 32, device=dev)
    ga = PointnetSAModule(npoint=None, radius=100, nsample=64, mlp=[20, 32, 48], use_xyz=True, bn=False).to(dev).eval()
    with torch.no_grad():
        nx, nf = ga(xyz, feat)
    with torch.enable_grad():
        nx_c, nf_c = ga(xyz, feat.clone().requires_grad_(True))
    assert nx is None and nx_c is None and nf.shape == (3, 48, 1)
    np.testing.assert_allclose(nf.cpu().numpy(), nf_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)
    odd = PointnetSAModule(npoint=10, radius=0.5, nsample=12, mlp=[20, 16], use_xyz=True, bn=False).to(dev).eval()
    with torch.no_grad():
        _, nf = odd(xyz, feat)
    with torch.enable_grad():
        _, nf_c = odd(xyz, feat.clone().requires_grad_(True))
    np.testing.assert_allclose(nf.cpu().numpy(), nf_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)


def test_fp_module_fused_equals_composed_and_oracle(dev, cpu):
    from pointnet2_lib.pointnet2.pointnet2_modules import PointnetFPModule
    from pointrcnn_amd.rpn import randomize_bn_stats
    torch.manual_seed(2)
    mod = randomize_bn_stats(PointnetFPModule(mlp=[96 + 40, 64, 32])).to(dev).eval()
    unk, kn = unit_cloud(2, 700, seed=1), unit_cloud(2, 150, seed=2)
    uf = torch.randn(2, 40, 700, device=dev)
    kf = torch.randn(2, 96, 150, device=dev)
    with torch.no_grad():
        out_f = mod(T(unk, dev), T(kn, dev), uf, kf)
    with torch.enable_grad():
        out_c = mod(T(unk, dev), T(kn, dev), uf.clone().requires_grad_(True), kf)
    assert out_f.shape == (2, 32, 700)
    np.testing.assert_allclose(out_f.cpu().numpy(), out_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)
    # no skip features (level-0 of the RPN: pointnet2_msg.py:18,42-45)
    mod0 = PointnetFPModule(mlp=[96, 32], bn=False).to(dev).eval()
    with torch.no_grad():
        o_f = mod0(T(unk, dev), T(kn, dev), None, kf)
    with torch.enable_grad():
        o_c = mod0(T(unk, dev), T(kn, dev), None, kf.clone().requires_grad_(True))
    np.testing.assert_allclose(o_f.cpu().numpy(), o_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)


def test_conv1d_heads_fused_equals_torch(dev):
    """pt_utils.Conv1d stacks as the RPN heads build them (rpn.py:20-46): fused path == torch path"""
    import pointnet2_lib.pointnet2.pytorch_utils as pt_utils
    from pointrcnn_amd.rpn import randomize_bn_stats
    torch.manual_seed(3)
    head = torch.nn.Sequential(pt_utils.Conv1d(128, 128, bn=True), torch.nn.Dropout(0.5),
                               pt_utils.Conv1d(128, 76, activation=None))
    randomize_bn_stats(head).to(dev).eval()
    x = torch.randn(2, 128, 1000, device=dev)
    with torch.no_grad():
        y_f = head(x)
    with torch.enable_grad():
        y_c = head(x.clone().requires_grad_(True))
    assert y_f.shape == (2, 76, 1000)
    np.testing.assert_allclose(y_f.cpu().numpy(), y_c.detach().cpu().numpy(), atol=2e-4, rtol=1e-4)


def test_full_rpn_fused_equals_composed(dev):
    """the benchmark graph (default.yaml RPN, 16384 points): fused inference path vs the composed op-by-op path
    of the very same modules (HIP index ops + MIOpen conv/BN); index ops are shared so only fp32 summation
    order differs"""
    from pointrcnn_amd import rpn
    torch.manual_seed(4)
    model = rpn.randomize_bn_stats(rpn.RPN()).to(dev).eval()
    pts = rpn.synthetic_clouds(1, 16384, device=dev)
    with torch.no_grad():
        out_f = model({"pts_input": pts})
    with torch.enable_grad():
        out_c = model({"pts_input": pts.clone().requires_grad_(True)})
    assert out_f["rpn_cls"].shape == (1, 16384, 1) and out_f["rpn_reg"].shape == (1, 16384, 76)
    assert out_f["backbone_features"].shape == (1, 128, 16384)
    for k in ("backbone_features", "rpn_cls", "rpn_reg"):
        a, b = out_f[k].cpu().numpy(), out_c[k].detach().cpu().numpy()
        np.testing.assert_allclose(a, b, atol=5e-4 * max(1.0, float(np.abs(b).max())), rtol=0)


def _synthetic_rois(xyz, M, seed):
    """(B,M,7) car-sized boxes centred on random points of each frame (stand-in for the proposal layer)"""
    g = torch.Generator().manual_seed(seed)
    B, N, _ = xyz.shape
    pick = torch.randint(0, N, (B, M), generator=g)
    ctr = torch.gather(xyz.cpu(), 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    h = torch.rand(B, M, 1, generator=g) * 0.4 + 1.4
    w = torch.rand(B, M, 1, generator=g) * 0.3 + 1.5
    l = torch.rand(B, M, 1, generator=g) * 1.0 + 3.5
    ry = (torch.rand(B, M, 1, generator=g) - 0.5) * 6.28
    return torch.cat([ctr[..., 0:1], ctr[..., 1:2] + h / 2, ctr[..., 2:3], h, w, l, ry], 2)


def test_rcnn_stage_fused_equals_composed_and_roipool_oracle(dev, cpu):
    """BASELINE config 3 shape of the second stage (rcnn_net.py:115-190): roipool3d (M=20 RoIs x 512 pts x 133 ch)
    -> canonical transform -> SharedMLPs -> SA(npoint 128, ns 64) -> SA(32, 64) -> GroupAll -> heads, then the final
    rotated NMS (eval_rcnn.py:618-619, thr 0.1).  Pooled points vs the oracle exactly; fused vs composed path."""
    from pointrcnn_amd import rcnn
    torch.manual_seed(5)
    B, N, M = 2, 16384, 20
    xyz = T(kitti_cloud(B, N, seed=300), dev)
    rois = _synthetic_rois(xyz, M, seed=6).to(dev)
    data = {"rpn_xyz": xyz, "rpn_features": torch.randn(B, N, 128, device=dev), "seg_mask": (torch.rand(B, N, device=dev) > 0.5).float(),
            "pts_depth": torch.norm(xyz, p=2, dim=2), "roi_boxes3d": rois}
    net = rcnn.RCNNNet().to(dev).eval()
    with torch.no_grad():
        out_f = net(data)
    # roipool3d inside the stage == oracle on the same enlarged boxes and features
    feat = torch.cat([data["seg_mask"].unsqueeze(2), (data["pts_depth"] / 70.0 - 0.5).unsqueeze(2), data["rpn_features"]], 2)
    with torch.no_grad():
        pooled, empty = rcnn.roipool3d_gpu(xyz, feat, rois, 1.0, 512)
    want, wempty = cpu.roipool3d(xyz.cpu().numpy(), rcnn.enlarge_box3d(rois.view(-1, 7), 1.0).view(B, M, 7).cpu().numpy(),
                                 feat.cpu().numpy(), 512)
    assert np.array_equal(pooled.cpu().numpy(), want) and np.array_equal(empty.cpu().numpy(), wempty)
    assert out_f["rcnn_cls"].shape == (B * M, 1) and out_f["rcnn_reg"].shape == (B * M, 46)   # rcnn_net.py:184-185
    with torch.enable_grad():
        data_c = dict(data, rpn_features=data["rpn_features"].clone().requires_grad_(True))
        out_c = net(data_c)
    for k in ("rcnn_cls", "rcnn_reg"):
        a, b = out_f[k].cpu().numpy(), out_c[k].detach().cpu().numpy()
        np.testing.assert_allclose(a, b, atol=5e-4 * max(1.0, float(np.abs(b).max())), rtol=0)
    # final rotated NMS per frame on the RoIs with the stage's scores
    for b in range(B):
        bev = rcnn.boxes3d_to_bev_torch(rois[b])
        scores = out_f["rcnn_cls"].view(B, M)[b]
        keep = rcnn.nms_gpu(bev, scores, 0.1).cpu().numpy()
        order = torch.sort(scores, descending=True)[1].cpu().numpy()
        want_keep = order[cpu.nms(bev[torch.from_numpy(order).to(dev)].cpu().numpy(), 0.1, "rotated")]
        assert np.array_equal(keep, want_keep)


def test_rpn_backward_composed_path(dev):
    """config 4 building block: one training-style step of an SA+FP pair through the HIP backward kernels
    (group / gather / three_interpolate grads) -- gradients reach every parameter and are finite"""
    from pointnet2_lib.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG
    torch.manual_seed(6)
    sa = PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[8, 16, 32], [8, 16, 32]],
                             use_xyz=True, bn=True).to(dev).train()
    fp = PointnetFPModule(mlp=[64 + 8, 32, 16]).to(dev).train()
    xyz = T(unit_cloud(4, 1024, seed=9), dev)
    feat = torch.randn(4, 8, 1024, device=dev, requires_grad=True)
    new_xyz, f1 = sa(xyz, feat)
    out = fp(xyz, new_xyz, feat, f1)
    assert out.shape == (4, 16, 1024)
    out.square().mean().backward()
    params = list(sa.parameters()) + list(fp.parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    assert feat.grad is not None and torch.isfinite(feat.grad).all() and float(feat.grad.abs().sum()) > 0


@pytest.mark.parametrize("div", [1, 4, 32])
@pytest.mark.parametrize("cloud", ["sparse", "dense", "mixed"])
def test_sa_padding_free_grouping_is_bit_identical(dev, cloud, div, mlp_mode):
    """PRCNN_GROUP_DEDUP (csrc/dedup.hip): sparse groups (<= nsample/div hits) run as their real rows only, dense
    groups with all their rows -- the level's output must not change by one bit.  sparse cloud: nearly every group has
    one hit; dense: all full; mixed: both lists populated (and the MSG level's two radii split differently).
    div 1 = every group through the flat list, 32 = only single-hit groups."""
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm
    from pointrcnn_amd.rpn import randomize_bn_stats
    torch.manual_seed(7)
    B, N = 3, 3000
    base = unit_cloud(B, N, seed=11)
    if cloud == "sparse":
        base = base * 40.0
    elif cloud == "mixed":
        base[:, : N // 2] *= 0.15                     # half the points in a dense clump, the rest spread out
        base[:, N // 2:] *= 12.0
    xyz = T(base, dev)
    for mlps, cin in (([[16, 16, 32], [32, 32, 64]], 0), ([[24, 64, 128], [24, 96, 128]], 24), ([[24, 196, 256], [24, 160, 200]], 24)):
        mod = pm.PointnetSAModuleMSG(npoint=512, radii=[0.25, 0.6], nsamples=[16, 32], mlps=[list(m) for m in mlps], use_xyz=True, bn=True)
        randomize_bn_stats(mod).to(dev).eval()
        feat = torch.randn(B, cin, N, device=dev) if cin else None
        outs = {}
        for flag in (False, True):
            pm.GROUP_DEDUP, keep = flag, pm.DEDUP_SPARSE_DIV
            pm.DEDUP_SPARSE_DIV = div
            try:
                with torch.no_grad():
                    outs[flag] = mod(xyz, feat)[1].clone()
            finally:
                pm.GROUP_DEDUP, pm.DEDUP_SPARSE_DIV = True, keep
        assert_same_result(outs[False], outs[True], mlp_mode, (cloud, mlps))


@pytest.mark.parametrize("ns", [16, 12, 32])
@pytest.mark.parametrize("smax", [1, 4, 12])
def test_group_compact_lists(dev, smax, ns):
    """prcnn_group_compact: every group lands in exactly one list; sparse groups own cnt consecutive flat rows holding
    their real hits (global point indices) and their centroid; dense groups keep all nsample rows"""
    from pointrcnn_amd import ops
    B, N, M = 2, 4000, 700
    pts = kitti_cloud(B, N, seed=3)
    pts[:, : N // 2] *= 0.25                              # half the points 16x denser: full groups next to single-hit ones
    xyz = T(pts, dev)
    new_xyz = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, M))
    idx = ops.ball_query(1.2, ns, xyz, new_xyz)
    sp = ops.GroupSplit(idx, new_xyz, N, smax)
    rows, cd, cs = (int(v) for v in sp.counts.cpu())
    idx_c = idx.cpu().numpy().reshape(B * M, ns)
    cnt = np.array([ns if not (r[1:] == r[0]).any() else 1 + int(np.argmax(r[1:] == r[0])) for r in idx_c])
    sparse = cnt <= smax
    assert cs == sparse.sum() and cd == (~sparse).sum() and rows == cnt[sparse].sum() and cs > 0 and (cd > 0 or smax == ns)
    sl, so, sc = (t[:cs].cpu().numpy() for t in (sp.slist, sp.soff, sp.scnt))
    ln = sp.listn[:cd].cpu().numpy()
    assert np.array_equal(np.sort(sl), np.nonzero(sparse)[0]) and np.array_equal(np.sort(ln), np.nonzero(~sparse)[0])
    assert np.array_equal(sc, cnt[sl])
    order = np.argsort(so)
    assert so[order][0] == 0 and np.array_equal(so[order][1:], np.cumsum(sc[order])[:-1])       # a partition of the flat rows
    ridx, rnx, nx = sp.ridx.view(-1).cpu().numpy(), sp.rnx.view(-1, 3).cpu().numpy(), new_xyz.view(-1, 3).cpu().numpy()
    for j in range(0, cs, 7):
        g = sl[j]
        assert np.array_equal(ridx[so[j]:so[j] + sc[j]], (g // M) * N + idx_c[g, :sc[j]])
        assert np.array_equal(rnx[so[j]:so[j] + sc[j]], np.repeat(nx[g][None], sc[j], 0))
    assert np.array_equal(sp.idxn.view(-1, ns)[:cd].cpu().numpy(), (ln // M)[:, None] * N + idx_c[ln])
    assert torch.equal(sp.nxn.view(-1, 3)[:cd], new_xyz.view(-1, 3)[sp.listn[:cd].long()])


def test_rcnn_roi_duplicate_elimination_is_bit_identical(dev, cpu, mlp_mode):
    """PRCNN_ROI_DEDUP: an RoI with fewer than 512 points is padded with copies of its first rows; the fused stage skips the
    copies in the per-point layers and never gathers them in the first SA level.  RoIs from empty to > 512 points: the
    stage's outputs must not change by one bit, and `distinct` must be the oracle's point count."""
    from pointrcnn_amd import ops, rcnn
    torch.manual_seed(8)
    B, N, M = 2, 16384, 24
    pts = kitti_cloud(B, N, seed=310)
    pts[:, :3000] = pts[:, :3000] * np.float32(0.08) + np.array([5.0, 1.0, 20.0], np.float32)      # a dense clump: RoIs with > 512 points
    xyz = T(pts, dev)
    rois = _synthetic_rois(xyz, M, seed=9)
    rois[:, 0, :3] = torch.tensor([0.0, -30.0, 5.0])           # an RoI far above the scene: empty
    rois[:, 1, :3] = torch.tensor([5.3, 1.8, 21.5])            # inside the clump
    rois[:, 1, 3:6] = torch.tensor([2.5, 3.0, 6.0])
    rois[:, 2, :] = torch.tensor([5.0, 1.3, 22.8, 0.5, 0.2, 0.2, 0.3])        # a sliver of the clump: a few hundred points
    rois = rois.to(dev)
    data = {"rpn_xyz": xyz, "rpn_features": torch.randn(B, N, 128, device=dev), "seg_mask": (torch.rand(B, N, device=dev) > 0.5).float(),
            "pts_depth": torch.norm(xyz, p=2, dim=2), "roi_boxes3d": rois}
    net = rcnn.RCNNNet().to(dev).eval()
    outs = {}
    for flag in (False, True):
        rcnn.ROI_DEDUP = flag
        try:
            with torch.no_grad():
                outs[flag] = {k: v.clone() for k, v in net(data).items()}
        finally:
            rcnn.ROI_DEDUP = True
    for k in ("rcnn_cls", "rcnn_reg", "pooled_empty_flag"):
        assert_same_result(outs[False][k], outs[True][k], mlp_mode, k)
    # distinct == min(points inside the enlarged box, 512), 1 for the empty RoI
    pool_boxes = rcnn.enlarge_box3d(rois.view(-1, 7), 1.0).view(B, M, 7)
    feat_cl = data["rpn_features"]
    _, _, empty, distinct = ops.roipool3d_canonical(xyz, pool_boxes, rois, [data["seg_mask"]], feat_cl, 512, want_distinct=True)
    flags = np.stack([cpu.pts_in_boxes3d(pts[b], pool_boxes[b].cpu().numpy()) for b in range(B)])        # (B, M, N)
    want = np.minimum(flags.sum(2), 512)
    assert np.array_equal(empty.cpu().numpy(), (want == 0).astype(np.int32))
    assert np.array_equal(distinct.cpu().numpy(), np.maximum(want, 1))
    assert want.max() == 512 and want.min() == 0 and ((want > 0) & (want < 128)).any() and ((want > 128) & (want < 512)).any()


def test_padding_free_lists_walked_in_slices_are_bit_identical(dev, monkeypatch):
    """round 6 (two-stage detector memory): the dense and the flat list of a padding-free scale are walked in slices with a device-side
    count per slice (pointnet2_modules._run_scale) so that their worst-case-sized intermediates never exist at once.  Slices of a few
    hundred kilobytes (dozens of them, most past the lists' ends: zero-row launches) == one slice, bit for bit, on a cloud whose groups
    are part dense, part sparse -- the RCNN stage's first level in miniature (nsample 64, three 128-wide layers, no BatchNorm)."""
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm
    r = np.random.default_rng(4)
    B, N = 6, 512
    # half of every frame in a tight cluster (dense balls), half scattered (sparse balls)
    xyz = np.concatenate([r.normal(0, 0.08, (B, N // 2, 3)), r.uniform(-3, 3, (B, N // 2, 3))], 1).astype(np.float32)
    feat = r.normal(size=(B, 128, N)).astype(np.float32)
    sa = pm.PointnetSAModule(npoint=128, radius=0.2, nsample=64, mlp=[128, 128, 128, 128], use_xyz=True, bn=False).to(dev).eval()
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(feat).to(dev)
    outs = []
    for chunk in (1 << 62, 300 * 1024, 64 * 1024):
        monkeypatch.setattr(pm, "DENSE_CHUNK_BYTES", chunk)
        with torch.no_grad():
            outs.append(sa(x, f)[1].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert float(outs[0].abs().max()) > 0
