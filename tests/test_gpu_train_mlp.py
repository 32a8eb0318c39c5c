"""Training-mode SharedMLP kernels (csrc/mlp_train.h, pointrcnn_amd/train_mlp.py) against plain PyTorch fp32: the reference
runs these layers as nn.Conv2d(1x1) -> nn.BatchNorm2d (batch statistics) -> ReLU -> max over nsample through torch
(upstream pytorch_utils.SharedMLP; lib/net/pointnet2_msg.py:20-45), so torch's own modules + autograd ARE the reference here.

Tolerances (written where used): forward 1e-5 * scale (fp32 MFMA sums in a different k order than torch's GEMM);
parameter / input gradients 1e-4 relative to the tensor's largest entry where no arg-max is involved; with max-pooling a
near-tie between two rows can route one gradient entry differently in the two implementations (both are valid sub-gradients),
so pooled cases are held to 1e-5 in the median and 5e-3 in norm."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _close(a, b, rel, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= rel * scale, "%s: max abs err %.3e > %.1e * %.3e" % (what, err, rel, scale)


def _close_pooled(a, b, what):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    scale = max(1e-12, float(b.abs().max()))
    assert float((a - b).abs().median()) <= 1e-5 * scale, what
    assert float((a - b).norm()) <= 5e-3 * max(1e-12, float(b.norm())), what


class _TorchStack(nn.Module):
    """Conv(1x1, no bias) -> BN -> ReLU layers on (R, K) rows, the torch reference"""

    def __init__(self, chans, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.lin = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(chans[:-1], chans[1:])])
        self.bn = nn.ModuleList([nn.BatchNorm1d(b) for b in chans[1:]])
        with torch.no_grad():
            for l, b in zip(self.lin, self.bn):
                l.weight.copy_(torch.randn(l.weight.shape, generator=g) * (1.0 / np.sqrt(l.weight.shape[1])))
                b.weight.copy_(torch.rand(b.weight.shape, generator=g) + 0.5)
                b.bias.copy_(torch.randn(b.bias.shape, generator=g) * 0.2)
                b.running_mean.copy_(torch.randn(b.bias.shape, generator=g) * 0.1)

    def forward(self, x):
        for l, b in zip(self.lin, self.bn):
            x = torch.relu(b(l(x)))
        return x


def _run_ours(ref, src, x0, x1, pool_ns):
    from pointrcnn_amd import train_mlp
    bns = [copy.deepcopy(b) for b in ref.bn]
    params = []
    for l, b in zip(ref.lin, bns):
        params += [l.weight.detach().clone().requires_grad_(True), b.weight.detach().clone().requires_grad_(True),
                   b.bias.detach().clone().requires_grad_(True)]
    out = train_mlp.SharedMLPTrain.apply(src, bns, pool_ns, x0, x1, *params)
    return out, params, bns


@pytest.mark.parametrize("rows,chans", [(1000, [128, 128]), (4099, [99, 64, 96, 128]), (300, [3, 16, 16, 32]), (2048, [259, 128, 196, 256]),
                                        (777, [515, 256, 384, 512])])
def test_plain_stack_forward_backward_equal_torch(dev, rows, chans):
    from pointrcnn_amd import train_mlp
    g = torch.Generator().manual_seed(rows)
    ref = _TorchStack(chans, seed=rows).to(dev).train()
    x = (torch.randn(rows, chans[0], generator=g) * 1.5 + 0.3).to(dev)
    xo = x.clone().requires_grad_(True)
    got, params, bns = _run_ours(ref, train_mlp.Source("plain"), xo, None, 0)      # (copies the BatchNorm buffers BEFORE the reference runs)
    xr = x.clone().requires_grad_(True)
    want = ref(xr)
    gout = torch.randn(want.shape, generator=g).to(dev)
    want.backward(gout)
    got.backward(gout)
    _close(got, want, 1e-5, "forward")
    _close(xo.grad, xr.grad, 1e-4, "input gradient")
    for l, (lin, bn) in enumerate(zip(ref.lin, ref.bn)):
        _close(params[3 * l].grad, lin.weight.grad, 1e-4, "dW layer %d" % l)
        _close(params[3 * l + 1].grad, bn.weight.grad, 1e-4, "dgamma layer %d" % l)
        _close(params[3 * l + 2].grad, bn.bias.grad, 1e-4, "dbeta layer %d" % l)
        _close(bns[l].running_mean, bn.running_mean, 1e-5, "running_mean layer %d" % l)
        _close(bns[l].running_var, bn.running_var, 1e-5, "running_var layer %d" % l)
        assert int(bns[l].num_batches_tracked) == int(bn.num_batches_tracked) == 1


def test_statistics_survive_a_large_mean(dev):
    """batch variance from (mean, M2) slab partials: no E[y^2] - mean^2 cancellation when |mean| >> std"""
    from pointrcnn_amd import train_mlp
    ref = _TorchStack([64, 64], seed=3).to(dev).train()
    with torch.no_grad():
        ref.lin[0].weight.copy_(torch.eye(64, device=dev))
    x = (torch.randn(70000, 64, generator=torch.Generator().manual_seed(1)) * 0.05 + 30.0).to(dev)
    got, _, bns = _run_ours(ref, train_mlp.Source("plain"), x.clone().requires_grad_(True), None, 0)
    want = ref(x)
    _close(bns[0].running_var, ref.bn[0].running_var, 1e-4, "running_var")
    _close(got, want, 2e-4, "normalised output")        # xhat = (y - 30) / 0.05: input rounding is amplified 600 x


def _modules(kind, dev, seed, **kw):
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm
    torch.manual_seed(seed)
    m = (pm.PointnetSAModuleMSG(**kw) if kind == "sa" else pm.PointnetFPModule(**kw)).to(dev).train()
    with torch.no_grad():
        for p in m.modules():
            if isinstance(p, (nn.BatchNorm1d, nn.BatchNorm2d)):
                p.weight.uniform_(0.5, 1.5)
                p.bias.normal_(0, 0.2)
    return pm, m, copy.deepcopy(m)


def _compare_modules(ma, mb, pooled):
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    assert set(pa) == set(pb)
    for n in pa:
        assert pa[n].grad is not None and pb[n].grad is not None, n
        (_close_pooled if pooled else (lambda a, b, w: _close(a, b, 1e-4, w)))(pa[n].grad, pb[n].grad, n)
    for (na, ba), (nb, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        if ba.dtype.is_floating_point:
            _close(ba, bb, 1e-5, na)
        else:
            assert int(ba) == int(bb)


@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("C", [0, 8, 96])
def test_sa_module_fused_training_equals_composed_torch_path(dev, C, dedup, monkeypatch):
    """PointnetSAModuleMSG in training mode: hand-written path == the composed path (HIP grouping ops + torch Conv2d /
    BatchNorm2d / ReLU / max, the reference's own structure): outputs, feature gradient, every parameter gradient and
    BatchNorm buffer.  Clouds dense enough that groups are partly padded, partly full.  dedup: on the padding-free rows
    (distinct rows + multiplicities) or on the nsample-padded rows."""
    B, N = 3, 1500
    pm, fused, comp = _modules("sa", dev, 5 + C, npoint=200, radii=[0.15, 0.3], nsamples=[16, 32],
                               mlps=[[C, 16, 16, 32], [C, 32, 48, 64]], use_xyz=True, bn=True)
    g = torch.Generator().manual_seed(C)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = None if C == 0 else torch.randn(B, C, N, generator=g).to(dev)
    fa = None if feat is None else feat.clone().requires_grad_(True)
    fb = None if feat is None else feat.clone().requires_grad_(True)
    assert fused._train_ok(xyz, fa)
    monkeypatch.setattr(pm, "TRAIN_DEDUP", dedup)
    nx_a, out_a = fused(xyz, fa)
    pm.TRAIN_FUSED = False
    # the composed side through ATen's own convolution / batch-norm kernels: MIOpen picks its convolution algorithm from run-time
    # measurements, and one of its choices for the 3-channel first layer moved this comparison by 1e-4 once in round 4
    # (tools/train_flake_probe.py measures both sides against float64; the fused side is additionally held to a float64 reference
    # 50 times over in test_sa_module_fused_training_equals_float64_reference_50_times)
    with torch.backends.cudnn.flags(enabled=False):
        try:
            nx_b, out_b = comp(xyz, fb)
        finally:
            pm.TRAIN_FUSED = True
        assert torch.equal(nx_a, nx_b)
        _close(out_a, out_b, 1e-5, "pooled features")
        gout = torch.randn(out_b.shape, generator=g).to(dev)
        out_a.backward(gout)
        out_b.backward(gout)
    if feat is not None:
        _close_pooled(fa.grad, fb.grad, "feature gradient")
    _compare_modules(fused, comp, pooled=True)


def sa_reference_f64(module, xyz, feat, gout):
    """The reference's SA-module training step restated in float64 on the CPU with torch autograd -- no MIOpen, no rocBLAS, no HIP
    arithmetic in the loop: grouped rows [xyz[idx] - new_xyz | feat[idx]] (upstream QueryAndGroup order; the fp32 difference is the
    kernels' own input) -> per layer y = rows W^T, BatchNorm with batch statistics (biased variance, running stats with momentum and
    the unbiased variance), ReLU -> max over nsample (lib/net/pointnet2_msg.py:20-34).  Sampling and ball-query indices come from
    the device operators (index ops, bit-exact against the oracle elsewhere).
    -> out (B,C,M) f64, feature gradient or None, {parameter name: gradient}, {buffer name: value after the step}"""
    from pointrcnn_amd import ops
    with torch.no_grad():
        new_xyz = ops.gather_rows(xyz.contiguous(), ops.furthest_point_sample(xyz.contiguous(), module.npoint))
        idxs = [ops.ball_query(g.radius, g.nsample, xyz.contiguous(), new_xyz.contiguous()) for g in module.groupers]
    names = {id(p): n for n, p in module.named_parameters()}
    bnames = {id(b): n for n, b in module.named_buffers()}
    x32, c32 = xyz.cpu(), new_xyz.cpu()
    f64 = None if feat is None else feat.detach().double().cpu().requires_grad_(True)
    B = xyz.shape[0]
    bi = torch.arange(B)[:, None, None]
    leaves, buffers, outs = {}, {}, []
    for g, mlp, idx in zip(module.groupers, module.mlps, idxs):
        idx = idx.long().cpu()
        M, ns = idx.shape[1:]
        rows = (x32[bi, idx] - c32[:, :, None, :]).double()
        if f64 is not None:
            rows = torch.cat([rows, f64.transpose(1, 2)[bi, idx]], dim=3)
        x = rows.reshape(B * M * ns, -1)
        for layer in mlp.layers():
            bn = layer.bn.bn
            W = layer.conv.weight.detach().double().cpu().flatten(1).requires_grad_(True)
            gamma, beta = bn.weight.detach().double().cpu().requires_grad_(True), bn.bias.detach().double().cpu().requires_grad_(True)
            leaves[names[id(layer.conv.weight)]], leaves[names[id(bn.weight)]], leaves[names[id(bn.bias)]] = W, gamma, beta
            y = x @ W.t()
            mean, var = y.mean(0), y.var(0, unbiased=False)
            n = y.shape[0]
            buffers[bnames[id(bn.running_mean)]] = ((1 - bn.momentum) * bn.running_mean.double().cpu() + bn.momentum * mean).detach()
            buffers[bnames[id(bn.running_var)]] = ((1 - bn.momentum) * bn.running_var.double().cpu() + bn.momentum * var * n / (n - 1)).detach()
            x = torch.relu((y - mean) / torch.sqrt(var + bn.eps) * gamma + beta)
        outs.append(x.view(B, M, ns, -1).max(dim=2)[0])
    out = torch.cat(outs, dim=2).transpose(1, 2)
    out.backward(gout.double().cpu())
    return out.detach(), (None if f64 is None else f64.grad), {n: t.grad for n, t in leaves.items()}, buffers


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("C", [0, 8, 96])
def test_sa_module_fused_training_equals_float64_reference_50_times(dev, C, dedup, monkeypatch):
    """the fused training path against a float64 CPU autograd restatement (no MIOpen on the other side), the
    same step 50 times on fresh copies of the module -- every repetition inside the written bars (forward 1e-5 * scale; gradients
    and the feature gradient 1e-5 in the median and 5e-3 in norm, the pooled bar: near-ties of the max may route single entries
    differently; running statistics 1e-5), and every repetition's output and parameter gradients BIT-IDENTICAL to the first one's
    (fixed summation orders); the feature gradient too on the padding-free rows (gather form), within 1e-5 on the padded rows (their
    scatter accumulates with float atomics)."""
    B, N = 3, 1500
    pm, m0, _ = _modules("sa", dev, 5 + C, npoint=200, radii=[0.15, 0.3], nsamples=[16, 32],
                         mlps=[[C, 16, 16, 32], [C, 32, 48, 64]], use_xyz=True, bn=True)
    g = torch.Generator().manual_seed(C)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = None if C == 0 else torch.randn(B, C, N, generator=g).to(dev)
    monkeypatch.setattr(pm, "TRAIN_DEDUP", dedup)
    gout = torch.randn((B, 96, 200), generator=g).to(dev)
    out64, fgrad64, grads64, buffers64 = sa_reference_f64(m0, xyz, feat, gout)
    first = None
    for rep in range(50):
        m = copy.deepcopy(m0)
        fa = None if feat is None else feat.clone().requires_grad_(True)
        assert m._train_ok(xyz, fa)
        _, out = m(xyz, fa)
        out.backward(gout)
        grads = {n: p.grad for n, p in m.named_parameters()}
        if rep == 0:
            _close(out, out64, 1e-5, "pooled features")
            if fa is not None:
                _close_pooled(fa.grad, fgrad64, "feature gradient")
            for n in grads:
                _close_pooled(grads[n], grads64[n], n)
            for n, b in m.named_buffers():
                if b.dtype.is_floating_point:
                    _close(b, buffers64[n], 1e-5, n)
            first = (out.clone(), None if fa is None else fa.grad.clone(), {n: v.clone() for n, v in grads.items()})
        else:
            assert torch.equal(out, first[0]), rep
            if fa is not None and dedup:
                assert torch.equal(fa.grad, first[1]), rep
            elif fa is not None:                   # the padded-rows route scatters the feature gradient with float atomics
                _close(fa.grad, first[1], 1e-5, "feature gradient, repetition %d" % rep)
            for n in grads:
                assert torch.equal(grads[n], first[2][n]), (rep, n)


@pytest.mark.parametrize("C1", [0, 32])
def test_fp_module_fused_training_equals_composed_torch_path(dev, C1):
    B, n, m, C2 = 2, 900, 250, 64
    pm, fused, comp = _modules("fp", dev, 11 + C1, mlp=[C2 + C1, 128, 64])
    g = torch.Generator().manual_seed(C1 + 1)
    unknown, known = torch.rand(B, n, 3, generator=g).to(dev), torch.rand(B, m, 3, generator=g).to(dev)
    kf = torch.randn(B, C2, m, generator=g).to(dev)
    uf = None if C1 == 0 else torch.randn(B, C1, n, generator=g).to(dev)
    ka, kb = kf.clone().requires_grad_(True), kf.clone().requires_grad_(True)
    ua = None if uf is None else uf.clone().requires_grad_(True)
    ub = None if uf is None else uf.clone().requires_grad_(True)
    out_a = fused(unknown, known, ua, ka)
    pm.TRAIN_FUSED = False
    try:
        out_b = comp(unknown, known, ub, kb)
    finally:
        pm.TRAIN_FUSED = True
    _close(out_a, out_b, 1e-5, "propagated features")
    gout = torch.randn(out_b.shape, generator=g).to(dev)
    out_a.backward(gout)
    out_b.backward(gout)
    _close(ka.grad, kb.grad, 1e-4, "known-feature gradient")
    if uf is not None:
        _close(ua.grad, ub.grad, 1e-4, "skip-feature gradient")
    _compare_modules(fused, comp, pooled=False)


def test_conv1d_head_layer_fused_training_equals_torch(dev):
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm, pytorch_utils as pt
    torch.manual_seed(0)
    a = pt.Conv1d(128, 128, bn=True).to(dev).train()
    b = copy.deepcopy(a)
    x = torch.randn(2, 128, 3000, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = a(xa)
    pm.TRAIN_FUSED = False
    try:
        yb = b(xb)
    finally:
        pm.TRAIN_FUSED = True
    _close(ya, yb, 1e-5, "head layer")
    gout = torch.randn_like(yb)
    ya.backward(gout)
    yb.backward(gout)
    _close(xa.grad, xb.grad, 1e-4, "input gradient")
    _compare_modules(a, b, pooled=False)


def test_interp_rows_grad_gather_form_equals_the_atomic_form_and_is_repeatable(dev):
    """prcnn_interp_rows_grad_ws: references bucketed per known point, summed in ascending row order (bit-repeatable), against the
    atomic scatter and a float64 reference; a hub referenced by > 256 rows, unreferenced points, m not a multiple of anything"""
    from pointrcnn_amd import _cabi
    from pointrcnn_amd.ops import _p, _stream
    L = _cabi.lib()
    g = torch.Generator().manual_seed(5)
    B, n, m, C = 3, 5000, 777, 96
    idx3 = torch.randint(0, m - 40, (B, n, 3), generator=g, dtype=torch.int32)
    idx3[0, :450, 1] = 5                                       # a hub (sorted in LDS up to 512 references: repeatable)
    w3 = torch.rand(B, n, 3, generator=g)
    G = torch.randn(B * n, C + 4, generator=g)
    idx3, w3, G = idx3.to(dev), w3.to(dev), G.to(dev)
    want = torch.zeros(B, m, C, dtype=torch.float64, device=dev)
    Gv = G[:, :C].view(B, n, C).double()
    for t in range(3):
        want.scatter_add_(1, idx3[:, :, t].long().unsqueeze(-1).expand(-1, -1, C), Gv * w3[:, :, t].double().unsqueeze(-1))
    outs = []
    for rep in range(2):
        out = torch.full((B, m, C), float("nan"), device=dev)
        nbytes = L.prcnn_interp_rows_grad_work_bytes(B, n, m)
        assert nbytes > 0
        work = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _cabi.check(L.prcnn_interp_rows_grad_ws(_p(G), C + 4, _p(idx3), _p(w3), B, n, m, C, _p(out), C, _p(work), nbytes, _stream()), "ws")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    atom = torch.full((B, m, C), float("nan"), device=dev)
    _cabi.check(L.prcnn_interp_rows_grad_ws(_p(G), C + 4, _p(idx3), _p(w3), B, n, m, C, _p(atom), C, None, 0, _stream()), "fallback")
    scale = want.abs().max().item()
    assert (outs[0].double() - want).abs().max().item() <= 2e-6 * scale and (atom.double() - want).abs().max().item() <= 2e-6 * scale
    assert (outs[0][:, m - 40:] == 0).all()                    # unreferenced known points: exact zeros, written
    idx3[1, :1500, 2] = 9                                      # beyond the sort capacity: bucket order, same sum to rounding
    want[1].zero_()
    for t in range(3):
        want[1].scatter_add_(0, idx3[1, :, t].long().unsqueeze(-1).expand(-1, C), Gv[1] * w3[1, :, t].double().unsqueeze(-1))
    out = torch.empty((B, m, C), device=dev)
    _cabi.check(L.prcnn_interp_rows_grad_ws(_p(G), C + 4, _p(idx3), _p(w3), B, n, m, C, _p(out), C, _p(work), nbytes, _stream()), "ws")
    assert (out.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_flat_rows_grad_gather_form_equals_the_atomic_form_and_is_repeatable(dev):
    """prcnn_flat_rows_grad_ws (gradient of the padding-free rows' gather: rows bucketed by source point, frames delimited by their
    groups' first rows on the device) against the atomic scatter and a float64 reference"""
    from pointrcnn_amd import _cabi, ops, train_mlp
    from pointrcnn_amd.ops import _p, _stream
    L = _cabi.lib()
    g = torch.Generator().manual_seed(9)
    B, N, M, ns, C = 3, 900, 70, 16, 40
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    new_xyz = xyz[:, :M].contiguous()
    idx = ops.ball_query(0.12, ns, xyz, new_xyz)
    rows = train_mlp.GroupRows(idx, new_xyz, N)
    live = int(rows.rows_dev.item())
    assert 0 < live < B * M * ns
    G = torch.randn(rows.max_rows, C + 4, generator=g).to(dev)
    want = torch.zeros(B * N, C, dtype=torch.float64, device=dev)
    want.index_add_(0, rows.ridx[:live].long(), G[:live, :C].double())
    nbytes = L.prcnn_flat_rows_grad_work_bytes(B, N, rows.max_rows)
    assert nbytes > 0
    outs = []
    for rep in range(2):
        out = torch.full((B, N, C), float("nan"), device=dev)
        work = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _cabi.check(L.prcnn_flat_rows_grad_ws(_p(G), C + 4, _p(rows.ridx), _p(rows.rows_dev), rows.max_rows, C, _p(out), C, _p(rows.off),
                                              B, N, M, _p(work), nbytes, _stream()), "ws")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    atom = torch.full((B, N, C), float("nan"), device=dev)
    _cabi.check(L.prcnn_flat_rows_grad_ws(_p(G), C + 4, _p(rows.ridx), _p(rows.rows_dev), rows.max_rows, C, _p(atom), C, _p(rows.off),
                                          B, N, M, None, 0, _stream()), "fallback")
    scale = want.abs().max().item()
    assert (outs[0].view(-1, C).double() - want).abs().max().item() <= 2e-6 * scale
    assert (atom.view(-1, C).double() - want).abs().max().item() <= 2e-6 * scale


def test_head_sequential_on_rows_equals_module_by_module(dev):
    """pt_utils.fused_sequential in training mode (the RPN heads: Conv1d+BN+ReLU -> Dropout -> Conv1d with bias): the rows path
    (one fused node, elementwise dropout, F.linear; output a view whose transpose is contiguous) == running the Sequential
    module by module through torch, for outputs, input gradient and every parameter gradient; with p > 0 the masks differ (a
    different element order draws them) but the zero fraction and the scaling are dropout's"""
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointnet2_lib.pointnet2 import pointnet2_modules as pm, pytorch_utils as pt
    torch.manual_seed(3)
    seq = nn.Sequential(pt.Conv1d(128, 128, bn=True), nn.Dropout(0.0), pt.Conv1d(128, 76, activation=None)).to(dev).train()
    ref = copy.deepcopy(seq)
    base = torch.randn(2, 3000, 128, device=dev)
    xa, xb = base.clone().requires_grad_(True), base.clone().requires_grad_(True)
    ya = pt.fused_sequential(seq, xa.transpose(1, 2))
    assert ya.shape == (2, 76, 3000) and ya.transpose(1, 2).is_contiguous()
    pm.TRAIN_FUSED = False
    try:
        yb = ref(xb.transpose(1, 2))
    finally:
        pm.TRAIN_FUSED = True
    _close(ya, yb, 1e-5, "head output")
    gout = torch.randn_like(yb)
    ya.backward(gout)
    yb.backward(gout)
    _close(xa.grad, xb.grad, 1e-4, "input gradient")
    _compare_modules(seq, ref, pooled=False)
    seq[1].p = 0.5
    y = pt.fused_sequential(seq, base.transpose(1, 2))
    seq[2]._parts()[0].weight.data.fill_(1.0)
    seq[2]._parts()[0].bias.data.zero_()
    h = pt.fused_sequential(nn.Sequential(seq[0], seq[1]), base.transpose(1, 2))
    frac = (h == 0).float().mean().item()
    assert 0.5 < frac < 0.9 and torch.isfinite(y).all()          # relu zeros + half of the rest dropped


def test_eval_mode_and_unsupported_layers_keep_their_paths(dev):
    """BatchNorm in eval mode / a layer without activation / odd channel counts are not this path's business: stack_ok says no,
    modules fall back.  (Conv + bias -> ReLU WITHOUT BatchNorm is covered since the RCNN stage trains on it: test_gpu_train_rcnn.py)"""
    import pointrcnn_amd
    pointrcnn_amd.install()
    from pointrcnn_amd import train_mlp
    from pointnet2_lib.pointnet2 import pytorch_utils as pt
    a = pt.Conv1d(16, 16, bn=True).to(dev)
    assert train_mlp.stack_ok([a.train()]) and not train_mlp.stack_ok([a.eval()])
    assert train_mlp.stack_ok([pt.Conv1d(16, 16, bn=False).to(dev).train()])
    assert not train_mlp.stack_ok([pt.Conv1d(16, 16, bn=False, activation=None).to(dev).train()])
    assert not train_mlp.stack_ok([pt.Conv1d(16, 1, bn=True).to(dev).train()])            # channel count not a multiple of 4
