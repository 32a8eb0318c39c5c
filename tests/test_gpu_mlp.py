"""GPU parity: fused per-point MLP layers (fp32 MFMA) vs the double-accumulated CPU oracle.
The gathered / interpolated A rows are exact; the GEMM sums in a different k order than a sequential
reference, so the contract is |hip - oracle| <= 1e-5 * max(1, max|oracle|) (util.mlp_tol)."""
import numpy as np
import pytest
import torch

from util import mlp_tol, unit_cloud

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def lin(dev, w, b, relu, k_rot=0):
    from pointrcnn_amd import ops
    return ops.PackedLinear(T(w, dev), None if b is None else T(b, dev), relu=relu, k_rot=k_rot)


def assert_elementwise(got, rows, layers, pool_ns=0, eps=1e-5, rows_err=None):
    """north_star's 1e-5 read PER ELEMENT for a stack of layers (round 6: the review asked for the bound of test_mlp_rows_matches_oracle
    on the chain / grouped / interpolated launches too).  rows: the fp32 input rows of the first layer (as the kernel builds them: gathered /
    interpolated values are exact copies or exactly specified fp32 expressions); layers: [(w, b, relu), ...].  The stack is evaluated in
    float64; a layer's own rounding is allowed eps * (sum_k |h_k||w_k| + |b|) per output, the error it inherits from its input is pushed
    through |W| (ReLU and max are 1-Lipschitz).  rows_err: per-element error already in the rows (interpolated rows: a few fp32 roundings)."""
    h = rows.astype(np.float64)
    err = np.zeros_like(h) if rows_err is None else rows_err.astype(np.float64)
    for w, b, relu in layers:
        wd = w.astype(np.float64)
        bd = 0.0 if b is None else b.astype(np.float64)
        s_ = np.abs(h) @ np.abs(wd).T + (0.0 if b is None else np.abs(bd))
        err = err @ np.abs(wd).T + eps * s_
        h = h @ wd.T + bd
        if relu:
            h = np.maximum(h, 0.0)
    if pool_ns:
        h = h.reshape(-1, pool_ns, h.shape[-1]).max(1)
        err = err.reshape(-1, pool_ns, err.shape[-1]).max(1)
    d = np.abs(got.astype(np.float64) - h)
    assert d.shape == err.shape and (d <= err + 1e-30).all(), float((d / (err + 1e-30)).max())


@pytest.mark.parametrize("rows,K,Nout,relu,bias", [(128, 32, 64, True, True), (1000, 3, 16, True, False),
                                                   (77, 5, 1, False, True), (300, 99, 76, False, True),
                                                   (513, 131, 196, True, True), (256, 515, 512, True, True),
                                                   (40, 1536, 33, True, False), (129, 8, 32, False, False)])
def test_mlp_rows_matches_oracle(dev, cpu, rows, K, Nout, relu, bias):
    from pointrcnn_amd import ops
    r = np.random.default_rng(rows + K)
    a = r.normal(size=(rows, K)).astype(np.float32)
    w = (r.normal(size=(Nout, K)) * 0.2).astype(np.float32)
    b = r.normal(size=(Nout,)).astype(np.float32) if bias else None
    want = cpu.linear_rows(a, w, b, relu)
    got = ops.mlp_rows(T(a, dev), lin(dev, w, b, relu)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)
    # the same contract PER ELEMENT (north_star's 1e-5 read literally): |err| <= 1e-5 * (sum_k |x_k||w_k| + |b|) against float64, in
    # both arithmetics of the fused inference MLPs (conftest mlp_mode) -- the tensor-scale bound above is the looser of the two
    exact = a.astype(np.float64) @ w.astype(np.float64).T + (0.0 if b is None else b.astype(np.float64))
    if relu:
        exact = np.maximum(exact, 0.0)
    elem = 1e-5 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + (0.0 if b is None else np.abs(b).astype(np.float64))) + 1e-30
    assert (np.abs(got.astype(np.float64) - exact) <= elem).all(), float((np.abs(got - exact) / elem).max())


def test_mlp_rows_identity_is_transpose_safe(dev):
    """A = identity with an ASYMMETRIC W: catches row/col swaps in the MFMA output mapping"""
    from pointrcnn_amd import ops
    K, Nout = 64, 96
    w = (np.arange(Nout * K, dtype=np.float32).reshape(Nout, K) % 251) / 16.0
    got = ops.mlp_rows(T(np.eye(K, dtype=np.float32), dev), lin(dev, w, None, False)).cpu().numpy()
    assert np.array_equal(got, w.T)      # single non-zero product per output: exact


def test_mlp_rows_strided_io_and_col_offset(dev, cpu):
    """input rows taken out of a wider buffer (ld_in > K, unaligned) and output written at a column offset"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(7)
    wide = r.normal(size=(2, 150, 133)).astype(np.float32)
    w = (r.normal(size=(40, 5)) * 0.3).astype(np.float32)
    b = r.normal(size=(40,)).astype(np.float32)
    x = T(wide, dev)[..., 0:5]                       # (2,150,5) view, row stride 133
    out = torch.full((300, 100), -7.0, device=dev)
    ops.mlp_rows(x, lin(dev, w, b, True), out=(out, 30))
    want = cpu.linear_rows(wide[..., 0:5].reshape(-1, 5), w, b, True)
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 30:70], want, atol=mlp_tol(want), rtol=0)
    assert (got[:, :30] == -7.0).all() and (got[:, 70:] == -7.0).all()


@pytest.mark.parametrize("ns", [16, 32, 64])
def test_mlp_rows_fused_maxpool(dev, cpu, ns):
    from pointrcnn_amd import ops
    r = np.random.default_rng(ns)
    groups, K, Nout = 37, 24, 70
    a = r.normal(size=(groups * ns, K)).astype(np.float32)
    w = (r.normal(size=(Nout, K)) * 0.3).astype(np.float32)
    b = r.normal(size=(Nout,)).astype(np.float32)
    want = cpu.linear_rows(a, w, b, True).reshape(groups, ns, Nout).max(1)
    got = ops.mlp_rows(T(a, dev), lin(dev, w, b, True), pool_ns=ns).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)
    # generic pooling kernel (any nsample) agrees with the fused epilogue bit for bit
    full = ops.mlp_rows(T(a, dev), lin(dev, w, b, True))
    assert np.array_equal(ops.maxpool_rows(full, ns).cpu().numpy(), got)


@pytest.mark.parametrize("C,Nout,ns", [(0, 16, 16), (3, 32, 32), (96, 64, 16), (13, 40, 8), (256, 128, 64)])
def test_mlp_group_matches_oracle(dev, cpu, C, Nout, ns):
    """first SA layer fused with grouping: rows = [dxyz, feat] in torch order, weights in torch order"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(C + ns)
    B, N, M = 2, 600, 50
    xyz = unit_cloud(B, N, seed=C)
    new_xyz = xyz[:, :M].copy()
    idx = cpu.ball_query(0.25, ns, xyz, new_xyz)
    feat = r.normal(size=(B, C, N)).astype(np.float32) if C else None
    w = (r.normal(size=(Nout, C + 3)) * 0.3).astype(np.float32)
    b = r.normal(size=(Nout,)).astype(np.float32)
    # oracle: grouping_operation on xyz^T and features, centre, concat [dxyz, feat], then the layer
    gx = cpu.group(xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    g = gx if feat is None else np.concatenate([gx, cpu.group(feat, idx)], 1)      # (B,3+C,M,ns)
    rows = g.transpose(0, 2, 3, 1).reshape(-1, C + 3)
    want = cpu.linear_rows(rows, w, b, True)
    feat_cl = None if feat is None else T(feat.transpose(0, 2, 1), dev)
    got = ops.mlp_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), feat_cl, lin(dev, w, b, True, k_rot=3 if C else 0))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=mlp_tol(want), rtol=0)
    assert_elementwise(got.cpu().numpy(), rows, [(w, b, True)])


def test_mlp_group_gathered_rows_are_exact(dev, cpu):
    """identity weights expose the A tile itself: grouped features must be EXACT copies (north star: 1e-5;
    here bit equal), dxyz an exactly rounded fp32 subtraction"""
    from pointrcnn_amd import ops
    B, N, M, ns, C = 1, 400, 30, 16, 29
    r = np.random.default_rng(3)
    xyz = unit_cloud(B, N, seed=8)
    new_xyz = xyz[:, :M].copy()
    idx = cpu.ball_query(0.3, ns, xyz, new_xyz)
    feat = r.normal(size=(B, C, N)).astype(np.float32)
    w = np.eye(C + 3, dtype=np.float32)
    got = ops.mlp_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), T(feat.transpose(0, 2, 1), dev),
                        lin(dev, w, None, False, k_rot=3)).cpu().numpy()
    gx = cpu.group(xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    g = np.concatenate([gx, cpu.group(feat, idx)], 1).transpose(0, 2, 3, 1).reshape(-1, C + 3)
    assert np.array_equal(got, g)


@pytest.mark.parametrize("C2,C1,Nout", [(64, 0, 32), (256, 96, 128), (30, 7, 20), (512, 256, 64)])
def test_mlp_interp_matches_oracle(dev, cpu, C2, C1, Nout):
    """first FP layer fused with three_interpolate + skip concat"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(C2 + C1)
    B, n, m = 2, 333, 80
    unk, kn = unit_cloud(B, n, seed=1), unit_cloud(B, m, seed=2)
    d2, idx3 = cpu.three_nn(unk, kn)
    w3 = cpu.three_weights(d2)
    kf = r.normal(size=(B, C2, m)).astype(np.float32)
    sf = r.normal(size=(B, C1, n)).astype(np.float32) if C1 else None
    w = (r.normal(size=(Nout, C2 + C1)) * 0.2).astype(np.float32)
    b = r.normal(size=(Nout,)).astype(np.float32)
    interp = cpu.three_interp(kf, idx3, w3)
    cat = interp if sf is None else np.concatenate([interp, sf], 1)
    want = cpu.linear_rows(cat.transpose(0, 2, 1).reshape(-1, C2 + C1), w, b, True)
    got = ops.mlp_interp(T(kf.transpose(0, 2, 1), dev), T(idx3, dev), T(w3, dev),
                         None if sf is None else T(sf.transpose(0, 2, 1), dev), lin(dev, w, b, True))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=mlp_tol(want), rtol=0)
    # (the kernel's interpolated rows ARE the oracle's three_interp values bit for bit: test_gather_group_interp_forward_exact)
    assert_elementwise(got.cpu().numpy(), cat.transpose(0, 2, 1).reshape(-1, C2 + C1), [(w, b, True)])


# ------------------------------------------------------------------ register-resident layer chains
def _stack(r, dims, scale=0.25):
    ws = [(r.normal(size=(dims[i + 1], dims[i])) * scale).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [r.normal(size=(dims[i + 1],)).astype(np.float32) * 0.2 for i in range(len(dims) - 1)]
    return ws, bs


@pytest.mark.parametrize("C,widths,ns", [(0, (16, 16, 32), 16), (0, (32, 32, 64), 32), (96, (64, 64, 128), 16),
                                         (96, (64, 96, 128), 32)])
def test_chain_group_equals_oracle_and_per_layer(dev, cpu, C, widths, ns):
    """the four RPN SA1/SA2 stacks (default.yaml:41-44) as ONE kernel: vs the oracle and vs the per-layer kernels"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(C + ns)
    B, N, M = 2, 700, 45                               # rows = 2*45*ns: not a multiple of the 128-row workgroup
    xyz = unit_cloud(B, N, seed=ns)
    new_xyz = xyz[:, :M].copy()
    idx = cpu.ball_query(0.3, ns, xyz, new_xyz)
    feat = r.normal(size=(B, C, N)).astype(np.float32) if C else None
    ws, bs = _stack(r, (C + 3,) + widths)
    gx = cpu.group(xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    g = gx if feat is None else np.concatenate([gx, cpu.group(feat, idx)], 1)
    rows = g.transpose(0, 2, 3, 1).reshape(-1, C + 3)
    rows0 = rows
    for w, b in zip(ws, bs):
        rows = cpu.linear_rows(rows, w, b, True)
    want = rows.reshape(B * M, ns, -1).max(1)
    layers = [lin(dev, ws[0], bs[0], True, k_rot=3 if C else 0)] + [lin(dev, w, b, True) for w, b in zip(ws[1:], bs[1:])]
    assert ops.chain_supported(1, layers, ns)
    feat_cl = None if feat is None else T(feat.transpose(0, 2, 1), dev)
    out = torch.full((B * M, widths[-1] + 8), -3.0, device=dev)
    ops.mlp_chain_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), feat_cl, layers, out=(out, 4), pool_ns=ns)
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 4:4 + widths[-1]], want, atol=mlp_tol(want), rtol=0)
    assert_elementwise(got[:, 4:4 + widths[-1]], rows0, [(w, b, True) for w, b in zip(ws, bs)], pool_ns=ns)
    assert (got[:, :4] == -3.0).all() and (got[:, 4 + widths[-1]:] == -3.0).all()
    x = ops.mlp_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), feat_cl, layers[0])
    x = ops.mlp_rows(x, layers[1])
    x = ops.mlp_rows(x, layers[2], pool_ns=ns).cpu().numpy()
    np.testing.assert_allclose(got[:, 4:4 + widths[-1]], x, atol=mlp_tol(want), rtol=0)


def test_chain_interp_and_rows_equal_oracle(dev, cpu):
    """FP0 stack (256 -> 128 -> 128, no skip) and the two RPN head stacks (128 -> 128 -> {1, 76})"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(11)
    B, n, m, C2 = 2, 301, 90, 256
    unk, kn = unit_cloud(B, n, seed=1), unit_cloud(B, m, seed=2)
    d2, idx3 = cpu.three_nn(unk, kn)
    w3 = cpu.three_weights(d2)
    kf = r.normal(size=(B, C2, m)).astype(np.float32)
    ws, bs = _stack(r, (C2, 128, 128), 0.1)
    rows = cpu.three_interp(kf, idx3, w3).transpose(0, 2, 1).reshape(-1, C2)
    want = cpu.linear_rows(cpu.linear_rows(rows, ws[0], bs[0], True), ws[1], bs[1], True)
    layers = [lin(dev, w, b, True) for w, b in zip(ws, bs)]
    assert ops.chain_supported(2, layers, 0)
    got = ops.mlp_chain_interp(T(kf.transpose(0, 2, 1), dev), T(idx3, dev), T(w3, dev), None, layers).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)
    assert_elementwise(got, rows, [(ws[0], bs[0], True), (ws[1], bs[1], True)])
    feats = want                                            # (602,128) rows feed the heads
    for nout in (1, 76, 128):
        hw, hb = _stack(r, (128, 128, nout), 0.1)
        ref = cpu.linear_rows(cpu.linear_rows(feats, hw[0], hb[0], True), hw[1], hb[1], False)
        hl = [lin(dev, hw[0], hb[0], True), lin(dev, hw[1], hb[1], False)]
        assert ops.chain_supported(0, hl, 0)
        out = ops.mlp_chain_rows(T(feats, dev), hl).cpu().numpy()
        assert out.shape == (602, nout)
        np.testing.assert_allclose(out, ref, atol=mlp_tol(ref), rtol=0)
        assert_elementwise(out, feats, [(hw[0], hb[0], True), (hw[1], hb[1], False)])


def test_chain_unsupported_shapes_are_reported_not_guessed(dev):
    from pointrcnn_amd import ops, _cabi
    r = np.random.default_rng(0)
    ws, bs = _stack(r, (20, 256, 64))
    layers = [lin(dev, w, b, True) for w, b in zip(ws, bs)]
    assert not ops.chain_supported(0, layers, 0)            # 256 > 128
    with pytest.raises(_cabi.PointOpsError):
        ops.mlp_chain_rows(torch.zeros(64, 20, device=dev), layers)


# ------------------------------------------------------------------ first-layer hoisting (exact algebra, fewer FLOPs)
@pytest.mark.parametrize("chain", [False, True])
def test_hoisted_group_equals_unhoisted_oracle(dev, cpu, chain):
    """W.[dxyz; feat[idx]] computed as relu((W_f.feat)[idx] + W_x.dxyz + b): compared against the oracle's plain
    group -> layer -> layer -> max-pool on the ORIGINAL weights (SA2 widths 99 -> 64 -> 96 -> 128, nsample 32)"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(21)
    B, N, M, ns, C = 2, 900, 40, 32, 96
    widths = (64, 96, 128) if chain else (64, 160, 72)      # (3,4,0)/32 has a chain instance; 160 forces the tiled path
    xyz = unit_cloud(B, N, seed=4)
    new_xyz = xyz[:, :M].copy()
    idx = cpu.ball_query(0.3, ns, xyz, new_xyz)
    feat = r.normal(size=(B, C, N)).astype(np.float32)
    ws, bs = _stack(r, (C + 3,) + widths, 0.15)
    gx = cpu.group(xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    rows = np.concatenate([gx, cpu.group(feat, idx)], 1).transpose(0, 2, 3, 1).reshape(-1, C + 3)
    for w, b in zip(ws, bs):
        rows = cpu.linear_rows(rows, w, b, True)
    want = rows.reshape(B * M, ns, -1).max(1)
    # hoisted: Z per source point, second/third layers on activated gathered rows
    feat_cl = T(feat.transpose(0, 2, 1), dev)
    z = ops.mlp_rows(feat_cl, lin(dev, ws[0][:, 3:].copy(), None, False)).view(B, N, -1)
    act = (T(ws[0][:, :3].copy(), dev), T(bs[0], dev))
    rest = [lin(dev, w, b, True) for w, b in zip(ws[1:], bs[1:])]
    if chain:
        assert ops.chain_supported(1, rest, ns)
        got = ops.mlp_chain_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), z, rest, pool_ns=ns, act=act)
    else:
        x = ops.mlp_group(T(xyz, dev), T(new_xyz, dev), T(idx, dev), z, rest[0], act=act)
        got = ops.mlp_rows(x, rest[1], pool_ns=ns)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=mlp_tol(want), rtol=0)


def test_hoisted_interp_equals_unhoisted_oracle(dev, cpu):
    """W.[interp(x); skip] computed as interp(W_a.x) + W_b.skip (epilogue add), and the no-skip form
    relu(interp(Y) + b) feeding the second layer (tiled and chain), vs the oracle's plain interpolate -> layers"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(22)
    B, n, m, C2, C1 = 2, 500, 130, 256, 96
    unk, kn = unit_cloud(B, n, seed=5), unit_cloud(B, m, seed=6)
    d2, idx3 = cpu.three_nn(unk, kn)
    w3 = cpu.three_weights(d2)
    kf = r.normal(size=(B, C2, m)).astype(np.float32)
    sf = r.normal(size=(B, C1, n)).astype(np.float32)
    interp = cpu.three_interp(kf, idx3, w3)
    # with skip: 352 -> 256 -> 256 (FP1 shape class)
    ws, bs = _stack(r, (C2 + C1, 256, 256), 0.08)
    rows = np.concatenate([interp, sf], 1).transpose(0, 2, 1).reshape(-1, C2 + C1)
    want = cpu.linear_rows(cpu.linear_rows(rows, ws[0], bs[0], True), ws[1], bs[1], True)
    y = ops.mlp_rows(T(kf.transpose(0, 2, 1), dev), lin(dev, ws[0][:, :C2].copy(), None, False)).view(B, m, -1)
    x = ops.mlp_rows_addinterp(T(sf.transpose(0, 2, 1), dev), lin(dev, ws[0][:, C2:].copy(), bs[0], True), y, T(idx3, dev), T(w3, dev))
    got = ops.mlp_rows(x, lin(dev, ws[1], bs[1], True)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)
    # without skip: 256 -> 128 -> 128 (FP0): chain (4,0,0) and tiled
    ws, bs = _stack(r, (C2, 128, 128), 0.08)
    rows = interp.transpose(0, 2, 1).reshape(-1, C2)
    want = cpu.linear_rows(cpu.linear_rows(rows, ws[0], bs[0], True), ws[1], bs[1], True)
    y = ops.mlp_rows(T(kf.transpose(0, 2, 1), dev), lin(dev, ws[0], None, False)).view(B, m, -1)
    l1 = lin(dev, ws[1], bs[1], True)
    b0 = T(bs[0], dev)
    assert ops.chain_supported(2, [l1], 0)
    got_c = ops.mlp_chain_interp(y, T(idx3, dev), T(w3, dev), None, [l1], act_bias=b0).cpu().numpy()
    got_t = ops.mlp_interp(y, T(idx3, dev), T(w3, dev), None, l1, act_bias=b0).cpu().numpy()
    np.testing.assert_allclose(got_c, want, atol=mlp_tol(want), rtol=0)
    np.testing.assert_allclose(got_t, want, atol=mlp_tol(want), rtol=0)


def test_chain_kernel_variants_are_bit_identical(dev, cpu, monkeypatch):
    """The register-chain has four code paths for the same arithmetic -- generic (bounds-checked), straight-line "fast",
    persistent with LDS-resident weights (opt-in) and the persistent register-weight kernel of SA level 0 -- selected
    by shape and environment.  They must agree bit for bit (same MFMA order, same activation expressions)."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(33)
    B, N, M = 2, 1500, 70

    def run_all(fn):
        outs = {}
        for name, env in (("default", {}), ("generic", {"PRCNN_NO_FAST_CHAIN": "1", "PRCNN_NO_SA0": "1"}),
                          ("persistent", {"PRCNN_PERSISTENT_CHAIN": "1"})):
            for k in ("PRCNN_NO_FAST_CHAIN", "PRCNN_NO_SA0", "PRCNN_PERSISTENT_CHAIN"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            outs[name] = fn().clone()
        for k in ("PRCNN_NO_FAST_CHAIN", "PRCNN_NO_SA0", "PRCNN_PERSISTENT_CHAIN"):
            monkeypatch.delenv(k, raising=False)
        assert torch.equal(outs["default"], outs["generic"]) and torch.equal(outs["default"], outs["persistent"])
        return outs["default"]

    xyz = T(unit_cloud(B, N, seed=8), dev)
    new_xyz = xyz[:, :M].contiguous()
    # hoisted SA2-type stacks (activated gather -> 2 layers -> pool), both scale shapes
    for ns, widths in ((16, (64, 64, 128)), (32, (64, 96, 128))):
        idx = ops.ball_query(0.3, ns, xyz, new_xyz)
        z = T(r.normal(size=(B, N, widths[0])).astype(np.float32), dev)
        ws, bs = _stack(r, widths, 0.15)
        act = (T(r.normal(size=(widths[0], 3)).astype(np.float32), dev), T(r.normal(size=(widths[0],)).astype(np.float32), dev))
        layers = [lin(dev, w, b, True) for w, b in zip(ws, bs)]
        run_all(lambda: ops.mlp_chain_group(xyz, new_xyz, idx, z, layers, pool_ns=ns, act=act))
    # SA level 0 stacks (xyz only)
    for ns, widths in ((16, (3, 16, 16, 32)), (32, (3, 32, 32, 64))):
        idx = ops.ball_query(0.3, ns, xyz, new_xyz)
        ws, bs = _stack(r, widths, 0.3)
        layers = [lin(dev, w, b, True) for w, b in zip(ws, bs)]
        got = run_all(lambda: ops.mlp_chain_group(xyz, new_xyz, idx, None, layers, pool_ns=ns))
        gx = cpu.group(xyz.cpu().numpy().transpose(0, 2, 1), idx.cpu().numpy()) - new_xyz.cpu().numpy().transpose(0, 2, 1)[..., None]
        rows = gx.transpose(0, 2, 3, 1).reshape(-1, 3)
        for w, b in zip(ws, bs):
            rows = cpu.linear_rows(rows, w, b, True)
        want = rows.reshape(B * M, ns, -1).max(1)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=mlp_tol(want), rtol=0)
    # hoisted FP0 (interp of Y, one layer) and the two heads (plain rows, 128 -> 128 -> 1 / 76)
    kn = T(unit_cloud(B, 300, seed=9), dev)
    d2, idx3, w3 = ops.three_nn(xyz, kn, want_weight=True)
    y = T(r.normal(size=(B, 300, 128)).astype(np.float32), dev)
    ws, bs = _stack(r, (128, 128), 0.1)
    b0 = T(r.normal(size=(128,)).astype(np.float32), dev)
    l1 = lin(dev, ws[0], bs[0], True)
    feat = run_all(lambda: ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0))
    for nout in (1, 76):
        ws, bs = _stack(r, (128, 128, nout), 0.1)
        layers = [lin(dev, ws[0], bs[0], True), lin(dev, ws[1], bs[1], False)]
        run_all(lambda: ops.mlp_chain_rows(feat, layers))


def test_tile_order_remaps_are_bit_identical(dev, monkeypatch):
    """Workgroup -> tile remaps must not change a byte: the XCD-aware frame order of the interp chain (B % 8 == 0, n % 128 == 0:
    XCD x walks frames x, x+8, ...) and the live-prefix order of segment-skipping launches (tile q of segment s taken by
    workgroup q * nseg + s).  Dead rows are compared only where they are live."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(44)
    B, n, m = 16, 384, 96                        # 3 tiles per frame, 16 frames: every XCD gets two frames
    unknown, known = T(unit_cloud(B, n, seed=3), dev), T(unit_cloud(B, m, seed=4), dev)
    _, idx3, w3 = ops.three_nn(unknown, known, want_weight=True)
    y = T(r.normal(size=(B, m, 128)).astype(np.float32), dev)
    ws, bs = _stack(r, (128, 128), 0.1)
    b0 = T(r.normal(size=(128,)).astype(np.float32), dev)
    l1 = lin(dev, ws[0], bs[0], True)
    monkeypatch.delenv("PRCNN_NO_XCD_ORDER", raising=False)
    a = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0).clone()
    monkeypatch.setenv("PRCNN_NO_XCD_ORDER", "1")
    b = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0).clone()
    monkeypatch.delenv("PRCNN_NO_XCD_ORDER", raising=False)
    assert torch.equal(a, b)
    # segment-prefix live rows: 40 segments of 256 rows, live counts from 1 to 256 (both tiles of a segment live or not)
    S, nseg = 256, 40
    cnt = torch.tensor([1, 127, 128, 129, 255, 256] + list(r.integers(1, 257, nseg - 6)), dtype=torch.int32, device=dev)
    x = T(r.normal(size=(nseg * S, 96)).astype(np.float32), dev)
    w2, b2 = _stack(r, (96, 160), 0.1)
    l2 = lin(dev, w2[0], b2[0], True)
    live = (torch.arange(S, device=dev)[None, :] < cnt[:, None].long()).view(-1)
    full = ops.mlp_rows(x, l2)
    part = ops.mlp_rows(x, l2, seg=(cnt, S))
    assert torch.equal(full[live], part[live])
    up = [lin(dev, w, b, True) for w, b in zip(*_stack(r, (5, 128, 128), 0.2))]
    pts = T(r.normal(size=(nseg * S, 5)).astype(np.float32), dev)
    assert torch.equal(ops.mlp_chain_rows(pts, up)[live], ops.mlp_chain_rows(pts, up, seg=(cnt, S))[live])


# ---------------------------------------------------------------- split-bf16 variant (opt-in, never the default arithmetic)
def _scaled_rows(r, rows, K):
    """rows whose magnitudes span several binades (the split must be exact whatever the exponent)"""
    return (r.normal(size=(rows, K)) * np.exp2(r.integers(-12, 12, size=(rows, 1)))).astype(np.float32)


@pytest.mark.parametrize("rows,K,Nout,relu,bias", [(24576, 32, 128, True, True), (25000, 96, 256, True, True), (12288, 512, 512, False, True),
                                                   (24576, 64, 64, True, False), (24576 + 77, 64, 96, True, True), (16384, 128, 200, False, True),
                                                   (12288 + 5, 32, 76, True, True), (3000, 256, 96, True, True), (200, 64, 64, True, True)])
@pytest.mark.own_arithmetic
@pytest.mark.parametrize("terms", [6, 3])
def test_split_bf16_rows_against_float64(dev, monkeypatch, rows, K, Nout, relu, bias, terms):
    """the split-bf16 layer against a float64 product.  Six terms: every dropped partial product is below 2^-24 |x||w|, so the
    result must sit within the fp32 kernel's own contract (1e-5 of the scale) -- checked here against the tighter, per-element
    bound 2e-6 * (|x| . |w| + |b|).  Three terms: below 2^-16 |x||w| per product, bound 4e-5 * (|x| . |w| + |b|).  The last two
    shapes are launches of a few tiles: they take the split kernel like any other (the arithmetic of a layer does not depend on
    its row count)."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(rows + K + terms)
    a = _scaled_rows(r, rows, K)
    w = (r.normal(size=(Nout, K)) * 0.2).astype(np.float32)
    b = r.normal(size=(Nout,)).astype(np.float32) if bias else None
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", terms)
    got = ops.mlp_rows(T(a, dev), lin(dev, w, b, relu)).cpu().numpy().astype(np.float64)
    want = a.astype(np.float64) @ w.astype(np.float64).T + (0.0 if b is None else b.astype(np.float64))
    if relu:
        want = np.maximum(want, 0.0)
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + (0.0 if b is None else np.abs(b).astype(np.float64))
    bound = (2e-6 if terms == 6 else 4e-5) * scale + 1e-30
    assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())
    if terms == 6:
        np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)


@pytest.mark.own_arithmetic
def test_split_bf16_identity_and_exact_pieces(dev, monkeypatch):
    """A = identity with an asymmetric W whose entries need all 24 significand bits: with six terms the three pieces of W are
    reassembled exactly (x = 1 has one piece), so the output IS W^T bit for bit -- catches a wrong lane / k mapping of the bf16
    operands and a lossy split; a strided input and a column offset in the output go through the same path"""
    from pointrcnn_amd import ops
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    K, Nout, rows = 128, 128, 24576
    r = np.random.default_rng(3)
    w = (r.integers(1 << 23, 1 << 24, size=(Nout, K)).astype(np.float32) * np.exp2(r.integers(-30, 10, size=(Nout, K)))).astype(np.float32)
    a = np.zeros((rows, K + 4), np.float32)
    a[np.arange(rows), np.arange(rows) % K] = 1.0
    out = torch.full((rows, Nout + 8), -7.0, device=dev)
    ops.mlp_rows(T(a, dev)[:, :K], lin(dev, w, None, False), out=(out, 4))
    got = out.cpu().numpy()
    assert np.array_equal(got[:, 4:4 + Nout], w.T[np.arange(rows) % K])
    assert (got[:, :4] == -7.0).all() and (got[:, 4 + Nout:] == -7.0).all()


@pytest.mark.own_arithmetic
def test_split_bf16_addinterp_equals_fp32_layer(dev, monkeypatch):
    """hoisted FP first layer with the interpolated addend: the split variant against the fp32 kernel on the same inputs"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(11)
    B, n, m, C1, Nout = 4, 8192, 2048, 96, 256
    skip = T(r.normal(size=(B, n, C1)).astype(np.float32), dev)
    y = T(r.normal(size=(B, m, Nout)).astype(np.float32), dev)
    idx3 = T(r.integers(0, m, size=(B, n, 3)).astype(np.int32), dev)
    w3 = r.random(size=(B, n, 3)).astype(np.float32)
    w3 = T(w3 / w3.sum(-1, keepdims=True), dev)
    l = lin(dev, (r.normal(size=(Nout, C1)) * 0.2).astype(np.float32), r.normal(size=(Nout,)).astype(np.float32), True)
    want = ops.mlp_rows_addinterp(skip, l, y, idx3, w3).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_rows_addinterp(skip, l, y, idx3, w3).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=mlp_tol(want), rtol=0)


@pytest.mark.own_arithmetic
def test_split_bf16_full_rpn_stays_within_the_fp32_contract(dev, monkeypatch):
    """the whole RPN graph (default.yaml, 4 frames of 16384 points) with the plain-row layers on the six-term split kernel: sample
    sets and neighbour lists are untouched (index operators see the same coordinates), features and head outputs stay within the
    fused-vs-composed bound of test_full_rpn_fused_equals_composed -- and far inside it: 1e-5 of the output scale"""
    from pointrcnn_amd import ops, rpn
    torch.manual_seed(4)
    model = rpn.randomize_bn_stats(rpn.RPN()).to(dev).eval()
    pts = rpn.synthetic_clouds(4, 16384, device=dev)
    with torch.no_grad():
        ref = {k: v.clone() for k, v in model({"pts_input": pts}).items() if torch.is_tensor(v)}
        monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
        out = model({"pts_input": pts})
    differs = False
    for k in ("backbone_features", "rpn_cls", "rpn_reg"):
        a, b = out[k].cpu().numpy(), ref[k].cpu().numpy()
        np.testing.assert_allclose(a, b, atol=1e-5 * max(1.0, float(np.abs(b).max())), rtol=0)
        differs = differs or not np.array_equal(a, b)
    assert differs, "the split kernel did not run (outputs are bit-identical to the fp32 path)"


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("n1,relu1", [(76, False), (1, False), (128, True), (96, True)])
@pytest.mark.parametrize("terms", [6, 3])
def test_split_bf16_two_layer_chain_against_float64(dev, monkeypatch, n1, relu1, terms):
    """the heads' two-layer chain (128 -> 128 -> n1) on the split-bf16 chain kernel against a float64 evaluation of the same
    stack; bound per element: the first layer's error bound pushed through |W2|, plus the second layer's own"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(n1 + terms)
    rows = 40000 + 17                                     # a ragged last tile
    x = _scaled_rows(r, rows, 128)
    w0 = (r.normal(size=(128, 128)) * 0.1).astype(np.float32)
    b0 = r.normal(size=(128,)).astype(np.float32)
    w1 = (r.normal(size=(n1, 128)) * 0.1).astype(np.float32)
    b1 = r.normal(size=(n1,)).astype(np.float32)
    layers = [lin(dev, w0, b0, True), lin(dev, w1, b1, relu1)]
    ref32 = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", terms)
    got = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy().astype(np.float64)
    assert got.shape == (rows, n1)
    xd, w0d, w1d = x.astype(np.float64), w0.astype(np.float64), w1.astype(np.float64)
    h = np.maximum(xd @ w0d.T + b0, 0.0)
    want = h @ w1d.T + b1
    if relu1:
        want = np.maximum(want, 0.0)
    eps = 2e-6 if terms == 6 else 4e-5
    s0 = np.abs(xd) @ np.abs(w0d).T + np.abs(b0)
    bound = eps * (s0 @ np.abs(w1d).T) + eps * (h @ np.abs(w1d).T + np.abs(b1)) + 1e-30
    assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())
    assert not np.array_equal(got.astype(np.float32), ref32), "the split chain did not run"
    if terms == 6:
        np.testing.assert_allclose(got, ref32, atol=mlp_tol(ref32), rtol=0)


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("terms", [6, 3])
def test_split_bf16_hoisted_fp0_chain(dev, monkeypatch, terms):
    """hoisted FP0 (rows relu(interp(Y) + b0) through one 128 -> 128 layer) on the split chain kernel, B = 8 frames so that the
    XCD-aware tile order is exercised: against float64 on the SAME interpolated rows (the prologue's fp32 arithmetic is the fp32
    chain's: the row values are taken from a float64-free restatement in torch) and against the fp32 chain kernel"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(40 + terms)
    B, n, m = 8, 4096, 1024
    y = T(r.normal(size=(B, m, 128)).astype(np.float32), dev)
    idx3 = T(r.integers(0, m, size=(B, n, 3)).astype(np.int32), dev)
    w3 = r.random(size=(B, n, 3)).astype(np.float32)
    w3 = T(w3 / w3.sum(-1, keepdims=True), dev)
    b0 = T(r.normal(size=(128,)).astype(np.float32), dev)
    w1 = (r.normal(size=(128, 128)) * 0.1).astype(np.float32)
    b1 = r.normal(size=(128,)).astype(np.float32)
    l1 = lin(dev, w1, b1, True)
    ref32 = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", terms)
    got = ops.mlp_chain_interp(y, idx3, w3, None, [l1], act_bias=b0).cpu().numpy().astype(np.float64)
    # the interpolated, activated rows in fp32 (same expression order as interp1: (w0 a + w1 b) + w2 c), then float64 for the layer
    g = torch.gather(y.unsqueeze(1).expand(B, 3, m, 128), 2, idx3.long().permute(0, 2, 1).unsqueeze(-1).expand(B, 3, n, 128))
    rows = torch.relu((w3[..., 0:1] * g[:, 0] + w3[..., 1:2] * g[:, 1]) + w3[..., 2:3] * g[:, 2] + b0).reshape(-1, 128).cpu().numpy().astype(np.float64)
    want = np.maximum(rows @ w1.astype(np.float64).T + b1, 0.0)
    scale = np.abs(rows) @ np.abs(w1).astype(np.float64).T + np.abs(b1)
    # (the rows themselves may differ from the kernel's by an fp32 rounding of the interpolation: one more 1e-7-class term)
    bound = ((2e-6 if terms == 6 else 4e-5) + 4e-7) * scale + 1e-30
    assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())
    assert not np.array_equal(got.astype(np.float32), ref32), "the split chain did not run"
    if terms == 6:
        np.testing.assert_allclose(got, ref32, atol=mlp_tol(ref32), rtol=0)


@pytest.mark.own_arithmetic
def test_split_bf16_rows_with_device_row_count_and_live_segments(dev, monkeypatch):
    """the split layer kernel under a device-side row count and under segment-prefix live rows (the RCNN stage's launches): the same
    tiles are written as by the fp32 kernel, to the fp32 contract; everything else keeps the caller's bytes"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(91)
    rows, K, N = 64 * 512, 128, 128
    x = T(r.normal(size=(rows, K)).astype(np.float32), dev)
    l = lin(dev, (r.normal(size=(N, K)) * 0.1).astype(np.float32), r.normal(size=(N,)).astype(np.float32), True)
    cnt = T(r.integers(1, 513, size=(64,)).astype(np.int32), dev)
    live = T(np.array([100], np.int32), dev)

    def run(**kw):
        out = torch.full((rows, N), -7.0, device=dev)
        ops.mlp_rows(x, l, out=(out, 0), **kw)
        return out.cpu().numpy()
    ref_seg, ref_dev = run(seg=(cnt, 512)), run(rows_dev=live, rows_unit=128)
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got_seg, got_dev = run(seg=(cnt, 512)), run(rows_dev=live, rows_unit=128)
    for got, ref in ((got_seg, ref_seg), (got_dev, ref_dev)):
        assert np.array_equal(got == -7.0, ref == -7.0)                  # the same rows written
        np.testing.assert_allclose(got, ref, atol=mlp_tol(ref[ref != -7.0]), rtol=0)
        assert not np.array_equal(got, ref), "the split kernel did not run"
    assert (got_dev[12800:] == -7.0).all() and (got_dev[:12800] != -7.0).any()


def _same_class(got, ref):
    """the non-finite entries agree: NaN where NaN, +inf where +inf, -inf where -inf"""
    return (np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isposinf(got), np.isposinf(ref))
            and np.array_equal(np.isneginf(got), np.isneginf(ref)))


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("Nout,relu,pool", [(256, False, 0), (64, True, 0), (128, False, 16)])
def test_split_bf16_layer_nonfinite_rows_are_the_fp32_kernels_rows(dev, monkeypatch, Nout, relu, pool):
    """The split of an infinity is (inf, NaN, NaN): left alone it would turn a row that the fp32 kernels return as +-inf into NaN
    (and inf x 0 pieces into NaN where fp32 has inf).  A wave that sees a non-finite accumulator redoes its 64-row block with fp32
    MFMAs in mlp_layer_b_kernel's k order: those blocks are the fp32 kernel's BITS, every other block stays a split result within
    the contract.  Rows with +inf, -inf, NaN, +inf and -inf together, and a finite product that overflows."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(500 + Nout)
    rows, K = 24576 + 64, 128
    a = r.normal(size=(rows, K)).astype(np.float32)
    w = (r.normal(size=(Nout, K)) * 0.2).astype(np.float32)
    w[3, :] = np.abs(w[3, :])                                 # one all-positive weight row: +inf and -inf in a row give NaN there too
    w[5, 7] = 0.0                                             # inf x 0 = NaN in the fp32 product as well
    w[9, 9] = w[9, 10] = 1.0                                  # 2e38 + 2e38 overflows in column 9 of row 20000 (finite operands)
    b = r.normal(size=(Nout,)).astype(np.float32)
    bad = {3: [(5, np.inf)], 700: [(0, -np.inf)], 5000: [(127, np.nan)], 9000: [(1, np.inf), (2, -np.inf)], 12000: [(7, np.inf)],
           20000: [(9, 2e38), (10, 2e38)], rows - 1: [(64, np.inf)]}
    for row, ents in bad.items():
        for c, v in ents:
            a[row, c] = v
    l = lin(dev, w, b, relu)
    ref = ops.mlp_rows(T(a, dev), l, pool_ns=pool).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_rows(T(a, dev), l, pool_ns=pool).cpu().numpy()
    assert np.isinf(ref).any() and (relu or pool or np.isnan(ref).any())
    assert _same_class(got, ref)
    unit = 64 // pool if pool else 64                         # output rows of a wave's 64-row block
    redone, partly = np.zeros(got.shape[0], bool), np.zeros(got.shape[0], bool)
    for row in bad:
        q = row // 64
        # (the overflowing row makes ONE column non-finite: only the wave that owns that column block redoes its part of the rows)
        (partly if row == 20000 else redone)[q * unit:(q + 1) * unit] = True
    assert np.array_equal(got[redone], ref[redone], equal_nan=True), "a redone block is not the fp32 kernel's bits"
    fin = np.isfinite(ref[partly])
    assert np.abs(got[partly][fin] - ref[partly][fin]).max() <= mlp_tol(ref[partly][fin])
    rest_g, rest_r = got[~(redone | partly)], ref[~(redone | partly)]
    assert np.isfinite(rest_r).all()
    np.testing.assert_allclose(rest_g, rest_r, atol=mlp_tol(rest_r), rtol=0)
    assert not np.array_equal(rest_g, rest_r), "the split kernel did not run"


@pytest.mark.own_arithmetic
def test_split_bf16_addinterp_nonfinite_rows(dev, monkeypatch):
    """hoisted FP first layer: non-finite skip features (the matrix product) and a non-finite interpolated addend row (added in
    the shared fp32 epilogue): the same entries are NaN / +inf / -inf as in the fp32 kernel, finite entries within the contract"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(611)
    B, n, m, C1, Nout = 2, 8192, 2048, 96, 256
    skip = r.normal(size=(B, n, C1)).astype(np.float32)
    skip[0, 17, 3] = np.inf
    skip[1, 4000, 95] = -np.inf
    skip[1, 8191, 0] = np.nan
    y = r.normal(size=(B, m, Nout)).astype(np.float32)
    y[0, 5, 100] = np.inf
    idx3 = r.integers(0, m, size=(B, n, 3)).astype(np.int32)
    idx3[0, 300] = (5, 6, 7)
    w3 = r.random(size=(B, n, 3)).astype(np.float32)
    w3 = w3 / w3.sum(-1, keepdims=True)
    l = lin(dev, (r.normal(size=(Nout, C1)) * 0.2).astype(np.float32), r.normal(size=(Nout,)).astype(np.float32), False)
    args = (T(skip, dev), l, T(y, dev), T(idx3, dev), T(w3, dev))
    ref = ops.mlp_rows_addinterp(*args).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_rows_addinterp(*args).cpu().numpy()
    assert np.isinf(ref).any() and np.isnan(ref).any() and _same_class(got, ref)
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= mlp_tol(ref[fin])


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("n1,relu1", [(76, False), (1, False), (128, True)])
def test_split_bf16_chain_nonfinite_rows(dev, monkeypatch, n1, relu1):
    """the heads' chain with +inf / -inf / NaN among the input rows: the wave redoes its 32 rows on the fp32 pipe (the fp32 chain's
    products in its k order) -- the same entries are non-finite as in the fp32 chain, their finite neighbours and every other tile
    within the contract.  (After the first layer's ReLU an infinity survives, a NaN does not: max(NaN, 0) = 0 in both kernels.)"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(700 + n1)
    rows = 40000 + 17
    x = r.normal(size=(rows, 128)).astype(np.float32)
    bad = {0: (3, np.inf), 33: (127, -np.inf), 4100: (64, np.nan), 20000: (1, np.inf), rows - 1: (5, -np.inf)}
    for row, (c, v) in bad.items():
        x[row, c] = v
    w0 = (r.normal(size=(128, 128)) * 0.1).astype(np.float32)
    w1 = (r.normal(size=(n1, 128)) * 0.1).astype(np.float32)
    layers = [lin(dev, w0, r.normal(size=(128,)).astype(np.float32), True), lin(dev, w1, r.normal(size=(n1,)).astype(np.float32), relu1)]
    ref = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy()
    # (two ReLU layers: +inf hidden units of both signs of weight sum to NaN, which the output ReLU returns as 0 -- all finite)
    assert (relu1 or (~np.isfinite(ref)).any()) and _same_class(got, ref)
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= mlp_tol(ref[fin])
    clean = np.ones(rows, bool)
    for row in bad:
        clean[(row // 32) * 32:(row // 32 + 1) * 32] = False
    assert not np.array_equal(got[clean], ref[clean]), "the split chain did not run"


@pytest.mark.own_arithmetic
@pytest.mark.parametrize("ns,Nout,pool", [(16, 128, True), (32, 256, True), (64, 64, True), (32, 96, False)])
def test_split_bf16_hoisted_group_layer(dev, cpu, monkeypatch, ns, Nout, pool):
    """prcnn_mlp_group_split (round 5): the hoisted grouped layer -- rows relu(Z[idx] + W_x.dxyz + b) activated on their way into the bf16
    split -- against the fp32 layer kernel on the same inputs (contract 1e-5 of the scale, and provably the other kernel), pooled over 16 /
    32 / 64 samples and unpooled, with a device-side group count smaller than the launch's capacity (the bounded-grid form's row count),
    and with a +inf in one source row of Z: the groups that gather it come out as the fp32 kernel's rows (same non-finite entries)."""
    from pointrcnn_amd import ops
    r = np.random.default_rng(ns + Nout)
    B, N, M, C = 3, 700, 150, 64
    xyz = unit_cloud(B, N, seed=ns)
    new_xyz = xyz[:, :M].copy()
    idx = cpu.ball_query(0.25, ns, xyz, new_xyz)
    z = (r.normal(size=(B, N, C)) * 0.7).astype(np.float32)
    wx = (r.normal(size=(C, 3)) * 0.5).astype(np.float32)
    b0 = r.normal(size=(C,)).astype(np.float32)
    w1 = (r.normal(size=(Nout, C)) * 0.15).astype(np.float32)
    b1 = r.normal(size=(Nout,)).astype(np.float32)
    layer = lin(dev, w1, b1, True)
    act = (T(wx, dev), T(b0, dev))
    live = torch.tensor([B * M - 37], dtype=torch.int32, device=dev)          # groups actually present (device-side count)
    args = (T(xyz, dev), T(new_xyz, dev), T(idx, dev))

    def run(zz, terms, groups_dev=None):
        monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", terms)
        return ops.mlp_group(*args, T(zz, dev), layer, pool_ns=ns if pool else 0, act=act, groups_dev=groups_dev).cpu().numpy()

    ref, got = run(z, 0), run(z, 6)
    assert got.shape == ref.shape == ((B * M if pool else B * M * ns), Nout)
    np.testing.assert_allclose(got, ref, atol=mlp_tol(ref), rtol=0)
    assert not np.array_equal(got, ref), "the split kernel did not run"
    # device-side group count: the live groups' rows are the full launch's rows, bit for bit
    part = run(z, 6, live)
    nlive = (B * M - 37) * (1 if pool else ns)
    assert np.array_equal(part[:nlive], got[:nlive])
    # a non-finite source row
    zb = z.copy()
    zb[1, int(idx[1, 5, 0]), 7] = np.inf
    refb, gotb = run(zb, 0), run(zb, 6)
    assert not np.isfinite(refb).all() and _same_class(gotb, refb)
    fin = np.isfinite(refb)
    assert np.abs(gotb[fin] - refb[fin]).max() <= mlp_tol(refb[fin])


@pytest.mark.own_arithmetic
def test_split_bf16_chain_layer1_overflow_is_the_fp32_chains_infinity(dev, monkeypatch):
    """finite inputs, finite hidden units, but a layer-1 product that overflows fp32: the fp32 chain returns +-inf there; the split
    of the overflowing accumulation holds NaN pieces, so the wave must notice it on the layer-1 accumulators too (round-4 advisor
    finding: only layer 0 was tested) and redo its rows on the fp32 pipe -- same non-finite entries, same class, as the fp32 chain"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(77)
    rows = 4096 + 5
    x = np.abs(r.normal(size=(rows, 128))).astype(np.float32)
    x[70] *= 1e18; x[3000] *= 1e18                                       # hidden units ~1e19 (finite), times weights ~1e20: overflow in layer 1
    w0 = np.abs(r.normal(size=(128, 128)) * 0.1).astype(np.float32)
    w1 = (r.normal(size=(76, 128)) * 1e20).astype(np.float32)
    w1[:38] = np.abs(w1[:38])                                            # these outputs overflow to +inf, the mixed-sign ones to NaN
    layers = [lin(dev, w0, np.zeros(128, np.float32), True), lin(dev, w1, np.zeros(76, np.float32), False)]
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 0)
    ref = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_chain_rows(T(x, dev), layers).cpu().numpy()
    assert np.isinf(ref[70]).any() and np.isinf(ref[3000]).any()
    assert _same_class(got, ref)
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= mlp_tol(ref[fin])


@pytest.mark.own_arithmetic
def test_split_bf16_hoisted_fp0_nonfinite_rows(dev, monkeypatch):
    """hoisted FP0 with an infinite known-point row: every row interpolating from it is relu(inf + b) = inf going into the layer"""
    from pointrcnn_amd import ops
    r = np.random.default_rng(801)
    B, n, m = 8, 4096, 1024
    y = r.normal(size=(B, m, 128)).astype(np.float32)
    y[2, 10, 7] = np.inf
    y[5, 1000, 100] = -np.inf                              # relu(-inf + b) = 0: finite
    idx3 = r.integers(0, m, size=(B, n, 3)).astype(np.int32)
    idx3[2, 50] = (10, 11, 12)
    idx3[5, 60] = (1000, 3, 4)
    w3 = r.random(size=(B, n, 3)).astype(np.float32)
    w3 = w3 / w3.sum(-1, keepdims=True)
    b0 = T(r.normal(size=(128,)).astype(np.float32), dev)
    l1 = lin(dev, (r.normal(size=(128, 128)) * 0.1).astype(np.float32), r.normal(size=(128,)).astype(np.float32), False)
    args = (T(y, dev), T(idx3, dev), T(w3, dev), None, [l1])
    ref = ops.mlp_chain_interp(*args, act_bias=b0).cpu().numpy()
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    got = ops.mlp_chain_interp(*args, act_bias=b0).cpu().numpy()
    assert np.isinf(ref).any() and _same_class(got, ref)
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= mlp_tol(ref[fin])


@pytest.mark.own_arithmetic
def test_split_chain_forms_are_bit_identical(dev, monkeypatch):
    """round 6: the split-bf16 chain runs in three forms -- lane-is-a-row loads (mlp_chain_s_kernel, PRCNN_CHAIN_COOP=0), cooperative
    row access through an LDS transposition tile (mlp_chain_c_kernel, PRCNN_CHAIN_PERSIST=0; PRCNN_CHAIN_COOP=2 also for the two-layer heads) and the persistent weights-resident form
    (mlp_chain_p_kernel: hoisted FP0 and the single-channel head).  Same products in the same order: the outputs must be the same
    BITS, on ragged row counts (last tile partly empty, fewer tiles than waves), frames a multiple of 8 (XCD-aware tile order) and
    not, and with a non-finite input row (every form redoes the wave's rows on the fp32 pipe)."""
    from pointrcnn_amd import ops
    monkeypatch.setattr(ops, "MLP_SPLIT_TERMS", 6)
    r = np.random.default_rng(77)
    w0 = (r.normal(size=(128, 128)) * 0.1).astype(np.float32)
    b0 = r.normal(size=(128,)).astype(np.float32)
    l0 = lin(dev, w0, b0, True)
    heads = [lin(dev, (r.normal(size=(n1, 128)) * 0.1).astype(np.float32), r.normal(size=(n1,)).astype(np.float32), False) for n1 in (1, 76, 128)]

    def forms(fn):
        outs = []
        for coop, persist in (("0", "0"), ("1", "0"), ("2", "0"), ("1", "1")):
            monkeypatch.setenv("PRCNN_CHAIN_COOP", coop)
            monkeypatch.setenv("PRCNN_CHAIN_PERSIST", persist)
            outs.append(fn().clone())
        monkeypatch.delenv("PRCNN_CHAIN_COOP")
        monkeypatch.delenv("PRCNN_CHAIN_PERSIST")
        return outs

    for rows in (40000 + 17, 300, 31, 128 * 2048):
        x = _scaled_rows(r, rows, 128)
        if rows > 1000:
            x[1234, 5] = np.inf                               # one non-finite row: its wave takes the fp32 redo path in every form
        xt = T(x, dev)
        for l1 in heads:
            outs = forms(lambda: ops.mlp_chain_rows(xt, [l0, l1]))
            assert all(torch.equal(outs[0].view(torch.int32), o.view(torch.int32)) for o in outs[1:]), (rows, l1.nout)
    for B, n, m in ((8, 4096, 1024), (3, 1000, 250), (16, 256, 64), (1, 40, 7)):
        y = T(r.normal(size=(B, m, 128)).astype(np.float32), dev)
        idx3 = T(r.integers(0, m, size=(B, n, 3)).astype(np.int32), dev)
        w3 = r.random(size=(B, n, 3)).astype(np.float32)
        w3 = T(w3 / w3.sum(-1, keepdims=True), dev)
        bb = T(r.normal(size=(128,)).astype(np.float32), dev)
        outs = forms(lambda: ops.mlp_chain_interp(y, idx3, w3, None, [l0], act_bias=bb))
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (B, n, m)
        # strided output: the chain writes into a column window of a wider buffer
        wide = [torch.zeros((B * n, 160), device=dev) for _ in range(4)]
        k = [0]

        def into():
            o = wide[k[0]]
            k[0] += 1
            ops.mlp_chain_interp(y, idx3, w3, None, [l0], out=(o, 32), act_bias=bb)
            return o
        outs = forms(into)
        assert all(torch.equal(outs[0], o) for o in outs[1:]) and float(outs[0][:, :32].abs().max()) == 0.0
