"""`pointnet2_lib.pointnet2.pointnet2_modules` -- PointnetSAModuleMSG / PointnetSAModule / PointnetFPModule with
the constructor and forward signatures the reference uses (lib/net/pointnet2_msg.py:27-34,44,61,66-68;
lib/net/rcnn_net.py:33-41,180) [UPSTREAM module, absent from the tree; semantics per SURVEY.md Appendix A].

Three execution paths with identical results:
  * composed path (autograd, anything the kernels do not cover): FPS -> gather -> ball_query -> grouping -> SharedMLP ->
    max_pool, every point op a HIP kernel behind an autograd Function (pointnet2_utils), convolutions / BatchNorm in torch;
  * fused TRAINING path (autograd on, BatchNorm in training mode; pointrcnn_amd/train_mlp.py, csrc/mlp_train.h): the whole
    SharedMLP of a scale -- gather, 1x1 convs on the MFMA pipe, batch statistics, ReLU, max-pool with arg-max -- and its
    backward (BatchNorm reduction, dgrad, wgrad, scatter through the gather) are hand-written kernels on channels-last
    rows; only each layer's pre-normalisation output is kept for backward.  PRCNN_TRAIN_FUSED=0 is the A/B switch;
  * fused inference path (no autograd, BN in eval mode): features stay channels-last, the two radii of an MSG
    level are queried in one scan, grouping / 3-NN interpolation is folded into the first MLP layer's MFMA
    A-tile, max-pool into the last layer's epilogue, and both scales write straight into the concatenated
    output -- no (B,C+3,npoint,nsample) tensor, no cat, no transposes.  Outputs are returned as (B,C,N)
    VIEWS of channels-last buffers so the next module picks them up without a copy.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from pointrcnn_amd import ops, train_mlp
from . import pointnet2_utils
from . import pytorch_utils as pt_utils

_POOL_FUSED = (16, 32, 64)
# exact-algebra optimisation of the fused inference path (see _forward_fused); switchable for A/B tests
HOIST_FIRST_LAYER = True
GROUP_DEDUP = os.environ.get("PRCNN_GROUP_DEDUP", "1") != "0"       # padding-free grouping (csrc/dedup.hip); 0 = A/B switch, same bits
DEDUP_SPARSE_DIV = int(os.environ.get("PRCNN_DEDUP_SPARSE_DIV", "4"))  # groups with <= nsample/DIV hits run as flat rows
STACK_ALL_FLAT = os.environ.get("PRCNN_STACK_ALL_FLAT", "1") != "0"      # A/B switch: stack-kernel scales keep a dense list when off
STACK_ALL_FLAT_MAX_ROWS = 1 << 20
# largest layer-to-layer intermediate of a padding-free scale's dense list allocated at once (see _run_scale); 0 = never slice (A/B)
DENSE_CHUNK_BYTES = int(float(os.environ.get("PRCNN_DENSE_CHUNK_MB", "2048")) * 2 ** 20) or (1 << 62)
TRAIN_FUSED = os.environ.get("PRCNN_TRAIN_FUSED", "1") != "0"          # hand-written training-mode SharedMLP (train_mlp.py)
TRAIN_DEDUP = os.environ.get("PRCNN_TRAIN_DEDUP", "1") != "0"          # ... on padding-free rows (exact; A/B switch)


def _channels_last(features):
    """(B,C,N) -> (B,N,C) tensor with unit last stride and uniform row stride (view when possible)."""
    if features is None:
        return None
    return pt_utils._rows_view(features.transpose(1, 2))


def _run_mlp_tail(x_rows, layers, start, out, pool_ns):
    """layers[start:] applied to channels-last rows; the last layer pools/stores into `out`."""
    n = len(layers)
    for li in range(start, n):
        last = li == n - 1
        x_rows = ops.mlp_rows(x_rows, layers[li], out=out if last else None, pool_ns=pool_ns if last else 0)
    return x_rows


def _flatten_frames(x):
    """(B, N, C) rows with a uniform row stride -> (1, B*N, C) view (frames become one long cloud; indices b*N + p)"""
    B, N, C = x.shape
    if C == 1 and x.stride(2) != 1:       # a size-1 dimension carries whatever stride its history left it (a (B, 1, N) intensity plane)
        x = x.as_strided((B, N, 1), (x.stride(0), x.stride(1), 1), x.storage_offset())
    assert x.stride(2) == 1 and (B == 1 or x.stride(0) == N * x.stride(1)), "rows must have one uniform stride"
    return x.as_strided((1, B * N, C), (B * N * x.stride(1), x.stride(1), 1), x.storage_offset())


def _stack_scale(layers, act, src):
    """True when the flat (nsample 1, un-pooled) row list of this scale runs on the two-layer stack kernel"""
    return (any(l.nout > 128 for l in layers) and act is not None and src is not None and src.shape[-1] <= 256
            and src.shape[-1] % 8 == 0 and ops.chain_supported(1, layers, 0))


class _SliceCount:
    """what bench.py's instrumented pass needs of a slice's device-side count: the tensor, kept alive until it is read"""
    slice_count = True

    def __init__(self, counts):
        self.counts = counts


def _run_scale(xyz, ctr, idx, src, layers, act, out, ns, pool, groups_dev):
    """One grouping scale: gather + SharedMLP (+ max-pool over the nsample rows when `pool`), the register-resident chain
    kernel where an instance exists, LDS-tiled layer kernels otherwise.  groups_dev: device-side group count (dedup)."""
    pool_ns = ns if pool else 0
    # layers wider than 128 channels: only the two-layer stack kernel takes them (hoisted form on a flat, un-pooled row list)
    wide = any(l.nout > 128 for l in layers)
    if (ns == 1 or (pool and ns in (16, 32))) and (not wide or (ns == 1 and _stack_scale(layers, act, src))) \
            and ops.chain_supported(1, layers, pool_ns):
        return ops.mlp_chain_group(xyz, ctr, idx, src, layers, out=out, pool_ns=pool_ns, act=act, groups_dev=groups_dev)
    if len(layers) == 1:
        return ops.mlp_group(xyz, ctr, idx, src, layers[0], out=out, pool_ns=pool_ns, act=act, groups_dev=groups_dev)
    n = len(layers)
    G = idx.shape[1] if idx.shape[0] == 1 else 0
    inter_bytes = idx.numel() * max(l.nout for l in layers[:-1]) * 4
    if groups_dev is not None and (pool or ns == 1) and out is not None and G > 1 and inter_bytes > DENSE_CHUNK_BYTES:
        # The dense list of a padding-free scale has a device-side length, but the layer-to-layer intermediate is allocated for the
        # host-side bound: every group dense, G x nsample rows (the RCNN stage's first level: 409 600 groups x 64 x 128 floats =
        # 13.4 GB, for a list that holds 0-200 groups on the benchmark clouds) -- 20 GB per in-flight batch capped the two-stage
        # pipeline at 10 slots.  The list is walked in slices of at most DENSE_CHUNK_BYTES of intermediate instead: slice k covers
        # groups [g0, g1) and runs with the device-side count clamp(count - g0, 0, g1 - g0), so slices beyond the list's end are
        # launches of zero rows.  The flat list of the sparse groups (nsample 1, un-pooled: a "group" is a row) is walked the same way.
        # Same rows through the same kernels: same bits (tests/test_gpu_round2.py, PRCNN_DENSE_CHUNK_MB).
        nchunk = -(-inter_bytes // DENSE_CHUNK_BYTES)
        gc = -(-G // nchunk)
        dst, col = out
        for g0 in range(0, G, gc):
            g1 = min(G, g0 + gc)
            cnt = (groups_dev - g0).clamp_(0, g1 - g0)
            if ops._split_log is not None:                    # (bench.py's instrumented pass reads the device-side counts afterwards)
                ops._split_log.append(_SliceCount(cnt))
            x = ops.mlp_group(xyz, None if ctr is None else ctr[:, g0:g1], idx[:, g0:g1], src, layers[0], act=act, groups_dev=cnt)
            for li in range(1, n):
                last = li == n - 1
                x = ops.mlp_rows(x, layers[li], out=(dst[g0:g1], col) if last else None, pool_ns=pool_ns if last else 0, rows_dev=cnt, rows_unit=ns)
            del x
        return dst
    x = ops.mlp_group(xyz, ctr, idx, src, layers[0], act=act, groups_dev=groups_dev)
    for li in range(1, n):
        last = li == n - 1
        x = ops.mlp_rows(x, layers[li], out=out if last else None, pool_ns=pool_ns if last else 0, rows_dev=groups_dev, rows_unit=ns)
    return x


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = "max_pool"

    def _fused_ok(self, xyz, features):
        if torch.is_grad_enabled() or not xyz.is_cuda or self.pool_method != "max_pool":
            return False
        if xyz.dtype != torch.float32 or (features is not None and features.dtype != torch.float32):
            return False
        for g, m in zip(self.groupers, self.mlps):
            if not getattr(g, "use_xyz", True) or not m.fusable():
                return False
        return True

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3) or None, new_features (B,sum C_out,npoint)"""
        if self._fused_ok(xyz, features):
            return self._forward_fused(xyz, features, new_xyz)
        if self._train_ok(xyz, features):
            return self._forward_train(xyz, features, new_xyz)
        new_features_list = []
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if new_xyz is None:
            new_xyz = pointnet2_utils.gather_operation(
                xyz_flipped, pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            ).transpose(1, 2).contiguous() if self.npoint is not None else None
        for i in range(len(self.groupers)):
            new_features = self.groupers[i](xyz, new_xyz, features)         # (B,C,npoint,nsample)
            new_features = self.mlps[i](new_features)
            if self.pool_method == "max_pool":
                # == F.max_pool2d(new_features, kernel_size=[1, nsample]) (the upstream module's call), forward and backward:
                # both take the FIRST maximal entry of a row, so the gradient goes to the same element.  The reduction over the
                # contiguous last dimension is several times faster than the generic NCHW pooling kernel, which was 8 % + 3 %
                # (forward + backward) of the RPN training step.
                new_features = new_features.max(dim=3, keepdim=True)[0]
            elif self.pool_method == "avg_pool":
                new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
            else:
                raise NotImplementedError
            new_features_list.append(new_features.squeeze(-1))
        return new_xyz, torch.cat(new_features_list, dim=1)

    # ---- fused training path ------------------------------------------------------------------
    def _train_ok(self, xyz, features):
        if not TRAIN_FUSED or not train_mlp.usable(xyz, features) or self.pool_method != "max_pool":
            return False
        if self.npoint is None:         # GroupAll (the RCNN stage's last level): one group of all N points, no centring
            g = self.groupers[0]
            return len(self.groupers) == 1 and isinstance(g, pointnet2_utils.GroupAll) and g.use_xyz and features is not None \
                and xyz.shape[1] <= 255 and train_mlp.stack_ok(self.mlps[0].layers())
        for g, m in zip(self.groupers, self.mlps):
            if not isinstance(g, pointnet2_utils.QueryAndGroup) or not g.use_xyz or g.nsample > 255 or not train_mlp.stack_ok(m.layers()):
                return False
        return True

    def _forward_train(self, xyz, features, new_xyz):
        """sampling and neighbour search carry no gradient; the grouped SharedMLP + max-pool of every scale is ONE autograd
        node over the channels-last features (train_mlp.SharedMLPTrain)"""
        xyz = xyz.contiguous()
        B = xyz.shape[0]
        if self.npoint is None:
            # GroupAll: rows [feat | xyz - 0] of every point, one group per frame (pointnet2_utils.GroupAll: xyz channels first, no
            # centring), max over the N points
            N = xyz.shape[1]
            idx = torch.arange(N, dtype=torch.int32, device=xyz.device).view(1, 1, N).expand(B, 1, N).contiguous()
            src = train_mlp.Source("group", xyz=xyz, new_xyz=torch.zeros((B, 1, 3), dtype=xyz.dtype, device=xyz.device), idx=idx)
            out = train_mlp.run_stack(self.mlps[0].layers(), src, _channels_last(features), None, pool_ns=N)
            return None, out.view(B, 1, -1).transpose(1, 2)
        with torch.no_grad():
            if new_xyz is None:
                new_xyz = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, self.npoint))
            new_xyz = new_xyz.contiguous()
            if len(self.groupers) == 2:
                a, b = self.groupers
                idxs = list(ops.ball_query2(a.radius, a.nsample, b.radius, b.nsample, xyz, new_xyz))
            else:
                idxs = [ops.ball_query(g.radius, g.nsample, xyz, new_xyz) for g in self.groupers]
        feat_cl = _channels_last(features)
        M = new_xyz.shape[1]
        outs = []
        for g, mlp, idx in zip(self.groupers, self.mlps, idxs):
            if TRAIN_DEDUP:
                # padding-free rows: the distinct rows of every group + multiplicities instead of the nsample-padded rows
                with torch.no_grad():
                    rows = train_mlp.GroupRows(idx, new_xyz, xyz.shape[1])
                src = train_mlp.Source("group", xyz=xyz.view(1, -1, 3), rows=rows)
            else:
                src = train_mlp.Source("group", xyz=xyz, new_xyz=new_xyz, idx=idx)
            outs.append(train_mlp.run_stack(mlp.layers(), src, feat_cl, None, pool_ns=g.nsample).view(B, M, -1))
        out_cl = outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)
        return new_xyz, out_cl.transpose(1, 2)

    # ---- fused inference path -----------------------------------------------------------------
    def _forward_fused(self, xyz, features, new_xyz):
        # rcnn.py tags a roipool3d-padded cloud with the number of distinct points per frame (the rest are wrap-copies whose
        # feature rows were never computed): only usable when every scale takes the padding-free path below
        valid_n = getattr(xyz, "_prcnn_valid_n", None)
        xyz = xyz.contiguous()
        B, N, _ = xyz.shape
        feat_cl = _channels_last(features)
        if new_xyz is None and self.npoint is not None:
            new_xyz = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, self.npoint))
        M = new_xyz.shape[1] if new_xyz is not None else 1
        c_outs = [m.layers()[-1]._parts()[0].out_channels for m in self.mlps]
        out_cl = torch.empty((B, M, sum(c_outs)), dtype=torch.float32, device=xyz.device)

        # neighbour search: both radii of an MSG level in one scan
        idxs = [None] * len(self.groupers)
        qg = [i for i, g in enumerate(self.groupers) if isinstance(g, pointnet2_utils.QueryAndGroup)]
        if len(qg) == 2 and new_xyz is not None:
            a, b = self.groupers[qg[0]], self.groupers[qg[1]]
            idxs[qg[0]], idxs[qg[1]] = ops.ball_query2(a.radius, a.nsample, b.radius, b.nsample, xyz, new_xyz)
        for i, g in enumerate(self.groupers):
            if idxs[i] is not None:
                continue
            if isinstance(g, pointnet2_utils.QueryAndGroup):
                idxs[i] = ops.ball_query(g.radius, g.nsample, xyz, new_xyz)
            else:                                   # GroupAll: one group holding every point, no centring
                idxs[i] = torch.arange(N, dtype=torch.int32, device=xyz.device).view(1, 1, N).expand(B, 1, N).contiguous()

        # First-layer hoisting: the first conv of every scale is linear and grouping is a linear gather, so
        # W.[dxyz; feat[idx]] = W_x.dxyz + (W_f.feat)[idx].  Z = W_f.feat is computed ONCE per source point for all
        # scales in a single GEMM (N rows instead of M*nsample rows: 4-8x fewer), the grouped rows then enter the
        # second layer as relu(Z[idx] + W_x.dxyz + b).  Same result up to fp32 reassociation.
        hoist = None
        if feat_cl is not None and HOIST_FIRST_LAYER:
            parts = [mlp.layers()[0].hoisted_group() if len(mlp.layers()) >= 2 else None for mlp in self.mlps]
            if all(p is not None for p in parts):
                key = tuple(id(p[0]) for p in parts)
                if getattr(self, "_prcnn_zlin", (None, None))[0] != key:
                    self._prcnn_zlin = (key, ops.PackedLinear(torch.cat([p[0] for p in parts], 0).contiguous(), None, relu=False))
                z_all = ops.mlp_rows(feat_cl, self._prcnn_zlin[1], seg=None if valid_n is None else (valid_n, N)).view(B, N, -1)
                hoist, zoff = [], 0
                for p in parts:
                    hoist.append((z_all[..., zoff:zoff + p[0].shape[0]], (p[1], p[2])))
                    zoff += p[0].shape[0]

        col = 0
        for i, (g, mlp) in enumerate(zip(self.groupers, self.mlps)):
            group_all = not isinstance(g, pointnet2_utils.QueryAndGroup)
            ns = idxs[i].shape[2]
            mods = mlp.layers()
            fused_pool = ns in _POOL_FUSED
            dst = (out_cl.view(B * M, -1), col)
            ctr = None if group_all else new_xyz
            if hoist is not None:
                src, act = hoist[i]
                layers = [m.packed() for m in mods[1:]]
            else:
                src, act = feat_cl, None
                # torch's grouped channel order is [dxyz(3), feat(C)]; the kernel's A row is [feat(C), dxyz(3)]
                layers = [mods[0].packed(k_rot=3 if feat_cl is not None else 0)] + [m.packed() for m in mods[1:]]
            if valid_n is not None and not (GROUP_DEDUP and not group_all and fused_pool):
                raise RuntimeError("a cloud tagged with _prcnn_valid_n needs the padding-free grouping path on every scale")
            if GROUP_DEDUP and not group_all and fused_pool:
                # Padding-free grouping: a group with fewer than nsample neighbours repeats its first hit, and the max
                # over copies of a row is the row -- sparse groups contribute only their real rows to one flat row list
                # (segmented max afterwards), dense groups run as they are, each list with a device-side length.
                # Same bits, far fewer rows.
                # Scales the two-layer stack kernel takes (wide SA3 / SA4 stacks in hoisted form) send EVERY group through the
                # flat list: the stack kernel carries a row through both layers whatever the list length, so the dense list
                # would buy nothing there and its three launches per scale (two layer kernels, one scatter) would run empty on
                # most clouds.  Only for short levels (the RPN's SA3 / SA4): where the padded bound runs into millions of rows
                # (the RCNN stage's 64-sample groups over 100 RoIs per frame) many groups ARE dense, and pooling them in the
                # layer kernel's epilogue beats a round trip of their rows through HBM (two-stage detector 5.1 k vs 4.4 k frames/s).
                all_flat = STACK_ALL_FLAT and B * M * ns <= STACK_ALL_FLAT_MAX_ROWS and _stack_scale(layers, act, src)
                sp = ops.GroupSplit(idxs[i], new_xyz, N, ns if all_flat else max(1, ns // DEDUP_SPARSE_DIV), valid_n=valid_n)
                xyz_f = xyz.view(1, B * N, 3)
                src_f = None if src is None else _flatten_frames(src)
                t1 = torch.empty((sp.max_rows, c_outs[i]), dtype=torch.float32, device=xyz.device)
                _run_scale(xyz_f, sp.rnx, sp.ridx, src_f, layers, act, (t1, 0), 1, False, sp.rows)
                ops.segmax_scatter(t1, sp, dst[0], col)
                del t1                                        # (worst-case sized: gone before the dense list's intermediates are allocated)
                if not all_flat:
                    tn = torch.empty((sp.G, c_outs[i]), dtype=torch.float32, device=xyz.device)
                    _run_scale(xyz_f, sp.nxn, sp.idxn, src_f, layers, act, (tn, 0), ns, True, sp.count_dense)
                    ops.scatter_rows(tn, sp.listn, sp.count_dense, dst[0], col)
            else:
                x = _run_scale(xyz, ctr, idxs[i], src, layers, act, dst if fused_pool else None, ns, fused_pool, None)
                if not fused_pool:
                    ops.maxpool_rows(x, ns, out=dst)
            col += c_outs[i]
        return new_xyz, out_cl.transpose(1, 2)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (lib/net/pointnet2_msg.py:27-34)."""

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method="max_pool",
                 instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for i in range(len(radii)):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radii[i], nsamples[i], use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            mlp_spec = list(mlps[i])
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(mlp_spec, bn=bn, instance_norm=instance_norm))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction; npoint=None groups all points (lib/net/rcnn_net.py:31-41)."""

    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True,
                 pool_method="max_pool", instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """Feature propagation: 3-NN inverse-distance interpolation + skip concat + SharedMLP
    (lib/net/pointnet2_msg.py:43-45,66-68)."""

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) or None, known_feats (B,C2,m) -> (B,mlp[-1],n)"""
        if (not torch.is_grad_enabled()) and known is not None and unknown.is_cuda and self.mlp.fusable() \
                and known_feats.dtype == torch.float32:
            return self._forward_fused(unknown, known, unknow_feats, known_feats)
        if TRAIN_FUSED and known is not None and train_mlp.usable(unknown, known_feats, unknow_feats) \
                and train_mlp.stack_ok(self.mlp.layers()):
            # fused training path: 3-NN search without gradient, interpolation + concat + SharedMLP as one autograd node
            unknown, known = unknown.contiguous(), known.contiguous()
            B, n, _ = unknown.shape
            with torch.no_grad():
                _, idx3, w3 = ops.three_nn(unknown, known, want_weight=True)
            src = train_mlp.Source("interp", idx3=idx3, w3=w3)
            x = train_mlp.run_stack(self.mlp.layers(), src, _channels_last(known_feats), _channels_last(unknow_feats))
            return x.view(B, n, -1).transpose(1, 2)
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)
        else:
            new_features = interpolated_feats
        new_features = self.mlp(new_features.unsqueeze(-1))
        return new_features.squeeze(-1)

    def _forward_fused(self, unknown, known, unknow_feats, known_feats):
        unknown, known = unknown.contiguous(), known.contiguous()
        B, n, _ = unknown.shape
        _, idx3, w3 = ops.three_nn(unknown, known, want_weight=True)
        known_cl = _channels_last(known_feats)
        skip_cl = _channels_last(unknow_feats)
        mods = self.mlp.layers()
        # First-layer hoisting: interpolation is linear, so W.[interp(x); skip] = interp(W_a.x) + W_b.skip.  W_a.x is
        # computed per KNOWN point (m rows instead of n: 4x fewer); the interpolated rows are added in the epilogue of
        # the skip GEMM, or (no skip features) enter the second layer as relu(interp(Y) + b).
        parts = mods[0].hoisted_interp(known_cl.shape[-1]) if HOIST_FIRST_LAYER else None
        if parts is not None and (skip_cl is not None or len(mods) >= 2):
            lin_a, lin_b, bias0 = parts
            y = ops.mlp_rows(known_cl, lin_a).view(B, known_cl.shape[1], -1)
            rest = [m.packed() for m in mods[1:]]
            if skip_cl is not None:
                x = ops.mlp_rows_addinterp(skip_cl, lin_b, y, idx3, w3)
                x = _run_mlp_tail(x, rest, 0, None, 0) if rest else x
            elif ops.chain_supported(2, rest, 0):
                x = ops.mlp_chain_interp(y, idx3, w3, None, rest, act_bias=bias0)
            else:
                x = ops.mlp_interp(y, idx3, w3, None, rest[0], act_bias=bias0)
                x = _run_mlp_tail(x, rest, 1, None, 0)
            return x.view(B, n, -1).transpose(1, 2)
        layers = [m.packed() for m in mods]
        if ops.chain_supported(2, layers, 0):
            x = ops.mlp_chain_interp(known_cl, idx3, w3, skip_cl, layers)
        else:
            x = ops.mlp_interp(known_cl, idx3, w3, skip_cl, layers[0])
            x = _run_mlp_tail(x, layers, 1, None, 0)
        return x.view(B, n, -1).transpose(1, 2)
