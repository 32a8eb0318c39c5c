"""Training-mode SharedMLP on the hand-written kernels of csrc/mlp_train.h (BASELINE config 4).

One autograd Function runs a whole SharedMLP stack -- every layer Conv(1x1, no bias) -> BatchNorm (batch statistics) -> ReLU,
optionally max-pooled over `nsample` -- on channels-last rows, for the three ways the reference produces those rows:

  * "group":  PointnetSAModule*  rows = [xyz[idx] - new_xyz | feat[idx]]      (pointnet2_utils.QueryAndGroup + SharedMLP +
              F.max_pool2d, upstream pointnet2_modules; call sites lib/net/pointnet2_msg.py:27-34,61)
  * "interp": PointnetFPModule   rows = [three_interpolate(known) | skip]     (lib/net/pointnet2_msg.py:43-45,66-68)
  * "plain":  pt_utils.Conv1d heads on the backbone features                   (lib/net/rpn.py:20-46)

What the reference's path keeps per layer (conv output, BatchNorm output, ReLU output, plus the grouped tensor) shrinks to
the pre-normalisation output y of each layer; normalisation + ReLU are applied where y is consumed (next layer's operand
staging, pooling, backward).  The backward pass is hand-written too: BatchNorm reduction, dgrad and wgrad MFMA kernels, and the
scatter of the first layer's row gradients back through the gather.  Running statistics are updated exactly as nn.BatchNorm
does (momentum, unbiased variance, num_batches_tracked).  No CPU / torch fallback inside: the modules decide (`stack_ok`)
whether a stack takes this path or the composed torch path.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _cabi, ops
from .ops import _p, _stream

_F32 = torch.float32
INTERP_GRAD_GATHER = os.environ.get("PRCNN_INTERP_GRAD_GATHER", "1") != "0"      # A/B switch: 0 = atomics
# bench.py's accounting: when a list, every stack forward appends (rows (int) or live-row counter (device tensor), [(K, N) per layer],
# first layer takes an input gradient) -- nothing is read back here
FLOP_LOG = None


def logged_flops(log):
    """executed GEMM FLOPs of the logged stack calls for one step: forward + weight gradient for every layer, input gradient for
    every layer but a first layer whose input takes none (2 rows K N each)"""
    total = 0.0
    for rows, dims, need_x in log:
        r = float(rows.item()) if torch.is_tensor(rows) else float(rows)
        for li, (k, n) in enumerate(dims):
            total += 2.0 * r * k * n * (3.0 if (li > 0 or need_x) else 2.0)
    return total
MODE = {"plain": 0, "group": 1, "interp": 2}


def _up(x, m):
    return (x + m - 1) // m * m


def stack_ok(mods):
    """True when every layer of a pt_utils conv stack is one of the two forms the kernels implement, on CUDA fp32:
    Conv(1x1, bias=False) -> BatchNorm(training, affine, running stats, momentum) -> ReLU, or -- the RCNN stage, USE_BN False --
    Conv(1x1, with or without bias) -> ReLU."""
    from pointnet2_lib.pointnet2 import pytorch_utils as pt
    for m in mods:
        if not isinstance(m, pt._ConvBase) or not m._prcnn_fusable:
            return False
        conv, bn, act = m._parts()
        if act is None or not isinstance(act, nn.ReLU):
            return False
        if bn is not None:
            if conv.bias is not None or not bn.training:
                return False
            if not bn.affine or not bn.track_running_stats or bn.momentum is None:
                return False
        if any(k != 1 for k in conv.kernel_size) or any(s != 1 for s in conv.stride) or any(p != 0 for p in conv.padding):
            return False
        if not conv.weight.is_cuda or conv.weight.dtype != _F32 or conv.out_channels % 4:
            return False
    return len(mods) > 0


class Source:
    """how the rows entering the first layer are produced (non-differentiable pieces; the differentiable tensors go through
    Function.apply)"""

    def __init__(self, mode, **kw):
        self.mode = mode
        self.__dict__.update(kw)


class GroupRows:
    """Padding-free rows of one grouping scale for training (prcnn_train_group_rows): ball_query pads a group with copies of its
    first hit; the distinct rows + a multiplicity per row carry the same batch statistics, maxima and gradients (exact algebra,
    csrc/mlp_train.h).  Deterministic layout (group order).  The live row count stays on the device."""

    def __init__(self, idx, new_xyz, N):
        B, M, ns = idx.shape
        dev = idx.device
        G, R = B * M, B * M * ns
        self.B, self.M, self.ns, self.N, self.G, self.max_rows = B, M, ns, N, G, R
        ibuf = torch.empty((2 * G + 1 + 2 * R,), dtype=torch.int32, device=dev)
        self.cnt, self.off, self.rows_dev = ibuf[:G], ibuf[G:2 * G], ibuf[2 * G:2 * G + 1]
        self.ridx, self.row_grp = ibuf[2 * G + 1:2 * G + 1 + R], ibuf[2 * G + 1 + R:]
        fbuf = torch.empty((4 * R,), dtype=_F32, device=dev)
        self.rnx, self.mult = fbuf[:3 * R].view(R, 3), fbuf[3 * R:]
        _cabi.check(_cabi.lib().prcnn_train_group_rows(_p(idx), _p(new_xyz), B, N, M, ns, _p(self.cnt), _p(self.off), _p(self.rows_dev),
                                                       _p(self.ridx), _p(self.rnx), _p(self.mult), _p(self.row_grp), _stream()),
                    "prcnn_train_group_rows")


def _pack(w2d, k_rot=0):
    L = _cabi.lib()
    nout, k = w2d.shape
    wp = torch.empty((L.prcnn_wpack_floats(nout, k),), dtype=_F32, device=w2d.device)
    _cabi.check(L.prcnn_pack_weight(_p(w2d), nout, k, k_rot, _p(wp), _stream()), "prcnn_pack_weight")
    return wp


def _fill_src(S, src, x0, x1):
    if src.mode == "plain":
        S.mode, S.rows, S.K = 0, x0.shape[0], x0.shape[1]
        S.in_, S.ld_in = _p(x0), x0.stride(0)
    elif src.mode == "group" and getattr(src, "rows", None) is not None:
        gr = src.rows                                  # padding-free rows: one long "frame" of distinct rows, nsample 1
        C = 0 if x0 is None else x0.shape[-1]
        S.mode, S.rows, S.K = 1, gr.max_rows, C + 3
        S.xyz, S.new_xyz, S.idx = _p(src.xyz), _p(gr.rnx), _p(gr.ridx)
        S.feat, S.ld_feat = _p(x0), (0 if x0 is None else x0.stride(1))
        S.B, S.N, S.M, S.ns, S.C = 1, gr.B * gr.N, gr.max_rows, 1, C
        S.mult, S.rows_dev, S.norm_rows = _p(gr.mult), _p(gr.rows_dev), gr.max_rows
        S.seg_off, S.seg_cnt, S.row_grp, S.groups = _p(gr.off), _p(gr.cnt), _p(gr.row_grp), gr.G
    elif src.mode == "group":
        B, M, ns = src.idx.shape
        C = 0 if x0 is None else x0.shape[-1]
        S.mode, S.rows, S.K = 1, B * M * ns, C + 3
        S.xyz, S.new_xyz, S.idx = _p(src.xyz), _p(src.new_xyz), _p(src.idx)
        S.feat, S.ld_feat = _p(x0), (0 if x0 is None else x0.stride(1))
        S.B, S.N, S.M, S.ns, S.C = B, src.xyz.shape[1], M, ns, C
    else:
        B, n, _ = src.idx3.shape
        C2 = x0.shape[-1]
        C1 = 0 if x1 is None else x1.shape[-1]
        S.mode, S.rows, S.K = 2, B * n, C2 + C1
        S.known, S.idx3, S.w3, S.skip = _p(x0), _p(src.idx3), _p(src.w3), _p(x1)
        S.ld_known, S.ld_skip = x0.stride(1), (0 if x1 is None else x1.stride(1))
        S.B, S.n, S.m, S.C2, S.C1 = B, n, x0.shape[1], C2, C1


def _rows3(t, name):
    """(B, n, C) tensor whose rows have unit channel stride and one uniform row stride (frames back to back)"""
    if t is None:
        return None
    if t.dtype != _F32 or not t.is_cuda:
        raise RuntimeError("%s must be a float32 CUDA tensor" % name)
    B, n, C = t.shape
    if t.stride(2) != 1 or (B > 1 and t.stride(0) != n * t.stride(1)) or t.stride(1) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
        if C % 4:                                     # 16-byte rows for the vector fetchers
            pad = torch.zeros((B, n, _up(C, 4)), dtype=_F32, device=t.device)
            pad[..., :C] = t
            t = pad[..., :C]
    return t


def _rows2(t):
    """(R, K) rows with unit channel stride, 16-byte aligned rows (zero-padded copy when K is not a multiple of 4)"""
    R, K = t.shape
    if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    pad = torch.zeros((R, _up(K, 4)), dtype=_F32, device=t.device)
    pad[:, :K] = t
    return pad[:, :K]


def _wpack_floats(nout, k):
    return ((nout + 31) // 32) * ((k + 7) // 8) * 256


class _Stack:
    """device buffers + the prcnn_train_layer_t array of one stack invocation (kept alive by the autograd context)"""

    def __init__(self, params, bns, rows, K0, kin0, need_x, dev):
        nl = len(bns)
        nout = [int(params[3 * l].shape[0]) for l in range(nl)]
        ks = [K0] + nout[:-1]
        kins = [kin0] + nout[:-1]
        self.nl, self.nout, self.ks, self.rows = nl, nout, ks, rows
        ld_c = [_up(n, 128) for n in nout]
        # one allocation each for: saved outputs, constant tables (zeroed), weight images, parameter gradients
        self.ybuf = torch.empty((rows * sum(nout),), dtype=_F32, device=dev)
        self.cbuf = torch.zeros((6 * sum(ld_c),), dtype=_F32, device=dev)
        wt_need = [need_x or l > 0 for l in range(nl)]
        wsz = [_wpack_floats(nout[l], ks[l]) for l in range(nl)]
        wtsz = [_wpack_floats(kins[l], nout[l]) if (wt_need[l] and kins[l] > 0) else 0 for l in range(nl)]
        self.wbuf = torch.empty((sum(wsz) + sum(wtsz),), dtype=_F32, device=dev)
        gsz = [nout[l] * ks[l] + 2 * nout[l] for l in range(nl)]
        self.gbuf = torch.empty((sum(gsz),), dtype=_F32, device=dev)
        self.layers = (_cabi.TrainLayer * nl)()
        yo = co = wo = go = 0
        self.grad_views = []
        for l in range(nl):
            W, gamma, beta = params[3 * l: 3 * l + 3]           # gamma None: no normalisation, beta = the conv bias (or None)
            bn, Ly = bns[l], self.layers[l]
            if not all(t is None or t.is_contiguous() for t in (W, gamma, beta)):
                raise RuntimeError("SharedMLP parameters must be contiguous")
            Ly.Nout, Ly.W, Ly.gamma, Ly.beta = nout[l], _p(W), _p(gamma), _p(beta)
            if bn is not None:
                Ly.eps, Ly.momentum = float(bn.eps), float(bn.momentum)
                Ly.running_mean, Ly.running_var = _p(bn.running_mean), _p(bn.running_var)
            Ly.y = self.ybuf.data_ptr() + 4 * yo
            yo += rows * nout[l]
            Ly.cst, Ly.ld_c = self.cbuf.data_ptr() + 4 * co, ld_c[l]
            co += 6 * ld_c[l]
            Ly.wpack = self.wbuf.data_ptr() + 4 * wo
            wo += wsz[l]
            Ly.wpack_t = (self.wbuf.data_ptr() + 4 * wo) if wtsz[l] else None
            wo += wtsz[l]
            n, k = nout[l], ks[l]
            dW, dg, db = self.gbuf[go: go + n * k], self.gbuf[go + n * k: go + n * k + n], self.gbuf[go + n * k + n: go + n * k + 2 * n]
            go += gsz[l]
            Ly.dW, Ly.dgamma, Ly.dbeta = _p(dW), (_p(dg) if gamma is not None else None), (_p(db) if beta is not None else None)
            self.grad_views.append((dW, dg if gamma is not None else None, db if beta is not None else None))

    def work(self, K0, backward, dev):
        nbytes = _cabi.lib().prcnn_train_stack_work_bytes(self.rows, self.layers, self.nl, K0, 1 if backward else 0)
        return torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev), nbytes


def _aligned_ptr(buf):
    return (buf.data_ptr() + 255) // 256 * 256


class SharedMLPTrain(torch.autograd.Function):
    """apply(src, bns, pool_ns, x0, x1, W_0, gamma_0, beta_0, W_1, ...) -> (groups, N_last) activated (and pooled) rows.
    x0: features (B,N,C) / known features (B,m,C2) / plain rows (R,K); x1: skip features (B,n,C1) or None.
    Forward and backward are ONE library call each (prcnn_train_stack_fwd / _bwd): the per-layer launch sequence is host C++."""

    @staticmethod
    def forward(ctx, src, bns, pool_ns, x0, x1, *params):
        L = _cabi.lib()
        nl = len(bns)
        assert len(params) == 3 * nl
        dev = params[0].device
        if src.mode != "plain":
            x0, x1 = _rows3(x0, "features"), _rows3(x1, "skip features")
        else:
            x0 = _rows2(x0)
        S = _cabi.TrainSrc()
        _fill_src(S, src, x0, x1)
        rows, K0 = int(S.rows), int(S.K)
        kin0 = K0 - 3 if src.mode == "group" else K0
        need_x = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        ns = pool_ns if pool_ns and pool_ns > 1 else 1
        flat = getattr(src, "rows", None)
        with torch.no_grad():
            st = _Stack(params, bns, rows, K0, kin0, need_x, dev)
            a_dump, ld_dump = None, 0
            if src.mode != "plain":
                ld_dump = _up(K0, 4)
                a_dump = torch.empty((rows, ld_dump), dtype=_F32, device=dev)
            groups, N = (flat.G if flat is not None else rows // ns), st.nout[-1]
            out = torch.empty((groups, N), dtype=_F32, device=dev)
            arg = torch.empty((groups, N), dtype=torch.uint8, device=dev) if (ns > 1 or flat is not None) else None
            work, nbytes = st.work(K0, False, dev)
            _cabi.check(L.prcnn_train_stack_fwd(ctypes.byref(S), st.layers, nl, ns, _p(a_dump), ld_dump, _p(out), N, 0, _p(arg),
                                                _aligned_ptr(work), nbytes, _stream()), "prcnn_train_stack_fwd")
            for bn in bns:
                if bn is not None:
                    bn.num_batches_tracked += 1
            if FLOP_LOG is not None:
                FLOP_LOG.append((flat.rows_dev if flat is not None else rows, list(zip(st.ks, st.nout)), need_x))
        ctx.src, ctx.S, ctx.st, ctx.ns, ctx.K0, ctx.kin0, ctx.rows = src, S, st, ns, K0, kin0, rows
        ctx.a_dump, ctx.ld_dump, ctx.arg = a_dump, ld_dump, arg
        ctx.x_shapes = (None if x0 is None else tuple(x0.shape), None if x1 is None else tuple(x1.shape))
        ctx.keep = (x0, x1, params)                 # the descriptor holds raw pointers into these
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _cabi.lib()
        st, src, rows = ctx.st, ctx.src, ctx.rows
        dev = gout.device
        need_x = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        G = gout                                      # rows of any 16-byte-aligned stride are taken as they are (slices of a concatenation)
        if G.stride(1) != 1 or G.stride(0) % 4 or G.data_ptr() % 16 or G.dtype != _F32:
            G = gout.contiguous().float()
        gin, ld_gin = None, 0
        if need_x:
            ld_gin = _up(ctx.kin0, 4)
            gin = torch.empty((rows, ld_gin), dtype=_F32, device=dev)
        work, nbytes = st.work(ctx.K0, True, dev)
        _cabi.check(L.prcnn_train_stack_bwd(ctypes.byref(ctx.S), st.layers, st.nl, ctx.ns, _p(ctx.a_dump), ctx.ld_dump, _p(G), G.stride(0),
                                            _p(ctx.arg), _p(gin), ld_gin, _aligned_ptr(work), nbytes, _stream()), "prcnn_train_stack_bwd")
        grads = []
        params = ctx.keep[2]
        for l in range(st.nl):
            dW, dg, db = st.grad_views[l]
            grads += [dW.view_as(params[3 * l]), dg, db]      # (None for the parameters a layer does not have)
        gx0 = gx1 = None
        if need_x:
            if src.mode == "plain":
                gx0 = gin[:, :ctx.K0]
            elif src.mode == "group" and getattr(src, "rows", None) is not None:
                gr, C = src.rows, ctx.kin0
                # gather form (rows bucketed by source point, summed in row order): written, not accumulated
                gx0 = torch.empty((gr.B, gr.N, C), dtype=_F32, device=dev)
                nbytes = L.prcnn_flat_rows_grad_work_bytes(gr.B, gr.N, gr.max_rows) if INTERP_GRAD_GATHER else 0
                work = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
                _cabi.check(L.prcnn_flat_rows_grad_ws(_p(gin), ld_gin, _p(gr.ridx), _p(gr.rows_dev), gr.max_rows, C, _p(gx0), C, _p(gr.off),
                                                      gr.B, gr.N, gr.M, _p(work), nbytes, _stream()), "prcnn_flat_rows_grad")
            elif src.mode == "group":
                B, M, ns = src.idx.shape
                N0, C = src.xyz.shape[1], ctx.kin0
                gx0 = torch.zeros((B, N0, C), dtype=_F32, device=dev)
                _cabi.check(L.prcnn_group_rows_grad(_p(gin), ld_gin, _p(src.idx), B, M, ns, C, N0, _p(gx0), C, _stream()), "prcnn_group_rows_grad")
            else:
                B, n, _ = src.idx3.shape
                m, C2 = ctx.x_shapes[0][1], ctx.x_shapes[0][2]
                if ctx.needs_input_grad[3]:
                    # gather form (references bucketed by known point, summed in row order): written, not accumulated
                    gx0 = torch.empty((B, m, C2), dtype=_F32, device=dev)
                    nbytes = L.prcnn_interp_rows_grad_work_bytes(B, n, m) if INTERP_GRAD_GATHER else 0
                    work = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
                    _cabi.check(L.prcnn_interp_rows_grad_ws(_p(gin), ld_gin, _p(src.idx3), _p(src.w3), B, n, m, C2, _p(gx0), C2, _p(work), nbytes,
                                                            _stream()), "prcnn_interp_rows_grad")
                if ctx.x_shapes[1] is not None and ctx.needs_input_grad[4]:
                    gx1 = gin[:, C2:ctx.K0].reshape(B, n, ctx.K0 - C2)
        ctx.st = ctx.a_dump = ctx.arg = ctx.keep = None          # release the saved activations now
        return (None, None, None, gx0, gx1, *grads)


def run_stack(mods, src, x0, x1=None, pool_ns=0):
    """mods: pt_utils conv layers (stack_ok); -> (groups, N_last) rows"""
    params, bns = [], []
    for m in mods:
        conv, bn, _ = m._parts()
        params += [conv.weight, bn.weight, bn.bias] if bn is not None else [conv.weight, None, conv.bias]
        bns.append(bn)
    return SharedMLPTrain.apply(src, bns, pool_ns, x0, x1, *params)


def usable(*tensors):
    """the hand-written training path applies: autograd on, CUDA fp32 inputs"""
    return torch.is_grad_enabled() and all(t is None or (t.is_cuda and t.dtype == _F32) for t in tensors)


__all__ = ["SharedMLPTrain", "Source", "run_stack", "stack_ok", "usable", "nn"]
