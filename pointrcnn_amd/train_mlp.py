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

import torch
import torch.nn as nn

from . import _cabi, ops
from .ops import _p, _stream

_F32 = torch.float32
MODE = {"plain": 0, "group": 1, "interp": 2}


def _up(x, m):
    return (x + m - 1) // m * m


def stack_ok(mods):
    """True when every layer of a pt_utils conv stack is Conv(1x1, bias=False) -> BatchNorm(training, affine, running stats,
    momentum) -> ReLU on CUDA fp32 -- what the kernels implement."""
    from pointnet2_lib.pointnet2 import pytorch_utils as pt
    for m in mods:
        if not isinstance(m, pt._ConvBase) or not m._prcnn_fusable:
            return False
        conv, bn, act = m._parts()
        if bn is None or act is None or conv.bias is not None or not bn.training:
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


class SharedMLPTrain(torch.autograd.Function):
    """apply(src, bns, pool_ns, x0, x1, W_0, gamma_0, beta_0, W_1, ...) -> (groups, N_last) activated (and pooled) rows.
    x0: features (B,N,C) / known features (B,m,C2) / plain rows (R,K); x1: skip features (B,n,C1) or None."""

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
        ys, csts = [], []
        a_dump, ld_dump = None, 0
        with torch.no_grad():
            for l in range(nl):
                W, gamma, beta = params[3 * l: 3 * l + 3]
                bn = bns[l]
                N = W.shape[0]
                K = K0 if l == 0 else ys[-1].shape[1]
                w2 = W.reshape(N, K)
                wp = _pack(w2, 3 if (l == 0 and src.mode == "group") else 0)
                y = torch.empty((rows, N), dtype=_F32, device=dev)
                ld_c = _up(N, 128)
                cst = torch.zeros((6, ld_c), dtype=_F32, device=dev)
                part = torch.empty((L.prcnn_train_part_floats(rows, N),), dtype=_F32, device=dev)
                if l == 0:
                    Sl = S
                    if src.mode != "plain":
                        ld_dump = _up(K0, 4)
                        a_dump = torch.empty((rows, ld_dump), dtype=_F32, device=dev)
                else:
                    Sl = _cabi.TrainSrc()
                    Sl.mode, Sl.rows, Sl.K = 0, rows, K
                    Sl.in_, Sl.ld_in = _p(ys[-1]), ys[-1].stride(0)
                    Sl.pro_scale, Sl.pro_shift = _p(csts[-1][0]), _p(csts[-1][1])
                _cabi.check(L.prcnn_train_fwd(ctypes.byref(Sl), _p(wp), N, _p(y), N, _p(a_dump) if l == 0 else None, ld_dump if l == 0 else 0,
                                              _p(part), N, _stream()), "prcnn_train_fwd")
                _cabi.check(L.prcnn_train_bn_finalize(_p(part), N, rows, N, _p(gamma), _p(beta), float(bn.eps), float(bn.momentum),
                                                      _p(bn.running_mean), _p(bn.running_var), _p(cst), ld_c, _stream()),
                            "prcnn_train_bn_finalize")
                bn.num_batches_tracked += 1
                ys.append(y)
                csts.append(cst)
            ns = pool_ns if pool_ns else 1
            groups, N = rows // ns, ys[-1].shape[1]
            out = torch.empty((groups, N), dtype=_F32, device=dev)
            arg = torch.empty((groups, N), dtype=torch.uint8, device=dev) if ns > 1 else None
            _cabi.check(L.prcnn_train_pool(_p(ys[-1]), N, groups, ns, N, _p(csts[-1]), csts[-1].stride(0), _p(out), N, 0, _p(arg),
                                           _stream()), "prcnn_train_pool")
        ctx.src, ctx.nl, ctx.pool_ns, ctx.K0, ctx.rows = src, nl, (ns if ns > 1 else 0), K0, rows
        ctx.x_shapes = (None if x0 is None else tuple(x0.shape), None if x1 is None else tuple(x1.shape))
        ctx.ld_dump = ld_dump
        ctx.save_for_backward(*[t for t in (x0, x1, a_dump, arg) if t is not None], *ys, *csts, *params)
        ctx.have = (x0 is not None, x1 is not None, a_dump is not None, arg is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _cabi.lib()
        saved = list(ctx.saved_tensors)
        x0 = saved.pop(0) if ctx.have[0] else None
        x1 = saved.pop(0) if ctx.have[1] else None
        a_dump = saved.pop(0) if ctx.have[2] else None
        arg = saved.pop(0) if ctx.have[3] else None
        nl, rows, src = ctx.nl, ctx.rows, ctx.src
        ys, csts, params = saved[:nl], saved[nl:2 * nl], saved[2 * nl:]
        dev = gout.device
        need_x = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        grads = [None] * (3 * nl)
        G = gout.contiguous()
        ldG, pool = G.stride(0), ctx.pool_ns
        for l in range(nl - 1, -1, -1):
            W = params[3 * l]
            N = W.shape[0]
            K = ctx.K0 if l == 0 else ys[l - 1].shape[1]
            y, cst = ys[l], csts[l]
            g = _cabi.TrainGrad()
            g.rows, g.N, g.G, g.ldG = rows, N, _p(G), ldG
            g.arg, g.pool_ns = (_p(arg), pool) if (l == nl - 1 and pool) else (None, 0)
            g.y, g.ld_y, g.cst, g.ld_c = _p(y), y.stride(0), _p(cst), cst.stride(0)
            part = torch.empty((L.prcnn_train_bwd_part_floats(rows, N),), dtype=_F32, device=dev)
            dgamma, dbeta = torch.empty((N,), dtype=_F32, device=dev), torch.empty((N,), dtype=_F32, device=dev)
            _cabi.check(L.prcnn_train_bn_backward(ctypes.byref(g), _p(part), N, _p(dgamma), _p(dbeta), _stream()), "prcnn_train_bn_backward")
            # wgrad: dW = dy^T . a, a = the rows that entered this layer's convolution
            if l > 0:
                a, lda, ps, pb = ys[l - 1], ys[l - 1].stride(0), _p(csts[l - 1][0]), _p(csts[l - 1][1])
            elif a_dump is not None:
                a, lda, ps, pb = a_dump, ctx.ld_dump, None, None
            else:
                a, lda, ps, pb = x0, x0.stride(0), None, None
            splits = L.prcnn_train_wgrad_splits(rows, N, K)
            wpart = torch.empty((splits, N, K), dtype=_F32, device=dev)
            dW = torch.empty((N, K), dtype=_F32, device=dev)
            _cabi.check(L.prcnn_train_wgrad(ctypes.byref(g), _p(a), lda, K, ps, pb, _p(wpart), splits, _p(dW), _stream()), "prcnn_train_wgrad")
            if l == 0 and src.mode == "group" and K > 3:      # kernel order [feat | dxyz] -> torch order [dxyz | feat]
                dW = torch.cat([dW[:, K - 3:], dW[:, :K - 3]], 1)
            grads[3 * l], grads[3 * l + 1], grads[3 * l + 2] = dW.view_as(W), dgamma, dbeta
            if l == 0 and not need_x:
                break
            # dgrad: gradient w.r.t. the rows that entered the convolution
            w2 = W.reshape(N, K)
            if l == 0 and src.mode == "group":
                kin = K - 3                                   # only the feature columns have a consumer (xyz carries no gradient)
                wt = w2[:, 3:].t().contiguous()
            else:
                kin = K
                wt = w2.t().contiguous()
            wpt = _pack(wt)
            Gp = torch.empty((rows, _up(kin, 4)), dtype=_F32, device=dev)
            _cabi.check(L.prcnn_train_dgrad(ctypes.byref(g), _p(wpt), kin, _p(Gp), Gp.stride(0), _stream()), "prcnn_train_dgrad")
            G, ldG = Gp, Gp.stride(0)
        gx0 = gx1 = None
        if need_x:
            if src.mode == "plain":
                gx0 = G[:, :ctx.K0]
            elif src.mode == "group":
                B, M, ns = src.idx.shape
                N0, C = src.xyz.shape[1], ctx.K0 - 3
                gx0 = torch.zeros((B, N0, C), dtype=_F32, device=dev)
                _cabi.check(L.prcnn_group_rows_grad(_p(G), ldG, _p(src.idx), B, M, ns, C, N0, _p(gx0), C, _stream()), "prcnn_group_rows_grad")
            else:
                B, n, _ = src.idx3.shape
                m, C2 = ctx.x_shapes[0][1], ctx.x_shapes[0][2]
                if ctx.needs_input_grad[3]:
                    gx0 = torch.zeros((B, m, C2), dtype=_F32, device=dev)
                    _cabi.check(L.prcnn_interp_rows_grad(_p(G), ldG, _p(src.idx3), _p(src.w3), B, n, m, C2, _p(gx0), C2, _stream()),
                                "prcnn_interp_rows_grad")
                if x1 is not None and ctx.needs_input_grad[4]:
                    gx1 = G[:, C2:ctx.K0].reshape(B, n, ctx.K0 - C2)
        return (None, None, None, gx0, gx1, *grads)


def run_stack(mods, src, x0, x1=None, pool_ns=0):
    """mods: pt_utils conv layers (stack_ok); -> (groups, N_last) rows"""
    params, bns = [], []
    for m in mods:
        conv, bn, _ = m._parts()
        params += [conv.weight, bn.weight, bn.bias]
        bns.append(bn)
    return SharedMLPTrain.apply(src, bns, pool_ns, x0, x1, *params)


def usable(*tensors):
    """the hand-written training path applies: autograd on, CUDA fp32 inputs"""
    return torch.is_grad_enabled() and all(t is None or (t.is_cuda and t.dtype == _F32) for t in tensors)


__all__ = ["SharedMLPTrain", "Source", "run_stack", "stack_ok", "usable", "nn"]
