"""Tensor-level host layer over the C ABI: argument checking, output allocation, stream plumbing.

PyTorch is used for device memory and streams only; every computation below is one call into
libprcnn_pointops.so on torch's CURRENT stream (so it composes with torch.cuda.graphs and side streams).
Shapes and argument meaning mirror the reference op surface (see each function's citation).
"""
import ctypes
import os
import threading

import torch

from . import _cabi

_INT = torch.int32
_F32 = torch.float32
# The spatially pruned FPS kernel (Morton pre-sort + exact bounding-box skip) for 2048 < N <= 16384.  Bit-identical
# results, ~35 % shorter serial chain.  With the neighbour searches on the grid the FPS chain is the longest dependency of
# a batch, so it is the default (+8 % RPN throughput at 3 batches in flight); PRCNN_FPS_PRUNED=0 selects the plain kernel.
FPS_PRUNED = os.environ.get("PRCNN_FPS_PRUNED", "1") == "1"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=_F32, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA(HIP) tensor: the HIP kernels are the only implementation" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must have %d dims, got shape %s" % (name, ndim, tuple(t.shape)))
    return t


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------ PointNet++ operators
def furthest_point_sample(xyz, npoint, order="canonical"):
    """xyz (B,N,3) f32 -> idx (B,npoint) i32   [pointnet2_utils.furthest_point_sample]
    order: "canonical" (ties among equal running min-distances -> lowest point index; the tested contract) or "upstream"
    (the order the upstream CUDA kernel's thread layout produces, SURVEY Appendix A.1: argmin (k mod T, k)) -- they differ
    only on clouds with duplicate points / exact lattices; "upstream" is a comparison mode, not a fast path."""
    _chk(xyz, "xyz", ndim=3)
    B, N, _ = xyz.shape
    idx = torch.empty((B, npoint), dtype=_INT, device=xyz.device)
    if order != "canonical":
        if order != "upstream":
            raise ValueError("furthest_point_sample: order must be 'canonical' or 'upstream'")
        tmp = torch.empty((B, N), dtype=_F32, device=xyz.device)
        _cabi.check(_cabi.lib().prcnn_fps_order(_p(xyz), B, N, npoint, 1, _p(tmp), _p(idx), _stream()), "prcnn_fps_order")
        return idx
    # (B,N) 4-byte scratch: Morton-order permutation of the spatially pruned kernel (2048 < N <= 16384), or the
    # HBM-resident min-distance array (N > 16384); the small-N kernels need none
    tmp = torch.empty((B, N), dtype=_F32, device=xyz.device) if (N > 16384 or (N > 2048 and FPS_PRUNED)) else None
    L = _cabi.lib()
    _cabi.check(L.prcnn_fps(_p(xyz), B, N, npoint, _p(tmp), _p(idx), _stream()), "prcnn_fps")
    return idx


_ARITH = {"canonical": 0, "upstream": 1}
_ORDER = {"canonical": 0, "upstream": 1}


def furthest_point_sample_mode(xyz, npoint, order="canonical", arith="canonical"):
    """COMPARISON MODE (prcnn_fps_mode): FPS with a selectable tie order and squared-distance arithmetic -- "upstream" arithmetic is
    fma(dz,dz, fma(dy,dy, dx*dx)), what nvcc makes of the upstream kernel's expression.  Plain kernel, not a fast path."""
    _chk(xyz, "xyz", ndim=3)
    B, N, _ = xyz.shape
    idx = torch.empty((B, npoint), dtype=_INT, device=xyz.device)
    tmp = torch.empty((B, N), dtype=_F32, device=xyz.device)
    _cabi.check(_cabi.lib().prcnn_fps_mode(_p(xyz), B, N, npoint, _ORDER[order], _ARITH[arith], _p(tmp), _p(idx), _stream()), "prcnn_fps_mode")
    return idx


def ball_query_arith(radius, nsample, xyz, new_xyz, arith="upstream"):
    """COMPARISON MODE (prcnn_ball_query_arith): ball_query by a plain scan under the chosen squared-distance arithmetic"""
    _chk(xyz, "xyz", ndim=3); _chk(new_xyz, "new_xyz", ndim=3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=_INT, device=xyz.device)
    _cabi.check(_cabi.lib().prcnn_ball_query_arith(_p(xyz), _p(new_xyz), B, N, M, float(radius), nsample, _ARITH[arith], _p(idx), _stream()),
                "prcnn_ball_query_arith")
    return idx


def three_nn_arith(unknown, known, arith="upstream"):
    """COMPARISON MODE (prcnn_three_nn_arith): -> dist2 (B,n,3) SQUARED, idx (B,n,3) under the chosen squared-distance arithmetic"""
    _chk(unknown, "unknown", ndim=3); _chk(known, "known", ndim=3)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((B, n, 3), dtype=_F32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=_INT, device=unknown.device)
    _cabi.check(_cabi.lib().prcnn_three_nn_arith(_p(unknown), _p(known), B, n, m, _ARITH[arith], _p(d2), _p(idx), _stream()),
                "prcnn_three_nn_arith")
    return d2, idx


def fps_status():
    """Raise if a multi-workgroup furthest_point_sample launch (N > 16384) timed out waiting for a partner slice since the
    last check (its output then holds -1 indices).  Does not synchronise: call after the stream has (end of a step / test)."""
    _cabi.check(_cabi.lib().prcnn_fps_status(), "prcnn_fps_status")


def gather(features, idx):
    """features (B,C,N), idx (B,M) i32 -> (B,C,M)   [gather_operation]"""
    _chk(features, "features", ndim=3); _chk(idx, "idx", _INT, 2)
    B, C, N = features.shape
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=_F32, device=features.device)
    _cabi.check(_cabi.lib().prcnn_gather(_p(features), _p(idx), B, C, N, M, _p(out), _stream()), "prcnn_gather")
    return out


def gather_grad(grad_out, idx, N):
    _chk(grad_out, "grad_out", ndim=3); _chk(idx, "idx", _INT, 2)
    B, C, M = grad_out.shape
    g = torch.zeros((B, C, N), dtype=_F32, device=grad_out.device)
    _cabi.check(_cabi.lib().prcnn_gather_grad(_p(grad_out), _p(idx), B, C, N, M, _p(g), _stream()), "prcnn_gather_grad")
    return g


def gather_rows(in_cl, idx):
    """in_cl (B,N,C) channels-last (row stride may exceed C), idx (B,M) i32 -> (B,M,C)"""
    _chk(idx, "idx", _INT, 2)
    B, N, C = in_cl.shape
    M = idx.shape[1]
    out = torch.empty((B, M, C), dtype=_F32, device=in_cl.device)
    _cabi.check(_cabi.lib().prcnn_gather_rows(_p(in_cl), _row_stride(in_cl), _p(idx), B, N, M, C, _p(out), _stream()),
                "prcnn_gather_rows")
    return out


# Frames with at least this many points are searched through a per-frame grid (csrc/grid.hip) instead of a full scan:
# identical results, ~N/30 of the distance evaluations.  PRCNN_GRID_SEARCH=0 forces the scans (A/B, debugging).
GRID_MIN_POINTS = 2048 if os.environ.get("PRCNN_GRID_SEARCH", "1") != "0" else 1 << 62
THREE_NN_GRID_MIN = int(os.environ.get("PRCNN_THREE_NN_GRID_MIN", "1024")) if GRID_MIN_POINTS < (1 << 62) else 1 << 62   # known points
DENSE_SCAN = os.environ.get("PRCNN_DENSE_SCAN", "1") != "0"      # dense frames fall back to the scan (A/B switch, same results)
BQ_GRID_CELLS = int(os.environ.get("PRCNN_BQ_GRID_CELLS", "128"))       # cells per axis of the ball-query grid (64 | 128)


class Grid:
    """points of B frames binned into per-frame 64x64 x-z grids (prcnn_grid_build)"""

    def __init__(self, xyz, min_cell=0.0, cells_per_axis=64):
        _chk(xyz, "xyz", ndim=3)
        self.B, self.N = xyz.shape[0], xyz.shape[1]
        L = _cabi.lib()
        nbytes = L.prcnn_grid_bytes(self.B, self.N)
        self.buf = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=xyz.device)
        _cabi.check(L.prcnn_grid_build(_p(xyz), self.B, self.N, float(min_cell), int(cells_per_axis), _p(self.buf), nbytes, _stream()),
                    "prcnn_grid_build")


def ball_query_grid(grid, new_xyz, radius_a, nsample_a, radius_b=0.0, nsample_b=0, xyz=None):
    """== ball_query / ball_query2 on a Grid of the source points.  xyz: the points the grid was built from -- frames the grid
    build flags as dense are then answered by the index-order scan (same results, see prcnn_ball_query2_grid)"""
    _chk(new_xyz, "new_xyz", ndim=3)
    B, M = new_xyz.shape[0], new_xyz.shape[1]
    if B != grid.B:
        raise ValueError("ball_query_grid: %d frames of centroids vs %d frames in the grid" % (B, grid.B))
    ia = torch.empty((B, M, nsample_a), dtype=_INT, device=new_xyz.device)
    ib = torch.empty((B, M, nsample_b), dtype=_INT, device=new_xyz.device) if nsample_b else None
    _cabi.check(_cabi.lib().prcnn_ball_query2_grid(_p(grid.buf), _p(xyz), _p(new_xyz), B, grid.N, M, float(radius_a), nsample_a, _p(ia),
                                                   float(radius_b), nsample_b, _p(ib), _stream()), "prcnn_ball_query2_grid")
    return (ia, ib) if nsample_b else ia


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32   [ball_query]"""
    _chk(xyz, "xyz", ndim=3); _chk(new_xyz, "new_xyz", ndim=3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    if N >= GRID_MIN_POINTS and B > 0 and M > 0:
        return ball_query_grid(Grid(xyz, radius, BQ_GRID_CELLS), new_xyz, radius, nsample, xyz=xyz if DENSE_SCAN else None)
    idx = torch.empty((B, M, nsample), dtype=_INT, device=xyz.device)
    _cabi.check(_cabi.lib().prcnn_ball_query(_p(xyz), _p(new_xyz), B, N, M, float(radius), nsample, _p(idx), _stream()),
                "prcnn_ball_query")
    return idx


def ball_query2(radius_a, nsample_a, radius_b, nsample_b, xyz, new_xyz):
    """two radii in one scan -> (idx_a, idx_b)"""
    _chk(xyz, "xyz", ndim=3); _chk(new_xyz, "new_xyz", ndim=3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    if N >= GRID_MIN_POINTS and B > 0 and M > 0:
        return ball_query_grid(Grid(xyz, max(radius_a, radius_b), BQ_GRID_CELLS), new_xyz, radius_a, nsample_a, radius_b, nsample_b,
                               xyz=xyz if DENSE_SCAN else None)
    ia = torch.empty((B, M, nsample_a), dtype=_INT, device=xyz.device)
    ib = torch.empty((B, M, nsample_b), dtype=_INT, device=xyz.device)
    _cabi.check(_cabi.lib().prcnn_ball_query2(_p(xyz), _p(new_xyz), B, N, M, float(radius_a), nsample_a, _p(ia),
                                              float(radius_b), nsample_b, _p(ib), _stream()), "prcnn_ball_query2")
    return ia, ib


def group(features, idx):
    """features (B,C,N), idx (B,M,ns) -> (B,C,M,ns)   [grouping_operation]"""
    _chk(features, "features", ndim=3); _chk(idx, "idx", _INT, 3)
    B, C, N = features.shape
    _, M, ns = idx.shape
    out = torch.empty((B, C, M, ns), dtype=_F32, device=features.device)
    _cabi.check(_cabi.lib().prcnn_group(_p(features), _p(idx), B, C, N, M, ns, _p(out), _stream()), "prcnn_group")
    return out


def group_grad(grad_out, idx, N):
    _chk(grad_out, "grad_out", ndim=4); _chk(idx, "idx", _INT, 3)
    B, C, M, ns = grad_out.shape
    g = torch.zeros((B, C, N), dtype=_F32, device=grad_out.device)
    _cabi.check(_cabi.lib().prcnn_group_grad(_p(grad_out), _p(idx), B, C, N, M, ns, _p(g), _stream()), "prcnn_group_grad")
    return g


def three_nn(unknown, known, want_weight=False):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3), idx (B,n,3) i32 [, weight (B,n,3)]"""
    _chk(unknown, "unknown", ndim=3); _chk(known, "known", ndim=3)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty((B, n, 3), dtype=_F32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=_INT, device=unknown.device)
    w = torch.empty((B, n, 3), dtype=_F32, device=unknown.device) if want_weight else None
    if m >= THREE_NN_GRID_MIN and B > 0 and n > 0:
        g = Grid(known, 0.0)
        _cabi.check(_cabi.lib().prcnn_three_nn_grid(_p(g.buf), _p(unknown), B, n, m, _p(d2), _p(idx), _p(w), _stream()),
                    "prcnn_three_nn_grid")
        return (d2, idx, w) if want_weight else (d2, idx)
    _cabi.check(_cabi.lib().prcnn_three_nn(_p(unknown), _p(known), B, n, m, _p(d2), _p(idx), _p(w), _stream()),
                "prcnn_three_nn")
    return (d2, idx, w) if want_weight else (d2, idx)


def three_interpolate(features, idx, weight):
    """features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n)"""
    _chk(features, "features", ndim=3); _chk(idx, "idx", _INT, 3); _chk(weight, "weight", ndim=3)
    B, C, m = features.shape
    n = idx.shape[1]
    out = torch.empty((B, C, n), dtype=_F32, device=features.device)
    _cabi.check(_cabi.lib().prcnn_three_interp(_p(features), _p(idx), _p(weight), B, C, m, n, _p(out), _stream()),
                "prcnn_three_interp")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, "grad_out", ndim=3); _chk(idx, "idx", _INT, 3); _chk(weight, "weight", ndim=3)
    B, C, n = grad_out.shape
    g = torch.zeros((B, C, m), dtype=_F32, device=grad_out.device)
    ws = torch.empty((B, m, C), dtype=_F32, device=grad_out.device)        # channels-last accumulator (see the header)
    _cabi.check(_cabi.lib().prcnn_three_interp_grad(_p(grad_out), _p(idx), _p(weight), B, C, n, m, _p(g), _p(ws), _stream()),
                "prcnn_three_interp_grad")
    return g


# ------------------------------------------------------------------ fused per-point MLP layers
class PackedLinear:
    """One 1x1-conv layer with (eval-mode) BatchNorm folded in, packed for the MFMA kernel.

    weight: (Nout, K) f32 device tensor in torch conv layout; bias: (Nout) or None.
    k_rot: leading input channels moved to the end of K (3 for grouped [dxyz, feat] layers).
    """

    def __init__(self, weight, bias=None, relu=True, k_rot=0):
        _chk(weight, "weight", ndim=2)
        self.nout, self.k = weight.shape
        self.relu = bool(relu)
        L = _cabi.lib()
        self.wpack = torch.empty((L.prcnn_wpack_floats(self.nout, self.k),), dtype=_F32, device=weight.device)
        _cabi.check(L.prcnn_pack_weight(_p(weight), self.nout, self.k, k_rot, _p(self.wpack), _stream()),
                    "prcnn_pack_weight")
        self._w = weight if k_rot == 0 else None         # kept for the split-bf16 variant's own weight image (built on first use)
        self._wsplit = None
        # bias is kept zero-padded to a multiple of 32 entries (the chain kernel reads whole 32-channel blocks)
        self.bias = None
        if bias is not None:
            _chk(bias.contiguous(), "bias", ndim=1)
            self.bias = torch.zeros(((self.nout + 31) // 32 * 32,), dtype=_F32, device=weight.device)
            self.bias[: self.nout] = bias


    def wsplit(self, chain=0):
        """the pre-split bf16 weight image of the split-bf16 variant (prcnn_pack_weight_split; chain = 1: the chain kernel's
        image), or None when the layer does not qualify (K a multiple of 32, no rotated input channels)"""
        if self._wsplit is None:
            self._wsplit = {}
        if chain not in self._wsplit and self._w is not None and self.k % 32 == 0:
            L = _cabi.lib()
            img = torch.empty((L.prcnn_wsplit_bytes(self.nout, self.k),), dtype=torch.uint8, device=self._w.device)
            _cabi.check(L.prcnn_pack_weight_split(_p(self._w.contiguous()), self.nout, self.k, int(chain), _p(img), _stream()),
                        "prcnn_pack_weight_split")
            self._wsplit[chain] = img
        return self._wsplit.get(chain)


# Arithmetic of the fused inference MLPs' plain-row layers, heads and hoisted FP0 (everything that reaches prcnn_mlp_rows*,
# the two-layer head chains and the interpolating chain):
#   6 (default) -- split-bf16, six terms: every fp32 operand cut EXACTLY into three bf16 pieces, six bf16 MFMA products per fp32
#                  product, fp32 accumulate; drops partial products below 2^-24 |x||w|: fp32-grade results (measured 2.9e-7 of
#                  sum |x||w| against float64, the fp32-MFMA kernel 3.0e-7), rows with non-finite values recomputed on the fp32
#                  pipe.  Qualified by the whole GPU suite, which runs every test under both arithmetics (tests/conftest.py
#                  `mlp_mode`), and by bench.py's max_diff_vs_f32_mfma on uniform, lidar and saturated clouds (contract 1e-5).
#   0           -- fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout: PRCNN_MLP_SPLIT=0.
#   3           -- three terms (drops below 2^-16 |x||w|): OUTSIDE the 1e-5 contract, dev / trade-off measurements only.
# The choice never depends on row counts (a frame's bits are the same in a batch of 1 and of 32 -- with one exception: FINITE rows that
# share a 64-row block (32 in the chain kernels) with a row holding inf / NaN are recomputed with it on the fp32 pipe and carry the
# fp32-MFMA kernel's bits, within the same 1e-5 contract).  Training kernels are fp32 MFMA.
MLP_SPLIT_TERMS = int(os.environ.get("PRCNN_MLP_SPLIT", "6") or 0)
if MLP_SPLIT_TERMS not in (0, 3, 6):
    raise RuntimeError("PRCNN_MLP_SPLIT=%r: 6 (default) / 3 split-bf16 terms, 0 = fp32 MFMA" % os.environ.get("PRCNN_MLP_SPLIT"))

_MODE_ROWS, _MODE_GROUP, _MODE_INTERP = 0, 1, 2
_chain_ok = {}


def chain_supported(mode, layers, pool_ns=0):
    """True when the register-resident chain kernel has an instance for this stack of PackedLinear layers."""
    key = (mode, tuple(l.nout for l in layers), pool_ns)
    hit = _chain_ok.get(key)
    if hit is None:
        arr = (ctypes.c_int * len(layers))(*[l.nout for l in layers])
        hit = bool(_cabi.lib().prcnn_mlp_chain_supported(mode, len(layers), arr, pool_ns)) if 1 <= len(layers) <= 3 else False
        _chain_ok[key] = hit
    return hit


class _ChainArgs:
    """host arrays (wpack*, bias*, nout, relu) describing a stack of PackedLinear layers.  Holds the layers themselves:
    the raw pointers below stay valid and the id()s of the cache key cannot be recycled while this object lives."""

    def __init__(self, layers):
        n = len(layers)
        self.n = n
        self.layers = tuple(layers)
        self.wpack = (ctypes.c_void_p * n)(*[l.wpack.data_ptr() for l in layers])
        self.bias = (ctypes.c_void_p * n)(*[(l.bias.data_ptr() if l.bias is not None else None) for l in layers])
        self.nout = (ctypes.c_int * n)(*[l.nout for l in layers])
        self.relu = (ctypes.c_int * n)(*[int(l.relu) for l in layers])


def _chain_args(layers):
    key = tuple(id(l) for l in layers)
    hit = getattr(layers[0], "_chain_cache", None)
    if hit is None or hit[0] != key:
        hit = (key, _ChainArgs(layers))
        layers[0]._chain_cache = hit
    return hit[1]


def mlp_chain_rows(x, layers, out=None, pool_ns=0, seg=None):
    """whole stack on channels-last rows in one kernel (see mlp_rows for the layout contract; seg as there)"""
    if x.stride(-1) != 1:
        raise RuntimeError("mlp_chain_rows: last dim must be contiguous")
    K = x.shape[-1]
    rows = x.numel() // K
    ld_in = _row_stride(x)
    rows_out = rows // pool_ns if pool_ns else rows
    buf, ld_out, col_off = _out_buf(out, rows_out, layers[-1], x.device)
    a = _chain_args(layers)
    seg_cnt, seg_rows = (None, 0) if seg is None else (seg[0], int(seg[1]))
    if (MLP_SPLIT_TERMS and not pool_ns and seg is None and a.n == 2 and K == 128 and layers[0].nout == 128
            and (layers[1].nout == 1 or 64 < layers[1].nout <= 128) and layers[0].wsplit(1) is not None and layers[1].wsplit(1) is not None):
        if getattr(a, "wchain", None) is None:                 # (host array of the two chain images, kept with the cached args)
            a.wchain = (ctypes.c_void_p * 2)(layers[0].wsplit(1).data_ptr(), layers[1].wsplit(1).data_ptr())
        rc = _cabi.lib().prcnn_mlp_chain_rows_split(_p(x), ld_in, rows, K, a.wchain, a.wpack, a.bias, a.nout, a.relu,
                                                    MLP_SPLIT_TERMS, _p(buf), ld_out, col_off, _stream())
        if rc != _cabi.EUNSUPPORTED:
            _cabi.check(rc, "prcnn_mlp_chain_rows_split")
            return buf
    _cabi.check(_cabi.lib().prcnn_mlp_chain_rows(_p(x), ld_in, rows, K, a.n, a.wpack, a.bias, a.nout, a.relu, _p(buf),
                                                 ld_out, col_off, pool_ns, _p(seg_cnt), seg_rows, _stream()), "prcnn_mlp_chain_rows")
    return buf


def mlp_chain_group(xyz, new_xyz, idx, feat_cl, layers, out=None, pool_ns=0, act=None, groups_dev=None):
    B, N, _ = xyz.shape
    _, M, ns = idx.shape
    C = 0 if feat_cl is None else feat_cl.shape[-1]
    ld_feat = 0 if feat_cl is None else _row_stride(feat_cl)
    rows = B * M * ns
    rows_out = rows // pool_ns if pool_ns else rows
    buf, ld_out, col_off = _out_buf(out, rows_out, layers[-1], xyz.device)
    a = _chain_args(layers)
    awx, ab = (None, None) if act is None else act
    _cabi.check(_cabi.lib().prcnn_mlp_chain_group(_p(xyz), _p(new_xyz), _p(idx), _p(feat_cl), ld_feat, B, N, M, ns, C, _p(awx), _p(ab), a.n,
                                                  a.wpack, a.bias, a.nout, a.relu, _p(buf), ld_out, col_off, pool_ns,
                                                  _p(groups_dev), _stream()), "prcnn_mlp_chain_group")
    return buf


def mlp_chain_interp(known_cl, idx3, w3, skip_cl, layers, out=None, act_bias=None):
    B, m, C2 = known_cl.shape
    n = idx3.shape[1]
    C1 = 0 if skip_cl is None else skip_cl.shape[-1]
    ld_skip = 0 if skip_cl is None else _row_stride(skip_cl)
    buf, ld_out, col_off = _out_buf(out, B * n, layers[-1], known_cl.device)
    a = _chain_args(layers)
    if (MLP_SPLIT_TERMS and skip_cl is None and act_bias is not None and a.n == 1 and C2 == 128 and layers[0].nout == 128
            and layers[0].wsplit(1) is not None):
        l0 = layers[0]
        rc = _cabi.lib().prcnn_mlp_chain_interp_split(_p(known_cl), _row_stride(known_cl), _p(idx3), _p(w3), B, n, m, C2, _p(act_bias),
                                                      _p(l0.wsplit(1)), _p(l0.wpack), _p(l0.bias), l0.nout, int(l0.relu), MLP_SPLIT_TERMS, _p(buf),
                                                      ld_out, col_off, _stream())
        if rc != _cabi.EUNSUPPORTED:
            _cabi.check(rc, "prcnn_mlp_chain_interp_split")
            return buf
    _cabi.check(_cabi.lib().prcnn_mlp_chain_interp(_p(known_cl), _row_stride(known_cl), _p(idx3), _p(w3), _p(skip_cl), ld_skip,
                                                   B, n, m, C2, C1, _p(act_bias), a.n, a.wpack, a.bias, a.nout, a.relu, _p(buf),
                                                   ld_out, col_off, _stream()), "prcnn_mlp_chain_interp")
    return buf


def _row_stride(x):
    """row stride (in elements) of a channels-last rows view (..., K): the stride of the innermost leading dim of
    size > 1 (size-1 dims carry arbitrary strides); the caller guarantees uniform striding (pt_utils._rows_view)."""
    for size, stride in zip(reversed(x.shape[:-1]), reversed(x.stride()[:-1])):
        if size != 1:
            return stride
    return x.shape[-1]


def _out_buf(out, rows, lin, device):
    if out is None:
        return torch.empty((rows, lin.nout), dtype=_F32, device=device), lin.nout, 0
    buf, col_off = out
    return buf, _row_stride(buf), col_off


def mlp_rows(x, lin, out=None, pool_ns=0, rows_dev=None, rows_unit=1, seg=None):
    """x (..., K) channels-last rows (last dim contiguous, uniform row stride) -> (rows[/pool_ns], Nout).
    out = (buffer, col_off) writes into a wider channels-last buffer instead of allocating.
    rows_dev (1,) int32 device tensor: only the first rows_dev * rows_unit rows are processed (device-side count).
    seg = (seg_cnt (nseg,) int32 device tensor, seg_rows): segment-prefix live rows -- of every run of seg_rows rows only the
    first seg_cnt[s] are live (roipool3d wrap-copies); tiles in a segment's dead tail are skipped, their outputs unwritten."""
    if x.stride(-1) != 1:
        raise RuntimeError("mlp_rows: last dim must be contiguous")
    K = x.shape[-1]
    rows = x.numel() // K
    ld_in = _row_stride(x)
    rows_out = rows // pool_ns if pool_ns else rows
    buf, ld_out, col_off = _out_buf(out, rows_out, lin, x.device)
    if MLP_SPLIT_TERMS and ld_in % 4 == 0 and (seg is None or (rows_dev is None and not pool_ns)) and lin.wsplit() is not None:
        _cabi.check(_cabi.lib().prcnn_mlp_rows_split(_p(x), ld_in, rows, K, _p(lin.wpack), _p(lin.wsplit()), MLP_SPLIT_TERMS,
                                                     _p(lin.bias), lin.nout, int(lin.relu), _p(buf), ld_out, col_off, int(pool_ns), _p(rows_dev),
                                                     int(rows_unit), _p(None if seg is None else seg[0]), 0 if seg is None else int(seg[1]),
                                                     _stream()), "prcnn_mlp_rows_split")
        return buf
    _cabi.check(_cabi.lib().prcnn_mlp_rows(_p(x), ld_in, rows, K, _p(lin.wpack), _p(lin.bias), lin.nout, int(lin.relu),
                                           _p(buf), ld_out, col_off, pool_ns, _p(rows_dev), int(rows_unit),
                                           _p(None if seg is None else seg[0]), 0 if seg is None else int(seg[1]), _stream()), "prcnn_mlp_rows")
    return buf


def mlp_group(xyz, new_xyz, idx, feat_cl, lin, out=None, pool_ns=0, act=None, groups_dev=None):
    """First SA layer fused with ball-query grouping.  xyz (B,N,3), new_xyz (B,M,3) or None (GroupAll),
    idx (B,M,ns) i32, feat_cl (B,N,C) channels-last or None -> (B*M*ns[/pool_ns], Nout).
    act = (act_wx (C,3), act_bias (C)) selects the HOISTED form: feat_cl is Z = W_f.feat per source point and `lin` is
    the SECOND layer (see include/prcnn_pointops.h)."""
    B, N, _ = xyz.shape
    _, M, ns = idx.shape
    C = 0 if feat_cl is None else feat_cl.shape[-1]
    ld_feat = 0 if feat_cl is None else _row_stride(feat_cl)
    rows = B * M * ns
    rows_out = rows // pool_ns if pool_ns else rows
    buf, ld_out, col_off = _out_buf(out, rows_out, lin, xyz.device)
    awx, ab = (None, None) if act is None else act
    if MLP_SPLIT_TERMS and act is not None and C % 32 == 0 and ld_feat % 4 == 0 and lin.wsplit() is not None:
        # the hoisted grouped layer on the split-bf16 kernel (round 5; PRCNN_GROUP_SPLIT=0 inside the library is the A/B switch)
        _cabi.check(_cabi.lib().prcnn_mlp_group_split(_p(xyz), _p(new_xyz), _p(idx), _p(feat_cl), ld_feat, B, N, M, ns, C, _p(awx), _p(ab),
                                                      _p(lin.wpack), _p(lin.wsplit()), MLP_SPLIT_TERMS, _p(lin.bias), lin.nout, int(lin.relu),
                                                      _p(buf), ld_out, col_off, pool_ns, _p(groups_dev), _stream()), "prcnn_mlp_group_split")
        return buf
    _cabi.check(_cabi.lib().prcnn_mlp_group(_p(xyz), _p(new_xyz), _p(idx), _p(feat_cl), ld_feat, B, N, M, ns, C, _p(awx), _p(ab),
                                            _p(lin.wpack), _p(lin.bias), lin.nout, int(lin.relu), _p(buf), ld_out,
                                            col_off, pool_ns, _p(groups_dev), _stream()), "prcnn_mlp_group")
    return buf


def mlp_interp(known_cl, idx3, w3, skip_cl, lin, out=None, act_bias=None):
    """First FP layer fused with three_interpolate + skip concat.  known_cl (B,m,C2), idx3/w3 (B,n,3),
    skip_cl (B,n,C1) or None -> (B*n, Nout).  act_bias (C2) selects the HOISTED form (skip_cl must be None): known_cl
    is Y = W.known per known point, the row is relu(interp(Y) + act_bias) and `lin` is the second layer."""
    B, m, C2 = known_cl.shape
    n = idx3.shape[1]
    C1 = 0 if skip_cl is None else skip_cl.shape[-1]
    ld_skip = 0 if skip_cl is None else _row_stride(skip_cl)
    buf, ld_out, col_off = _out_buf(out, B * n, lin, known_cl.device)
    _cabi.check(_cabi.lib().prcnn_mlp_interp(_p(known_cl), _row_stride(known_cl), _p(idx3), _p(w3), _p(skip_cl), ld_skip,
                                             B, n, m, C2, C1, _p(act_bias), _p(lin.wpack), _p(lin.bias), lin.nout,
                                             int(lin.relu), _p(buf), ld_out, col_off, _stream()), "prcnn_mlp_interp")
    return buf


def mlp_rows_addinterp(skip_cl, lin_b, y_cl, idx3, w3, out=None):
    """Hoisted FP first layer with skip features: act(skip . W_b^T + bias + interp(y_cl)); skip_cl (B,n,C1),
    y_cl (B,m,Nout) = W_a.known, idx3/w3 (B,n,3) -> (B*n, Nout)."""
    B, n, C1 = skip_cl.shape
    m = y_cl.shape[1]
    buf, ld_out, col_off = _out_buf(out, B * n, lin_b, skip_cl.device)
    if MLP_SPLIT_TERMS and _row_stride(skip_cl) % 4 == 0 and lin_b.wsplit() is not None:
        _cabi.check(_cabi.lib().prcnn_mlp_rows_addinterp_split(_p(skip_cl), _row_stride(skip_cl), C1, _p(lin_b.wpack), _p(lin_b.wsplit()),
                                                               MLP_SPLIT_TERMS, _p(lin_b.bias), lin_b.nout, int(lin_b.relu), _p(y_cl),
                                                               _row_stride(y_cl), _p(idx3), _p(w3), B, n, m, _p(buf), ld_out, col_off,
                                                               _stream()), "prcnn_mlp_rows_addinterp_split")
        return buf
    _cabi.check(_cabi.lib().prcnn_mlp_rows_addinterp(_p(skip_cl), _row_stride(skip_cl), C1, _p(lin_b.wpack), _p(lin_b.bias),
                                                     lin_b.nout, int(lin_b.relu), _p(y_cl), _row_stride(y_cl), _p(idx3),
                                                     _p(w3), B, n, m, _p(buf), ld_out, col_off, _stream()),
                "prcnn_mlp_rows_addinterp")
    return buf


def maxpool_rows(x, ns, out=None):
    """x (rows, C) -> (rows/ns, C): max over every ns consecutive rows (generic nsample fallback)."""
    rows, C = x.shape
    rows_out = rows // ns
    if out is None:
        buf, ld_out, col_off = torch.empty((rows_out, C), dtype=_F32, device=x.device), C, 0
    else:
        buf, col_off = out
        ld_out = buf.stride(-2)
    _cabi.check(_cabi.lib().prcnn_maxpool_rows(_p(x), x.stride(0), rows_out, ns, C, _p(buf), ld_out, col_off, _stream()),
                "prcnn_maxpool_rows")
    return buf


# ------------------------------------------------------------------ roipool3d
ROIPOOL_BINS = os.environ.get("PRCNN_ROIPOOL_BINS", "1") != "0"        # binned point selection (A/B switch; same results)
ROIPOOL_BINS_MIN_BOXES = 8


def _roipool_work(B, N, M, dev):
    """scratch for the binned point selection of roipool3d (None: linear scan -- few RoIs, or N beyond the LDS bitmap)"""
    if not ROIPOOL_BINS or M < ROIPOOL_BINS_MIN_BOXES:
        return None, 0
    nbytes = _cabi.lib().prcnn_roipool3d_work_bytes(B, N)
    if nbytes == 0:
        return None, 0
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def roipool3d(xyz, boxes3d_enlarged, pts_feature, sampled_pt_num):
    """xyz (B,N,3), boxes (B,M,7) already enlarged, pts_feature (B,N,C) -> pooled (B,M,S,3+C), empty (B,M) i32
    [roipool3d_cuda.forward, lib/utils/roipool3d/src/roipool3d.cpp:48-79]"""
    _chk(xyz, "xyz", ndim=3); _chk(boxes3d_enlarged, "boxes3d", ndim=3); _chk(pts_feature, "pts_feature", ndim=3)
    B, N, _ = xyz.shape
    M, C = boxes3d_enlarged.shape[1], pts_feature.shape[2]
    pooled = torch.empty((B, M, sampled_pt_num, 3 + C), dtype=_F32, device=xyz.device)
    empty = torch.empty((B, M), dtype=_INT, device=xyz.device)
    work, nbytes = _roipool_work(B, N, M, xyz.device)
    _cabi.check(_cabi.lib().prcnn_roipool3d_ws(_p(xyz), _p(boxes3d_enlarged), _p(pts_feature), B, N, M, C, sampled_pt_num,
                                               _p(pooled), _p(empty), _p(work), nbytes, _stream()), "prcnn_roipool3d")
    return pooled, empty


def roipool3d_canonical(xyz, pool_boxes3d, rois, extras, feat_cl, sampled_pt_num, out_feat=None, want_distinct=False):
    """Fused lib/net/rcnn_net.py:127-154: pool [extras..., feat] per RoI and move the pooled xyz into the RoI's
    canonical frame, writing each consumer's operand directly.
    xyz (B,N,3); pool_boxes3d (B,M,7) enlarged; rois (B,M,7) or None; extras: list of <= 2 (B,N) tensors;
    feat_cl (B,N,C) channels-last rows (unit last stride).  out_feat = (buffer (B*M*S, W), col): write the C pooled
    features at that column of a wider rows buffer (default: a fresh (B*M*S, C) tensor).
    want_distinct: also return distinct (B,M) i32 = number of distinct rows per RoI (the rest are wrap-copies); the feature
    rows of the copies are then NOT written (see prcnn_roipool3d_canonical).
    -> pts (B*M, S, 3+len(extras)), feat buffer, empty (B,M) i32 [, distinct]"""
    _chk(xyz, "xyz", ndim=3); _chk(pool_boxes3d, "pool_boxes3d", ndim=3)
    B, N, _ = xyz.shape
    if not (isinstance(feat_cl, torch.Tensor) and feat_cl.is_cuda and feat_cl.dtype == _F32 and feat_cl.dim() == 3
            and feat_cl.stride(-1) == 1 and feat_cl.stride(0) == N * feat_cl.stride(1)):
        raise RuntimeError("roipool3d_canonical: feat must be a (B,N,C) fp32 device tensor of uniformly strided rows")
    M, C, S = pool_boxes3d.shape[1], feat_cl.shape[2], int(sampled_pt_num)
    ex = [e.contiguous() for e in extras]
    if len(ex) > 2:
        raise ValueError("roipool3d_canonical: at most 2 scalar channels")
    for e in ex:
        _chk(e, "extra", ndim=2)
    P = 3 + len(ex)
    dev = xyz.device
    pts = torch.empty((B * M, S, P), dtype=_F32, device=dev)
    if out_feat is None:
        fbuf, col = torch.empty((B * M * S, C), dtype=_F32, device=dev), 0
    else:
        fbuf, col = out_feat
        if fbuf.dim() != 2 or fbuf.shape[0] != B * M * S or fbuf.stride(1) != 1 or col + C > fbuf.shape[1]:
            raise ValueError("roipool3d_canonical: out_feat buffer must be (B*M*S, >= col + C) rows")
    empty = torch.empty((B, M), dtype=_INT, device=dev)
    distinct = torch.empty((B, M), dtype=_INT, device=dev) if want_distinct else None
    if rois is not None:
        _chk(rois, "rois", ndim=3)
    work, nbytes = _roipool_work(B, N, M, dev)
    _cabi.check(_cabi.lib().prcnn_roipool3d_canonical_ws(
        _p(xyz), _p(pool_boxes3d), _p(rois), _p(ex[0]) if ex else None, _p(ex[1]) if len(ex) > 1 else None, _p(feat_cl),
        _row_stride(feat_cl), B, N, M, C, S, _p(pts), P, fbuf.data_ptr() + 4 * col, fbuf.stride(0), _p(empty), _p(distinct),
        _p(work), nbytes, _stream()), "prcnn_roipool3d_canonical")
    return (pts, fbuf, empty, distinct) if want_distinct else (pts, fbuf, empty)


def rpn_labels(pts, gt_boxes3d, num_gt=None, extra_width=0.2):
    """KittiRCNNDataset.generate_rpn_training_labels (kitti_rcnn_dataset.py:365-394) for a whole batch on the device.
    pts (B,N,3), gt_boxes3d (B,G,7), num_gt (B) i32 or None -> cls_label (B,N) i32 {-1,0,1}, reg_label (B,N,7)"""
    _chk(pts, "pts", ndim=3); _chk(gt_boxes3d, "gt_boxes3d", ndim=3)
    B, N, _ = pts.shape
    G = gt_boxes3d.shape[1]
    if gt_boxes3d.shape[0] != B or gt_boxes3d.shape[2] != 7:
        raise ValueError("rpn_labels: gt_boxes3d must be (%d, G, 7)" % B)
    if num_gt is not None:
        _chk(num_gt, "num_gt", _INT, 1)
    cls = torch.empty((B, N), dtype=_INT, device=pts.device)
    reg = torch.empty((B, N, 7), dtype=_F32, device=pts.device)
    _cabi.check(_cabi.lib().prcnn_rpn_labels(_p(pts), _p(gt_boxes3d), _p(num_gt), B, N, G, float(extra_width), _p(cls), _p(reg),
                                             _stream()), "prcnn_rpn_labels")
    return cls, reg


def gt_aug_edit(pts, intensity, boxes3d, new_pts, new_intensity, num_pts=None, num_boxes=None, num_new=None, extra_h=2.0,
                want_removed=False):
    """The point work of KittiRCNNDataset.apply_gt_aug_to_one_scene (kitti_rcnn_dataset.py:484-507) for a batch of scenes.
    pts (B,N,3), intensity (B,N) or None, boxes3d (B,K,7) accepted objects (tested with h + extra_h), new_pts (B,P,3),
    new_intensity (B,P) or None; num_* (B) i32 live counts or None.
    -> out_pts (B,N+P,3), out_intensity (B,N+P) or None, count (B) i32 [, removed (B,N) i32]: surviving scene points in their
    original order, then the new points; rows past count are zero."""
    _chk(pts, "pts", ndim=3); _chk(boxes3d, "boxes3d", ndim=3); _chk(new_pts, "new_pts", ndim=3)
    B, N, _ = pts.shape
    K, P = boxes3d.shape[1], new_pts.shape[1]
    if boxes3d.shape[0] != B or boxes3d.shape[2] != 7 or new_pts.shape[0] != B or new_pts.shape[2] != 3:
        raise ValueError("gt_aug_edit: boxes3d must be (%d, K, 7), new_pts (%d, P, 3)" % (B, B))
    if (intensity is None) != (new_intensity is None):
        raise ValueError("gt_aug_edit: intensity and new_intensity go together")
    if intensity is not None:
        _chk(intensity, "intensity", ndim=2); _chk(new_intensity, "new_intensity", ndim=2)
        if tuple(intensity.shape) != (B, N) or tuple(new_intensity.shape) != (B, P):
            raise ValueError("gt_aug_edit: intensity must be (B, N), new_intensity (B, P)")
    for name, t in (("num_pts", num_pts), ("num_boxes", num_boxes), ("num_new", num_new)):
        if t is not None:
            _chk(t, name, _INT, 1)
            if t.shape[0] != B:
                raise ValueError("gt_aug_edit: %s must be (%d,)" % (name, B))
    dev = pts.device
    out_pts = torch.empty((B, N + P, 3), dtype=_F32, device=dev)
    out_int = torch.empty((B, N + P), dtype=_F32, device=dev) if intensity is not None else None
    count = torch.empty((B,), dtype=_INT, device=dev)
    removed = torch.empty((B, N), dtype=_INT, device=dev) if want_removed else None
    _cabi.check(_cabi.lib().prcnn_gt_aug_edit(_p(pts), _p(intensity), _p(num_pts), _p(boxes3d), _p(num_boxes), float(extra_h),
                                              _p(new_pts), _p(new_intensity), _p(num_new), B, N, K, P, _p(out_pts), _p(out_int),
                                              _p(count), _p(removed), _stream()), "prcnn_gt_aug_edit")
    return (out_pts, out_int, count, removed) if want_removed else (out_pts, out_int, count)


def pts_in_boxes3d(pts, boxes3d):
    """pts (N,3), boxes3d (M,7) -> flags (M,N) i32"""
    _chk(pts, "pts", ndim=2); _chk(boxes3d, "boxes3d", ndim=2)
    N, M = pts.shape[0], boxes3d.shape[0]
    flags = torch.empty((M, N), dtype=_INT, device=pts.device)
    _cabi.check(_cabi.lib().prcnn_pts_in_boxes3d(_p(pts), _p(boxes3d), N, M, _p(flags), _stream()), "prcnn_pts_in_boxes3d")
    return flags


# ------------------------------------------------------------------ iou3d
def boxes_overlap_bev(boxes_a, boxes_b, out=None):
    _chk(boxes_a, "boxes_a", ndim=2); _chk(boxes_b, "boxes_b", ndim=2)
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    if out is None:
        out = torch.empty((na, nb), dtype=_F32, device=boxes_a.device)
    _cabi.check(_cabi.lib().prcnn_boxes_overlap_bev(_p(boxes_a), na, _p(boxes_b), nb, _p(out), _stream()),
                "prcnn_boxes_overlap_bev")
    return out


def boxes_iou_bev(boxes_a, boxes_b, out=None):
    _chk(boxes_a, "boxes_a", ndim=2); _chk(boxes_b, "boxes_b", ndim=2)
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    if out is None:
        out = torch.empty((na, nb), dtype=_F32, device=boxes_a.device)
    _cabi.check(_cabi.lib().prcnn_boxes_iou_bev(_p(boxes_a), na, _p(boxes_b), nb, _p(out), _stream()),
                "prcnn_boxes_iou_bev")
    return out


def ref_trig(fn, a, b=None):
    """csrc/ref_trig.h on the device, element-wise: fn in {"sinf", "cosf", "atan2f"} (atan2f(a, b)); float32 tensors"""
    _chk(a, "a")
    out = torch.empty_like(a)
    _cabi.check(_cabi.lib().prcnn_ref_trig(_p(a), _p(b), a.numel(), ["sinf", "cosf", "atan2f"].index(fn), _p(out), _stream()), "prcnn_ref_trig")
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """boxes_a (N,7), boxes_b (M,7) [x, y, z, h, w, l, ry] -> 3-D IoU (N,M)   [iou3d_utils.boxes_iou3d_gpu, iou3d_utils.py:20-53]"""
    _chk(boxes_a, "boxes_a", ndim=2); _chk(boxes_b, "boxes_b", ndim=2)
    out = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=_F32, device=boxes_a.device)
    _cabi.check(_cabi.lib().prcnn_boxes_iou3d(_p(boxes_a), boxes_a.shape[0], _p(boxes_b), boxes_b.shape[0], _p(out), _stream()),
                "prcnn_boxes_iou3d")
    return out


def proposal_target_sample(roi_boxes3d, gt_boxes3d, roi_per_image=64, thresholds=(0.55, 0.6, 0.45, 0.05), fg_ratio=0.5, hard_bg_ratio=0.8,
                           aug_times=10, aug_method="multiple", seed=0):
    """ProposalTargetLayer.sample_rois_for_rcnn for the whole batch in one launch (lib/rpn/proposal_target_layer.py:75-250).
    roi_boxes3d (B,M,7), gt_boxes3d (B,G,>=7, zero rows at the end); thresholds = (REG_FG, CLS_FG, CLS_BG, CLS_BG_LO).
    -> dict: rois, gt_of_rois (B,R,7), roi_iou (B,R), src (B,R) i32, max_overlaps (B,M), gt_assignment (B,M) i32, counts (B,4), status (B)"""
    _chk(roi_boxes3d, "roi_boxes3d", ndim=3); _chk(gt_boxes3d, "gt_boxes3d", ndim=3)
    B, M, _ = roi_boxes3d.shape
    G, gc = gt_boxes3d.shape[1], gt_boxes3d.shape[2]
    R, dev = int(roi_per_image), roi_boxes3d.device
    o = {"rois": torch.empty((B, R, 7), dtype=_F32, device=dev), "gt_of_rois": torch.empty((B, R, 7), dtype=_F32, device=dev),
         "roi_iou": torch.empty((B, R), dtype=_F32, device=dev), "src": torch.empty((B, R), dtype=_INT, device=dev),
         "max_overlaps": torch.empty((B, M), dtype=_F32, device=dev), "gt_assignment": torch.empty((B, M), dtype=_INT, device=dev),
         "counts": torch.empty((B, 4), dtype=_INT, device=dev), "status": torch.empty((B,), dtype=_INT, device=dev)}
    cfg6 = (ctypes.c_double * 6)(*[float(v) for v in tuple(thresholds) + (fg_ratio, hard_bg_ratio)])
    _cabi.check(_cabi.lib().prcnn_proposal_target_sample(_p(roi_boxes3d), _p(gt_boxes3d), B, M, G, gc, R, ctypes.cast(cfg6, ctypes.c_void_p),
                                                         int(aug_times), {"multiple": 0, "single": 1}[aug_method], seed & 0xFFFFFFFF,
                                                         _p(o["rois"]), _p(o["gt_of_rois"]), _p(o["roi_iou"]), _p(o["src"]),
                                                         _p(o["max_overlaps"]), _p(o["gt_assignment"]), _p(o["counts"]), _p(o["status"]),
                                                         _stream()), "prcnn_proposal_target_sample")
    return o


def nms_sorted(boxes_sorted, thresh, rotated=True, max_keep=0):
    """Greedy NMS over boxes already sorted by descending score, fully on device.
    -> keep (N) int64 (first num entries valid), num (1) int32.  No host sync.
    max_keep > 0 stops the sweep once that many boxes are kept (the proposal layer only uses the first
    RPN_POST_NMS_TOP_N of them, lib/rpn/proposal_layer.py:112): same leading entries, a fraction of the work."""
    _chk(boxes_sorted, "boxes", ndim=2)
    N = boxes_sorted.shape[0]
    L = _cabi.lib()
    keep = torch.empty((max(N, 1),), dtype=torch.int64, device=boxes_sorted.device)
    num = torch.empty((1,), dtype=_INT, device=boxes_sorted.device)
    wsb = L.prcnn_nms_workspace_bytes(N)
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=boxes_sorted.device)
    _cabi.check(L.prcnn_nms(_p(boxes_sorted), N, float(thresh), 0 if rotated else 1, int(max_keep), _p(keep), _p(num),
                            _p(ws), wsb, _stream()), "prcnn_nms")
    return keep, num


_NMS_STAGE = threading.local()


def nms_sorted_to_host(boxes_sorted, thresh, rotated, keep_cpu):
    """The shape the reference's pybind functions force (iou3d.cpp:73-120: `keep` is a CPU int64 tensor, the return value the number of
    kept boxes): greedy NMS over boxes sorted by descending score, the kept indices copied into `keep_cpu[:n]`, -> n (host int).
    ONE host synchronisation: the kept indices and their count leave the device in one asynchronous copy into a pinned staging buffer
    (one per host thread and device) queued behind the sweep, the host waits for the stream once and copies the n entries."""
    _chk(boxes_sorted, "boxes", ndim=2)
    N = boxes_sorted.shape[0]
    if N == 0:
        return 0
    dev = boxes_sorted.device
    stages = getattr(_NMS_STAGE, "buf", None)
    if stages is None:
        stages = _NMS_STAGE.buf = {}
    stage = stages.get(dev.index)
    if stage is None or stage.numel() < N + 1:
        stage = stages[dev.index] = torch.empty((max(2 * N, 8192) + 1,), dtype=torch.int64).pin_memory()
    L = _cabi.lib()
    wsb = L.prcnn_nms_workspace_bytes(N)
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=dev)
    out = torch.empty((N + 1,), dtype=torch.int64, device=dev)          # [0, N): kept indices, [N]: their count (low int32)
    stream = torch.cuda.current_stream(dev)
    _cabi.check(L.prcnn_nms(_p(boxes_sorted), N, float(thresh), 0 if rotated else 1, 0, _p(out), _p(out) + 8 * N,
                            _p(ws), wsb, stream.cuda_stream), "prcnn_nms")
    stage[:N + 1].copy_(out, non_blocking=True)
    stream.synchronize()
    n = int(stage[N:N + 1].view(torch.int32)[0])
    if n:
        keep_cpu[:n].copy_(stage[:n])
    return n


# ---------------------------------------------------------------------------------------------------------
# proposal stage
# ---------------------------------------------------------------------------------------------------------
def decode_bbox_target(roi, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size, get_xz_fine=True,
                       get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=False, y_to_bottom=False):
    """lib/utils/bbox_transform.py:24-121 as one kernel.  roi (N,3|7), pred_reg (N,C) -> (N,7).
    anchor_size: 3 host floats (h, w, l)."""
    _chk(roi, "roi", ndim=2)
    _chk(pred_reg, "pred_reg", ndim=2)
    if roi.shape[0] != pred_reg.shape[0]:
        raise ValueError("decode_bbox_target: %d rois vs %d regression rows" % (roi.shape[0], pred_reg.shape[0]))
    N, C = pred_reg.shape
    out = torch.empty((N, 7), dtype=torch.float32, device=pred_reg.device)
    anchor = (ctypes.c_float * 3)(*[float(a) for a in anchor_size])
    _cabi.check(_cabi.lib().prcnn_decode_bbox_target(
        _p(roi), roi.shape[1], _p(pred_reg), N, C, float(loc_scope), float(loc_bin_size), int(num_head_bin),
        ctypes.cast(anchor, ctypes.c_void_p), int(bool(get_xz_fine)), int(bool(get_y_by_bin)), float(loc_y_scope),
        float(loc_y_bin_size), int(bool(get_ry_fine)), int(bool(y_to_bottom)), _p(out), _stream()), "prcnn_decode_bbox_target")
    return out


def proposal_layer(scores, boxes3d, pre, post, nms_thresh, rotated=False, ranges=(0.0, 40.0, 80.0)):
    """lib/rpn/proposal_layer.py:35-141 for the whole batch, no host sync.
    scores (B,N) raw, boxes3d (B,N,7) decoded; pre/post = (area 1, area 2) top-n; ranges=None -> score based.
    -> rois (B, post[0]+post[1], 7), roi_scores (B, post[0]+post[1]) zero padded, count (B) int32."""
    _chk(scores, "scores", ndim=2)
    _chk(boxes3d, "boxes3d", ndim=3)
    B, N = scores.shape
    if tuple(boxes3d.shape) != (B, N, 7):
        raise ValueError("proposal_layer: boxes3d must be (%d, %d, 7), got %s" % (B, N, tuple(boxes3d.shape)))
    L = _cabi.lib()
    tot = int(post[0]) + int(post[1])
    dev = scores.device
    rois = torch.empty((B, tot, 7), dtype=torch.float32, device=dev)
    roi_scores = torch.empty((B, tot), dtype=torch.float32, device=dev)
    count = torch.empty((B,), dtype=_INT, device=dev)
    wsb = L.prcnn_proposal_workspace_bytes(B, N, max(pre), max(post))
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=dev)
    r = ranges if ranges is not None else (0.0, 0.0, 0.0)
    _cabi.check(L.prcnn_proposal_layer(_p(scores), _p(boxes3d), B, N, int(ranges is not None), float(r[0]), float(r[1]),
                                       float(r[2]), int(pre[0]), int(pre[1]), int(post[0]), int(post[1]), float(nms_thresh),
                                       0 if rotated else 1, _p(rois), _p(roi_scores), _p(count), _p(ws), wsb, _stream()),
                "prcnn_proposal_layer")
    return rois, roi_scores, count


def nms_batched(boxes3d, scores, valid=None, thresh=0.1, rotated=True, max_keep=0):
    """tools/eval_rcnn.py:600-614 for the whole batch: per frame select valid rows, order by score, greedy NMS.
    boxes3d (B,M,7), scores (B,M), valid (B,M) bool/uint8 or None -> keep (B,max_keep) int32 (-1 padded), num (B)."""
    _chk(boxes3d, "boxes3d", ndim=3)
    _chk(scores, "scores", ndim=2)
    B, M = scores.shape
    if tuple(boxes3d.shape) != (B, M, 7):
        raise ValueError("nms_batched: boxes3d must be (%d, %d, 7), got %s" % (B, M, tuple(boxes3d.shape)))
    if valid is not None:
        if tuple(valid.shape) != (B, M) or not valid.is_cuda:
            raise ValueError("nms_batched: valid must be a (%d, %d) device tensor" % (B, M))
        valid = valid.to(torch.uint8).contiguous()
    mk = M if max_keep <= 0 or max_keep > M else int(max_keep)
    L = _cabi.lib()
    dev = scores.device
    keep = torch.empty((B, max(mk, 1)), dtype=_INT, device=dev)
    num = torch.empty((B,), dtype=_INT, device=dev)
    wsb = L.prcnn_nms_batched_workspace_bytes(B, M)
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=dev)
    _cabi.check(L.prcnn_nms_batched(_p(boxes3d), _p(scores), _p(valid) if valid is not None else None, B, M, float(thresh),
                                    0 if rotated else 1, mk, _p(keep), _p(num), _p(ws), wsb, _stream()), "prcnn_nms_batched")
    return keep[:, :mk], num


# ---------------------------------------------------------------------------------------------------------
# KITTI evaluation kernels (host logic in pointrcnn_amd/kitti_eval.py)
# ---------------------------------------------------------------------------------------------------------
def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """rotate_iou.py:287-329 rotate_iou_gpu_eval: (N,5), (K,5) [cx, cy, dx, dy, angle] -> (N,K)"""
    _chk(boxes, "boxes", ndim=2); _chk(query_boxes, "query_boxes", ndim=2)
    N, K = boxes.shape[0], query_boxes.shape[0]
    out = torch.zeros((N, K), dtype=_F32, device=boxes.device)
    _cabi.check(_cabi.lib().prcnn_rotate_iou_eval(_p(boxes), N, _p(query_boxes), K, int(criterion), _p(out), _stream()),
                "prcnn_rotate_iou_eval")
    return out


def kitti_overlaps(metric, dt, dt_off, gt, gt_off, ov_off, total):
    """per-frame overlap blocks (rows = dt, columns = gt) of all frames, flat float64 (see include/prcnn_pointops.h)"""
    F = dt_off.shape[0] - 1
    out = torch.zeros((max(int(total), 1),), dtype=torch.float64, device=dt_off.device)
    _cabi.check(_cabi.lib().prcnn_kitti_overlaps(int(metric), _p(dt), _p(dt_off), _p(gt), _p(gt_off), _p(ov_off), F, _p(out), _stream()),
                "prcnn_kitti_overlaps")
    return out[:int(total)]


def kitti_statistics(overlaps, ov_off, gt_datas, gt_off, dt_datas, dt_off, ign_gt, ign_det, dc, dc_off, metric, min_overlap,
                     thresholds, compute_fp, compute_aos):
    """compute_statistics_jit for every (frame, threshold) -> res (F,T,4) f64, matched (G) f64"""
    F, T = gt_off.shape[0] - 1, thresholds.shape[0]
    dev = gt_off.device
    res = torch.zeros((F, T, 4), dtype=torch.float64, device=dev)
    G = gt_datas.shape[0]
    matched = torch.full((max(G, 1),), float("nan"), dtype=torch.float64, device=dev)
    max_det = int((dt_off[1:] - dt_off[:-1]).max().item()) if F > 0 else 0
    _cabi.check(_cabi.lib().prcnn_kitti_statistics(_p(overlaps), _p(ov_off), _p(gt_datas), _p(gt_off), _p(dt_datas), _p(dt_off),
                                                   _p(ign_gt), _p(ign_det), _p(dc), _p(dc_off), F, max_det, int(metric),
                                                   float(min_overlap), _p(thresholds), T, int(compute_fp), int(compute_aos), _p(res),
                                                   _p(matched), _stream()), "prcnn_kitti_statistics")
    return res, matched[:G]


# ---------------------------------------------------------------------------------------------------------
# padding-free grouping (csrc/dedup.hip)
# ---------------------------------------------------------------------------------------------------------
_split_log = None      # bench.py's instrumented pass sets this to a list to read the device-side list lengths afterwards


class GroupSplit:
    """The G = B*M groups of one ball query split on the device (prcnn_group_compact) into a flat list of the real rows of
    the sparse groups (at most sparse_max hits) and the list of dense groups.  All tensors are worst-case sized; counts
    (3,) int32 holds [flat rows, dense groups, sparse groups]."""

    def __init__(self, idx, new_xyz, N, sparse_max, valid_n=None):
        _chk(idx, "idx", _INT, 3)
        _chk(new_xyz, "new_xyz", ndim=3)
        B, M, ns = idx.shape
        G, dev = B * M, idx.device
        T = max(1, min(int(sparse_max), ns))
        self.G, self.ns, self.max_rows = G, ns, G * T
        e = lambda shape, dt=_INT: torch.empty(shape, dtype=dt, device=dev)      # noqa: E731
        self.ridx, self.rnx = e((1, G * T, 1)), e((1, G * T, 3), _F32)
        self.slist, self.soff, self.scnt = e((G,)), e((G,)), e((G,))
        self.idxn, self.nxn, self.listn = e((1, G, ns)), e((1, G, 3), _F32), e((G,))
        self.counts = e((3,))
        if valid_n is not None and (valid_n.dtype != _INT or valid_n.numel() != B or not valid_n.is_contiguous()):
            raise RuntimeError("valid_n must be a contiguous (B,) int32 tensor")
        _cabi.check(_cabi.lib().prcnn_group_compact(_p(idx), _p(new_xyz), B, N, M, ns, T, _p(valid_n), _p(self.ridx), _p(self.rnx), _p(self.slist),
                                                    _p(self.soff), _p(self.scnt), _p(self.idxn), _p(self.nxn), _p(self.listn),
                                                    _p(self.counts), _stream()), "prcnn_group_compact")
        self.rows, self.count_dense, self.count_sparse = self.counts[0:1], self.counts[1:2], self.counts[2:3]
        if _split_log is not None:
            _split_log.append(self)


def segmax_scatter(src, sp, dst, col_off):
    """dst[g, col_off : col_off + C] = max over the flat result rows of every sparse group g of the GroupSplit sp"""
    C = src.shape[-1]
    _cabi.check(_cabi.lib().prcnn_segmax_scatter(_p(src), src.stride(-2), _p(sp.slist), _p(sp.soff), _p(sp.scnt), _p(sp.count_sparse),
                                                 sp.G, C, _p(dst), dst.stride(-2), int(col_off), _stream()), "prcnn_segmax_scatter")


def scatter_rows(src, rows_list, count, dst, col_off):
    """dst[rows_list[r], col_off : col_off + C] = src[r] for r < count (count: (1,) int32 device tensor)"""
    C = src.shape[-1]
    _cabi.check(_cabi.lib().prcnn_scatter_rows(_p(src), src.stride(-2), _p(rows_list), _p(count), rows_list.shape[0], C, _p(dst),
                                               dst.stride(-2), int(col_off), _stream()), "prcnn_scatter_rows")


# ---------------------------------------------------------------------------------------------------------
# RPN input builder (csrc/scene.hip)
# ---------------------------------------------------------------------------------------------------------
def scene_prepare(raw, offsets, max_points_per_frame, calib, img_hw, scope, npoints, seed):
    """raw (total,4) f32 scans back to back, offsets (B+1) i64, calib (B,24) f32 [M 4x3 | P2 3x4], img_hw (B,2) i32 -- all on
    the device; scope: 6 host floats [x0 x1 y0 y1 z0 z1] or None.
    -> pts_rect (B,npoints,3), intensity - 0.5 (B,npoints), src index (B,npoints) i32, nvalid (B) i32, status (B) i32
    [lib/datasets/kitti_rcnn_dataset.py:246-310 for a whole batch; see prcnn_scene_prepare]"""
    _chk(raw, "raw", ndim=2)
    _chk(calib, "calib", ndim=2)
    _chk(img_hw, "img_hw", _INT, 2)
    if offsets.dtype != torch.int64 or not offsets.is_contiguous() or offsets.device != raw.device:
        raise RuntimeError("offsets must be a contiguous int64 tensor on the device of raw")
    if raw.shape[1] != 4 or calib.shape[1] != 24 or img_hw.shape[1] != 2:
        raise RuntimeError("expected raw (total,4), calib (B,24), img_hw (B,2)")
    B, total, dev = offsets.shape[0] - 1, raw.shape[0], raw.device
    if calib.shape[0] != B or img_hw.shape[0] != B:
        raise RuntimeError("calib / img_hw must have one row per frame")
    L = _cabi.lib()
    ws = torch.empty((int(L.prcnn_scene_workspace_bytes(total, B)),), dtype=torch.uint8, device=dev)
    xyz = torch.empty((B, npoints, 3), dtype=_F32, device=dev)
    inten = torch.empty((B, npoints), dtype=_F32, device=dev)
    src = torch.empty((B, npoints), dtype=_INT, device=dev)
    nvalid = torch.empty((B,), dtype=_INT, device=dev)
    status = torch.empty((B,), dtype=_INT, device=dev)
    sc = None if scope is None else (ctypes.c_double * 6)(*[float(v) for v in scope])
    _cabi.check(L.prcnn_scene_prepare(_p(raw), _p(offsets), B, total, int(max_points_per_frame), _p(calib), _p(img_hw), sc, int(npoints),
                                      int(seed) & 0xFFFFFFFF, _p(xyz), _p(inten), _p(src), _p(nvalid), _p(status), _p(ws), ws.numel(),
                                      _stream()), "prcnn_scene_prepare")
    return xyz, inten, src, nvalid, status
