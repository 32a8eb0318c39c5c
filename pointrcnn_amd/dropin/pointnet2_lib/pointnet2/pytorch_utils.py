"""`pointnet2_lib.pointnet2.pytorch_utils` -- SharedMLP / Conv1d / Conv2d / FC building blocks with the
attribute names the released checkpoint's state-dict keys need (`layer{k}.conv.weight`, `layer{k}.bn.bn.*`;
in-tree evidence lib/net/rpn.py:64,66, lib/net/rcnn_net.py:104) [UPSTREAM module, absent from the tree].

Inference fast path (MI355X): when autograd is off, BatchNorm is in eval mode and the input lives on the GPU,
a 1x1 conv (+BN +ReLU) layer is ONE fused fp32-MFMA kernel on channels-last rows (pointrcnn_amd.ops.mlp_rows)
with BN folded into the packed weights; activations stay channels-last between layers and are handed to the
caller as transposed VIEWS, so a chain of these modules never pays a layout change.  Otherwise (training,
CPU construction) the modules behave as plain torch Conv/BN/ReLU stacks.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from pointrcnn_amd import ops


def _fold_bn(conv, bn):
    """(Nout,K) weight and (Nout) bias of conv followed by eval-mode batch norm."""
    w = conv.weight.detach().reshape(conv.out_channels, -1).float()
    b = conv.bias.detach().float() if conv.bias is not None else None
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w = w * scale[:, None]
        b = shift if b is None else b * scale + shift
    return w.contiguous(), (None if b is None else b.contiguous())


def _rows_view(x_cl):
    """x_cl: (..., C) view with unit last stride -> tensor usable as uniformly strided rows (copy if needed)."""
    if x_cl.stride(-1) != 1:
        if x_cl.shape[-1] != 1:
            return x_cl.contiguous()
        # a single channel: the size-1 last dimension keeps whatever stride its history left it (torch calls the tensor contiguous
        # and .contiguous() returns it unchanged) -- give it the unit stride the kernels' callers test for
        x_cl = x_cl.as_strided(x_cl.shape, tuple(x_cl.stride()[:-1]) + (1,), x_cl.storage_offset())
    ld = None
    expect = None
    for size, stride in zip(reversed(x_cl.shape[:-1]), reversed(x_cl.stride()[:-1])):
        if size == 1:
            continue
        if ld is None:
            ld, expect = stride, stride * size
        elif stride != expect:
            return x_cl.contiguous()
        else:
            expect = stride * size
    if ld is not None and ld < x_cl.shape[-1]:
        return x_cl.contiguous()
    return x_cl


def run_conv_stack(mods, x):
    """Fused inference of consecutive 1x1 conv(+BN+ReLU) layers `mods` (all `fusable()`), x = (B,C,L) or (B,C,H,W).
    One register-chain kernel for the whole stack when an instance exists, else one kernel per layer.
    Returns a (B,C_out,...) VIEW of a channels-last buffer."""
    nd = x.dim()
    perm = (0, 2, 1) if nd == 3 else (0, 2, 3, 1)
    x_cl = _rows_view(x.permute(*perm))
    layers = [m.packed() for m in mods]
    if len(layers) > 1 and ops.chain_supported(0, layers, 0):
        y = ops.mlp_chain_rows(x_cl, layers)
    else:
        y = x_cl
        for lin in layers:
            y = ops.mlp_rows(y, lin)
    y = y.view(*x_cl.shape[:-1], y.shape[-1])
    return y.permute(0, 2, 1) if nd == 3 else y.permute(0, 3, 1, 2)


def _inference_input(x):
    return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32


def fused_sequential(seq, x):
    """Run an nn.Sequential of pt_utils conv layers (+ eval-mode Dropout) -- e.g. the RPN heads, lib/net/rpn.py:20-46
    -- through run_conv_stack when every member allows it; otherwise exactly seq(x)."""
    if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3:
        y = _train_rows_sequential(seq, x)
        return seq(x) if y is None else y
    mods = []
    for m in seq:
        if isinstance(m, nn.Dropout):
            if m.training:
                return seq(x)
            continue
        if not (isinstance(m, _ConvBase) and m.fusable()):
            return seq(x)
        mods.append(m)
    if not mods or not _inference_input(x):
        return seq(x)
    return run_conv_stack(mods, x)


class _LinearRows(torch.autograd.Function):
    """y = x W^T + b on (R, K) rows with R >> K, N (an output layer over every point of the batch).  The forward and the input
    gradient are plain library GEMMs; the weight gradient g^T x reduces over R = 262144 rows into a 76 x 128 result, which the
    library runs as ONE skinny GEMM without splitting the reduction (640 us, 8 TFLOP/s): here it is a batched product over row
    chunks followed by a sum over the chunks (split-K by hand, deterministic)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            R = x.shape[0]
            chunks = 1
            while chunks < 512 and R % (chunks * 2) == 0 and R // (chunks * 2) >= 512:
                chunks *= 2
            if chunks > 1 and x.is_contiguous():
                gw = torch.bmm(g.view(chunks, R // chunks, -1).transpose(1, 2), x.view(chunks, R // chunks, -1)).sum(0)
            else:
                gw = g.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def _train_rows_sequential(seq, x):
    """Training: a Sequential of 1 x 1 Conv1d layers and Dropout on (B, C, L) (the RPN / RCNN heads, lib/net/rpn.py:20-46) computed
    on channels-last ROWS from end to end: runs of Conv -> BatchNorm -> ReLU are one hand-written autograd node each
    (train_mlp.SharedMLPTrain), Dropout is elementwise on the rows, a bare Conv (bias, no norm: the output layer) is F.linear on the
    rows.  Module by module the Sequential would transpose (B, C, L) <-> rows around every fused layer and run the output layer as
    an MIOpen convolution forward + backward: four full-size copies and 1.2 ms of a 15 ms step for a 128 -> 76 product.
    The result is returned as a (B, C_out, L) VIEW of the (B, L, C_out) rows, so the caller's `.transpose(1, 2).contiguous()` is
    free.  None: some member is not covered (the caller runs seq(x))."""
    from pointrcnn_amd import train_mlp
    from . import pointnet2_modules
    if not pointnet2_modules.TRAIN_FUSED:
        return None
    plan, run = [], []
    for m in seq:
        if isinstance(m, nn.Dropout):
            if run:
                plan.append(("stack", run))
                run = []
            plan.append(("dropout", m))
        elif isinstance(m, _ConvBase) and isinstance(m._parts()[0], nn.Conv1d) and train_mlp.stack_ok([m]):
            run.append(m)
        elif isinstance(m, _ConvBase) and isinstance(m._parts()[0], nn.Conv1d) and m._prcnn_fusable and m._parts()[1] is None:
            conv = m._parts()[0]
            if conv.kernel_size != (1,) or conv.stride != (1,) or conv.padding != (0,) or conv.groups != 1:
                return None
            if run:
                plan.append(("stack", run))
                run = []
            plan.append(("linear", m))
        else:
            return None
    if run:
        plan.append(("stack", run))
    if not any(k == "stack" for k, _ in plan):
        return None
    B, C, L = x.shape
    rows = _rows_view(x.permute(0, 2, 1)).reshape(B * L, C)
    for kind, m in plan:
        if kind == "stack":
            rows = train_mlp.run_stack(m, train_mlp.Source("plain"), rows)
        elif kind == "dropout":
            rows = F.dropout(rows, m.p, m.training, False)
        else:
            conv, _, act = m._parts()
            rows = _LinearRows.apply(rows, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
            if act is not None:
                rows = F.relu(rows) if isinstance(act, nn.ReLU) else act(rows)
    return rows.view(B, L, -1).permute(0, 2, 1)


class _BNBase(nn.Sequential):
    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class _ConvBase(nn.Sequential):
    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=None,
                 batch_norm=None, bias=True, preact=False, name="", instance_norm=False, instance_norm_func=None):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        bn_unit = None
        if bn:
            bn_unit = batch_norm(out_size if not preact else in_size)
        in_unit = None
        if instance_norm:
            in_unit = instance_norm_func(out_size if not preact else in_size, affine=False,
                                         track_running_stats=False)
        if preact:
            if bn:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)
            if not bn and instance_norm:
                self.add_module(name + "in", in_unit)
        self.add_module(name + "conv", conv_unit)
        if not preact:
            if bn:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)
            if not bn and instance_norm:
                self.add_module(name + "in", in_unit)
        self._prcnn_name = name
        self._prcnn_fusable = (not preact) and (not instance_norm) and \
            (activation is None or isinstance(activation, nn.ReLU))
        self._prcnn_cache = {}

    # ---- fused inference path -------------------------------------------------------------
    def _parts(self):
        n = self._prcnn_name
        conv = getattr(self, n + "conv")
        bnw = getattr(self, n + "bn", None)
        bn = bnw[0] if bnw is not None else None
        act = getattr(self, n + "activation", None)
        return conv, bn, act

    def fusable(self):
        if not self._prcnn_fusable:
            return False
        conv, bn, _ = self._parts()
        ks = conv.kernel_size if isinstance(conv.kernel_size, tuple) else (conv.kernel_size,)
        st = conv.stride if isinstance(conv.stride, tuple) else (conv.stride,)
        pd = conv.padding if isinstance(conv.padding, tuple) else (conv.padding,)
        if any(k != 1 for k in ks) or any(s != 1 for s in st) or any(p != 0 for p in pd):
            return False
        if bn is not None and bn.training:
            return False
        return conv.weight.is_cuda and conv.weight.dtype == torch.float32

    def packed(self, k_rot=0):
        """PackedLinear of this layer (BN folded); cached until a parameter/buffer changes."""
        conv, bn, act = self._parts()
        tensors = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if bn is not None:
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = (k_rot,) + tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._prcnn_cache.get(k_rot)
        if hit is not None and hit[0] == key:
            return hit[1]
        w, b = _fold_bn(conv, bn)
        lin = ops.PackedLinear(w, b, relu=act is not None, k_rot=k_rot)
        self._prcnn_cache[k_rot] = (key, lin)
        return lin

    def _folded(self):
        """(w (Nout,K), b (Nout) or None, has_relu, cache key) of conv + eval-BN"""
        conv, bn, act = self._parts()
        tensors = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if bn is not None:
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        w, b = _fold_bn(conv, bn)
        return w, b, act is not None, key

    def _cached(self, tag, build):
        conv, bn, _ = self._parts()
        tensors = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if bn is not None:
            tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._prcnn_cache.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        val = build()
        self._prcnn_cache[tag] = (key, val)
        return val

    def hoisted_group(self):
        """First layer of an SA stack, split for hoisting: W = [W_x (dxyz, 3 cols) | W_f (features)] in torch's grouped
        channel order.  -> (w_f (N0,C) tensor, act_wx (N0,3), act_bias (N0)) or None when not applicable."""
        def build():
            w, b, relu, _ = self._folded()
            n0, k = w.shape
            if not relu or k <= 3 or n0 % 4 or (k - 3) % 4:
                return None
            bias = b if b is not None else torch.zeros(n0, device=w.device)
            return (w[:, 3:].contiguous(), w[:, :3].contiguous(), bias.contiguous())
        return self._cached("hoist_group", build)

    def hoisted_interp(self, c2):
        """First layer of an FP stack, split for hoisting: W = [W_a (interpolated, c2 cols) | W_b (skip)].
        -> (lin_a: PackedLinear(W_a, no bias, no act), lin_b: PackedLinear(W_b, bias, relu) or None, bias (N0)) or None"""
        def build():
            w, b, relu, _ = self._folded()
            n0, k = w.shape
            if not relu or n0 % 4 or c2 % 4 or c2 > k:
                return None
            bias = (b if b is not None else torch.zeros(n0, device=w.device)).contiguous()
            lin_a = ops.PackedLinear(w[:, :c2].contiguous(), None, relu=False)
            lin_b = ops.PackedLinear(w[:, c2:].contiguous(), bias, relu=True) if k > c2 else None
            return (lin_a, lin_b, bias)
        return self._cached(("hoist_interp", c2), build)

    def forward(self, x):
        if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3:
            # training: Conv1d -> BatchNorm1d -> ReLU on (B, C, L) as one hand-written autograd node (train_mlp.py)
            from pointrcnn_amd import train_mlp
            from . import pointnet2_modules
            if pointnet2_modules.TRAIN_FUSED and train_mlp.stack_ok([self]):
                B, C, L = x.shape
                rows = _rows_view(x.permute(0, 2, 1))
                y = train_mlp.run_stack([self], train_mlp.Source("plain"), rows.reshape(B * L, C))
                return y.view(B, L, -1).permute(0, 2, 1)
        if torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or not self.fusable():
            return super().forward(x)
        return run_conv_stack([self], x)


class Conv1d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0, activation=nn.ReLU(inplace=True),
                 bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False, name="", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv1d,
                         batch_norm=BatchNorm1d, bias=bias, preact=preact, name=name,
                         instance_norm=instance_norm, instance_norm_func=nn.InstanceNorm1d)


class Conv2d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True, preact=False,
                 name="", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv2d,
                         batch_norm=BatchNorm2d, bias=bias, preact=preact, name=name,
                         instance_norm=instance_norm, instance_norm_func=nn.InstanceNorm2d)


class SharedMLP(nn.Sequential):
    """args = [C_in, C_1, ..., C_out]: a stack of 1x1 Conv2d(+BN)+ReLU layers named layer0, layer1, ..."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False, name="",
                 instance_norm=False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=plain and bn, activation=activation if plain else None,
                                   preact=preact, instance_norm=instance_norm))

    def layers(self):
        return [m for m in self.children()]

    def fusable(self):
        return all(isinstance(m, _ConvBase) and m.fusable() for m in self.children())

    def forward(self, x):
        if _inference_input(x) and self.fusable():
            return run_conv_stack(self.layers(), x)
        if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[3] == 1:
            # training on a (B, C, N, 1) "image" of per-point rows (rcnn_net.py:160-166 xyz_up_layer / merge_down_layer): the whole
            # stack as one hand-written autograd node on channels-last rows
            from pointrcnn_amd import train_mlp
            from . import pointnet2_modules
            if pointnet2_modules.TRAIN_FUSED and train_mlp.stack_ok(self.layers()):
                B, C, N, _ = x.shape
                rows = _rows_view(x.squeeze(3).permute(0, 2, 1)).reshape(B * N, C)
                y = train_mlp.run_stack(self.layers(), train_mlp.Source("plain"), rows)
                return y.view(B, N, -1).permute(0, 2, 1).unsqueeze(3)
        return super().forward(x)


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None, preact=False,
                 name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(in_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "fc", fc)
        if not preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
