"""CPU port of ONE frame of the RPN TRAINING step -- TEST / MEASUREMENT INFRASTRUCTURE (bench.py's `cpu_baseline` leg of the
training lines only; nothing in pointrcnn_amd/ imports this).

The reference has no CPU path for training (its point operators are CUDA-only, lib/net/train_functions.py:13-120 runs on the GPU),
so the baseline is assembled like the inference one (oracle/rpn_cpu.py): index operators = the C oracle (FPS, ball query, 3-NN: no
gradient flows through them), everything that carries gradients = torch on the CPU with autograd -- grouping as an index_select,
SharedMLP layers as linear + training-mode batch_norm + ReLU on the nsample-PADDED grouped rows (what the reference's
Conv2d(1x1) + BatchNorm2d + ReLU compute), max-pool over nsample, 3-NN interpolation, the two heads -- then backward through all of
it and a plain SGD update of every weight.  Follows lib/net/pointnet2_msg.py:56-70 and lib/net/rpn.py:68-82 for the graph and
SURVEY.md Appendix A.4-A.6 for the module semantics.  The loss is a PROXY (mean squares of the head outputs: the reference's
focal + bin-based regression loss is a few hundred kFLOP per frame, immaterial next to the 45 GFLOP of the stacks); batch
statistics are taken over the frame's own rows (one frame per worker process).  Same operations, same shapes, same FLOPs as the
GPU step: a timing baseline, not a parity reference."""
import time

import numpy as np


def make_params(spec):
    """spec: oracle.rpn_cpu.extract_rpn_weights(model) -> the same nesting with torch leaf tensors (weight, gamma, beta)"""
    import torch

    def layer(l):
        w, b, relu = l
        n = w.shape[0]
        return {"w": torch.from_numpy(np.ascontiguousarray(w)).requires_grad_(True), "g": torch.ones(n, requires_grad=True),
                "b": torch.zeros(n, requires_grad=True), "relu": relu, "bn": relu}
    return {"sa": [{"npoint": lv["npoint"], "scales": [{"radius": sc["radius"], "nsample": sc["nsample"], "layers": [layer(l) for l in sc["layers"]]}
                                                       for sc in lv["scales"]]} for lv in spec["sa"]],
            "fp": [[layer(l) for l in fp] for fp in spec["fp"]], "cls": [layer(l) for l in spec["cls"]], "reg": [layer(l) for l in spec["reg"]]}


def leaves(params):
    out = []
    for lv in params["sa"]:
        for sc in lv["scales"]:
            out += sc["layers"]
    for fp in params["fp"]:
        out += fp
    out += params["cls"] + params["reg"]
    return [t for l in out for t in (l["w"], l["g"], l["b"])]


def _stack(x, layers):
    import torch
    import torch.nn.functional as F
    for l in layers:
        x = F.linear(x, l["w"])
        if l["bn"]:
            x = F.batch_norm(x, None, None, l["g"], l["b"], True, 0.1, 1e-5)
            x = torch.relu(x)
        else:
            x = x + l["b"]
    return x


def rpn_train_frame(cpu, xyz, params, lr=1e-7, timings=None):
    """one frame: forward, proxy loss, backward, SGD update in place; -> loss (float)"""
    import torch
    t = timings if timings is not None else {}

    def tick(name, t0):
        t[name] = t.get(name, 0.0) + time.perf_counter() - t0
    l_xyz = [np.ascontiguousarray(xyz, dtype=np.float32)]
    l_feat = [None]
    t_fwd = time.perf_counter()
    for lv in params["sa"]:
        p = l_xyz[-1]
        t0 = time.perf_counter()
        fidx = cpu.fps(p[None], lv["npoint"])[0]
        tick("fps", t0)
        new_xyz = p[fidx]
        outs = []
        for sc in lv["scales"]:
            t0 = time.perf_counter()
            idx = cpu.ball_query(sc["radius"], sc["nsample"], p[None], new_xyz[None])[0]              # (M, ns)
            tick("ball_query", t0)
            t0 = time.perf_counter()
            it = torch.from_numpy(idx.astype(np.int64)).reshape(-1)
            gx = torch.from_numpy(p[idx] - new_xyz[:, None, :]).reshape(-1, 3)
            rows = gx if l_feat[-1] is None else torch.cat([gx, l_feat[-1].index_select(0, it)], 1)
            rows = _stack(rows, sc["layers"])
            outs.append(rows.view(lv["npoint"], sc["nsample"], -1).max(1)[0])
            tick("stack_fwd", t0)
        l_xyz.append(new_xyz)
        l_feat.append(torch.cat(outs, 1))
    for i in range(-1, -(len(params["fp"]) + 1), -1):
        unknown, known = l_xyz[i - 1], l_xyz[i]
        t0 = time.perf_counter()
        d2, idx3 = cpu.three_nn(unknown[None], known[None])
        w3 = torch.from_numpy(cpu.three_weights(d2)[0])
        tick("three_nn", t0)
        t0 = time.perf_counter()
        i3 = torch.from_numpy(idx3[0].astype(np.int64))
        f = l_feat[i]
        interp = (f.index_select(0, i3[:, 0]) * w3[:, 0:1] + f.index_select(0, i3[:, 1]) * w3[:, 1:2]) + f.index_select(0, i3[:, 2]) * w3[:, 2:3]
        rows = interp if l_feat[i - 1] is None else torch.cat([interp, l_feat[i - 1]], 1)
        l_feat[i - 1] = _stack(rows, params["fp"][i])
        tick("stack_fwd", t0)
    t0 = time.perf_counter()
    feats = l_feat[0]
    cls, reg = _stack(feats, params["cls"]), _stack(feats, params["reg"])
    loss = cls.square().mean() + reg.square().mean()
    tick("stack_fwd", t0)
    t["forward"] = t.get("forward", 0.0) + time.perf_counter() - t_fwd
    t0 = time.perf_counter()
    ps = leaves(params)
    grads = torch.autograd.grad(loss, ps, allow_unused=True)
    tick("backward", t0)
    t0 = time.perf_counter()
    with torch.no_grad():
        for p_, g_ in zip(ps, grads):
            if g_ is not None:
                p_.add_(g_, alpha=-lr)
    tick("update", t0)
    return float(loss.detach())
