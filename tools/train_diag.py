"""dev probe: where do the GPU training-step gradients differ from the CPU golden (tests/golden/train_ref.npz)?"""
import os
import sys
import numpy as np
import torch
sys.path[:0] = [".", "tests", "tests/golden"]
from make_golden import TRAIN_CASE, train_batch
import cpu_ops
from pointrcnn_amd import rpn, train_functions as tf

dev = torch.device("cuda:0")
g = np.load("tests/golden/train_ref.npz")
pts, gt, cls, reg = train_batch(TRAIN_CASE)


class W(torch.nn.Module):
    def __init__(self, r):
        super().__init__()
        self.rpn = r


def run(tag):
    model = rpn.RPN()
    cpu_ops.fill_params_by_name(W(model), TRAIN_CASE["wseed"])
    model = model.to(dev)
    tr = tf.RPNTrainer(model, ddp=False)
    tr.model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    T = lambda a: torch.from_numpy(a).to(dev)
    loss = tr.loss({"pts_input": T(pts), "rpn_cls_label": T(cls), "rpn_reg_label": T(reg)})
    loss.backward()
    params = {"rpn." + n: p for n, p in model.named_parameters()}
    names = g["step_names"].tolist()
    rel_norm, rel_samp, worst = 0.0, 0.0, None
    table = []
    for i, n in enumerate(names):
        gr = params[n].grad
        rn = abs(float(gr.double().norm()) - float(g["step_gnorm"][i])) / float(g["step_gnorm"][i])
        got = np.resize(gr.reshape(-1)[:8].cpu().numpy(), 8)
        rs = float(np.abs(got - g["step_gsample"][i]).max() / max(np.abs(g["step_gsample"][i]).max(), 1e-12))
        if rs > rel_samp:
            rel_samp, worst = rs, n
        rel_norm = max(rel_norm, rn)
        table.append((n, rn, rs, float(g["step_gnorm"][i])))
    if tag == "default":
        for n, rn, rs, gn in table:
            print("   %-62s |g| %.3e  rel norm err %.1e  rel sample err %.1e" % (n, gn, rn, rs))
    print("%-28s loss %.7f (golden %.7f)  worst rel grad-norm err %.2e  worst rel sample err %.2e at %s" %
          (tag, float(loss.item()), float(g["step_loss"]), rel_norm, rel_samp, worst), flush=True)


run("default")
run("default again (run-to-run)")
