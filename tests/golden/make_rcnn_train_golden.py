#!/usr/bin/env python3
"""Generate tests/golden/train_rcnn_ref.npz: the reference's OWN RCNN-stage loss (lib/net/train_functions.py:122-214 get_rcnn_loss
+ lib/utils/loss_utils.py, run on CPU through tests/golden/ref_net.py) on seeded network outputs and ProposalTargetLayer-style
targets: loss value, tensorboard entries, gradients.  Needs /root/reference; the fixture travels."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_net  # noqa: E402
from make_golden import crc  # noqa: E402

CASES = {"bce": (11, "BinaryCrossEntropy", 40), "focal": (12, "SigmoidFocalLoss", 40), "nofg": (13, "BinaryCrossEntropy", 0)}


def main():
    g = {}
    for name, (seed, loss_cls, nfg) in CASES.items():
        case = ref_net.rcnn_loss_case(seed, nfg=nfg, nign=16 if nfg else 0)
        loss, tb, gc, gr = ref_net.reference_rcnn_loss(*case, loss_cls=loss_cls)
        g[name + "_loss"], g[name + "_gcls"], g[name + "_greg"] = np.float64(loss), gc, gr
        g[name + "_tb_keys"], g[name + "_tb_vals"] = np.array(sorted(tb)), np.array([float(tb[k]) for k in sorted(tb)])
        g[name + "_crc"] = crc(*case)
        print(name, loss, {k: round(float(v), 5) for k, v in tb.items() if k.startswith("rcnn")})
    np.savez_compressed(os.path.join(HERE, "train_rcnn_ref.npz"), **g)


if __name__ == "__main__":
    main()
