"""Shared by the CPU and GPU proposal-stage tests: the golden cases (tests/golden/make_golden.py) and the
reference's top-n arithmetic (lib/rpn/proposal_layer.py:64-68)."""
import os
import sys
import zlib

import numpy as np

from util import GOLDEN, rpn_like_scene

sys.path.insert(0, GOLDEN)
from make_golden import proposal_cases  # noqa: E402

# tools/cfgs/default.yaml:156-158,163-165
MODES = {"TEST": dict(pre=9000, post=100, thresh=0.8), "TRAIN": dict(pre=9000, post=512, thresh=0.85)}


def split_top_n(tot):
    """proposal_layer.py:66,68: [int(tot * 0.7), tot - int(tot * 0.7)]"""
    a = int(tot * 0.7)
    return a, tot - a


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


def golden():
    return np.load(os.path.join(GOLDEN, "proposal_ref.npz"))


def case_inputs(name):
    c = proposal_cases()[name]
    xyz, sc, reg = rpn_like_scene(c["B"], c["N"], seed=c["seed"], z_max=c["z_max"])
    m = MODES[c["mode"]]
    if c["distance"]:
        pre, post, ranges = split_top_n(m["pre"]), split_top_n(m["post"]), (0.0, 40.0, 80.0)
        kind = "rotated" if c["nms"] == "rotate" else "normal"
    else:                                    # score_based_proposal always calls nms_gpu (proposal_layer.py:135)
        pre, post, ranges, kind = (m["pre"], 0), (m["post"], 0), None, "rotated"
    return xyz, sc, reg, dict(pre=pre, post=post, thresh=m["thresh"], kind=kind, ranges=ranges)
