"""bench.py's cpu_baseline legs (oracle/cpu_baseline.py, oracle/rpn_cpu.py, oracle/rpn_train_cpu.py) on a small configuration:
they must run on the host alone (no HIP state), print one JSON line, and the training port must actually train (gradients reach
every stack, the update changes the loss)."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_spec():
    import pointrcnn_amd
    pointrcnn_amd.install()
    from oracle import rpn_cpu
    from pointrcnn_amd import rpn
    cfg = type("Cfg", (rpn.RPNConfig,), {"SA_NPOINTS": [256, 64, 16, 4], "NUM_POINTS": 1024})
    import torch
    torch.manual_seed(0)
    return rpn_cpu.extract_rpn_weights(rpn.randomize_bn_stats(rpn.RPN(cfg=cfg), seed=1)), rpn.synthetic_clouds(4, 1024).numpy()


def test_training_port_has_gradients_everywhere_and_updates(cpu):
    import torch
    from oracle import rpn_train_cpu
    spec, clouds = _small_spec()
    params = rpn_train_cpu.make_params(spec)
    before = [p.detach().clone() for p in rpn_train_cpu.leaves(params)]
    t = {}
    l1 = rpn_train_cpu.rpn_train_frame(cpu, clouds[0], params, lr=10.0, timings=t)        # (a large step: tiny gradients must still move fp32 weights)
    after = rpn_train_cpu.leaves(params)
    assert np.isfinite(l1) and {"fps", "ball_query", "three_nn", "stack_fwd", "backward", "update"} <= set(t)
    moved = [not torch.equal(a, b) for a, b in zip(after, before)]
    # gradients reach the stacks of every kind (on this toy cloud a few are legitimately zero: 0.1 m balls around 1024 points hold
    # one point each, so SA1's first scale sees identical rows; 4-point levels have degenerate batch statistics)
    assert sum(moved) >= int(0.6 * len(moved)), "only %d of %d tensors changed" % (sum(moved), len(moved))
    n_layers = sum(len(sc["layers"]) for lv in spec["sa"] for sc in lv["scales"]) + sum(len(f) for f in spec["fp"]) + len(spec["cls"]) + len(spec["reg"])
    assert len(after) == 3 * n_layers


@pytest.mark.parametrize("train", [False, True])
def test_cpu_baseline_subprocess_prints_one_json_line(tmp_path, train):
    spec, clouds = _small_spec()
    with open(tmp_path / "spec.pkl", "wb") as f:
        pickle.dump(spec, f)
    np.save(tmp_path / "clouds.npy", clouds)
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--spec", str(tmp_path / "spec.pkl"), "--clouds", str(tmp_path / "clouds.npy"),
           "--workers", "2", "--repeats", "2", "--budget-s", "5"] + (["--train"] if train else [])
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["frames"] == 4 and line["workers"] == 2 and line["frames_per_s"] > 0 and len(line["runs_s"]) == 2
    assert ("backward" in line["cpu_seconds_by_op"]) == train
