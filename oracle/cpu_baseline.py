#!/usr/bin/env python3
"""CPU baseline of the RPN inference graph -- TEST / MEASUREMENT INFRASTRUCTURE (bench.py's `cpu_baseline` leg only).

The reference has no CPU path for this graph (its only CPU fallback is roipool3d_cpu), so the baseline SURVEY 8(d)
specifies is assembled from the best CPU pieces available on the box: the C oracle for the index operators (FPS, ball
query, grouping, three_nn, interpolation: single-threaded C, one frame per process) and torch's CPU sgemm for the
SharedMLP layers, with one batch of frames spread over worker processes whose thread counts add up to the host's cores.
One warm-up run, then `--repeats` timed runs of the whole batch; the median is reported.

Run as a SUBPROCESS of bench.py (fork pool, no HIP in this process):
    python -m oracle.cpu_baseline --spec weights.pkl --clouds clouds.npy [--workers W] [--repeats 5]
prints one JSON line.
"""
import argparse
import json
import multiprocessing as mp
import os
import pickle
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_STATE = {}


def _init(spec_path, threads):
    import torch
    torch.set_num_threads(threads)
    import oracle
    from oracle import rpn_cpu
    rpn_cpu._TORCH_MLP = True
    with open(spec_path, "rb") as f:
        _STATE["spec"] = pickle.load(f)
    _STATE["cpu"] = oracle.cpu()
    _STATE["rpn_cpu"] = rpn_cpu


def _frame(xyz):
    timings = {}
    out = _STATE["rpn_cpu"].rpn_forward_frame(_STATE["cpu"], xyz, _STATE["spec"], timings)
    return float(out["rpn_cls"].sum()), float(np.abs(out["rpn_reg"]).sum()), timings


def _train_frame(xyz):
    """--train: forward + proxy loss + backward + SGD update of one frame (oracle/rpn_train_cpu.py)"""
    from oracle import rpn_train_cpu
    if "params" not in _STATE:
        _STATE["params"] = rpn_train_cpu.make_params(_STATE["spec"])
    timings = {}
    loss = rpn_train_cpu.rpn_train_frame(_STATE["cpu"], xyz, _STATE["params"], timings=timings)
    return loss, 0.0, timings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", required=True)
    ap.add_argument("--clouds", required=True, help=".npy (F, N, 3) float32: the frames of one batch")
    ap.add_argument("--workers", type=int, default=0, help="worker processes (default: min(frames, cores))")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--budget-s", type=float, default=40.0, help="stop repeating once this much wall time is spent (>= 2 timed runs)")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step of every frame (forward, backward, update) instead of inference")
    a = ap.parse_args()
    frame_fn = _train_frame if a.train else _frame
    clouds = np.load(a.clouds)
    cores = os.cpu_count() or 1
    workers = a.workers or max(1, min(clouds.shape[0], cores))
    threads = max(1, cores // workers)
    frames = [np.ascontiguousarray(f) for f in clouds]
    t_start = time.perf_counter()
    with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(a.spec, threads)) as pool:
        first = pool.map(frame_fn, frames, chunksize=1)                    # warm-up (library loads, BLAS thread pools)
        runs = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            res = pool.map(frame_fn, frames, chunksize=1)
            runs.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > a.budget_s and len(runs) >= 2:
                break
    med = statistics.median(runs)
    breakdown = {}
    for _, _, t in res:
        for k, v in t.items():
            breakdown[k] = breakdown.get(k, 0.0) + v
    print(json.dumps({"frames": len(frames), "workers": workers, "threads_per_worker": threads, "cores": workers * threads,
                      "host_cores_available": cores, "runs_s": [round(r, 3) for r in runs], "median_s": round(med, 4),
                      "frames_per_s": round(len(frames) / med, 3),
                      "cpu_seconds_by_op": {k: round(v, 2) for k, v in sorted(breakdown.items())},
                      "checksum": [round(sum(r[0] for r in first), 4), round(sum(r[1] for r in first), 2)]}), flush=True)


if __name__ == "__main__":
    main()
