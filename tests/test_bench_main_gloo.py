"""bench.py's main() end to end on CPU, world size 2 over gloo -- the path the driver runs on an 8-GPU node and the builder
never could: self-launch (`--gpus 2` without a launcher environment re-executes under torch.distributed.run) -> RANK /
WORLD_SIZE from the environment -> init_process_group -> barrier -> timed steps -> barrier -> MAX over ranks -> ONE JSON line
from rank 0, for the inference workload and for `--workload train`.  tests/bench_main_harness.py replaces device management
and the compute with stubs (rank r sleeps 5 (r + 1) ms per step); the control flow is bench.py's."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(HERE, "bench_main_harness.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("workload", ["rpn", "train", "train-rcnn"])
def test_main_under_torch_distributed_run(workload):
    """launched exactly as the driver launches it: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py"""
    steps = 4
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), HARNESS, "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--workload", workload,
           "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == "weak"
    # the slow rank (rank 1: 10 ms per step) sets the time; value = frames of BOTH ranks over it
    assert line["ms_per_step"] >= 9.5
    batch = {"rpn": 32, "train": 16, "train-rcnn": 4}[workload]
    assert abs(line["value"] - batch * 2 * 1e3 / line["ms_per_step"]) <= 0.01 * line["value"]
    if workload == "rpn":
        assert line["config"]["parallelism"] == "frames sharded, dp2" and line["value_h2d_inclusive"] > 0 and line["value_latency_mode"] > 0


def test_main_self_launches_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no launcher environment: re-executes itself as one rank per device"""
    r = subprocess.run([sys.executable, HARNESS, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                        "--no-variants"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert "re-executing" in r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3


def test_main_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, HARNESS, "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_inference_pipeline_bookkeeping_under_torch_distributed_run():
    """bench.py's REAL InferenceBench + pointrcnn_amd.pipeline.InferencePipeline (device "cpu": the slot / ticket /
    bounded-queue logic, eager steps) under `python -m torch.distributed.run`, world 2, gloo: 3 slots, more steps than slots (slots are
    reused, results collected in order), H2D-style submits and the one-batch-in-flight loop included (the variants' loops); rank 1's
    model is the slow one and sets the time"""
    steps = 7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), HARNESS, "--gpus", "2", "--steps", str(steps), "--warmup", "2", "--batch", "4", "--npoints", "256",
           "--streams", "3", "--proposals", "off", "--graph", "off", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(_env(), PRCNN_HARNESS_REAL_PIPELINE="1", PRCNN_BENCH_NO_SPLIT="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["config"]["streams"] == 3 and line["config"]["launch"] == "eager"
    assert line["ms_per_step"] >= 5.5                                  # rank 1: 6 ms per step
    assert abs(line["value"] - 4 * 2 * 1e3 / line["ms_per_step"]) <= 0.01 * line["value"]
    assert line["value_h2d_inclusive"] > 0 and line["value_latency_mode"] > 0
