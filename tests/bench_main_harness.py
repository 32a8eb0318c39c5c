"""TEST HARNESS (not product code): runs bench.main() -- the real argument parsing, world resolution, self-launch,
process-group initialisation, barrier / max-over-ranks timing and rank-0 JSON line -- on a machine without GPUs.

Launched by tests/test_bench_main_gloo.py, directly (self-launch path) or under `python -m torch.distributed.run`.  What is
replaced, in THIS process only:
  * torch.cuda device management (is_available / device_count / set_device / synchronize) -> no-ops reporting 2 devices,
    torch.device("cuda", r) -> "cpu";
  * dist.init_process_group("nccl", device_id=...) -> the same call with backend "gloo";
  * the library loader and the two workloads' compute: InferenceBench / run_train become stubs that keep the real
    barrier + reduce_elapsed sequence and sleep (rank + 1) * 5 ms per step, so the slow rank is rank 1;
  * bench.self_launch_command -> the same command line pointing at this harness instead of bench.py;
  * PRCNN_HARNESS_REAL_PIPELINE=1: InferenceBench is NOT replaced -- bench.py's own client code drives the real
    pointrcnn_amd.pipeline.InferencePipeline (device "cpu": slots, tickets, bounded queue, in-order results) around a toy model.
Everything else -- main()'s control flow -- is bench.py's own code."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 2
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
_real_device = torch.device
bench.torch = type(sys)("torch_for_bench")
bench.torch.__dict__.update(torch.__dict__)
bench.torch.device = lambda kind, index=None: _real_device("cpu")
_real_init = dist.init_process_group


def _init(backend=None, **kw):
    assert backend == "nccl", "bench.main must ask for RCCL (backend 'nccl'), got %r" % backend
    kw.pop("device_id", None)
    return _real_init("gloo", **kw)


dist.init_process_group = _init
from pointrcnn_amd import _cabi  # noqa: E402

_cabi.lib = lambda: None
_real_cmd = bench.self_launch_command


def _cmd(gpus, argv, port=None):
    cmd = _real_cmd(gpus, argv, port)
    cmd[cmd.index(os.path.abspath(bench.__file__))] = os.path.abspath(__file__)
    return cmd


bench.self_launch_command = _cmd


def _rank():
    return int(os.environ.get("RANK", "0"))


class StubBench:
    """InferenceBench with the same timed()/timed_single() sequence (barrier, steps, barrier, max over ranks), no kernels"""

    def __init__(self, args, model, dev, rank, world, clouds_kind, proposal_layer=None, raw=None):
        self.args, self.rank, self.world = args, rank, world
        self.out, self.clouds_cpu, self.graphs = {k: torch.ones(2) for k in ("backbone_features", "rpn_cls", "rpn_reg")}, None, None
        self.seed0 = bench.shard_seed0(rank, world, 0, args.batch)

    def prepare(self):
        return self

    def warm(self):
        return self

    def _timed(self, steps, d):
        if d is not None:
            d.barrier()
        t0 = time.perf_counter()
        time.sleep(0.005 * (self.rank + 1) * steps)
        if d is not None:
            d.barrier()
        return bench.reduce_elapsed(time.perf_counter() - t0, d, "cpu")

    def timed(self, steps, warmup, dist=None, h2d=False):
        return self._timed(steps, dist)

    def timed_single(self, steps, dist=None):
        return self._timed(steps, dist)

    def timed_depth(self, steps, depth, dist=None):
        return self._timed(steps, dist)

    def release(self):
        pass


def _stub_train(args, dev, rank, world, local_rank, d):
    assert (d is None) == (world == 1)
    if d is not None:
        d.barrier()
    t0 = time.perf_counter()
    time.sleep(0.005 * (rank + 1) * args.steps)
    if d is not None:
        d.barrier()
    elapsed = bench.reduce_elapsed(time.perf_counter() - t0, d, "cpu")
    if d is not None:                                   # the gradient all-reduce the real step performs through DDP
        t = torch.ones(4) * (rank + 1)
        d.all_reduce(t)
        assert float(t[0]) == sum(range(1, world + 1))
    return {"metric": "stub train", "value": bench.whole_job_value(args.batch, world, args.steps, elapsed), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "scaling": "weak"}


class _Model:
    def to(self, d):
        return self

    def eval(self):
        return self


class _TinyRPN(torch.nn.Module):
    """stand-in model for PRCNN_HARNESS_REAL_PIPELINE=1: the REAL bench.InferenceBench and pointrcnn_amd.pipeline.InferencePipeline
    (device "cpu": ticket / slot bookkeeping with eager steps) run under torch.distributed.run; only the network is a toy that
    takes (rank + 1) * 3 ms per step and returns the keys InferenceBench checks"""

    def forward(self, inp):
        x = inp["pts_input"]
        time.sleep(0.003 * (_rank() + 1))
        f = x.sum(2, keepdim=True)
        return {"rpn_cls": f * 0.5, "rpn_reg": f.expand(-1, -1, 4).contiguous() + 1.0, "backbone_features": f.transpose(1, 2), "backbone_xyz": x}


if os.environ.get("PRCNN_HARNESS_REAL_PIPELINE") == "1":
    torch.Tensor.pin_memory = lambda self, *a, **k: self          # no CUDA runtime here: "pinned" host tensors are plain ones
else:
    bench.InferenceBench = StubBench
bench.run_train = _stub_train
bench.run_train_rcnn = _stub_train
from pointrcnn_amd import rpn  # noqa: E402

rpn.RPN = (lambda *a, **k: _TinyRPN()) if os.environ.get("PRCNN_HARNESS_REAL_PIPELINE") == "1" else (lambda *a, **k: _Model())
rpn.randomize_bn_stats = lambda m, seed=0: m

if __name__ == "__main__":
    bench.main()
