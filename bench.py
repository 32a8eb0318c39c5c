#!/usr/bin/env python3
"""bench.py -- RPN inference throughput of the MI355X point-ops hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- the full RPN inference graph (PointNet++ MSG backbone: 4 SA + 4 FP
levels, + cls/reg heads; lib/net/rpn.py:68-82 via pointrcnn_amd/rpn.py) -- over one batch of synthetic
16 384-point clouds already resident in HBM.  Frames shard across ranks (one process per GPU, weak scaling:
`--batch` frames per GPU); inference needs no collective, so the only distributed traffic is the barrier
and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline     -- for the dominant kernel family (the fused per-point MLP, fp32 MFMA): the FLOPs its launches execute
                  (rows actually processed -- device-side counts read back -- x layer widths) / their GPU time, measured
                  with HIP events on the launch stream in an instrumented pass of the same steps; peak = 157.3 TFLOP/s
                  dense fp32 MFMA.  reference_graph_TFLOPs prices the reference's dense flop count over the same time.
  cpu_baseline -- the SAME graph on this box's host cores the way SURVEY 8(d) specifies: index operators = the C oracle
                  (single-threaded C, one frame per process), MLPs = torch-CPU sgemm, one batch of frames spread over worker
                  processes that together use every core; median of 5 runs after a warm-up (oracle/cpu_baseline.py, run as a
                  subprocess so that it forks before any HIP state exists); plus the reference's own roipool3d_cpu.
  kernels      -- per-op-family GPU time of one step (ms) from the same event pass.
  value_latency_mode -- one batch in flight (what a --batch_size-1-style caller sees: the FPS serial chain is exposed);
                        value_by_batches_in_flight: the same at 2 / 4 / 8 batches in flight (a double-buffered eval loop is depth 2).
  value_h2d_inclusive, value_dedup_off, value_saturated -- the same command with every batch's clouds copied from pinned host
                  memory inside the timed region / with padding-free grouping switched off / on clouds whose every ball is
                  full: the throughput is data-dependent (exact first-layer hoisting + padding-free grouping remove most of the
                  reference graph's MLP rows on sparse clouds) and these are its best / typical / worst cases.

`--workload train` (BASELINE config 4): one RPN training iteration per step (forward, the reference's loss, backward through
the HIP operator kernels, DistributedDataParallel gradient all-reduce over RCCL, grad-norm clip, optimizer step), 16 frames
per GPU.
"""
import argparse
import json
import os
import sys
import time

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
# serialise: with the default, a 4th in-flight batch doubled the step time.  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_T0 = time.perf_counter()
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md); the split-bf16 kernels' pipe


def trace(msg):
    """stage markers on stderr (PRCNN_BENCH_TRACE=1): where a run that dies without a JSON line got to"""
    if os.environ.get("PRCNN_BENCH_TRACE"):
        print("[bench %7.2fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32-input MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; with 16 batches in flight the first/last steps fill and drain the pipeline (one batch's "
                         "latency is ~45 ms under load), so short runs under-report the steady state by a few percent")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (BASELINE metric: bs32; train: 16 = config 4)")
    ap.add_argument("--npoints", type=int, default=16384)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto", help="replay the step from a hipGraph")
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight per GPU, default 20 (rpn) / 10 (rcnn) (each on its own HIP stream AND hardware queue, see GPU_MAX_HW_QUEUES "
                         "above): step k runs on HIP stream k %% streams, so one batch's FPS "
                         "(1 workgroup per frame = 32 of 256 CUs) overlaps another batch's MLP / neighbour kernels")
    ap.add_argument("--proposals", choices=["off", "normal", "rotate"], default="normal",
                    help="also run the proposal layer (decode + sort + distance split + NMS + top-100, lib/rpn/proposal_layer.py) "
                         "inside every step, as tools/eval_rcnn.py --eval_mode rpn does after the heads")
    ap.add_argument("--workload", choices=["rpn", "rcnn", "train", "train-rcnn", "reference"], default="rpn",
                    help="rpn = the BASELINE metric (RPN inference end-to-end); rcnn = BASELINE config 3, the whole two-stage detector "
                         "(RPN -> proposals -> roipool3d -> RCNN -> box decode -> rotated NMS), 100 RoIs per frame; train = BASELINE "
                         "config 4, one RPN training iteration per step under DistributedDataParallel (RCCL); train-rcnn = one RCNN-stage "
                         "training iteration per step (train_rcnn.py --train_mode rcnn: fixed RPN, proposals, RoI sampling, RCNN forward / "
                         "backward), batch 4 per GPU as the reference's script")
    ap.add_argument("--clouds", choices=["uniform", "lidar", "saturated"], default="uniform",
                    help="uniform = the BASELINE metric's synthetic clouds; lidar = range-dependent density + ground band + car "
                         "clusters (same bounds): a robustness check for the spatially pruned / grid kernels; saturated = 16384 "
                         "points in a 1.6 m cube: every ball of every level is full (the worst case for padding-free grouping)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra timed loops (H2D-inclusive, dedup off, saturated clouds) that report the data dependence")
    ap.add_argument("--h2d", action="store_true",
                    help="also copy every batch's clouds from pinned host memory inside the timed region (PCIe-inclusive rate; "
                         "the default keeps inputs resident in HBM, as the bench contract asks)")
    ap.add_argument("--input", choices=["clouds", "raw"], default="clouds",
                    help="clouds = prepared (npoints,3) clouds resident in HBM (the bench contract); raw = every step starts from "
                         "raw velodyne scans (~118k points x 16 B per frame) in pinned host memory: H2D copy + the on-device input "
                         "builder (prcnn_scene_prepare: lidar->rect, image/range crop, 16384-point draw) inside the timed region")
    ap.add_argument("--raw-points", type=int, default=118000, help="raw points per synthetic scan for --input raw")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc subprocess passes behind roofline.traffic")
    ap.add_argument("--dump-launches", action="store_true", help="print every library launch of one instrumented step to stderr "
                    "(name, live rows, executed GFLOP, us, TFLOP/s)")
    args = ap.parse_args(argv)
    train = args.workload in ("train", "train-rcnn")
    if args.steps is None:
        args.steps = 20 if (train or args.workload == "reference") else 320
    if args.warmup is None:
        args.warmup = 6 if train else (3 if args.workload == "reference" else 16)
    if args.batch is None:
        args.batch = (16 if args.workload == "train" else 4) if train else 32
    return args


# ---- multi-GPU plumbing (exercised on CPU by tests/test_dist_gloo.py) ------------------------------------------------
def shard_seed0(rank, world, slot, batch):
    """first frame seed of the batch that in-flight slot `slot` of rank `rank` owns: frames are sharded by rank (weak
    scaling, `batch` frames per GPU), disjoint across ranks and slots: rank r, slot s holds global batch s * world + r"""
    return 100 + (world * slot + rank) * batch


def reduce_elapsed(elapsed, dist=None, device="cpu"):
    """the job's step time is the slowest rank's: MAX over ranks of the barrier-bracketed elapsed time"""
    if dist is None:
        return float(elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_value(batch, world, steps, elapsed):
    """frames of ALL ranks per second"""
    return batch * world * steps / elapsed


def self_launch_command(gpus, argv, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself as one rank per GPU"""
    import socket
    if port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(args, env, ndev):
    """-> (world, rank, local_rank, relaunch): validates --gpus against the launcher environment and the visible devices"""
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
        rank, local_rank = int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
        if local_rank >= ndev:
            raise SystemExit("bench.py: local rank %d but only %d visible GPU(s)" % (local_rank, ndev))
        return world, rank, local_rank, False
    if args.gpus > 1:
        if ndev < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible: refusing to report a %d-GPU number from fewer devices"
                             % (args.gpus, ndev, args.gpus))
        return args.gpus, 0, 0, True
    return 1, 0, 0, False


def distributed_facts(dist, world):
    """what the line's N > 1 claims rest on: the world size the process group actually has, backend and RCCL version"""
    facts = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(), "n_gpus_requested": world}
    try:
        facts["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        facts["rccl_version"] = "unavailable (%s)" % type(e).__name__
    return facts


def distributed_preflight(dist, world, rank, dev):
    """first contact with RCCL happens HERE, before anything is timed: every rank reports its device, one all-reduce checks the
    ring, rank 0 prints what the N > 1 line will rest on (stderr) -- a mis-sized world or a dead link fails fast with a message
    instead of hanging inside the timed loop"""
    import socket
    try:
        name = torch.cuda.get_device_name(dev)
    except Exception as e:  # noqa: BLE001
        name = "unknown (%s)" % type(e).__name__
    mine = {"rank": rank, "host": socket.gethostname(), "device": str(dev), "name": name, "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    t = torch.full((1,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    want = world * (world + 1) / 2
    if abs(float(t.item()) - want) > 1e-6:
        raise SystemExit("bench.py: all-reduce over %d ranks returned %r, expected %r" % (world, float(t.item()), want))
    facts = distributed_facts(dist, world)
    facts["ranks"] = sorted(everyone, key=lambda r: r["rank"])
    if dev.type == "cuda" and len({(r["host"], r["device"]) for r in everyone}) != world:
        raise SystemExit("bench.py: %d ranks but only %d distinct devices: %s" % (world, len({(r["host"], r["device"]) for r in everyone}), everyone))
    if rank == 0:
        print("[bench] distributed preflight ok: %s" % json.dumps(facts), file=sys.stderr, flush=True)
    return facts


class EventProfiler:
    """Wraps the ctypes library: every prcnn_* launch is bracketed by HIP events recorded on the stream the
    kernel is launched on (torch's current stream).  Used only in the instrumented pass, never in the timed one."""

    def __init__(self, lib):
        self._lib = lib
        self.records = []          # (name, start, end, (flops per row, static rows, device row-count pointer, rows per count))
        self.splits = []           # GroupSplit objects of the pass (kept alive: their device counts are read in summary())

    @staticmethod
    def _work(name, a):
        """(fp32-equivalent flops per row, rows, rows_dev pointer, rows per device count, label, bf16 terms, MFMA flops per row ON THE
        PIPE THE LAUNCH USES) of one call.  EXECUTED work: first-layer hoisting and padding-free grouping make it smaller than the
        reference graph's count.  MFMA flops: what the matrix pipe is given -- output widths in whole 32-column blocks (a 76-wide
        layer runs as 96 columns, the cls head's single output channel is a VALU dot product: 0), times the number of bf16 products
        per fp32 product for the split kernels -- the quantity SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} x 512 counts."""
        def pad(n):
            return 0 if n == 1 else -(-int(n) // 32) * 32
        def chain(k0, nout, padded=False):
            widths = [k0] + list(nout)
            return 2.0 * sum(x * (pad(y) if padded else y) for x, y in zip(widths[:-1], widths[1:]))
        def widths(k0, nout, n):
            return "->".join(str(int(v)) for v in [k0] + [nout[i] for i in range(n)])
        if name == "prcnn_mlp_rows":
            return 2.0 * a[3] * a[6], a[2], a[12], a[13], "%d->%d" % (a[3], a[6]), 0, 2.0 * a[3] * pad(a[6])
        if name == "prcnn_mlp_rows_addinterp":
            return 2.0 * (a[2] + 3) * a[5], a[11] * a[12], None, 1, "%d->%d + interpolated addend" % (a[2], a[5]), 0, 2.0 * a[2] * pad(a[5])
        if name == "prcnn_mlp_rows_split":              # (fp32-EQUIVALENT flops: 2 K N per row, whatever the number of bf16 terms)
            return 2.0 * a[3] * a[8], a[2], a[14], a[15], "%d->%d (bf16x%d)" % (a[3], a[8], a[6]), a[6], 2.0 * a[3] * pad(a[8]) * a[6]
        if name == "prcnn_mlp_chain_rows_split":
            return (chain(a[3], [a[7][0], a[7][1]]), a[2], None, 1, widths(a[3], a[7], 2) + " (bf16x%d)" % a[9], a[9],
                    chain(a[3], [a[7][0], a[7][1]], True) * a[9])
        if name == "prcnn_mlp_chain_interp_split":
            return 2.0 * a[7] * a[12], a[4] * a[5], None, 1, "%d->%d (bf16x%d)" % (a[7], a[12], a[14]), a[14], 2.0 * a[7] * pad(a[12]) * a[14]
        if name == "prcnn_mlp_rows_addinterp_split":
            return (2.0 * (a[2] + 3) * a[7], a[13] * a[14], None, 1, "%d->%d + interpolated addend (bf16x%d)" % (a[2], a[7], a[5]), a[5],
                    2.0 * a[2] * pad(a[7]) * a[5])
        if name == "prcnn_mlp_group_split":             # hoisted form: K = C
            return (2.0 * a[9] * a[16], a[5] * a[7] * a[8], a[22], a[8], "%d->%d hoisted group (bf16x%d)" % (a[9], a[16], a[14]), a[14],
                    2.0 * a[9] * pad(a[16]) * a[14])
        if name == "prcnn_mlp_group":
            k = a[9] + (0 if a[10] else 3)
            return 2.0 * k * a[14], a[5] * a[7] * a[8], a[20], a[8], "%d->%d" % (k, a[14]), 0, 2.0 * k * pad(a[14])
        if name == "prcnn_mlp_interp":
            return 2.0 * (a[9] + a[10]) * a[14], a[6] * a[7], None, 1, "%d->%d" % (a[9] + a[10], a[14]), 0, 2.0 * (a[9] + a[10]) * pad(a[14])
        if name == "prcnn_mlp_chain_rows":
            return chain(a[3], a[7]), a[2], None, 1, widths(a[3], a[7], a[4]), 0, chain(a[3], [a[7][i] for i in range(a[4])], True)
        if name == "prcnn_mlp_chain_group":
            k = a[9] + (0 if a[10] else 3)
            return chain(k, a[15]), a[5] * a[7] * a[8], a[21], a[8], widths(k, a[15], a[12]), 0, chain(k, [a[15][i] for i in range(a[12])], True)
        if name == "prcnn_mlp_chain_interp":
            return (chain(a[9] + a[10], a[15]), a[6] * a[7], None, 1, widths(a[9] + a[10], a[15], a[12]), 0,
                    chain(a[9] + a[10], [a[15][i] for i in range(a[12])], True))
        return 0.0, 0, None, 1, "", 0, 0.0

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("prcnn_") or name.endswith("_bytes") or name in ("prcnn_last_error", "prcnn_abi_version",
                                                                              "prcnn_wpack_floats", "prcnn_wsplit_bytes"):
            return fn

        def wrapped(*args):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            self.records.append((name, s, e, self._work(name, args)))
            return rc
        return wrapped

    def summary(self, dump=None):
        torch.cuda.synchronize()
        dev_counts, undedup = {}, {}
        for sp in self.splits:
            c = sp.counts.cpu().tolist()
            for k, v in enumerate(c):
                dev_counts[sp.counts.data_ptr() + 4 * k] = v
            if getattr(sp, "slice_count", False):       # the device-side count of one slice of a list walked in slices (pointnet2_modules._run_scale)
                continue
            # rows the list's launch stands for in the graph without padding-free grouping: all G * nsample rows of the
            # scale, minus what the dense list (its own launches) carries
            undedup[sp.counts.data_ptr()] = sp.G * sp.ns - c[1] * sp.ns
        fam = {}
        self.mlp_launches = []        # (entry point, live rows, flops per row, us, label, bf16 terms, MFMA flops per row, rows without padding-free grouping) of every MLP-family launch, in launch order
        for name, s, e, (per_row, rows, ptr, unit, label, terms, pipe_per_row) in self.records:
            key = "mlp" if name.startswith("prcnn_mlp_") else name[len("prcnn_"):]
            d = fam.setdefault(key, {"ms": 0.0, "launches": 0, "flops": 0.0, "rows": 0, "rows_launched": 0, "pipe_seconds_at_peak": 0.0})
            ptr = getattr(ptr, "value", ptr)
            live = min(rows, dev_counts[ptr] * unit) if ptr else rows
            if dump is not None:
                us = s.elapsed_time(e) * 1e3
                print("%-28s rows %8d / %8d  flop/row %8.0f  %7.2f GFLOP %8.1f us %6.1f TF/s" %
                      (name, live, rows, per_row, per_row * live / 1e9, us, per_row * live / us / 1e6 if us > 0 else 0), file=dump)
            if key == "mlp":
                self.mlp_launches.append((name[len("prcnn_"):], live, per_row, s.elapsed_time(e) * 1e3, label, terms, pipe_per_row,
                                          undedup[ptr] if ptr in undedup else live))
                # time the launch's MFMA work would take at the dense peak of the pipe it runs on
                d["pipe_seconds_at_peak"] += pipe_per_row * live / ((BF16_MFMA_PEAK_TFLOPS if terms else FP32_MFMA_PEAK_TFLOPS) * 1e12)
            d["ms"] += s.elapsed_time(e)
            d["launches"] += 1
            d["flops"] += per_row * live
            d["rows"] += live
            d["rows_launched"] += rows
            d["rows_undedup"] = d.get("rows_undedup", 0) + (undedup[ptr] if ptr in undedup else live)
        return fam


def cpu_baseline(model, clouds_cpu, gpu_out):
    """SURVEY 8(d) CPU baseline of the same graph (see module docstring) + the reference's own roipool3d_cpu + the relative
    error of the GPU outputs against the double-accumulating oracle on frame 0"""
    import pickle
    import subprocess
    import tempfile
    import numpy as np
    import oracle
    from oracle import rpn_cpu
    cpu = oracle.cpu()
    spec = rpn_cpu.extract_rpn_weights(model)
    cores = os.cpu_count() or 1
    # "oracle ops x nproc frames in parallel processes" -- one single-threaded frame per core (capped at 128
    # worker processes; above that every worker gets cores // 128 BLAS threads)
    nframes = int(max(8, min(cores, 128)))
    if nframes > clouds_cpu.shape[0]:
        from pointrcnn_amd import rpn as _rpn
        clouds_cpu = torch.cat([clouds_cpu, _rpn.synthetic_clouds(nframes - clouds_cpu.shape[0], clouds_cpu.shape[1], seed0=90000)])
    res = None
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "spec.pkl"), "wb") as f:
            pickle.dump(spec, f)
        np.save(os.path.join(td, "clouds.npy"), clouds_cpu[:nframes].numpy())
        env = dict(os.environ)
        env.pop("OMP_NUM_THREADS", None)
        try:
            p = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--spec", os.path.join(td, "spec.pkl"), "--clouds",
                                os.path.join(td, "clouds.npy"), "--repeats", "5", "--budget-s", "40"], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=600)
            res = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            print("[bench] cpu_baseline subprocess failed: %s" % e, file=sys.stderr)
    # frame 0 through the double-accumulating oracle: the parity figure of the line
    out = rpn_cpu.rpn_forward_frame(cpu, clouds_cpu[0].numpy(), spec, {})
    err = {}
    for k in ("rpn_cls", "rpn_reg"):
        ref = out[k]
        got = gpu_out[k][0].float().cpu().numpy()
        err[k] = float(abs(got - ref).max() / max(1.0, abs(ref).max()))
    # the reference's ONLY CPU fallback on the path, its own lib/utils/roipool3d/src/roipool3d.cpp compiled for the host
    # (oracle/_ref): one frame of config-3 pooling (16384 pts x 130 features, 100 RoIs, 512 samples), next to the HIP kernel
    ref_roipool = None
    ref = oracle.ref()
    if ref is not None:
        from pointrcnn_amd import ops
        rng = np.random.default_rng(0)
        pts = clouds_cpu[0].numpy()
        ctr = pts[rng.integers(0, pts.shape[0], 100)]
        boxes = np.concatenate([ctr[:, :1], ctr[:, 1:2] + 1.8, ctr[:, 2:3], np.tile([3.6, 3.7, 6.0], (100, 1)),
                                rng.uniform(-3.14, 3.14, (100, 1))], 1).astype(np.float32)
        feat = rng.normal(size=(pts.shape[0], 130)).astype(np.float32)
        times = []
        for _ in range(6):
            t1 = time.perf_counter()
            pp, pf, pe = ref.roipool3d_cpu(pts, boxes, feat, 512)
            times.append(time.perf_counter() - t1)
        cpu_s = sorted(times[1:])[len(times[1:]) // 2]
        dev = gpu_out["rpn_cls"].device
        tx, tb, tf = (torch.from_numpy(a[None]).to(dev) for a in (pts, boxes, feat))
        pooled, _ = ops.roipool3d(tx, tb, tf, 512)
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            ops.roipool3d(tx, tb, tf, 512)
        e_.record()
        torch.cuda.synchronize()
        same = bool(np.array_equal(pooled[0, :, :, 3:].cpu().numpy(), pf) and np.array_equal(pooled[0, :, :, :3].cpu().numpy(), pp))
        ref_roipool = {"kind": "reference", "op": "roipool3d_cpu (lib/utils/roipool3d/src/roipool3d.cpp:127-195), 1 frame, 100 RoIs x 512 x 133",
                       "cpu_ms_per_frame": round(cpu_s * 1e3, 2), "cores": 1, "runs": "median of 5 after 1 warm-up",
                       "cpu_frames_per_s_all_cores": round(cores / cpu_s, 1),
                       "gpu_ms_per_frame_single_frame_launch": round(s_.elapsed_time(e_) / 20, 4), "outputs_identical": same}
    line = {"value": None, "unit": "frames/s", "cores": cores, "kind": "port",
            "what": "index operators: C oracle (oracle/prcnn_oracle.c), one frame per worker process; SharedMLP layers: torch-CPU "
                    "sgemm; worker threads add up to the host's cores (the reference has no CPU path for this graph)",
            "reference_roipool3d": ref_roipool, "host_cores_available": cores, "gpu_vs_oracle_rel_err_frame0": err}
    if res is not None:
        line.update({"value": res["frames_per_s"], "cores": res["cores"],
                     "sample": "%d frames (%d pts each) of the same RPN graph, %d worker processes x %d threads, median of %d runs "
                               "after 1 warm-up (%s s)" % (res["frames"], clouds_cpu.shape[1], res["workers"], res["threads_per_worker"], len(res["runs_s"]),
                                                          res["runs_s"]),
                     "cpu_seconds_by_op": res["cpu_seconds_by_op"]})
    return line


def arithmetic_label(terms):
    """`dtype` of the line: the arithmetic the fused inference MLPs compute in"""
    if terms == 0:
        return "f32 (fp32 MFMA: v_mfma_f32_32x32x2_f32; PRCNN_MLP_SPLIT=0)"
    return ("f32 via split-bf16x%d MFMA (every fp32 operand cut exactly into 3 bf16 pieces, %d bf16 MFMA products per fp32 product, fp32 "
            "accumulate: %s; hoisted grouped layers on the split kernel too, unhoisted grouped first layers (xyz-only input) and training on "
            "fp32 MFMA; PRCNN_MLP_SPLIT=0 selects fp32 MFMA throughout)" %
            (terms, terms, "fp32-grade, measured error 2.9e-7 of sum|x||w| vs 3.0e-7 for the fp32-MFMA kernel" if terms == 6 else
             "error 1e-5 of sum|x||w|: outside the 1e-5 output contract, dev only"))


def rel_diff(a, b, keys=("backbone_features", "rpn_cls", "rpn_reg")):
    """max over the keys of max |a - b| / max |b|"""
    return max(float((a[k] - b[k]).abs().max() / b[k].abs().max()) for k in keys)


def make_cloud_fn(kind):
    from pointrcnn_amd import rpn
    return {"uniform": rpn.synthetic_clouds, "lidar": rpn.lidar_like_clouds, "saturated": rpn.saturated_clouds}[kind]


class InferenceBench:
    """the timed inference loop: a client of pointrcnn_amd.pipeline.InferencePipeline (S in-flight slots, each a captured hipGraph
    of one step on its own stream, results collected in submission order through a bounded queue)"""

    def __init__(self, args, model, dev, rank, world, clouds_kind, proposal_layer=None, raw=None):
        self.args, self.model, self.dev, self.proposal_layer, self.raw = args, model, dev, proposal_layer, raw
        self.nstreams = max(1, args.streams)
        make_clouds = make_cloud_fn(clouds_kind)
        self.clouds_cpu = make_clouds(args.batch, args.npoints, seed0=shard_seed0(rank, world, 0, args.batch))
        self.pipe, self.out, self.host, self.graphs = None, None, None, None
        self._slot_clouds = lambda s_: (self.clouds_cpu if s_ == 0 else                                              # noqa: E731
                                        make_clouds(args.batch, args.npoints, seed0=shard_seed0(rank, world, s_, args.batch)))
        self._eager_inputs = None

    def _example(self):
        """per-slot initial inputs: every in-flight slot owns its (resident) batch and is captured on it"""
        if os.environ.get("PRCNN_BENCH_SAME_EXAMPLE"):          # dev A/B: every slot captured on slot 0's batch
            if self.raw is not None:
                return {k: self.raw["slots"][0]["dev"][k] for k in ("raw", "offsets", "calib", "img_hw")}
            return {"pts_input": self.clouds_cpu}
        if self.raw is not None:
            return [{k: self.raw["slots"][s_]["dev"][k] for k in ("raw", "offsets", "calib", "img_hw")} for s_ in range(self.nstreams)]
        return [{"pts_input": self._slot_clouds(s_)} for s_ in range(self.nstreams)]

    def _example0(self):
        ex = self._example()
        return ex[0] if isinstance(ex, list) else ex

    def step_from(self, inputs, slot=0):
        """one step of the workload on a slot's (static) input tensors: what the pipeline captures"""
        args, model = self.args, self.model
        with torch.no_grad():
            if self.raw is not None:
                r = self.raw["slots"][slot]
                xyz, _, _, _, r["status"] = self.raw["ops"].scene_prepare(inputs["raw"], inputs["offsets"], r["max_points"], inputs["calib"],
                                                                          inputs["img_hw"], self.raw["prep"].scope, args.npoints, r["seed"])
                o = model({"pts_input": xyz})
            else:
                o = model(inputs)
            if self.proposal_layer is not None:
                o["rois"], o["roi_scores_raw"] = self.proposal_layer(o["rpn_cls"][:, :, 0], o["rpn_reg"], o["backbone_xyz"])
            if args.workload == "rcnn":
                o["pred_boxes3d"], o["raw_scores"], o["keep"], o["num_keep"] = model.detections(o)
            return o

    def step(self, slot=0):
        """eager step on slot `slot`'s resident inputs, on the current stream (warm-up, instrumented pass)"""
        if self.pipe is not None:
            return self.step_from(self.pipe.inputs[slot], slot)
        if self._eager_inputs is None:
            self._eager_inputs = {k: v.to(self.dev) for k, v in self._example0().items()}
        return self.step_from(self._eager_inputs, slot)

    def warm(self):
        for _ in range(max(1, min(self.args.warmup, 4))):   # packs weights, fills the caching allocator
            self.out = self.step(0)
        torch.cuda.synchronize()
        return self

    def prepare(self):
        from pointrcnn_amd.pipeline import InferencePipeline
        args = self.args
        if self.out is None:
            self.warm()
        eager_out = self.out
        try:
            self.pipe = InferencePipeline(self.step_from, self._example(), slots=self.nstreams, device=self.dev, graph=args.graph != "off", copy_stream=True)
        except Exception as e:  # noqa: BLE001
            if args.graph == "on":
                raise
            print("[bench] hipGraph capture unavailable (%s); timing eager launches" % str(e).split("\n")[0], file=sys.stderr)
            self.pipe = InferencePipeline(self.step_from, self._example(), slots=self.nstreams, device=self.dev, graph=False, copy_stream=True)
        self._eager_inputs = None
        trace("pipeline built")
        if os.environ.get("PRCNN_BENCH_SAME_EXAMPLE"):
            for s_ in range(1, self.nstreams):
                if self.raw is not None:
                    for k in ("raw", "offsets", "calib", "img_hw"):
                        self.pipe.inputs[s_][k].copy_(self.raw["slots"][s_]["dev"][k])
                else:
                    self.pipe.inputs[s_]["pts_input"].copy_(self._slot_clouds(s_))
        torch.cuda.synchronize()
        self.graphs = self.pipe.graphs if self.pipe.graphed else None
        self.pipe.submit(None)
        out0 = self.pipe.result()
        trace("slot 0 replayed")
        for k in ("rpn_cls", "rpn_reg"):                 # the replayed graph must reproduce the eager result
            assert torch.equal(out0[k], eager_out[k]), "graph replay differs from eager (%s)" % k
        self.out = out0
        # slot 0's ticket was consumed: re-align ticket numbering so that step k runs on slot k % S
        for _ in range(1, self.nstreams):
            self.pipe.submit(None)
        self.pipe.drain()
        return self

    def run(self, k, h2d=False):
        """step k: collect the oldest result when the pipeline is full (bounded queue), then submit -- with the batch's inputs from
        pinned host memory when `h2d` (asynchronous copy on the slot's stream), on the resident inputs otherwise"""
        pipe = self.pipe
        if pipe.outstanding >= pipe.slots:
            pipe.result()
        slot = pipe._next_ticket % pipe.slots
        if self.raw is not None:
            pipe.submit({"raw": self.raw["slots"][slot]["host"]["raw"]})
        elif h2d:
            pipe.submit({"pts_input": self.host[slot]})
        else:
            pipe.submit(None)

    def timed(self, steps, warmup, dist=None, h2d=False):
        """`warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; -> max-over-ranks seconds"""
        if h2d and self.host is None:
            self.host = [self.pipe.inputs[s_]["pts_input"].cpu().pin_memory() for s_ in range(self.nstreams)]
        for k in range(warmup):
            self.run(k, h2d)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            self.run(k, h2d)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = reduce_elapsed(time.perf_counter() - t0, dist, self.dev)
        self.pipe.drain()
        return elapsed

    def timed_single(self, steps, dist=None):
        """latency mode: ONE batch in flight -- submit, wait for its result, submit the next (what a tools/eval_rcnn.py-style loop
        sees: the FPS serial chain is exposed)"""
        self.pipe.drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.pipe.submit(None)
            self.pipe.result()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return reduce_elapsed(time.perf_counter() - t0, dist, self.dev)

    def timed_depth(self, steps, depth, dist=None):
        """`depth` batches in flight: what a caller sees that keeps a bounded number of batches outstanding (depth 1 = timed_single)"""
        self.pipe.drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        inflight = 0
        for _ in range(steps):
            if inflight == depth:
                self.pipe.result()
                inflight -= 1
            self.pipe.submit(None)
            inflight += 1
        while inflight:
            self.pipe.result()
            inflight -= 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return reduce_elapsed(time.perf_counter() - t0, dist, self.dev)

    def release(self):
        if self.pipe is not None:
            self.pipe.close()
        self.pipe, self.graphs, self.out, self._eager_inputs = None, None, None, None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def event_pair_overhead_us(n=200):
    """what an empty HIP event pair on the current stream measures: subtracted from every bracketed launch"""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return v[len(v) // 2]


def instrumented_pass(args, bench, nprof, dump=None):
    """per-op-family GPU time + EXECUTED MLP flops of `nprof` eager steps (HIP events on the launch stream, one batch on the chip)"""
    from pointrcnn_amd import _cabi, ops as _ops
    bench.step(0)                           # one unprofiled eager step first (allocator / lazy state after the graph replays)
    torch.cuda.synchronize()
    prof = EventProfiler(_cabi._lib)
    real = _cabi._lib
    _cabi._lib, _ops._split_log = prof, prof.splits
    try:
        for _ in range(nprof):
            bench.step(0)
        fam = prof.summary(dump)
        fam["event_pair_overhead_us"] = event_pair_overhead_us()
        # per-launch table, averaged over the nprof repeats (the launch sequence of a step is static)
        per_step = len(prof.mlp_launches) // max(1, nprof)
        if per_step and per_step * nprof == len(prof.mlp_launches):
            table = []
            for i in range(per_step):
                reps = prof.mlp_launches[i::per_step]
                us = sum(r[3] for r in reps) / len(reps)
                name, rows, per_row, _, label, terms, pipe_per_row, rows_ref = reps[0]
                gflop, pipe_gflop = rows * per_row / 1e9, rows * pipe_per_row / 1e9
                peak = BF16_MFMA_PEAK_TFLOPS if terms else FP32_MFMA_PEAK_TFLOPS
                table.append({"launch": name, "widths": label, "rows": rows, "rows_in_the_reference_graph": rows_ref, "us": round(us, 1),
                              "fp32_equivalent_GFLOP": round(gflop, 3), "fp32_equivalent_TFLOPs": round(gflop * 1e3 / us, 1) if us > 0 else None,
                              "pipe": "bf16 MFMA x%d" % terms if terms else "fp32 MFMA", "mfma_GFLOP_on_pipe": round(pipe_gflop, 3),
                              "pipe_peak_TFLOPs": peak,
                              "frac_of_pipe_peak": round(min(1.0, pipe_gflop * 1e3 / us / peak), 4) if us > 0 else None})
            fam["mlp_by_launch"] = table
        return fam
    finally:
        _cabi._lib, _ops._split_log = real, None


def measure_hbm_traffic(args, timeout_s=240):
    """roofline.traffic, measured in this run: HBM bytes per launch of the MLP family from two rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in SEPARATE runs, kernel trace only -- the recipe of /opt/skills/guides/MI355X_MICROARCH.md, gfx950 correction
    included) of a short single-stream eager run of this script, as subprocesses.  -> fields for the roofline object; traffic is null
    (with the reason) when rocprofv3 is missing or a pass fails -- never a number from another run."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"traffic": None, "traffic_note": "rocprofv3 not found on this box"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(args.batch), "--npoints", str(args.npoints),
           "--clouds", args.clouds, "--proposals", args.proposals, "--no-cpu-baseline", "--no-roofline", "--no-variants", "--no-traffic",
           "--graph", "off", "--streams", "1"]
    t0 = time.perf_counter()
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                p = subprocess.run([exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(td, "pmc_" + c), "--"] + cmd,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                if p.returncode != 0:
                    return {"traffic": None, "traffic_note": "rocprofv3 --pmc %s pass failed (rc %d): %s" % (c, p.returncode, (p.stderr or "")[-300:])}
            spec = importlib.util.spec_from_file_location("derive_hbm_traffic", os.path.join(ROOT, "profiles", "derive_hbm_traffic.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            res = mod.derive(os.path.join(td, "pmc_"))
    except Exception as e:  # noqa: BLE001  (a missing counter, a timeout: the line must still be printed)
        return {"traffic": None, "traffic_note": "hbm traffic passes failed: %s: %s" % (type(e).__name__, str(e)[:300])}
    mlp = res.get("mlp")
    if not mlp:
        return {"traffic": None, "traffic_note": "no MLP-family dispatch in the counter collection"}
    return {"traffic": mlp["hbm_bytes_per_launch_corrected"], "traffic_unit": "HBM bytes per launch of the MLP family, measured in this run",
            "traffic_detail": {"launches_per_step": mlp["launches_per_step"], "hbm_bytes_per_step": mlp["hbm_bytes_per_step_corrected"],
                               "FETCH_SIZE_KiB": mlp["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": mlp["WRITE_SIZE_KiB"], "correction": mod.CORRECTION,
                               "command": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace -- python bench.py --steps 2 --warmup 1 --graph off "
                                          "--streams 1 (two separate subprocess passes, %.0f s)" % (time.perf_counter() - t0),
                               "all_families_hbm_bytes_per_step": {k: v["hbm_bytes_per_step_corrected"] for k, v in res.items()}}}


def stack_gemm_account(step, seconds_per_step):
    """executed GEMM FLOPs of the hand-written training stacks in one step (forward, weight gradient, input gradient; on the LIVE rows
    of padding-free lists) and what they amount to over the measured step time: a whole-step figure -- the step also holds the
    BatchNorm reductions, gathers, the loss and the optimizer -- next to the 157.3 TFLOP/s fp32-MFMA peak"""
    from pointrcnn_amd import train_mlp
    train_mlp.FLOP_LOG = []
    try:
        step()
        torch.cuda.synchronize()
        flops = train_mlp.logged_flops(train_mlp.FLOP_LOG)
        calls = len(train_mlp.FLOP_LOG)
    finally:
        train_mlp.FLOP_LOG = None
    if calls == 0:
        return None
    tf = flops / seconds_per_step / 1e12
    return {"stack_calls_per_step": calls, "executed_GFLOP_per_step": round(flops / 1e9, 2), "TFLOPs_over_the_whole_step": round(tf, 2),
            "frac_of_fp32_mfma_peak_over_the_whole_step": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}


def train_roofline(step, gemm):
    """roofline object of a training line: the hand-written SharedMLP stacks (prcnn_train_stack_fwd / _bwd: MFMA forward, dgrad, wgrad
    + the BatchNorm reductions, packing and pooling launched inside those calls) are the dominant family; achieved = the executed GEMM
    flops of the step (stack_gemm_account) / the GPU time of those library calls, HIP events on the launch stream around every call"""
    from pointrcnn_amd import _cabi
    if gemm is None:
        return None
    step()
    torch.cuda.synchronize()
    prof = EventProfiler(_cabi._lib)
    real = _cabi._lib
    _cabi._lib = prof
    nrep = 2
    try:
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
    finally:
        _cabi._lib = real
    fam = {}
    for name, s_, e_, _ in prof.records:
        d = fam.setdefault(name[len("prcnn_"):], [0.0, 0])
        d[0] += s_.elapsed_time(e_) / nrep
        d[1] += 1
    stack_ms = sum(v[0] for k, v in fam.items() if k.startswith("train_stack_"))
    lib_ms = sum(v[0] for v in fam.values())
    if stack_ms <= 0:
        return None
    achieved = gemm["executed_GFLOP_per_step"] / stack_ms            # GFLOP / ms = TFLOP/s
    return {"kernel": "train_fwd_kernel / train_dgrad_kernel / train_wgrad(_lds)_kernel (fp32 MFMA) + BatchNorm reductions, pack, pool inside "
                      "prcnn_train_stack_fwd / prcnn_train_stack_bwd",
            "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "flops_per_step": gemm["executed_GFLOP_per_step"] * 1e9, "stack_calls_ms_per_step": round(stack_ms, 3),
            "all_library_calls_ms_per_step": round(lib_ms, 3),
            "by_call_ms_per_step": {k: {"ms": round(v[0], 3), "calls": v[1] // nrep} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:12]},
            "note": "achieved = executed GEMM flops (forward + input gradient + weight gradient on the live rows) / GPU time of the stack "
                    "library calls (which also hold the BatchNorm statistics / backward reductions, weight packing, pooling and the split "
                    "wgrad reduction -- not GEMM flops); HIP events on the launch stream, eager step"}


def train_cpu_baseline(model_rpn, clouds_cpu, gpu_frames_per_s):
    """the RPN training step on the host cores (oracle/rpn_train_cpu.py: C-oracle index operators + torch-CPU autograd for everything
    that carries gradients, one frame per worker process), on a bounded sample; run as a subprocess before... no HIP state in it"""
    import pickle
    import subprocess
    import tempfile
    import numpy as np
    from oracle import rpn_cpu
    spec = rpn_cpu.extract_rpn_weights(model_rpn)
    cores = os.cpu_count() or 1
    nframes = int(max(4, min(cores // 2, 64)))          # (a training frame keeps ~6 GB of activations: fewer, fatter workers)
    if nframes > clouds_cpu.shape[0]:
        from pointrcnn_amd import rpn as _rpn
        clouds_cpu = torch.cat([clouds_cpu, _rpn.synthetic_clouds(nframes - clouds_cpu.shape[0], clouds_cpu.shape[1], seed0=91000)])
    res = None
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "spec.pkl"), "wb") as f:
            pickle.dump(spec, f)
        np.save(os.path.join(td, "clouds.npy"), clouds_cpu[:nframes].cpu().numpy())
        env = dict(os.environ)
        env.pop("OMP_NUM_THREADS", None)
        try:
            p = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--train", "--spec", os.path.join(td, "spec.pkl"), "--clouds",
                                os.path.join(td, "clouds.npy"), "--repeats", "2", "--budget-s", "30"], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=900)
            res = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            print("[bench] train cpu_baseline subprocess failed: %s" % e, file=sys.stderr)
    line = {"value": None, "unit": "frames/s", "cores": cores, "kind": "port",
            "what": "RPN training step per frame on the host: index operators = C oracle, gradient-carrying part (grouping, SharedMLP + "
                    "training-mode BatchNorm + ReLU on nsample-padded rows, max-pool, interpolation, heads, backward, SGD update) = torch-CPU "
                    "autograd; proxy loss; the reference has no CPU path for training (oracle/rpn_train_cpu.py)"}
    if res is not None:
        line.update({"value": res["frames_per_s"], "cores": res["cores"],
                     "sample": "%d frames (%d pts each), %d worker processes x %d threads, median of %d runs after 1 warm-up (%s s)" %
                               (res["frames"], clouds_cpu.shape[1], res["workers"], res["threads_per_worker"], len(res["runs_s"]), res["runs_s"]),
                     "cpu_seconds_by_op": res["cpu_seconds_by_op"],
                     "gpu_over_cpu": round(gpu_frames_per_s / res["frames_per_s"], 1) if res["frames_per_s"] else None})
    return line


def run_train(args, dev, rank, world, local_rank, dist):
    """BASELINE config 4: RPN training iterations under DDP; -> the JSON line"""
    from pointrcnn_amd import ops, rpn, train_functions as tf
    torch.manual_seed(1234)
    model = tf.init_rpn_head_weights(rpn.randomize_bn_stats(rpn.RPN(), seed=7)).to(dev)
    trainer = tf.RPNTrainer(model, ddp=dist is not None, device_ids=[local_rank] if dist is not None else None)
    make_clouds = make_cloud_fn(args.clouds)
    nslots = 4                                              # a few different resident batches, cycled
    batches = []
    g = torch.Generator().manual_seed(99 + rank)
    for s_ in range(nslots):
        pts = make_clouds(args.batch, args.npoints, seed0=shard_seed0(rank, world, s_, args.batch)).to(dev)
        # synthetic GT: 12 car-sized boxes per frame centred on cloud points (enlarged x2 so that the sparse synthetic cloud
        # yields a few hundred foreground points per frame), labels from the device label kernel (prcnn_rpn_labels)
        pick = torch.randint(0, args.npoints, (args.batch, 12), generator=g).to(dev)
        ctr = torch.gather(pts, 1, pick[..., None].expand(-1, -1, 3))
        hwl = torch.tensor([1.56, 1.6, 3.9], device=dev) * 2.0
        ry = (torch.rand((args.batch, 12, 1), generator=g) * 6.283 - 3.1416).to(dev)
        gt = torch.cat([ctr[..., 0:1], ctr[..., 1:2] + hwl[0] / 2, ctr[..., 2:3], hwl.expand(args.batch, 12, 3), ry], 2).contiguous()
        cls, reg = ops.rpn_labels(pts, gt)
        batches.append({"pts_input": pts, "rpn_cls_label": cls.long(), "rpn_reg_label": reg})
    prefetch = os.environ.get("PRCNN_TRAIN_PREFETCH", "1") != "0"

    def timed_loop(steps, ahead):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            loss_ = trainer.step(batches[k % nslots], next_batch=batches[(k + 1) % nslots] if ahead else None)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return reduce_elapsed(time.perf_counter() - t0, dist, dev), loss_
    # warm-up: MIOpen's kernel selection for the two torch convolutions of the heads and the caching allocator's block pool
    # settle over the first handful of steps (three were not enough: 43 ms per step measured against 28.5 ms in steady state)
    timed_loop(max(1, args.warmup), prefetch)
    elapsed, loss = timed_loop(args.steps, prefetch)
    elapsed_other, _ = timed_loop(min(args.steps, 10), not prefetch)
    nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
    allreduce_ms = None
    if dist is not None:                                     # the gradient all-reduce on its own: one flat fp32 buffer of the same size
        flat = torch.zeros(nparam, device=dev)
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        allreduce_ms = round((time.perf_counter() - t1) / 20 * 1e3, 4)
    # forward / backward / optimizer split of one step (events, rank 0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    trainer.model.train()
    trainer.optimizer.zero_grad(set_to_none=True)
    ev[0].record()
    loss = trainer.loss(batches[0])
    ev[1].record()
    loss.backward()
    ev[2].record()
    torch.nn.utils.clip_grad_norm_(model.parameters(), trainer.cfg.GRAD_NORM_CLIP)
    trainer.optimizer.step()
    ev[3].record()
    torch.cuda.synchronize()
    fg = int((batches[0]["rpn_cls_label"] > 0).sum().item())
    gemm = stack_gemm_account(lambda: trainer.step(batches[0]), elapsed / args.steps)
    extra = {}
    if rank == 0 and not args.no_roofline:
        extra["roofline"] = train_roofline(lambda: trainer.step(batches[0]), gemm)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        extra["cpu_baseline"] = train_cpu_baseline(model, batches[0]["pts_input"].cpu(), whole_job_value(args.batch, world, args.steps, elapsed))
    return {
        **extra,
        "metric": "KITTI frames/sec, RPN training step (%d pts/frame, bs%d per GPU, DDP over RCCL)" % (args.npoints, args.batch),
        "value": round(whole_job_value(args.batch, world, args.steps, elapsed), 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: train_rcnn.py --train_mode rpn, one iteration per step (forward, focal + bin-based "
                               "regression loss, backward, gradient all-reduce, grad-norm clip 1.0, AdamW step), tools/cfgs/default.yaml, "
                               "%d pts/frame, batch %d per GPU, synthetic labels (%d foreground points in batch 0)" % (args.npoints, args.batch, fg),
                   "frames_per_gpu": args.batch, "global_batch": args.batch * world, "npoints": args.npoints,
                   "parallelism": "dp%d (DistributedDataParallel, RCCL)" % world if world > 1 else "single GPU (no DDP wrapper)",
                   "clouds": args.clouds, "launch": "eager (autograd)", "inputs": "resident in HBM"},
        "train": {"loss_last": round(float(loss.item()), 4), "parameters": nparam, "gradient_bytes": nparam * 4,
                  "allreduce_ms_flat_buffer": allreduce_ms,
                  "forward_loss_ms": round(ev[0].elapsed_time(ev[1]), 3), "backward_ms": round(ev[1].elapsed_time(ev[2]), 3),
                  "clip_optimizer_ms": round(ev[2].elapsed_time(ev[3]), 3),
                  "stack_gemms": gemm,
                  "train_fused": os.environ.get("PRCNN_TRAIN_FUSED", "1") != "0",
                  "padding_free_rows": os.environ.get("PRCNN_TRAIN_DEDUP", "1") != "0",
                  "host_syncs_per_step": 0 if tf.SYNC_FREE_LOSS else 4,
                  "fps_prefetch": prefetch,
                  ("ms_per_step_without_fps_prefetch" if prefetch else "ms_per_step_with_fps_prefetch"):
                      round(1e3 * elapsed_other / min(args.steps, 10), 3),
                  "fps_prefetch_note": "the furthest-point sample sets of batch k+1 (a function of its coordinates only) are drawn on a "
                                       "side stream during step k, as a prefetching data loader allows; every step still does the same work",
                  "note": "every SharedMLP stack (gather / interpolation, 1x1 convs, training-mode BatchNorm, ReLU, max-pool) forward AND "
                          "backward is this package's hand-written kernels (csrc/mlp_train.h: MFMA forward / dgrad / wgrad, BatchNorm "
                          "reductions), one library call per stack and direction, on the padding-free rows of every group (distinct "
                          "rows + multiplicities: exact); the heads run on channels-last rows end to end; torch runs the two bias-only "
                          "output layers (library GEMMs, weight gradient split over row chunks), dropout, the loss (masked means: no "
                          "device-to-host read inside the step) and the optimizer.  A/B switches: PRCNN_TRAIN_FUSED=0 the composed "
                          "torch path (MIOpen / rocBLAS convolutions + BatchNorm), PRCNN_TRAIN_DEDUP=0 nsample-padded rows, "
                          "PRCNN_SYNC_FREE_LOSS=0 the reference's select-then-mean loss"}}


def run_train_rcnn(args, dev, rank, world, local_rank, dist):
    """RCNN-stage training iterations (`train_rcnn.py --train_mode rcnn`, tools/cfgs/default.yaml): the fixed RPN runs in eval mode
    without gradient (hand-written inference path), 512 proposals per frame, ProposalTargetLayer draws 64 RoIs per frame on the
    device, RCNNNet trains on their 512-point canonical clouds; -> the JSON line"""
    from pointrcnn_amd import point_rcnn, rpn, train_functions as tf
    torch.manual_seed(4321)
    model = point_rcnn.PointRCNN(mode="TRAIN").to(dev)
    rpn.randomize_bn_stats(model.rpn, seed=7)
    # The RPN is randomly initialised here (no checkpoint can exist in this environment), so its own proposals never overlap a
    # ground-truth box and every sampled RoI would be background.  The proposal layer still runs inside the step; its RoIs are then
    # replaced by jittered copies of the ground-truth boxes with a noise amplitude growing over the list, i.e. IoUs from ~0.9 down
    # to 0 -- foreground, hard and easy background all present, as a trained RPN's proposals are.
    class SyntheticProposals(torch.nn.Module):
        def __init__(self, real):
            super().__init__()
            self.real, self.gt = real, None

        def forward(self, scores, reg, xyz):
            rois, raw = self.real(scores, reg, xyz)
            B, M = rois.shape[:2]
            G = self.gt.shape[1]
            base = self.gt[:, torch.arange(M, device=rois.device) % G]
            gen = torch.Generator(device=rois.device).manual_seed(17)
            amp = torch.linspace(0.05, 2.5, M, device=rois.device).view(1, M, 1)
            noise = (torch.rand((B, M, 7), generator=gen, device=rois.device) - 0.5) * \
                torch.tensor([1.0, 0.3, 1.0, 0.2, 0.2, 0.4, 0.4], device=rois.device)
            return (base + noise * amp).contiguous(), raw
    model.rpn.proposal_layer = SyntheticProposals(model.rpn.proposal_layer)
    trainer = tf.RCNNTrainer(model, ddp=dist is not None, device_ids=[local_rank] if dist is not None else None)
    make_clouds = make_cloud_fn(args.clouds)
    nslots, batches = 4, []
    g = torch.Generator().manual_seed(199 + rank)
    for s_ in range(nslots):
        pts = make_clouds(args.batch, args.npoints, seed0=shard_seed0(rank, world, s_, args.batch)).to(dev)
        pick = torch.randint(0, args.npoints, (args.batch, 12), generator=g).to(dev)
        ctr = torch.gather(pts, 1, pick[..., None].expand(-1, -1, 3))
        hwl = torch.tensor([1.56, 1.6, 3.9], device=dev)
        ry = (torch.rand((args.batch, 12, 1), generator=g) * 6.283 - 3.1416).to(dev)
        gt = torch.cat([ctr[..., 0:1], ctr[..., 1:2] + hwl[0] / 2, ctr[..., 2:3], hwl.expand(args.batch, 12, 3), ry], 2).contiguous()
        batches.append({"pts_input": pts, "gt_boxes3d": gt})

    prefetch = os.environ.get("PRCNN_TRAIN_PREFETCH", "1") != "0"

    def timed_loop(steps):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            model.rpn.proposal_layer.gt = batches[k % nslots]["gt_boxes3d"]
            loss_ = trainer.step(batches[k % nslots], next_batch=batches[(k + 1) % nslots] if prefetch else None)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return reduce_elapsed(time.perf_counter() - t0, dist, dev), loss_
    timed_loop(max(1, args.warmup))
    elapsed, loss = timed_loop(args.steps)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    trainer.model.train()
    trainer.optimizer.zero_grad(set_to_none=True)
    model.rpn.proposal_layer.gt = batches[0]["gt_boxes3d"]
    ev[0].record()
    loss1 = trainer.loss(batches[0])
    ev[1].record()
    loss1.backward()
    ev[2].record()
    torch.nn.utils.clip_grad_norm_(trainer.params, trainer.grad_norm_clip)
    trainer.optimizer.step()
    ev[3].record()
    torch.cuda.synchronize()
    last = model.rcnn_net.proposal_target_layer.last
    nparam = sum(p.numel() for p in trainer.params)
    gemm = stack_gemm_account(lambda: trainer.step(batches[0]), elapsed / args.steps)
    extra = {}
    if rank == 0 and not args.no_roofline:
        extra["roofline"] = train_roofline(lambda: trainer.step(batches[0]), gemm)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample of the step on the host: its fixed-RPN forward (the step's first half) through the inference CPU baseline and the
        # reference's own roipool3d_cpu; the RCNN stage's forward / backward has no CPU path in the reference and no port here
        model.rpn.eval()
        with torch.no_grad():
            rpn_out = model.rpn({"pts_input": batches[0]["pts_input"]})
        extra["cpu_baseline"] = cpu_baseline(model.rpn, batches[0]["pts_input"].cpu(), rpn_out)
        extra["cpu_baseline"]["what"] = ("FIXED-RPN FORWARD of the step only (+ the reference's roipool3d_cpu in reference_roipool3d); the RCNN "
                                         "stage's forward / backward is not in `value`: " + extra["cpu_baseline"]["what"])
    return {
        **extra,
        "metric": "KITTI frames/sec, RCNN-stage training step (%d pts/frame, bs%d per GPU, 64 RoIs x 512 pts per frame)" % (args.npoints, args.batch),
        "value": round(whole_job_value(args.batch, world, args.steps, elapsed), 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "train_rcnn.py --train_mode rcnn (tools/cfgs/default.yaml: RCNN.ROI_SAMPLE_JIT, USE_BN False), one iteration "
                               "per step: fixed RPN forward (eval, no gradient), proposal layer (TRAIN: 512 RoIs), ProposalTargetLayer (64 "
                               "RoIs per frame sampled + augmented + pooled on the device), RCNNNet forward, BCE + bin-based regression "
                               "loss, backward, grad-norm clip 1.0, AdamW step; %d pts/frame, batch %d per GPU, 12 synthetic ground-truth "
                               "boxes per frame; the (untrained) RPN's RoIs are replaced after the proposal layer by jittered ground-truth "
                               "boxes so that the sampler sees foreground and hard / easy background" % (args.npoints, args.batch),
                   "frames_per_gpu": args.batch, "global_batch": args.batch * world, "npoints": args.npoints,
                   "rois_per_step": args.batch * 64,
                   "parallelism": "dp%d (DistributedDataParallel, RCCL)" % world if world > 1 else "single GPU (no DDP wrapper)",
                   "clouds": args.clouds, "launch": "eager (autograd)", "inputs": "resident in HBM"},
        "train": {"loss_last": round(float(loss.item()), 4), "parameters_trained": nparam,
                  "rpn_and_sampling_ms": None, "forward_loss_ms": round(ev[0].elapsed_time(ev[1]), 3),
                  "backward_ms": round(ev[1].elapsed_time(ev[2]), 3), "clip_optimizer_ms": round(ev[2].elapsed_time(ev[3]), 3),
                  "stack_gemms": gemm,
                  "train_fused": os.environ.get("PRCNN_TRAIN_FUSED", "1") != "0", "fps_prefetch": prefetch,
                  "sampler_status_frames_ok": int((last["status"] == 0).sum().item()),
                  "foreground_rois_last_step": int((last["counts"][:, 3]).sum().item()),
                  "note": "every RCNN SharedMLP stack (Conv with bias -> ReLU, no BatchNorm; gather + max-pool in the SA levels, the "
                          "GroupAll level, the two per-point stacks) forward AND backward is this package's hand-written kernels "
                          "(csrc/mlp_train.h); the heads run on channels-last rows; RoI sampling is one device launch per batch; "
                          "PRCNN_TRAIN_FUSED=0 = the composed torch path for A/B"}}


def reference_route(dev, clouds_cpu, mirror, mirror_out, nms_type, steps, warmup, latency_value=None):
    """frames/s of the reference's UNCHANGED lib/net + the loop body of tools/eval_rcnn.py on the drop-in (bench_reference.py), with the
    ratio to the mirror's one-batch-in-flight figure; None when no reference tree is on this box"""
    import bench_reference
    if not bench_reference.available():
        return None
    ref = bench_reference.measure(dev, clouds_cpu, mirror, steps=steps, warmup=warmup, nms_type=nms_type, mirror_out=mirror_out)
    if latency_value:
        ref["mirror_value_latency_mode"] = latency_value
        ref["eval_loop_vs_mirror_latency_mode"] = round(ref["value_eval_loop"] / latency_value, 3)
        ref["model_only_vs_mirror_latency_mode"] = round(ref["value_model_only"] / latency_value, 3)
    return ref


def run_reference(args, dev, rank, world, dist):
    """`--workload reference`: every rank runs the reference's own evaluation loop body on its shard of frames (no collective)"""
    from pointrcnn_amd import rpn
    torch.manual_seed(1234)
    mirror = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    clouds = make_cloud_fn(args.clouds)(args.batch, args.npoints, seed0=shard_seed0(rank, world, 0, args.batch))
    with torch.no_grad():
        mo = mirror({"pts_input": clouds.to(dev)})
    nms = args.proposals if args.proposals != "off" else "normal"
    if dist is not None:
        dist.barrier()
    ref = reference_route(dev, clouds, mirror, mo, nms, args.steps, args.warmup)
    if ref is None:
        raise SystemExit("bench.py --workload reference: no reference tree on this box (oracle/_ref/reference_py.tar.gz is staged by "
                         "__graft_entry__.build() in the build container)")
    elapsed = reduce_elapsed(ref["eval_loop_ms_per_batch"] * 1e-3 * args.steps, dist, dev)
    return {"metric": "KITTI frames/sec, RPN inference through the reference's unchanged lib/net + tools/eval_rcnn.py loop body on the drop-in "
                      "(%d pts/frame, bs%d per GPU)" % (args.npoints, args.batch),
            "value": round(whole_job_value(args.batch, world, args.steps, elapsed), 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": arithmetic_label(__import__("pointrcnn_amd.ops", fromlist=["x"]).MLP_SPLIT_TERMS), "data": "synthetic",
            "config": {"workload": "reference caller (lib/net/point_rcnn.py, lib/rpn/proposal_layer.py, unchanged) on pointrcnn_amd/dropin: H2D + "
                                   "backbone + heads + sigmoid + proposal layer (%s NMS) + read-back, one batch in flight, eager" % nms,
                       "frames_per_gpu": args.batch, "npoints": args.npoints, "clouds": args.clouds},
            "reference_lib_net": ref}


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only implementation of the hot path")
    world, rank, local_rank, relaunch = resolve_world(args, os.environ, torch.cuda.device_count())
    if relaunch:
        cmd = self_launch_command(args.gpus, sys.argv[1:])
        print("[bench] --gpus %d without a launcher environment: re-executing as %s" % (args.gpus, " ".join(cmd[:8]) + " ..."), file=sys.stderr)
        os.execv(cmd[0], cmd)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))          # RCCL over xGMI
        preflight = distributed_preflight(dist, world, rank, dev)

    from pointrcnn_amd import _cabi, rpn
    _cabi.lib()
    if args.workload == "reference":
        line = run_reference(args, dev, rank, world, dist)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload in ("train", "train-rcnn"):
        line = (run_train if args.workload == "train" else run_train_rcnn)(args, dev, rank, world, local_rank, dist)
        if dist is not None:
            line["distributed"] = preflight
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.manual_seed(1234)
    if args.workload == "rcnn":
        from pointrcnn_amd.point_rcnn import PointRCNN
        model = rpn.randomize_bn_stats(PointRCNN(mode="TEST"), seed=7).to(dev).eval()
        args.proposals = "off"                     # the two-stage model runs its own proposal layer
    else:
        model = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    if args.streams is None:      # rcnn: each in-flight batch holds ~9 GB of worst-case-sized RoI-stage buffers (20 GB until round 6: 10 slots)
        args.streams = 20 if args.workload == "rpn" else 16
    nstreams = max(1, args.streams)

    raw = None
    if args.input == "raw":
        # one packed batch of synthetic scans per in-flight slot: pinned on the host, preallocated on the device
        import numpy as np
        from pointrcnn_amd import kitti_input, ops as _pops
        calib0 = kitti_input.Calibration.from_text(kitti_input.KITTI_CALIB_TXT)
        prep = kitti_input.ScenePreparer(npoints=args.npoints, device=dev)
        raw_slots = []
        for s_ in range(nstreams):
            scans = [kitti_input.synthetic_scan(args.raw_points - 37 * f, seed=1000 + (world * s_ + rank) * args.batch + f, fov_frac=0.2,
                                                far_frac=0.1) for f in range(args.batch)]
            host_pack = prep.pack(scans, [calib0] * args.batch, [(375, 1242)] * args.batch)
            raw_slots.append({"host": host_pack, "dev": {k: host_pack[k].to(dev) for k in ("raw", "offsets", "calib", "img_hw")},
                              "max_points": host_pack["max_points"], "seed": 17 + s_})
        raw = {"slots": raw_slots, "prep": prep, "ops": _pops}

    proposal_layer = None
    if args.proposals != "off":
        from pointrcnn_amd.proposal_layer import ProposalConfig, ProposalLayer
        proposal_layer = ProposalLayer("TEST", cfg=type("Cfg", (ProposalConfig,), {"NMS_TYPE": args.proposals}))

    trace("model built; preparing %d slots" % nstreams)
    bench = InferenceBench(args, model, dev, rank, world, args.clouds, proposal_layer, raw).prepare()
    trace("slots captured; timed loop")
    hbm = None
    if dev.type == "cuda":
        # what the in-flight batches hold: every slot keeps a whole step's worst-case-sized buffers alive (a captured graph's private pool).
        # Beyond what fits the step rate falls off a cliff, not a slope (two-stage workload: 10 slots 7.2 k frames/s, 14 slots 47 -- DESIGN 8)
        free_b, total_b = torch.cuda.mem_get_info(dev)
        hbm = {"reserved_GB": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 1), "free_GB": round(free_b / 2 ** 30, 1),
               "total_GB": round(total_b / 2 ** 30, 1), "slots": nstreams}
        # every rank decides on the SAME flag (max over ranks): one rank leaving while the others enter the timed loop's barrier would hang
        # them until the collective times out
        too_full = free_b < 0.01 * total_b and not os.environ.get("PRCNN_BENCH_ALLOW_OVERSUBSCRIPTION")
        if dist is not None:
            flag = torch.tensor([int(too_full)], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            too_full_any = bool(flag.item())
        else:
            too_full_any = too_full
        if too_full_any and not too_full:
            raise SystemExit(1 if rank else "bench.py: another rank's in-flight batches do not fit its HBM; lower --streams")
        if too_full:
            # (the runtime pages oversubscribed memory instead of failing the allocation: a 14-slot two-stage run takes 680 ms per step)
            raise SystemExit(1 if rank else "bench.py: %d slots hold %.1f of %.1f GB of HBM (%.1f GB free): the in-flight batches do not fit and the step "
                             "rate would collapse; lower --streams (PRCNN_BENCH_ALLOW_OVERSUBSCRIPTION=1 runs anyway)"
                             % (nstreams, hbm["reserved_GB"], total_b / 2 ** 30, free_b / 2 ** 30))
        if free_b < 0.05 * total_b and rank == 0:
            print("[bench] WARNING: %d slots leave %.1f of %.1f GB of HBM free; expect the step rate to collapse -- lower --streams"
                  % (nstreams, free_b / 2 ** 30, total_b / 2 ** 30), file=sys.stderr, flush=True)
    elapsed = bench.timed(args.steps, args.warmup, dist, h2d=args.h2d)
    trace("timed loop done: %.3f ms/step" % (1e3 * elapsed / args.steps))
    out, clouds_cpu, graph = bench.out, bench.clouds_cpu, bench.graphs

    line = {
        "metric": ("KITTI frames/sec, RPN inference end-to-end (%d pts/frame, bs%d per GPU)" % (args.npoints, args.batch)) if args.workload == "rpn"
        else ("KITTI frames/sec, full two-stage PointRCNN inference (%d pts/frame, 100 RoIs/frame, bs%d per GPU)" % (args.npoints, args.batch)),
        "value": round(whole_job_value(args.batch, world, args.steps, elapsed), 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("Full RPN PointNet++ backbone (4 SA-MSG + 4 FP) + cls/reg heads, tools/cfgs/default.yaml, "
                                "%d pts/frame, batch %d per GPU, random-init weights, eval-mode BN" % (args.npoints, args.batch))
                   if args.workload == "rpn" else
                   ("BASELINE config 3: RPN + proposal layer + roipool3d + RCNN (3 SA levels on 100 RoIs x 512 pts) + box decode + "
                    "rotated NMS, tools/cfgs/default.yaml, %d pts/frame, batch %d per GPU, random-init weights" % (args.npoints, args.batch)),
                   "frames_per_gpu": args.batch, "npoints": args.npoints, "parallelism": "frames sharded, dp%d" % world,
                   "launch": "hipGraph replay" if graph is not None else "eager", "streams": nstreams,
                   "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "hbm_held_by_the_slots": hbm,
                   "proposal_layer": args.proposals,
                   "inputs": ("raw velodyne scans (%d pts x 16 B per frame) in pinned host memory -> H2D -> prcnn_scene_prepare, all inside "
                              "the timed region" % args.raw_points) if args.input == "raw" else
                             ("host (pinned) -> HBM copy inside the timed region" if args.h2d else
                              "resident in HBM (bench contract); the PCIe-inclusive rate of the same command is value_h2d_inclusive"),
                   "clouds": args.clouds, "group_dedup": os.environ.get("PRCNN_GROUP_DEDUP", "1") != "0",
                   "roi_dedup": os.environ.get("PRCNN_ROI_DEDUP", "1") != "0"},
    }
    from pointrcnn_amd import ops as _ops_split
    line["dtype"] = arithmetic_label(_ops_split.MLP_SPLIT_TERMS)
    plain = args.workload == "rpn" and args.input == "clouds" and not args.h2d
    if plain and not args.no_variants:
        # SURVEY 8(d)(i) counts the H2D copy: same graphs, every batch's clouds copied from pinned host memory on the batch's stream
        trace("h2d-inclusive loop")
        e2 = bench.timed(args.steps, min(args.warmup, nstreams), dist, h2d=True)
        trace("latency-mode loop")
        line["value_h2d_inclusive"] = round(whole_job_value(args.batch, world, args.steps, e2), 2)
        # a latency-sensitive caller keeps ONE batch in flight: then the FPS serial chain (1 workgroup per frame) is exposed
        lsteps = min(args.steps, 48)
        e3 = bench.timed_single(lsteps, dist)
        line["value_latency_mode"] = round(whole_job_value(args.batch, world, lsteps, e3), 2)
        line["latency_mode_ms_per_batch"] = round(1e3 * e3 / lsteps, 3)
        # between the one-batch caller and the 20-slot engine: the rate at a bounded number of batches in flight (a double-buffered
        # eval loop is depth 2: the next batch's FPS chain runs under this batch's MLPs)
        line["value_by_batches_in_flight"] = {"1": line["value_latency_mode"]}
        for depth in (2, 4, 8):
            if depth < nstreams:
                ed = bench.timed_depth(lsteps, depth, dist)
                line["value_by_batches_in_flight"][str(depth)] = round(whole_job_value(args.batch, world, lsteps, ed), 2)
        line["value_by_batches_in_flight"][str(nstreams)] = line["value"]
        if world == 1:
            # what an UNMODIFIED user of the reference gets: lib/net + the loop body of tools/eval_rcnn.py, unchanged, on the drop-in
            trace("reference route")
            try:
                ref = reference_route(dev, clouds_cpu, model, out, args.proposals if args.proposals != "off" else "normal", 12, 3,
                                      line["value_latency_mode"])
                line["value_reference_lib_net"] = ref["value_eval_loop"] if ref else None
                line["reference_lib_net"] = ref if ref else "no reference tree on this box (oracle/_ref/reference_py.tar.gz not staged)"
            except Exception as e:  # noqa: BLE001  (a side measurement must not cost the line; `--workload reference` raises)
                line["value_reference_lib_net"] = None
                line["reference_lib_net"] = "failed: %s: %s" % (type(e).__name__, str(e).split("\n")[0][:300])
                torch.cuda.synchronize()

    fam = None
    if rank == 0 and not args.no_roofline:
        nprof = min(3, args.steps)
        trace("instrumented pass")
        if args.dump_launches:
            instrumented_pass(args, bench, 1, dump=sys.stderr)
        fam = instrumented_pass(args, bench, nprof)                     # one batch on the chip (single stream, eager)
        by_launch = fam.pop("mlp_by_launch", None)
        ovh = fam.pop("event_pair_overhead_us", 0.0)
        mlp = fam.get("mlp", {"ms": 0.0, "launches": 1, "flops": 0.0, "rows": 0, "rows_launched": 0, "pipe_seconds_at_peak": 0.0})
        # Sum of the family's bracketed launch durations minus what an empty event pair measures; frac = the time its MFMA work takes at
        # the dense peak OF THE PIPE EACH LAUNCH USES (2.5 PFLOP/s bf16 for the split launches, 157.3 TFLOP/s for the fp32-MFMA
        # launches) / that time == the time-weighted mean of the launches' pipe fractions.
        # Durations are those of a launch ALONE on the chip: HIP events cannot time a kernel in the 20-stream configuration -- with 20
        # streams feeding the queues an event pair brackets the stream's SHARE of the chip, not the kernel (tried: 19.4 ms for the
        # family against 1.24 ms alone and 1.5 ms per step in the rocprofv3 kernel trace of the timed loop).  The timed configuration's
        # figure comes from that trace: profiles/recompute_roofline.py (same definition; the family runs ~10 % longer under load).
        # (No event-overhead correction: an empty event pair reads `event_pair_overhead_us`, but around a kernel most of that hides
        #  under the launch -- round 4 measured +1.5 us per launch, +3 % on the family, against the rocprofv3 trace; the figure is
        #  left conservative.)
        secs = max(1e-9, mlp["ms"] * 1e-3)
        frac = mlp["pipe_seconds_at_peak"] / secs
        # the algorithmic figure is the REFERENCE GRAPH's dense count (14.95 GFLOP/frame at 16 384 points).  Two exact
        # rewrites -- first-layer hoisting and padding-free grouping -- remove most of it before anything is multiplied, so the
        # reference-graph rate is a THROUGHPUT figure (it may exceed any peak); utilisation is priced on what the pipe is given.
        ref_flops = rpn.rpn_flops_per_frame() * args.batch * nprof if (args.npoints == 16384 and args.workload == "rpn") else None
        line["roofline"] = {
            "kernel": "fused per-point MLP family: mlp_layer_s_kernel / mlp_chain_s_kernel (bf16 MFMA, exact 3-way operand split, %d products per "
                      "fp32 product, fp32 accumulate) + fp32-MFMA grouped SA chains (sa_xyz_chain / mlp_chain_fast / mlp_stack2); fused gather / "
                      "interpolation / bias / ReLU / max-pool" % _ops_split.MLP_SPLIT_TERMS if _ops_split.MLP_SPLIT_TERMS else
                      "fused per-point MLP family on fp32 MFMA (mlp_layer_b / mlp_chain_fast / sa_xyz_chain / mlp_stack2; PRCNN_MLP_SPLIT=0)",
            "bound": "mfma", "unit": "TFLOP/s",
            "definition": "frac = sum over the family's launches of (MFMA flops given to the pipe / dense peak of THAT pipe: 2500 TFLOP/s bf16 for the "
                          "split launches, 157.3 TFLOP/s for the fp32-MFMA launches) / sum of the launches' durations (HIP events on the launch "
                          "stream, one batch on the chip, uncorrected: they read ~3 % long against the rocprofv3 trace); achieved = frac x peak of the pipe that holds "
                          "most of the time.  profiles/recompute_roofline.py derives the same number from the rocprofv3 kernel traces (single stream "
                          "and the 20-stream timed loop) and the MFMA instruction counters",
            "frac": round(frac, 4),
            "family_us_per_step": round(1e6 * secs / nprof, 1),
            "launches_per_step": mlp["launches"] // nprof, "event_pair_overhead_us": round(ovh, 2),
            "algorithmic_GFLOP_per_step_reference_graph": round(ref_flops / nprof / 1e9, 1) if ref_flops else None,
            "executed_fp32_equivalent_GFLOP_per_step": round(mlp["flops"] / nprof / 1e9, 2),
            "mfma_GFLOP_per_step_on_the_pipes": None,
            "fp32_equivalent_TFLOPs": round(mlp["flops"] / secs / 1e12, 2),
            "fp32_equivalent_vs_fp32_mfma_peak": round(mlp["flops"] / secs / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "reference_graph_TFLOPs": round(ref_flops / secs / 1e12, 1) if ref_flops else None,
            "rows_per_step": mlp["rows"] // nprof, "rows_per_step_without_dedup": mlp.get("rows_undedup", mlp["rows_launched"]) // nprof,
            "flops_per_step": mlp["flops"] / nprof,
            "traffic": None,
            "note": "fp32_equivalent_* = executed 2 K N flops per live row (after the exact rewrites) over the same time, against the 157.3 TFLOP/s "
                    "fp32-MFMA peak: what the layers would need on the fp32 matrix pipe -- a separately named figure, not the roofline fraction; "
                    "reference_graph_TFLOPs = SURVEY 8(d)'s dense count over the same time (a throughput, not a utilisation)"}
        if by_launch:
            for row in by_launch:
                if row["mfma_GFLOP_on_pipe"] > 0:
                    row["frac_of_pipe_peak"] = round(min(1.0, row["mfma_GFLOP_on_pipe"] * 1e3 / row["us"] / row["pipe_peak_TFLOPs"]), 4)
                    row["fp32_equivalent_TFLOPs"] = round(row["fp32_equivalent_GFLOP"] * 1e3 / row["us"], 1)
            line["roofline"]["by_kernel"] = by_launch
            pipes = {}
            for nm, peak in (("bf16", BF16_MFMA_PEAK_TFLOPS), ("fp32", FP32_MFMA_PEAK_TFLOPS)):
                rows_ = [r for r in by_launch if r["pipe"].startswith(nm)]
                us = sum(r["us"] for r in rows_)
                gf = sum(r["mfma_GFLOP_on_pipe"] for r in rows_)
                pipes[nm] = {"launches": len(rows_), "us_per_step": round(us, 1), "mfma_GFLOP_per_step": round(gf, 2),
                             "TFLOPs": round(gf * 1e3 / us, 1) if us > 0 else None, "peak_TFLOPs": peak,
                             "frac_of_pipe_peak": round(gf * 1e3 / us / peak, 4) if us > 0 else None}
            line["roofline"]["by_pipe"] = pipes
            line["roofline"]["mfma_GFLOP_per_step_on_the_pipes"] = {k: v["mfma_GFLOP_per_step"] for k, v in pipes.items()}
            dominant = "bf16" if pipes["bf16"]["us_per_step"] >= pipes["fp32"]["us_per_step"] else "fp32"
        else:
            dominant = "bf16" if _ops_split.MLP_SPLIT_TERMS else "fp32"
        line["roofline"]["peak"] = BF16_MFMA_PEAK_TFLOPS if dominant == "bf16" else FP32_MFMA_PEAK_TFLOPS
        line["roofline"]["achieved"] = round(line["roofline"]["frac"] * line["roofline"]["peak"], 2)
        fam_timed = fam
        line["kernels"] = {k: {"ms_per_step": round(v["ms"] / nprof, 3), "launches_per_step": v["launches"] // nprof}
                           for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        # HBM bytes per launch of the family, MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
        # MI355X_MICROARCH.md prescribes) of a short single-stream eager run of this same script as subprocesses; null when rocprofv3
        # is unavailable, fails or --no-traffic is given
        if world == 1 and not args.no_traffic and args.workload == "rpn" and args.input == "clouds":
            trace("hbm traffic passes")
            line["roofline"].update(measure_hbm_traffic(args))
        if "fps" in fam:      # the longest single kernel is latency/VALU-bound, neither HBM nor MFMA: report its rate
            evals = args.batch * sum(n * m for n, m in zip([args.npoints] + rpn.RPNConfig.SA_NPOINTS[:-1], rpn.RPNConfig.SA_NPOINTS))
            line["fps_kernel"] = {"ms_per_step": round(fam["fps"]["ms"] / nprof, 3), "distance_evals_per_step": evals,
                                  "Gevals_per_s": round(evals / (fam["fps"]["ms"] / nprof * 1e-3) / 1e9, 1),
                                  "note": "serial chain, 1 workgroup/frame (32 of 256 CUs); hidden by --streams"}
            if args.npoints == 16384 and list(rpn.RPNConfig.SA_NPOINTS) == [4096, 1024, 256, 64]:
                # Issue-bound model of the serial chain (round 5: DYNAMIC instruction counts).  Every sample is one trip of the kernel's sample
                # loop on the wave that finishes last -- an updating wave on its fast path; a lone wave issues at most one instruction per
                # 4 cycles, so the floor of a frame's chain is (instructions on that path) x 4 cycles x samples.  Counts from the gfx950 ISA
                # (hipcc -S, block by block): level 0 fps_slot_kernel<16,16>: 168 + 14 per reachable pair of slots, 3.2 pairs per update
                # measured (tools/fps_timing.py) = 213 (bound test 19, pair dispatch 23, maximum 8, slot masks + wave maximum 42, owner
                # search + selects 26, candidate 9, publish 11, barrier + collect 19, loop 11; the tie path, 190 instructions, runs only on
                # exactly equal distances; a wave that does not update issues 62); level 1 fps_pruned_kernel<4>: 170; level 2
                # fps_reg_kernel<64,16> (one wave, no exchange) 211; level 3 fps_reg_kernel<64,4> 95.  Rounds 3-4 priced the STATIC loop
                # (494: all eight pair blocks and the tie path), which the round-5 kernel undercuts.
                instr = (213 * 4095 + 170 * 1023 + 211 * 255 + 95 * 63)
                floor_ms = instr * 4 / 2.4e9 * 1e3
                line["fps_kernel"]["issue_model"] = {
                    "bound": "instruction issue of one wave (serial chain)", "instructions_per_frame_chain": instr,
                    "floor_ms_per_frame_chain": round(floor_ms, 3), "achieved_ms_per_frame_chain": round(fam["fps"]["ms"] / nprof, 3),
                    "frac": round(min(1.0, floor_ms / (fam["fps"]["ms"] / nprof)), 3),
                    "note": "dynamic instructions of the critical wave x 4 cycles at 2.4 GHz; all frames of a batch run their chains "
                            "concurrently (one workgroup each), so the per-step FPS time IS one frame's chain; frac = floor / measured.  "
                            "What separates the two: a dependent instruction issues after ~6 cycles, not 4 (measured: the one-accumulator "
                            "slot-mask chain against the two-accumulator one), two LDS round trips and one barrier per sample, and four "
                            "waves sharing a SIMD's VALU while all sixteen run the bound test"}
            # first-class roofline entry of the kernel that holds most of the kernel time (it is bound by neither HBM nor MFMA)
            tot = sum(v["ms"] for v in fam_timed.values())
            cus = min(256, args.batch)
            valu_peak = 256 * 4 * 16 * 2 * 2.4e9 / 10 / 1e9            # G evals/s: 256 CUs x 4 SIMDs x 16 lanes x 2 (packed fp32) x 2.4 GHz / 10 VALU ops per evaluation
            line["roofline_fps"] = {
                "kernel": "furthest-point-sampling chain (fps_sort + fps_slot_kernel<16,16> / fps_pruned_kernel<4> / fps_reg_kernel), one workgroup per frame",
                "bound": "instruction issue of one wave per sample (serial chain); neither HBM nor MFMA",
                "achieved": line["fps_kernel"]["Gevals_per_s"], "unit": "G distance-evaluations/s (algorithmic: N x npoint per level, SURVEY 8(d): 71.6 M per frame)",
                "ms_per_step": line["fps_kernel"]["ms_per_step"],
                "share_of_kernel_time_single_stream": round(fam_timed["fps"]["ms"] / tot, 4) if ("fps" in fam_timed and tot > 0) else None,
                "cu_occupancy": "%d of 256 CUs (one workgroup per frame of the batch)" % cus,
                "frac": line["fps_kernel"].get("issue_model", {}).get("frac"),
                "frac_definition": "issue-model floor of one frame's chain (dynamic instruction count of the critical wave x 4 cycles x samples at 2.4 GHz) / measured chain time",
                "valu_model": {"peak_Gevals_per_s_whole_chip": round(valu_peak, 0), "peak_Gevals_per_s_on_the_occupied_CUs": round(valu_peak * cus / 256, 0),
                               "frac_of_whole_chip": round(line["fps_kernel"]["Gevals_per_s"] / valu_peak, 4),
                               "frac_of_occupied_CUs": round(line["fps_kernel"]["Gevals_per_s"] / (valu_peak * cus / 256), 4),
                               "note": "10 VALU operations per evaluation (3 subtractions, 3 products, 2 sums, minimum, maximum-select) at the packed "
                                       "fp32 rate; algorithmic evaluations: the exact box pruning skips ~3/4 of them, so the fraction of the occupied "
                                       "CUs can exceed what the executed instructions alone would give"}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.input == "clouds":
        trace("cpu baseline")
        # two-stage workload: the RPN stage of the same graph on the host (its backbone is ~85 % of the two-stage step) and the
        # reference's own CPU code for the stage-2 pooling next to it; the RCNN stage's MLPs have no CPU path in the reference
        line["cpu_baseline"] = cpu_baseline(model.rpn if args.workload == "rcnn" else model, clouds_cpu, out)
        if args.workload == "rcnn":
            line["cpu_baseline"]["what"] = ("RPN STAGE ONLY of the two-stage graph (backbone + heads; the proposal layer, roipool3d and the "
                                            "RCNN stage are not in `value`; the reference's roipool3d_cpu is timed in reference_roipool3d): "
                                            + line["cpu_baseline"]["what"])
        trace("cpu baseline done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and raw is not None:
        # the input builder's CPU oracle (one thread) on the first frames of slot 0, and a bit-for-bit check of the GPU rows
        import numpy as np
        import oracle
        raw_slots, prep, _pops = raw["slots"], raw["prep"], raw["ops"]
        hp = raw_slots[0]["host"]
        nf = min(4, args.batch)
        off = hp["offsets"].numpy()[:nf + 1]
        t1 = time.perf_counter()
        ref = oracle.scene_prepare(hp["raw"].numpy()[:off[-1]], off, hp["calib"].numpy()[:nf], hp["img_hw"].numpy()[:nf], prep.scope,
                                   args.npoints, raw_slots[0]["seed"])
        dt = time.perf_counter() - t1
        with torch.no_grad():
            got = _pops.scene_prepare(raw_slots[0]["dev"]["raw"], raw_slots[0]["dev"]["offsets"], raw_slots[0]["max_points"],
                                      raw_slots[0]["dev"]["calib"], raw_slots[0]["dev"]["img_hw"], prep.scope, args.npoints, raw_slots[0]["seed"])
        line["cpu_baseline"] = {"value": round(nf / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                "sample": "%d raw scans through oracle scene_prepare (input builder only), %.2f s" % (nf, dt),
                                "gpu_rows_identical": bool(np.array_equal(got[0][:nf].cpu().numpy(), ref[0]) and
                                                           np.array_equal(got[2][:nf].cpu().numpy(), ref[2]))}

    if plain and not args.no_variants and args.clouds == "uniform":
        # The headline depends on the data: padding-free grouping removes the rows ball_query pads with, and the synthetic
        # uniform clouds are sparse (most balls hold 1-3 points).  Two more timed loops of the SAME command make that visible:
        #   value_dedup_off  -- every padded row computed, as the reference does (typical-cloud cost without the optimisation)
        #   value_saturated  -- clouds whose every ball is full: nothing to remove, the split is pure overhead (worst case)
        from pointnet2_lib.pointnet2 import pointnet2_modules as pm
        bench.release()
        vsteps, variants, arith_diff = min(args.steps, 96), {}, {}
        todo = (("dedup_off", "uniform", False), ("saturated", "saturated", True), ("lidar", "lidar", True))
        if os.environ.get("PRCNN_BENCH_VARIANT_SELFCHECK"):     # dev: the headline configuration again, as a variant
            todo = (("repeat", "uniform", True),) + todo + (("repeat2", "uniform", True),)
        for name, kind, dedup in todo:
            pm.GROUP_DEDUP = dedup
            trace("variant %s" % name)
            vb = InferenceBench(args, model, dev, rank, world, kind, proposal_layer, None).warm()
            if dedup and _ops_split.MLP_SPLIT_TERMS and world == 1:      # distance of this arithmetic from the fp32-MFMA graph on this kind of cloud
                mine = {k: vb.out[k].clone() for k in ("backbone_features", "rpn_cls", "rpn_reg")}
                keep_terms, _ops_split.MLP_SPLIT_TERMS = _ops_split.MLP_SPLIT_TERMS, 0
                try:
                    arith_diff[kind] = float("%.3g" % rel_diff(mine, vb.step(0)))
                finally:
                    _ops_split.MLP_SPLIT_TERMS = keep_terms
                del mine
            f2 = None
            if rank == 0 and not args.no_roofline:          # (eager pass before the graphs of this variant exist)
                f2 = instrumented_pass(args, vb, 1).get("mlp")
            vb.prepare()
            ev = vb.timed(vsteps, nstreams, dist)
            variants[name] = {"value": round(whole_job_value(args.batch, world, vsteps, ev), 2), "steps": vsteps}
            if f2 is not None:
                if f2:
                    variants[name].update({"mlp_rows_per_step": f2["rows"], "mlp_rows_launched_per_step": f2["rows_launched"],
                                           "mlp_ms_per_step": round(f2["ms"], 3),
                                           "mlp_TFLOPs_executed": round(f2["flops"] / (f2["ms"] * 1e-3) / 1e12, 2) if f2["ms"] > 0 else None})
            vb.release()
        pm.GROUP_DEDUP = os.environ.get("PRCNN_GROUP_DEDUP", "1") != "0"
        # The other arithmetics of the SAME command, beside the line's own (`dtype`): the fp32-MFMA graph (PRCNN_MLP_SPLIT=0) and the
        # three-term split (dev only: outside the contract).  max_diff_vs_f32_mfma = max |out - out_f32| / max |out_f32| over
        # backbone_features, rpn_cls, rpn_reg of slot 0's batch (contract: 1e-5), on uniform, lidar and saturated clouds.
        if world > 1:
            line["arithmetics"] = "measured on the single-GPU line only (python bench.py)"
        elif not os.environ.get("PRCNN_BENCH_NO_SPLIT"):
            from pointrcnn_amd import ops as _ops
            mine_terms = _ops.MLP_SPLIT_TERMS
            outs, arith = {}, {"note": "value / ms_per_step of the same timed loop under each arithmetic of the fused inference MLPs; mlp_* from an "
                                       "instrumented eager pass (fp32-EQUIVALENT flops: 2 K N per row whatever the number of bf16 terms)"}
            for terms in (mine_terms,) + tuple(t for t in (0, 6, 3) if t != mine_terms):
                _ops.MLP_SPLIT_TERMS = terms
                name = "f32_mfma" if terms == 0 else "split_bf16x%d" % terms
                trace("arithmetic %s" % name)
                try:
                    vb = InferenceBench(args, model, dev, rank, world, "uniform", proposal_layer, None).warm()
                    outs[terms] = {k: vb.out[k].clone() for k in ("backbone_features", "rpn_cls", "rpn_reg")}
                    if terms == mine_terms:                       # (its timed loop is the line's own)
                        vb.release()
                        continue
                    f2 = instrumented_pass(args, vb, 1).get("mlp") if (rank == 0 and not args.no_roofline) else None
                    vb.prepare()
                    ev = vb.timed(vsteps, nstreams, dist)
                    arith[name] = {"value": round(whole_job_value(args.batch, world, vsteps, ev), 2), "steps": vsteps, "ms_per_step": round(1e3 * ev / vsteps, 3)}
                    if f2:
                        arith[name].update({"mlp_ms_per_step": round(f2["ms"], 3),
                                            "mlp_fp32_equivalent_TFLOPs": round(f2["flops"] / (f2["ms"] * 1e-3) / 1e12, 2)})
                    vb.release()
                finally:
                    _ops.MLP_SPLIT_TERMS = mine_terms
            for terms, o in outs.items():
                if terms != 0 and 0 in outs:
                    d = {"uniform": float("%.3g" % rel_diff(o, outs[0]))}
                    if terms == mine_terms:
                        d.update(arith_diff)
                    (arith.setdefault("split_bf16x%d" % terms, {}))["max_diff_vs_f32_mfma"] = d
            line["arithmetics"] = arith
            if "f32_mfma" in arith:
                line["value_f32_mfma"] = arith["f32_mfma"]["value"]
        line["value_dedup_off"], line["value_saturated"] = variants["dedup_off"]["value"], variants["saturated"]["value"]
        line["value_lidar"] = variants["lidar"]["value"]
        if "repeat" in variants:
            line["value_repeat"] = [variants["repeat"]["value"], variants["repeat2"]["value"]]
        line["data_dependence"] = {"typical (uniform clouds, this line's value)": {"value": line["value"],
                                                                                   "mlp_rows_per_step": line.get("roofline", {}).get("rows_per_step")},
                                   "dedup_off (uniform clouds, every padded row computed)": variants["dedup_off"],
                                   "saturated (16384 pts in a 1.6 m cube: every ball full)": variants["saturated"],
                                   "lidar (range-dependent density, ground band + car clusters, same bounds)": variants["lidar"]}

    if dist is not None:
        line["distributed"] = preflight
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
