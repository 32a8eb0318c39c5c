#!/usr/bin/env python3
"""bench.py -- RPN inference throughput of the MI355X point-ops hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
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
  cpu_baseline -- the CPU oracle restatement of the SAME graph (oracle/rpn_cpu.py, kind "port", 1 thread)
                  timed on a bounded sample (one frame) on this box's host cores, rank 0 at N=1 only.
  kernels      -- per-op-family GPU time of one step (ms) from the same event pass.
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

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32-input MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320,
                    help="timed steps; with 16 batches in flight the first/last steps fill and drain the pipeline (one batch's "
                         "latency is ~45 ms under load), so short runs under-report the steady state by a few percent")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (BASELINE metric: bs32)")
    ap.add_argument("--npoints", type=int, default=16384)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto", help="replay the step from a hipGraph")
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight per GPU, default 16 (rpn) / 10 (rcnn) (each on its own HIP stream AND hardware queue, see GPU_MAX_HW_QUEUES "
                         "above): step k runs on HIP stream k %% streams, so one batch's FPS "
                         "(1 workgroup per frame = 32 of 256 CUs) overlaps another batch's MLP / neighbour kernels")
    ap.add_argument("--proposals", choices=["off", "normal", "rotate"], default="normal",
                    help="also run the proposal layer (decode + sort + distance split + NMS + top-100, lib/rpn/proposal_layer.py) "
                         "inside every step, as tools/eval_rcnn.py --eval_mode rpn does after the heads")
    ap.add_argument("--workload", choices=["rpn", "rcnn"], default="rpn",
                    help="rpn = the BASELINE metric (RPN inference end-to-end); rcnn = BASELINE config 3, the whole two-stage detector "
                         "(RPN -> proposals -> roipool3d -> RCNN -> box decode -> rotated NMS), 100 RoIs per frame")
    ap.add_argument("--clouds", choices=["uniform", "lidar"], default="uniform",
                    help="uniform = the BASELINE metric's synthetic clouds; lidar = range-dependent density + ground band + car "
                         "clusters (same bounds): a robustness check for the spatially pruned / grid kernels")
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
    return ap.parse_args()


class EventProfiler:
    """Wraps the ctypes library: every prcnn_* launch is bracketed by HIP events recorded on the stream the
    kernel is launched on (torch's current stream).  Used only in the instrumented pass, never in the timed one."""

    def __init__(self, lib):
        self._lib = lib
        self.records = []          # (name, start, end, (flops per row, static rows, device row-count pointer, rows per count))
        self.splits = []           # GroupSplit objects of the pass (kept alive: their device counts are read in summary())

    @staticmethod
    def _work(name, a):
        """(flops per row, rows, rows_dev pointer, rows per device count) of one call.  EXECUTED work: first-layer
        hoisting and padding-free grouping make it smaller than the reference graph's count."""
        def chain(k0, nout):
            widths = [k0] + list(nout)
            return 2.0 * sum(x * y for x, y in zip(widths[:-1], widths[1:]))
        if name == "prcnn_mlp_rows":
            return 2.0 * a[3] * a[6], a[2], a[12], a[13]
        if name == "prcnn_mlp_rows_addinterp":
            return 2.0 * (a[2] + 3) * a[5], a[11] * a[12], None, 1
        if name == "prcnn_mlp_group":
            return 2.0 * (a[9] + (0 if a[10] else 3)) * a[14], a[5] * a[7] * a[8], a[20], a[8]
        if name == "prcnn_mlp_interp":
            return 2.0 * (a[9] + a[10]) * a[14], a[6] * a[7], None, 1
        if name == "prcnn_mlp_chain_rows":
            return chain(a[3], a[7]), a[2], None, 1
        if name == "prcnn_mlp_chain_group":
            return chain(a[9] + (0 if a[10] else 3), a[15]), a[5] * a[7] * a[8], a[21], a[8]
        if name == "prcnn_mlp_chain_interp":
            return chain(a[9] + a[10], a[15]), a[6] * a[7], None, 1
        return 0.0, 0, None, 1

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("prcnn_") or name.endswith("_bytes") or name in ("prcnn_last_error", "prcnn_abi_version",
                                                                              "prcnn_wpack_floats"):
            return fn

        def wrapped(*args):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            self.records.append((name, s, e, self._work(name, args)))
            return rc
        return wrapped

    def summary(self):
        torch.cuda.synchronize()
        dev_counts = {}
        for sp in self.splits:
            for k, v in enumerate(sp.counts.cpu().tolist()):
                dev_counts[sp.counts.data_ptr() + 4 * k] = v
        fam = {}
        for name, s, e, (per_row, rows, ptr, unit) in self.records:
            key = "mlp" if name.startswith("prcnn_mlp_") else name[len("prcnn_"):]
            d = fam.setdefault(key, {"ms": 0.0, "launches": 0, "flops": 0.0, "rows": 0, "rows_launched": 0})
            ptr = getattr(ptr, "value", ptr)
            live = min(rows, dev_counts[ptr] * unit) if ptr else rows
            d["ms"] += s.elapsed_time(e)
            d["launches"] += 1
            d["flops"] += per_row * live
            d["rows"] += live
            d["rows_launched"] += rows
        return fam


def cpu_baseline(model, clouds_cpu, gpu_out):
    """the oracle port of the same graph on one frame (bounded sample), single thread"""
    import oracle
    from oracle import rpn_cpu
    cpu = oracle.cpu()
    spec = rpn_cpu.extract_rpn_weights(model)
    timings = {}
    nframes = min(3, clouds_cpu.shape[0])
    t0 = time.perf_counter()
    outs = [rpn_cpu.rpn_forward_frame(cpu, clouds_cpu[f].numpy(), spec, timings) for f in range(nframes)]
    dt = time.perf_counter() - t0
    out = outs[0]
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
        import numpy as np
        from pointrcnn_amd import ops
        rng = np.random.default_rng(0)
        pts = clouds_cpu[0].numpy()
        ctr = pts[rng.integers(0, pts.shape[0], 100)]
        boxes = np.concatenate([ctr[:, :1], ctr[:, 1:2] + 1.8, ctr[:, 2:3], np.tile([3.6, 3.7, 6.0], (100, 1)),
                                rng.uniform(-3.14, 3.14, (100, 1))], 1).astype(np.float32)
        feat = rng.normal(size=(pts.shape[0], 130)).astype(np.float32)
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pp, pf, pe = ref.roipool3d_cpu(pts, boxes, feat, 512)
        cpu_s = (time.perf_counter() - t1) / reps
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
                       "cpu_ms_per_frame": round(cpu_s * 1e3, 2), "cores": 1,
                       "gpu_ms_per_frame_single_frame_launch": round(s_.elapsed_time(e_) / 20, 4), "outputs_identical": same}
    return {"value": round(nframes / dt, 5), "unit": "frames/s", "cores": 1, "kind": "port", "reference_roipool3d": ref_roipool,
            "sample": "%d frames (16384 pts each) of the same RPN graph through oracle/rpn_cpu.py, %.1f s" % (nframes, dt),
            "host_cores_available": os.cpu_count(),
            "breakdown_s": {k: round(v, 2) for k, v in sorted(timings.items())},
            "gpu_vs_oracle_rel_err_frame0": err}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only implementation of the hot path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from pointrcnn_amd import _cabi, rpn
    _cabi.lib()
    torch.manual_seed(1234)
    if args.workload == "rcnn":
        from pointrcnn_amd.point_rcnn import PointRCNN
        model = rpn.randomize_bn_stats(PointRCNN(mode="TEST"), seed=7).to(dev).eval()
        args.proposals = "off"                     # the two-stage model runs its own proposal layer
    else:
        model = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    if args.streams is None:      # rcnn: each in-flight batch holds several GB of worst-case-sized RoI-stage buffers
        args.streams = 16 if args.workload == "rpn" else 10
    nstreams = max(1, args.streams)
    make_clouds = rpn.synthetic_clouds if args.clouds == "uniform" else rpn.lidar_like_clouds
    clouds_cpu = make_clouds(args.batch, args.npoints, seed0=100 + rank * args.batch)
    batches = [{"pts_input": clouds_cpu.to(dev)}]
    for s_ in range(1, nstreams):          # every in-flight slot owns its (resident) input batch
        batches.append({"pts_input": make_clouds(args.batch, args.npoints,
                                                          seed0=100 + (world * s_ + rank) * args.batch).to(dev)})
    streams = [torch.cuda.Stream() for _ in range(nstreams)]

    raw_slots = None
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

    proposal_layer = None
    if args.proposals != "off":
        from pointrcnn_amd.proposal_layer import ProposalConfig, ProposalLayer
        proposal_layer = ProposalLayer("TEST", cfg=type("Cfg", (ProposalConfig,), {"NMS_TYPE": args.proposals}))

    def step(slot=0):
        with torch.no_grad():
            if raw_slots is not None:
                r = raw_slots[slot]
                xyz, _, _, _, r["status"] = _pops.scene_prepare(r["dev"]["raw"], r["dev"]["offsets"], r["max_points"], r["dev"]["calib"],
                                                                r["dev"]["img_hw"], prep.scope, args.npoints, r["seed"])
                o = model({"pts_input": xyz})
            else:
                o = model(batches[slot])
            if proposal_layer is not None:
                o["rois"], o["roi_scores_raw"] = proposal_layer(o["rpn_cls"][:, :, 0], o["rpn_reg"], o["backbone_xyz"])
            if args.workload == "rcnn":
                o["pred_boxes3d"], o["raw_scores"], o["keep"], o["num_keep"] = model.detections(o)
            return o

    for _ in range(max(1, args.warmup)):        # packs weights, fills the caching allocator
        out = step(0)
    torch.cuda.synchronize()

    graphs = None
    if args.graph != "off":
        try:
            graphs, gouts = [], []
            for slot in range(nstreams):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[slot]):
                    step(slot)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=streams[slot]):
                    gouts.append(step(slot))
                graphs.append(g)
            graphs[0].replay()
            torch.cuda.synchronize()
            for k in ("rpn_cls", "rpn_reg"):                 # the replayed graph must reproduce the eager result
                assert torch.equal(gouts[0][k], out[k]), "graph replay differs from eager (%s)" % k
            out = gouts[0]
        except Exception as e:  # noqa: BLE001
            if args.graph == "on":
                raise
            print("[bench] hipGraph capture unavailable (%s); timing eager launches" % str(e).split("\n")[0],
                  file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()

    host = [b["pts_input"].cpu().pin_memory() for b in batches] if args.h2d else None

    def run(k):
        slot = k % nstreams
        with torch.cuda.stream(streams[slot]):
            if host is not None:
                batches[slot]["pts_input"].copy_(host[slot], non_blocking=True)
            if raw_slots is not None:
                raw_slots[slot]["dev"]["raw"].copy_(raw_slots[slot]["host"]["raw"], non_blocking=True)
            if graphs is not None:
                graphs[slot].replay()
            else:
                step(slot)

    for k in range(args.warmup):
        run(k)

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        run(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    graph = graphs

    frames = args.batch * world * args.steps
    line = {
        "metric": ("KITTI frames/sec, RPN inference end-to-end (16384 pts/frame, bs%d per GPU)" % args.batch) if args.workload == "rpn"
        else ("KITTI frames/sec, full two-stage PointRCNN inference (16384 pts/frame, 100 RoIs/frame, bs%d per GPU)" % args.batch),
        "value": round(frames / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("Full RPN PointNet++ backbone (4 SA-MSG + 4 FP) + cls/reg heads, tools/cfgs/default.yaml, "
                                "%d pts/frame, batch %d per GPU, random-init weights, eval-mode BN" % (args.npoints, args.batch))
                   if args.workload == "rpn" else
                   ("BASELINE config 3: RPN + proposal layer + roipool3d + RCNN (3 SA levels on 100 RoIs x 512 pts) + box decode + "
                    "rotated NMS, tools/cfgs/default.yaml, %d pts/frame, batch %d per GPU, random-init weights" % (args.npoints, args.batch)),
                   "frames_per_gpu": args.batch, "npoints": args.npoints, "parallelism": "frames sharded, dp%d" % world,
                   "launch": "hipGraph replay" if graph is not None else "eager", "streams": nstreams,
                   "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "proposal_layer": args.proposals,
                   "inputs": ("raw velodyne scans (%d pts x 16 B per frame) in pinned host memory -> H2D -> prcnn_scene_prepare, all inside "
                              "the timed region" % args.raw_points) if args.input == "raw" else
                             ("host (pinned) -> HBM copy inside the timed region" if args.h2d else "resident in HBM"),
                   "clouds": args.clouds, "group_dedup": os.environ.get("PRCNN_GROUP_DEDUP", "1") != "0",
                   "roi_dedup": os.environ.get("PRCNN_ROI_DEDUP", "1") != "0"},
    }

    if rank == 0 and not args.no_roofline and args.workload == "rpn":
        from pointrcnn_amd import ops as _ops
        prof = EventProfiler(_cabi._lib)
        real = _cabi._lib
        _cabi._lib, _ops._split_log = prof, prof.splits
        try:
            nprof = min(3, args.steps)
            for _ in range(nprof):
                step(0)
            fam = prof.summary()
        finally:
            _cabi._lib, _ops._split_log = real, None
        mlp = fam.get("mlp", {"ms": 0.0, "launches": 1, "flops": 0.0, "rows": 0, "rows_launched": 0})
        secs = mlp["ms"] * 1e-3
        # The roofline is priced on the flops the kernels EXECUTE: first-layer hoisting and padding-free grouping (both
        # exact) remove most of the reference graph's MLP work (14.95 GFLOP/frame, padding rows included),
        # so the reference-graph rate -- reported next to it -- is a throughput figure, not a utilisation, and may exceed
        # the peak.
        ref_flops = rpn.rpn_flops_per_frame() * args.batch * nprof if args.npoints == 16384 else None
        achieved = mlp["flops"] / secs / 1e12 if secs > 0 else 0.0
        line["roofline"] = {"kernel": "mlp_chain_* + mlp_layer_kernel (fused gather/interp + fp32 MFMA + bias/ReLU/max-pool)",
                            "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                            "launches_per_step": mlp["launches"] // nprof,
                            "avg_launch_us": round(1e3 * mlp["ms"] / max(1, mlp["launches"]), 2),
                            "flops_per_step": mlp["flops"] / nprof, "rows_per_step": mlp["rows"] // nprof,
                            "rows_per_step_without_dedup": mlp["rows_launched"] // nprof,
                            "reference_graph_flops_per_step": ref_flops / nprof if ref_flops else None,
                            "reference_graph_TFLOPs": round(ref_flops / secs / 1e12, 3) if ref_flops and secs > 0 else None,
                            "note": "achieved/frac = executed flops (after exact first-layer hoisting and padding-free grouping) / "
                                    "MLP-family GPU time; reference_graph_TFLOPs = the reference's dense flop count over the same time"}
        line["kernels"] = {k: {"ms_per_step": round(v["ms"] / nprof, 3), "launches_per_step": v["launches"] // nprof}
                           for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        # HBM bytes per launch of the same kernel family from the PMC passes committed under profiles/ (rocprofv3
        # --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied there); null when absent.
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath) and args.batch == 32 and args.npoints == 16384:
            try:
                line["roofline"]["traffic"] = json.load(open(tpath))["per_step_bs32"]["mlp"]["hbm_bytes_per_launch_corrected"]
                line["roofline"]["traffic_unit"] = "bytes/launch (PMC, profiles/r01_hbm_traffic.json)"
            except (KeyError, ValueError):
                pass
        if "fps" in fam:      # the longest single kernel is latency/VALU-bound, neither HBM nor MFMA: report its rate
            evals = args.batch * sum(n * m for n, m in zip([args.npoints] + rpn.RPNConfig.SA_NPOINTS[:-1], rpn.RPNConfig.SA_NPOINTS))
            line["fps_kernel"] = {"ms_per_step": round(fam["fps"]["ms"] / nprof, 3), "distance_evals_per_step": evals,
                                  "Gevals_per_s": round(evals / (fam["fps"]["ms"] / nprof * 1e-3) / 1e9, 1),
                                  "note": "serial chain, 1 workgroup/frame (32 of 256 CUs); hidden by --streams"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "rpn" and args.input == "clouds":
        line["cpu_baseline"] = cpu_baseline(model, clouds_cpu, out)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and raw_slots is not None:
        # the input builder's CPU oracle (one thread) on the first frames of slot 0, and a bit-for-bit check of the GPU rows
        import oracle
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

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
