#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c6; rm -rf $O; mkdir -p $O
timeout 900 python tools/graph_fault_probe2.py > $O/probe.log 2>&1; cat $O/probe.log | cut -c1-330
# the bench line in the new default arithmetic (split-bf16x6) with everything on
PRCNN_BENCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench.err; echo "bench rc=$?"; grep "^\[bench" $O/bench.err | tail -3
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r4c6/bench_steps20.json"))
    print({k: d.get(k) for k in ("value", "ms_per_step", "dtype", "value_f32_mfma", "value_h2d_inclusive", "value_latency_mode", "value_lidar", "value_saturated", "value_dedup_off")})
    print({k: v for k, v in d["roofline"].items() if k != "by_kernel"})
    print(json.dumps(d.get("arithmetics"))[:1500])
    for r in d["roofline"]["by_kernel"]:
        print("%-30s %-34s rows=%8d us=%7.1f frac=%.2f" % (r["launch"], r["widths"], r["rows"], r["us"], r["frac_of_peak"]))
    print(d["kernels"])
except Exception as e:
    print("no bench line:", e)
P
