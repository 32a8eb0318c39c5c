#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c10; rm -rf $O; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -5 $O/gputests.log | cut -c1-240
PRCNN_BENCH_TRACE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench.err; echo "bench rc=$?"
PRCNN_ADDY_PHASE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_addy0.json 2>> $O/bench.err; echo "bench addy0 rc=$?"
python - <<'P'
import json
for f in ("bench_steps20", "bench_addy0"):
    try:
        d = json.load(open("gpurun_out/r4c10/%s.json" % f))
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "value_latency_mode", "latency_mode_ms_per_batch", "latency_mode_forked_sampling")})
        print("  mlp", d["kernels"]["mlp"], [ (r["widths"][:28], r["us"]) for r in d["roofline"]["by_kernel"] if "addend" in r["widths"]])
    except Exception as e:
        print(f, "no line:", e)
P
