#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c7; rm -rf $O; mkdir -p $O
export PRCNN_MLP_SPLIT=0
timeout 300 rocgdb -batch -ex "set pagination off" -ex "run" -ex "info threads" -ex "bt" -ex "info registers pc" --args python tools/graph_fault_probe2.py bench sa01 > $O/rocgdb.log 2>&1
grep -a -n "received signal\|Thread.*stopped\|in .*kernel\|Memory\|fault\|#0\|#1\|AMDGPU Wave" $O/rocgdb.log | head -40 | cut -c1-260
unset PRCNN_MLP_SPLIT
timeout 400 python bench.py --workload rcnn --steps 40 --warmup 8 > $O/bench_rcnn.json 2> $O/bench_rcnn.err; echo "rcnn rc=$?"; tail -2 $O/bench_rcnn.err | cut -c1-300
timeout 400 python bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"; tail -2 $O/bench_train.err | cut -c1-300
python - <<'P'
import json
for f in ("bench_rcnn", "bench_train"):
    try:
        d = json.load(open("gpurun_out/r4c7/%s.json" % f))
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "dtype")})
        print({k: v for k, v in (d.get("roofline") or {}).items() if k not in ("by_kernel", "note", "kernel")})
        cb = d.get("cpu_baseline") or {}
        print({k: cb.get(k) for k in ("value", "cores", "kind", "sample", "gpu_over_cpu")})
    except Exception as e:
        print(f, "no line:", e)
P
