#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c8; rm -rf $O; mkdir -p $O
for st in sa01 backbone proposal; do
  timeout 200 python tools/graph_fault_probe2.py bench $st > $O/probe_$st.log 2>&1; echo "bench $st rc=$? $(grep -a 'CASE\|fault' $O/probe_$st.log | tail -1 | cut -c1-200)"
done
B="python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --no-roofline"
PRCNN_BENCH_SAME_EXAMPLE=1 PRCNN_BENCH_TRACE=1 timeout 200 $B > $O/same20.json 2> $O/same20.err; echo "same20 rc=$? $(grep '^\[bench' $O/same20.err | tail -1) $(cut -c1-140 $O/same20.json | grep -o '"value": [0-9.]*')"
timeout 700 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
tail -6 $O/gputests.log | cut -c1-240
