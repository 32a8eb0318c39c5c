#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c2; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_round4.py tests/test_pipeline.py -m gpu -q --tb=short > $O/newtests.log 2>&1; echo "rc=$?" >> $O/newtests.log
tail -5 $O/newtests.log
PRCNN_BENCH_TRACE=1 timeout 420 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench.err; rc=$?
echo "bench rc=$rc"; grep "^\[bench" $O/bench.err | tail -4; tail -c 300 $O/bench_steps20.json
if [ $rc -ne 0 ]; then
  PRCNN_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err; echo "streams1 rc=$?"; grep "^\[bench" $O/bench_s1.err | tail -3
  PRCNN_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline --graph off > $O/bench_eager.json 2> $O/bench_eager.err; echo "eager rc=$?"; grep "^\[bench" $O/bench_eager.err | tail -3
fi
PRCNN_MLP_SPLIT=6 timeout 400 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider > $O/split6_tests.log 2>&1; echo "split rc=$?" >> $O/split6_tests.log
tail -40 $O/split6_tests.log | cut -c1-220
