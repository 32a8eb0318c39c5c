#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c11; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_pipeline.py tests/test_gpu_round4.py -m gpu -q --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-200
B="python bench.py --no-cpu-baseline --no-roofline --no-variants --h2d"
for K in 20 320; do
  timeout 200 $B --steps $K --warmup 5 > $O/h2d_cs_k$K.json 2>> $O/err.txt; echo "copy stream  steps $K: $(cut -c1-200 $O/h2d_cs_k$K.json | grep -o '"value": [0-9.]*')"
  PRCNN_PIPELINE_COPY_STREAM=0 timeout 200 $B --steps $K --warmup 5 > $O/h2d_own_k$K.json 2>> $O/err.txt; echo "own streams  steps $K: $(cut -c1-200 $O/h2d_own_k$K.json | grep -o '"value": [0-9.]*')"
done
timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-variants --steps 20 --warmup 5 > $O/res_k20.json 2>> $O/err.txt; echo "resident steps 20: $(cut -c1-200 $O/res_k20.json | grep -o '"value": [0-9.]*')"
timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-variants --input raw --steps 320 > $O/raw_k320.json 2>> $O/err.txt; echo "raw input 320: $(cut -c1-200 $O/raw_k320.json | grep -o '"value": [0-9.]*')"
