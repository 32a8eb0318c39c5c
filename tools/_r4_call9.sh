#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c9; rm -rf $O; mkdir -p $O
B="python bench.py --no-variants --no-cpu-baseline --no-roofline"
for S in 8 10 12 14 16 20 24; do
  for K in 20 320; do
    timeout 200 $B --steps $K --warmup 5 --streams $S > $O/s${S}_k${K}.json 2> $O/err.txt
    echo "streams $S steps $K: $(cut -c1-200 $O/s${S}_k${K}.json | grep -o '"value": [0-9.]*')"
  done
done
