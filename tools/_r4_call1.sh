#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tail -5 $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench.err
tail -c 400 $O/bench_steps20.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python bench.py --no-cpu-baseline --no-roofline --no-variants > /dev/null 2>&1
python profiles/summarize_rocprof.py $O/kt3 "python bench.py (20 batches in flight, hipGraph replay) -- round 4 first run" > $O/kernel_stats.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -size +2M -delete
head -30 $O/kernel_stats.txt
