cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_train_mlp.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
python bench.py --workload train --steps 20 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python bench.py --workload train --steps 16 > /dev/null 2>&1
python profiles/summarize_rocprof.py /tmp/ktt "python bench.py --workload train --steps 16" > gpurun_out/r3_train_stats4.txt
grep -E "train_|bn_|interp_|flat_rows|group_rows" gpurun_out/r3_train_stats4.txt | head -30
