#!/bin/bash
# copy the summaries produced by tools/refresh_profiles.sh (gpurun_out/refresh) into profiles/
set -e
R=gpurun_out/refresh
cp $R/bench_default.json profiles/r01_final_bench.json
cp $R/bench_streams1.json profiles/r01_final_bench_streams1.json
cp $R/bench_lidar.json profiles/r01_final_bench_lidar.json
cp $R/bench_raw_input.json profiles/r01_final_bench_raw_input.json
cp $R/bench_rcnn.json profiles/r01_final_bench_rcnn.json
cp $R/kernel_stats.txt profiles/r01_final_kernel_stats.txt
cp $R/kernel_stats_streams1.txt profiles/r01_final_kernel_stats_streams1.txt
cp $R/hbm_traffic.json profiles/r01_hbm_traffic.json
cp $R/mfma_util.txt profiles/r01_mfma_util.txt
cp $R/opbench.jsonl profiles/r01_opbench.jsonl
ls -la profiles
