#!/bin/bash
# copy the summaries produced by tools/refresh_profiles.sh (gpurun_out/refresh) into profiles/ (round tag = $1, default r02)
set -e
R=gpurun_out/refresh
T=${1:-r06}
cp $R/bench_default.json profiles/${T}_final_bench.json
cp $R/bench_driver_flags.json profiles/${T}_final_bench_steps20.json
cp $R/bench_streams1.json profiles/${T}_final_bench_streams1.json
cp $R/bench_raw_input.json profiles/${T}_final_bench_raw_input.json
cp $R/bench_rcnn.json profiles/${T}_final_bench_rcnn.json
cp $R/bench_rcnn_f32_mfma.json profiles/${T}_final_bench_rcnn_f32_mfma.json
cp $R/bench_train.json profiles/${T}_final_bench_train.json
cp $R/bench_train_rcnn.json profiles/${T}_final_bench_train_rcnn.json
cp $R/kernel_stats_train_rcnn.txt profiles/${T}_final_kernel_stats_train_rcnn.txt
cp $R/bench_config5_rpn.json profiles/${T}_final_bench_config5_rpn.json
cp $R/kernel_stats.txt profiles/${T}_final_kernel_stats.txt
cp $R/kernel_stats_streams1.txt profiles/${T}_final_kernel_stats_streams1.txt
cp $R/kernel_stats_train.txt profiles/${T}_final_kernel_stats_train.txt
[ -f $R/kernel_stats_rcnn.txt ] && cp $R/kernel_stats_rcnn.txt profiles/${T}_final_kernel_stats_rcnn.txt
cp $R/kernel_stats_f32_mfma.txt profiles/${T}_final_kernel_stats_f32_mfma.txt
cp $R/hbm_traffic.json profiles/${T}_hbm_traffic.json
cp $R/mfma_util.txt profiles/${T}_mfma_util.txt
cp $R/opbench.jsonl profiles/${T}_opbench.jsonl
cp $R/bench_reference.json profiles/${T}_final_bench_reference.json
cp $R/bench_reference_rotate.json profiles/${T}_final_bench_reference_rotate.json
cp $R/kernel_stats_reference.txt profiles/${T}_final_kernel_stats_reference.txt
cp $R/arith_disagreement.json profiles/${T}_arith_disagreement.json
[ -s $R/mlp_chain_timing.txt ] && cp $R/mlp_chain_timing.txt profiles/${T}_mlp_chain_timing.txt
[ -s $R/nms_sweep_timing.txt ] && cp $R/nms_sweep_timing.txt profiles/${T}_nms_sweep_timing.txt
[ -s $R/fps_timing_all_waves.txt ] && cp $R/fps_timing_all_waves.txt profiles/${T}_fps_timing_all_waves.txt
[ -s $R/fps_batch_vs_slot.txt ] && cp $R/fps_batch_vs_slot.txt profiles/${T}_fps_batch_vs_slot.txt
ls -la profiles
