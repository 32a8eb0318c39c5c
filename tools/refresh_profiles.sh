#!/bin/bash
# Regenerates the artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh'
# Outputs land in gpurun_out/refresh/; copy the summaries into profiles/ afterwards (tools/collect_profiles.sh r06).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
# PMC passes first (separate runs per counter, no trace domains besides the kernel trace): the bench line below reads the
# traffic figure they produce
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants --graph off --streams 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
python profiles/derive_hbm_traffic.py $O/pmc_ $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
cp $O/hbm_traffic.json profiles/r06_hbm_traffic.json
python profiles/derive_mfma_util.py $O/pmc_mfma > $O/mfma_util.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline --no-variants > $O/bench_streams1.json 2>> $O/bench_default.err
python bench.py --input raw --no-variants > $O/bench_raw_input.json 2>> $O/bench_default.err
python bench.py --workload rcnn > $O/bench_rcnn.json 2>> $O/bench_default.err
PRCNN_MLP_SPLIT=0 python bench.py --workload rcnn --no-cpu-baseline > $O/bench_rcnn_f32_mfma.json 2>> $O/bench_default.err
python bench.py --workload train > $O/bench_train.json 2>> $O/bench_default.err
python bench.py --workload train-rcnn > $O/bench_train_rcnn.json 2>> $O/bench_default.err
python bench.py --npoints 65536 --batch 8 --steps 64 --no-variants > $O/bench_config5_rpn.json 2>> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python bench.py --no-cpu-baseline --no-roofline --no-variants > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python bench.py --no-cpu-baseline --no-roofline --no-variants --streams 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python bench.py --workload train --steps 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktr -- python bench.py --workload train-rcnn --steps 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc -- python bench.py --workload rcnn --steps 160 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python profiles/summarize_rocprof.py $O/ktc "python bench.py --workload rcnn --steps 160 (config 3: two-stage detector, 10 batches in flight, hipGraph replay)" > $O/kernel_stats_rcnn.txt
PRCNN_MLP_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python bench.py --no-cpu-baseline --no-roofline --no-variants > /dev/null 2>&1
python profiles/summarize_rocprof.py $O/kts "PRCNN_MLP_SPLIT=0 python bench.py (fp32-MFMA arithmetic throughout; 20 batches in flight, hipGraph replay)" > $O/kernel_stats_f32_mfma.txt
python profiles/summarize_rocprof.py $O/ktr "python bench.py --workload train-rcnn --steps 16 (RCNN-stage training step, bs4, eager, fused training path)" > $O/kernel_stats_train_rcnn.txt
python -m pointrcnn_amd.opbench > $O/opbench_raw.jsonl 2> $O/opbench.err
# the reference's unchanged lib/net + evaluation loop body on the drop-in (bench.py --workload reference), its kernel trace, and the rate
# at which the canonical and the upstream (nvcc-contracted) squared-distance arithmetics disagree
python bench.py --workload reference --steps 20 --warmup 3 > $O/bench_reference.json 2>> $O/bench_default.err
python bench.py --workload reference --proposals rotate --steps 20 --warmup 3 > $O/bench_reference_rotate.json 2>> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktref -- python bench_reference.py 12 > /dev/null 2>&1
python profiles/summarize_rocprof.py $O/ktref "python bench_reference.py 12 (the reference's unchanged lib/net + tools/eval_rcnn.py loop body on the drop-in, eager, one batch in flight)" > $O/kernel_stats_reference.txt
python tools/arith_disagreement.py --out $O/arith_disagreement.json > /dev/null 2>> $O/bench_default.err
PRCNN_POINTOPS_LIB=pointrcnn_amd/lib/libprcnn_mlptiming.so python tools/mlp_timing.py > $O/mlp_chain_timing.txt 2>/dev/null
PRCNN_POINTOPS_LIB=pointrcnn_amd/lib/libprcnn_sweeptiming.so python tools/nms_timing.py 6300 > $O/nms_sweep_timing.txt 2>/dev/null
[ -f pointrcnn_amd/lib/libprcnn_fpstiming.so ] && python tools/fps_timing.py --no-build --run 2>/dev/null | grep -v amdgpu.ids > $O/fps_timing_all_waves.txt
python tools/fps_batch_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/fps_batch_vs_slot.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/op_FETCH_SIZE -- python -m pointrcnn_amd.opbench > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/op_WRITE_SIZE -- python -m pointrcnn_amd.opbench > /dev/null 2>&1
python profiles/join_op_traffic.py $O/opbench_raw.jsonl /tmp/op_ $O/opbench.jsonl > /dev/null 2>> $O/opbench.err
python profiles/summarize_rocprof.py $O/kt3 "default: python bench.py (20 batches in flight, hipGraph replay)" > $O/kernel_stats.txt
python profiles/summarize_rocprof.py $O/kt1 "python bench.py --streams 1 (one batch in flight)" > $O/kernel_stats_streams1.txt
python profiles/summarize_rocprof.py $O/ktt "python bench.py --workload train --steps 16 (RPN training step, bs16, eager, fused training path)" > $O/kernel_stats_train.txt
# keep the merge small: only summaries travel back
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -size +2M -delete
tail -c 600 $O/bench_default.json; echo; tail -3 $O/hbm_traffic.txt; head -12 $O/kernel_stats_streams1.txt; tail -4 $O/mfma_util.txt
