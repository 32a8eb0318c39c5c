O=gpurun_out/refresh2; mkdir -p $O
python bench.py --workload rcnn > $O/bench_rcnn.json 2> $O/err.txt
PRCNN_MLP_SPLIT=0 python bench.py --workload rcnn --no-cpu-baseline > $O/bench_rcnn_f32_mfma.json 2>> $O/err.txt
python bench.py --workload train-rcnn > $O/bench_train_rcnn.json 2>> $O/err.txt
python bench.py --workload train-rcnn --no-cpu-baseline 2>> $O/err.txt | tail -c 300
tail -c 400 $O/bench_rcnn.json
