timeout 1500 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_modules.py tests/test_gpu_train_mlp.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench2.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step')}, {k:v for k,v in d.items() if k.startswith('value_')})
print({k:r[k] for k in ('achieved','frac','avg_launch_us')})
for x in r['by_kernel']: print(x['launch'][:28].ljust(28), str(x.get('widths'))[:26].ljust(26), x['rows'], round(x['us'],1), x['frac_of_peak'])
PY
python bench.py --workload train --steps 20 2>/dev/null | tail -1 | cut -c1-200
