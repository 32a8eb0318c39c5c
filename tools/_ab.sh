python -m pytest tests/test_gpu_mlp.py tests/test_gpu_reference_unchanged.py -x -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 16 --no-variants --no-traffic --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'family_us', r['family_us_per_step'], 'frac', r['frac'])
for row in r['by_kernel']:
    if 'chain' in row['launch'] and 'split' in row['launch']: print('   ', row['launch'], row['widths'], row['us'], row.get('frac_of_pipe_peak'))
"
