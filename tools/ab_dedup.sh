#!/bin/bash
# same-box A/B of the padding-free grouping switches (prints frames/s and MLP ms per step)
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 96 $EXTRA 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); k=l.get('kernels',{}); print('$* $EXTRA', l['value'], l['ms_per_step'], 'mlp', k.get('mlp',{}).get('ms_per_step'), 'compact', k.get('group_compact',{}).get('ms_per_step'), 'segmax', k.get('segmax_scatter',{}).get('ms_per_step'), 'scatter', k.get('scatter_rows',{}).get('ms_per_step'))"; }
for c in uniform lidar; do
  EXTRA="--clouds $c"
  run PRCNN_GROUP_DEDUP=0
  for d in 32 4 2 1; do run PRCNN_DEDUP_SPARSE_DIV=$d; done
done
EXTRA="--clouds uniform --streams 4"; run PRCNN_DEDUP_SPARSE_DIV=4
EXTRA="--clouds uniform --streams 2"; run PRCNN_DEDUP_SPARSE_DIV=4
EXTRA="--clouds lidar --streams 4"; run PRCNN_DEDUP_SPARSE_DIV=4
