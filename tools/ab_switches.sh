#!/bin/bash
# same-box A/B of the bit-identical implementation switches under the default bench command
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$* $EXTRA', l['value'], l['ms_per_step'])"; }
EXTRA=""
run A=default
run PRCNN_PERSISTENT_CHAIN=1
run PRCNN_FPS_PRUNED=0
run PRCNN_GROUP_DEDUP=0
run PRCNN_DEDUP_SPARSE_DIV=2
run PRCNN_DEDUP_SPARSE_DIV=8
run PRCNN_NO_SA0=1
run A=default
EXTRA="--clouds lidar"; run A=default
EXTRA="--h2d"; run A=default
EXTRA="--proposals rotate"; run A=default
EXTRA="--proposals off"; run A=default
