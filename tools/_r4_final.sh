#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.log | cut -c1-300
bash tools/refresh_profiles.sh > gpurun_out/final/refresh.log 2>&1
tail -c 300 gpurun_out/refresh/bench_driver_flags.json
