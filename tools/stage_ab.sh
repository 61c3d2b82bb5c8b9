#!/bin/bash
# stage times of the c3 bench with the default library and with variants (run on the GPU box): tools/stage_ab.sh [ab_libs/x.so ...]
for l in default "$@"; do
  if [ $l = default ]; then unset QUASAR_SLAM_LIB; else export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$l; fi
  python3 bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(j['value']/1e6,1), round(j['ms_per_step'],3), {k: round(v,3) for k,v in j['stages_ms_per_step'].items()})"
done
