#!/bin/bash
# A/B of a library variant against the default one, chain time per workload (run on the GPU box): tools/ab_lib.sh ab_libs/x.so [workloads...]
LIB=$1; shift
WLS=${@:-c1 g32 adv}
for wl in $WLS; do
  for l in default $LIB; do
    if [ $l = default ]; then unset QUASAR_SLAM_LIB; else export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$l; fi
    echo "$wl $l: $(QS_CHAIN_MODE=free timeout -k 10 200 python tools/slam_probe2.py 1048576 3 $wl 2>/dev/null | tail -1)"
  done
done
