#!/bin/bash
# A/B of library builds by HIP-event raycast stage time, several fresh processes per build
# (timing is bimodal per process: physical placement of the hot tiles).  usage: tools/ab_modes.sh RUNS lib...
RUNS=$1; shift
for lib in "$@"; do
  export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$lib
  for i in $(seq $RUNS); do
    echo -n "$(basename $lib .so) "; timeout -k 10 60 python3 tools/raycast_modes.py 2>/dev/null | cut -c1-80 || exit 1
  done
done
