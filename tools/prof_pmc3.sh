#!/bin/bash
# Several SQ counter passes over the raycast kernels, one rocprofv3 run per pass (counters only).
# usage: tools/prof_pmc3.sh [lib.so]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$1" ] && export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$1
k=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  k=$((k+1))
  bash tools/prof_pmc.sh "$set" sq3_$k | grep raster || exit 1
done
