#!/bin/bash
# usage: tools/pmc_sets.sh <lib.so|-> <kernel-substring> "<counter set 1>" ["<counter set 2>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ "$1" != "-" ] && export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$1
KSUB=$2; shift 2
k=0
for set in "$@"; do
  k=$((k+1))
  bash tools/prof_pmc.sh "$set" set_$k | grep "$KSUB" || exit 1
done
