#!/bin/bash
# build an A/B variant of the library into ab_libs/: tools/build_variant.sh NAME "-DMACRO ..." [file.hip ...]
# (objects of the listed files are rebuilt with the extra flags, the rest are taken from csrc/)
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/distributed-multi-agent-slam-swarm-robotics-system_amd/csrc
mkdir -p $ROOT/ab_libs /tmp/abobj_$NAME
OBJS=""
for f in qs_api decode slam raycast raycast_tiled grid_ops sparse_fuse ekf ekf_scan frontier icp diag rccl_fuse; do
  if [[ " $* " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $FLAGS -c $CS/$f.hip -o /tmp/abobj_$NAME/$f.o
    OBJS="$OBJS /tmp/abobj_$NAME/$f.o"
  else
    OBJS="$OBJS $CS/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/$NAME.so $OBJS
echo built ab_libs/$NAME.so
