#!/bin/bash
# raster A/B: production raster vs -DQT_CLIP=1 (closed-form entry into the tile; QS_RASTER_SORT=1: records of a
# tile pre-sorted by clipped span in an UNTIMED pass = the ideal regroup; =0: clip alone).  Prints the raster stage per workload.
#   tools/build_variant.sh clip "-DQT_CLIP=1" raycast_tiled.hip; cp csrc/libquasar_slam.so ab_libs/cur.so; tools/ab_raster_clip.sh
run() { QUASAR_SLAM_LIB=$1 QS_RASTER_SORT=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-micro --ekf 0 --steps 10 --warmup 2 $3 2>/dev/null | python -c "
import sys,json; r=json.load(sys.stdin); s=r['stages_ms_per_step']; print('$4', 'raster %.1f us  raycast stage %.1f us  parity %s' % (s['rc_raster']*1e3, s['raycast']*1e3, r['parity_checked']))"; }
for wl in "--workload c1" "--workload c3" "--workload adv" "--workload adv --grid 8192"; do
  echo "== $wl"
  run ab_libs/cur.so 0 "$wl" "production      "
  run ab_libs/clip.so 0 "$wl" "clip alone      "
  run ab_libs/clip.so 1 "$wl" "clip + regroup* "
done
