#!/bin/bash
# A/B of library builds: per-kernel average durations (rocprofv3 kernel trace) of the raycast kernels.
# usage: tools/ab_raycast.sh lib1.so lib2.so ...   (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so)
  OUT=gpurun_out/ab_$tag
  rm -rf $OUT && mkdir -p $OUT
  export QUASAR_SLAM_LIB=$GRAFT_REPO_ROOT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/prof_raycast.py 2 1048576 ${BOTS:-2} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for row in csv.DictReader(open(f)):
        k = row["Name"].split("(")[0]
        if any(s in k for s in ("raster", "rays", "scatter", "scan", "raycast")):
            print(f"{sys.argv[2]:14s} {k[:34]:36s} avg_us={float(row['AverageNs']) / 1e3:9.2f} calls={row['Calls']}")
            tot += float(row['AverageNs']) / 1e3
    print(f"{sys.argv[2]:14s} {'SUM':36s} avg_us={tot:9.2f}")
PY
done
