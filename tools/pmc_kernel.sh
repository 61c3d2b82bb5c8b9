#!/bin/bash
# SQ counters of one kernel: tools/pmc_kernel.sh <kernel substring> "<counters>" <python script + args...>
# (own run, counters only; prints per-dispatch averages)
K=$1; C=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_tmp
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" "$K" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if sys.argv[2] in k:
            agg[(k[:40], row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k[:40], row["Counter_Name"])][1] += 1
for (k, c), (s, n) in sorted(agg.items()):
    print(f"{k:42s} {c:28s} n={n:3d} per_dispatch={s / n:18.1f}")
PY
