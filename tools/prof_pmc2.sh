#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 tools/prof_ekf.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "ekf" in k or "chain" in k:
            agg[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(k, row["Counter_Name"])][1] += 1
for (k, c), (s, n) in sorted(agg.items()):
    print(f"{k[:30]:32s} {c:26s} per_dispatch={s / n:16.1f}")
PY
