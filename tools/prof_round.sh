#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (run on the GPU box from the repo root):
#   tools/prof_round.sh <tag> <bench.py args...>      (QS_PROF_ROUND=r03 by default)
#   -> gpurun_out/${QS_PROF_ROUND:-r03}/<tag>/{bench.json, kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv}
# kernel trace + stats in one run; FETCH_SIZE and WRITE_SIZE each in their own --pmc run (no tracing beside counters).
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${QS_PROF_ROUND:-r03}/$TAG
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-micro $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > /dev/null 2> $OUT/write.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for name, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    with open(f"{out}/pmc_{cname}.csv", "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "counter", "dispatches", "sum_KiB", "avg_KiB_per_dispatch"])
        for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            w.writerow([k, c, n, s, s / n])
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
head -12 $OUT/kernel_stats.csv | cut -c1-150
