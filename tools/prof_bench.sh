#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command  -> per-kernel average durations
#   2. PMC passes (own runs, no tracing): FETCH_SIZE, WRITE_SIZE per kernel dispatch
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<'PY'
import csv, glob, collections
for name in ("pmc_fetch", "pmc_write"):
    files = glob.glob(f"gpurun_out/prof_bench/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    with open(f"gpurun_out/prof_bench/{name}_summary.csv", "w") as out:
        out.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
        for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            out.write(f"{k},{c},{n},{s},{s / n}\n")
    print(open(f"gpurun_out/prof_bench/{name}_summary.csv").read()[:1500])
PY
cat $OUT/kernel_stats.csv | cut -c1-160
