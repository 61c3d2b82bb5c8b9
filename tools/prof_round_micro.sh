#!/bin/bash
# rocprofv3 evidence for the streaming kernels (K2 view, K3 fuse) and the MFMA correspondence search:
#   gpurun_out/${QS_PROF_ROUND:-r03}/c1_micro/{kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv}   (bench.py with its micro-benches)
#   gpurun_out/${QS_PROF_ROUND:-r03}/icp_nn/{kernel_stats.csv, sq_counters.csv}                              (tools/bench_icp_nn.py)
#   gpurun_out/${QS_PROF_ROUND:-r03}/c1_4096_sq/sq_counters.csv                                               (SQ counters of the chain kernel)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
summ() {  # dir, counter-run subdir, out csv
python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, collections, sys
out, sub, dst = sys.argv[1:4]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
with open(f"{out}/{dst}", "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["kernel", "counter", "dispatches", "sum", "avg_per_dispatch" if "sq" in dst else "avg_KiB_per_dispatch"])
    for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        w.writerow([k, c, n, s, s / n])
PY
}
O=gpurun_out/${QS_PROF_ROUND:-r03}/c1_micro; rm -rf $O; mkdir -p $O
A="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $A > $O/bench_under_trace.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 bench.py $A > /dev/null 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py $A > /dev/null 2> $O/w.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
summ $O pf pmc_FETCH_SIZE.csv; summ $O pw pmc_WRITE_SIZE.csv; rm -rf $O/trace $O/pf $O/pw
O=gpurun_out/${QS_PROF_ROUND:-r03}/icp_nn; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_icp_nn.py > $O/bench.json 2> $O/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -- python3 tools/bench_icp_nn.py > /dev/null 2> $O/sq.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
summ $O sq sq_counters.csv; rm -rf $O/trace $O/sq
O=gpurun_out/${QS_PROF_ROUND:-r03}/c1_4096_sq; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-micro > /dev/null 2> $O/sq.err
summ $O sq sq_counters.csv; rm -rf $O/sq
echo done
