#!/bin/bash
# idle time between consecutive kernels of the main stream in the steady-state steps of the 64-bot bench (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-micro > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY' | tee gpurun_out/gaps/c3_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps: from one qs_decode_kernel to the next; main stream = the queue the decode kernel runs on
dec = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("qs_decode_kernel")]
q_main = rows[dec[0]]["Queue_Id"]
res = []
for a, b in zip(dec[3:-1], dec[4:]):
    ks = [r for r in rows[a:b] if r["Queue_Id"] == q_main]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    span = int(ks[-1]["End_Timestamp"]) - int(ks[0]["Start_Timestamp"])
    gaps = [(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) for x, y in zip(ks[:-1], ks[1:])]
    res.append((len(ks), span / 1e3, busy / 1e3, sum(g for g in gaps if g > 0) / 1e3, max(gaps) / 1e3))
for r in res:
    print("main-stream kernels %d  span %.1f us  busy %.1f us  idle between kernels %.1f us  (largest gap %.1f us)" % r)
a, b = dec[4], dec[5]
ks = [r for r in rows[a:b] if r["Queue_Id"] == q_main]
print("one step, kernel by kernel (us: duration, gap before it):")
prev_end = None
for r in ks:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %-44s %8.1f %7.1f" % (r["Kernel_Name"].split("(")[0][:44], (en - st) / 1e3, 0.0 if prev_end is None else (st - prev_end) / 1e3))
    prev_end = en
PY
rm -rf $OUT/trace
