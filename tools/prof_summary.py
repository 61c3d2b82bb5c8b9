"""Per-kernel table from a profiles/r02/<tag>/ directory: average duration (kernel_stats.csv) next to the HBM bytes of
the PMC passes (FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md, + WRITE_SIZE).  usage: prof_summary.py <dir>..."""
import csv
import sys

for d in sys.argv[1:]:
    print("==", d)
    ks = {}
    for row in csv.DictReader(open(f"{d}/kernel_stats.csv")):
        ks[row["Name"].split("(")[0]] = (int(row["Calls"]), float(row["AverageNs"]), float(row["Percentage"]))
    f = {r["kernel"]: float(r["avg_KiB_per_dispatch"]) for r in csv.DictReader(open(f"{d}/pmc_FETCH_SIZE.csv", newline=""))}
    w = {r["kernel"]: float(r["avg_KiB_per_dispatch"]) for r in csv.DictReader(open(f"{d}/pmc_WRITE_SIZE.csv", newline=""))}
    tot = tt = 0.0
    for k, (c, avg, pct) in sorted(ks.items(), key=lambda kv: -kv[1][2])[:14]:
        fb, wb = f.get(k, 0) * 2048, w.get(k, 0) * 1024
        print(f"  {k[:44]:44s} calls={c:3d} avg={avg / 1e3:9.1f}us {pct:5.1f}%  fetchx2={fb / 1e6:8.1f}MB write={wb / 1e6:8.1f}MB -> {(fb + wb) / avg:7.1f} GB/s")
    for k in ks:
        if any(x in k for x in ("qs_rays", "qs_table_scan", "qs_scatter", "qs_raster")):
            tot += f.get(k, 0) * 2048 + w.get(k, 0) * 1024
            tt += ks[k][1]
    if tt:
        print(f"  raycast stage (4 kernels): {tot / 1e6:.1f} MB in {tt / 1e3:.1f} us = {tot / tt:.1f} GB/s")
