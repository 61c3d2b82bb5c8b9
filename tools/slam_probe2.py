#!/usr/bin/env python3
"""SLAM chain, owner wave 1 of graph 0, cycles per window by segment (QS_CHAIN_PROF2 build: tools/build_variant.sh chprof2
"-DQS_CHAIN_PROF2" slam.hip; QUASAR_SLAM_LIB=ab_libs/chprof2.so).  With the plain library: stage times only."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "distributed-multi-agent-slam-swarm-robotics-system_amd"
import torch
pkg = importlib.import_module(PKG)
replay = importlib.import_module(PKG + ".replay")
B = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
session, _ = replay.telemetry_csv_to_packets()
wl = sys.argv[3] if len(sys.argv) > 3 else "c1"           # c1: 2 bots; one64: 64 bots in ONE pose graph; g32: 64 bots in 32 graphs
if wl == "adv":
    d = torch.from_numpy(replay.adversarial_stream(B)).cuda()
    m = pkg.QuasarMapper(4096, 0.05, -102.4, -102.4, max_agent=2, exact_trig=False)
elif wl == "c1":
    d = torch.from_numpy(replay.cycle_stream(session, B)).cuda()
    m = pkg.QuasarMapper(4096, 0.05, -102.4, -102.4, max_agent=2, exact_trig=False)
else:
    d = torch.from_numpy(replay.multi_bot_stream(None, 64, B)).cuda()
    m = pkg.QuasarMapper(4096, 0.05, -102.4, -102.4, max_agent=64, bots_per_graph=0 if wl == "one64" else 2, exact_trig=False)
m.reset(); m.ingest_device(d.data_ptr(), B, 42, 0, 0, seq0=0); m.sync()
out = []
for r in range(reps):
    m.reset(); m.stage_times(reset=True); m.timing_enable(True); m.ingest_device(d.data_ptr(), B, 42, 0, 0, seq0=0); m.sync()
    c = m.counters(); st = m.stage_times()
    w = c["slam_windows"]
    out.append(round(st["slam_chain"][0], 3))
prof = "chprof2" in os.environ.get("QUASAR_SLAM_LIB", "")
res = {"workload": wl, "lib": os.path.basename(os.environ.get("QUASAR_SLAM_LIB", "default")), "chain_ms": out, "windows": w, "cyc_per_window": round(c["slam_cycles"] / w, 1),
       "closures": c["closures"]}
if prof:
    res.update({"head": c["slam_cyc_prepare"] / w, "q_setup": c["slam_cyc_query"] / w, "q_scan": c["slam_cyc_commit"] / w,
                "q_post": c["slam_misc_iters"] / w, "publish": c["ekf_wrap_clamp"] / w, "barrier": c["slam_rounds"] / w})
if "chprof3" in os.environ.get("QUASAR_SLAM_LIB", ""):
    res.update({"busy_wave0": c["slam_cyc_prepare"] / w, "busy_fetch": c["slam_cyc_query"] / w, "busy_insert": c["slam_cyc_commit"] / w,
                "busy_owner1": c["slam_misc_iters"] / w, "busy_other_owners_sum": c["ekf_wrap_clamp"] / w})
print(json.dumps(res))
