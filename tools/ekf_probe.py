#!/usr/bin/env python3
"""EKF stage time per step (2 bots and 64 bots)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "distributed-multi-agent-slam-swarm-robotics-system_amd"
import torch
pkg = importlib.import_module(PKG)
replay = importlib.import_module(PKG + ".replay")
session, _ = replay.telemetry_csv_to_packets()
for bots, B, bpg in ((2, 1 << 18, 0), (64, 1 << 20, 2)):
    stream = replay.cycle_stream(session, B) if bots == 2 else replay.multi_bot_stream(session, bots, B)
    d = torch.from_numpy(stream).cuda()
    t = torch.arange(B, dtype=torch.float64, device="cuda") * 0.25
    m = pkg.QuasarMapper(4096, 0.05, -102.4, -102.4, max_agent=bots, bots_per_graph=bpg, enable_ekf=True)
    for k in range(2):
        m.reset(); m.ingest_device(d.data_ptr(), B, 42, 0, t.data_ptr(), seq0=0)
    m.sync(); m.stage_times(reset=True); m.timing_enable(True)
    for k in range(3):
        m.reset(); m.ingest_device(d.data_ptr(), B, 42, 0, t.data_ptr(), seq0=0)
    st = m.stage_times(reset=True)
    ms = st["ekf"][0] / st["ekf"][1]
    print(json.dumps({"bots": bots, "B": B, "ekf_ms": round(ms, 3), "us_per_step_per_bot": round(ms * 1e3 / (B / bots), 4),
                      "slam_ms": round(st["slam"][0] / st["slam"][1], 3)}), flush=True)
    m.close()
