#!/usr/bin/env python3
"""Stage-time probe (GPU box): per-stage HIP-event times of the ingest pipeline for a few
batch sizes / modes.  Diagnostic only; bench.py is the judged measurement."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "distributed-multi-agent-slam-swarm-robotics-system_amd"


def main():
    import torch
    pkg = importlib.import_module(PKG)
    replay = importlib.import_module(PKG + ".replay")
    session, _ = replay.telemetry_csv_to_packets()
    dev = torch.device("cuda", 0)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    cases = []
    cases.append(dict(B=1 << 18, mode=2, counts=True, ekf=False, bots=2))
    cases.append(dict(B=1 << 20, mode=2, counts=True, ekf=False, bots=2))
    cases.append(dict(B=1 << 18, mode=2, counts=True, ekf=True, bots=2))
    cases.append(dict(B=1 << 18, mode=2, counts=True, ekf=False, bots=64))
    cases.append(dict(B=1 << 20, mode=2, counts=True, ekf=True, bots=64, bpg=2))
    for cs in cases:
        B = cs["B"]
        if cs["bots"] == 2:
            stream = replay.cycle_stream(session, B); max_agent = 2
        elif cs["bots"] == 0:
            stream = replay.adversarial_stream(B); max_agent = 2
        else:
            stream = replay.multi_bot_stream(session, cs["bots"], B); max_agent = cs["bots"]
        d = torch.from_numpy(stream).to(dev)
        dt = torch.arange(B, dtype=torch.float64, device=dev) * 0.25
        m = pkg.QuasarMapper(4096, 0.05, -102.4, -102.4, max_agent=max_agent, enable_counts=cs["counts"],
                             enable_ekf=cs["ekf"], raycast_mode=cs["mode"], bots_per_graph=cs.get("bpg", 0))
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for k in range(2):
            m.reset(); m.ingest_device(d.data_ptr(), B, 42, 0, dt.data_ptr(), seq0=0)
        torch.cuda.synchronize()
        m.stage_times(reset=True); m.timing_enable(True)
        t0 = time.perf_counter()
        steps = 3
        for k in range(steps):
            m.reset(); m.ingest_device(d.data_ptr(), B, 42, 0, dt.data_ptr(), seq0=0)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        st = m.stage_times(reset=True)
        cnt = m.counters()
        print(json.dumps({**cs, "ms_per_step": el * 1e3, "Mpkt_s": B / el / 1e6,
                          "stages_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1]},
                          "cnt": cnt}), flush=True)
        m.close()
        del d, dt


if __name__ == "__main__":
    main()
