#!/usr/bin/env python3
"""Fuzz of the HIP path against the CPU restatement (python tests/fuzz_gpu.py SEED CASES, on the GPU box;
uses the test-only checker, so it lives under tests/): random grid geometries, separations,
batch splits and adversarial / session streams; compares grid, counters, closures, drift, zone."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "distributed-multi-agent-slam-swarm-robotics-system_amd"
pkg = importlib.import_module(PKG)
replay = importlib.import_module(PKG + ".replay")
from oracle import oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
session, _ = replay.telemetry_csv_to_packets()
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for it in range(n_iter):
    size = int(rng.choice([64, 100, 200, 332, 512, 776, 1024]))
    res = float(rng.choice([0.05, 0.02, 0.1, 0.013]))
    span = size * res / 2
    ox, oy = -span + float(rng.normal(0, 0.3)), -span + float(rng.normal(0, 0.3))
    sep = float(rng.choice([0.0, 0.5, 5.0]))
    n = int(rng.integers(500, 6000))
    if rng.random() < 0.5:
        stream = replay.adversarial_stream(n, seed=int(rng.integers(1 << 30)), lo=-span * 1.2, hi=span * 1.2)
    else:
        stream = replay.cycle_stream(session, n)
    mode = int(rng.choice([0, 1, 2]))
    ekf = bool(rng.random() < 0.5)
    o = orc.OracleMapper(size, res, ox, oy, sep)
    if ekf:
        o.enable_ekf(0.0107)
    o.feed_stream(stream)
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n, int(rng.integers(0, 4)))]))
    with pkg.QuasarMapper(size, res, ox, oy, separation=sep, raycast_mode=mode, enable_ekf=ekf) as m:
        for a, b in zip(cuts[:-1], cuts[1:]):
            m.ingest_array(stream[a:b])
        ok = (m.grid_i8() == o.grid).all()
        h, mi = m.counts()
        ok &= (h == o.hits).all() and (mi == o.misses).all()
        ok &= m.counters()["cells"] == o.n_cells_written or len(cuts) > 2     # counters are per call... compare when one call
        ok &= (m.closures(0)[0].shape == o.closures(0)[0].shape) and (m.closures(0)[0] == o.closures(0)[0]).all()
        for b in (1, 2):
            ok &= np.allclose(m.drift(b), o.drift(b), rtol=0, atol=1e-9)
            z, zo = m.zone(b), o.zone(b)
            ok &= (z is None) == (zo is None) and (z is None or np.allclose(z, zo, rtol=0, atol=1e-9))
            if ekf:
                x, P = m.ekf_state(b); xo, Po = o.ekf_state(b)
                ok &= np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max()) and np.abs(P - Po).max() <= 1e-9 * max(1.0, np.abs(Po).max())
    if not ok:
        bad += 1
        print("MISMATCH", dict(it=it, size=size, res=res, ox=ox, oy=oy, sep=sep, n=n, mode=mode, ekf=ekf, cuts=cuts))
print("fuzz done:", n_iter, "cases,", bad, "mismatches")

# ---- loop-closure chain: many bots, graph partitions, closure constants, batch cuts -------------------
bad2 = 0
for it in range(n_iter):
    nb = int(rng.choice([1, 2, 3, 5, 13, 14, 20, 40]))
    bpg = int(rng.choice([0, 1, 2, 7, 13, 14]))
    radius = float(rng.choice([0.6, 0.3, 1.0, 0.45]))
    mb = int(rng.choice([30, 0, 1, 5, 31, 64]))
    corr = float(rng.choice([0.5, 1.0, 0.1]))
    n = int(rng.integers(2000, 24000))
    stream = replay.multi_bot_stream(session, nb, n, pitch=float(rng.choice([8.0, 1.0, 0.2])), origin=(-8.0, -8.0), tiles_per_row=5)
    o = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=nb, bots_per_graph=bpg)
    o.set_closure_params(radius, mb, corr)
    o.feed_stream(stream)
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n, int(rng.integers(0, 4)))]))
    form = str(rng.choice(["auto", "free", "free_posting", "window"]))
    with pkg.QuasarMapper(512, 0.05, -12.8, -12.8, max_agent=nb, bots_per_graph=bpg, closure_radius=radius,
                          min_poses_between=mb, closure_correction=corr) as m:
        m.set_chain_form(form)
        for a, b in zip(cuts[:-1], cuts[1:]):
            m.ingest_array(stream[a:b])
        ok = (m.grid_i8() == o.grid).all()
        ng = 1 if bpg == 0 else -(-nb // bpg)
        for g in range(ng):
            idx, cc = m.closures(g); oi, oc = o.closures(g)
            ok &= idx.shape == oi.shape and (idx == oi).all() and (len(oi) == 0 or np.abs(cc - oc).max() < 1e-9)
            xy, ti = m.landmarks(g); oxy, oti = o.landmarks(g)
            ok &= ti.shape == oti.shape and (ti == oti).all() and (len(oti) == 0 or np.abs(xy - oxy).max() < 1e-9)
        for b in range(1, nb + 1):
            ok &= np.allclose(m.drift(b), o.drift(b), rtol=0, atol=1e-9)
    if not ok:
        bad2 += 1
        print("MISMATCH(slam)", dict(it=it, nb=nb, bpg=bpg, radius=radius, mb=mb, corr=corr, n=n, cuts=cuts, form=form))
print("slam fuzz done:", n_iter, "cases,", bad2, "mismatches")
sys.exit(1 if bad or bad2 else 0)
