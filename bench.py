#!/usr/bin/env python3
"""bench.py -- QuasarPackets/s into a 4096x4096 occupancy grid on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input, from a fresh session:
    qs_reset -> decode (K0) -> landmark loop closure / drift (K4) -> 4-ray raycast into the
    grid (K1) [-> per-bot EKF (K5)] [-> RCCL fuse of the per-GPU grids when N > 1].
Inputs are resident in HBM before the timed region.

Workloads (--workload; "auto" picks by N):
  c1   BASELINE.json configs[1] (N = 1 default): the 2-bot generate_fake_dual_session.py session
       (the package's data/session_telemetry.csv, produced by the reference generator) cycled to B packets,
       4096^2 grid, res 0.05, origin -102.4, one pose graph (the reference's mapper).
  c3   configs[3] (N > 1 default; N = 1: configs[2]'s shape): 64 bots per GPU, bot i of rank r in room tile
       r*64 + i of the 25 x 25 lattice (8 m pitch), each replaying its own run of the reference generator (seed
       42 + lane, data/multibot_sessions.npz), one pose graph per 2 bots (the reference's deployment unit),
       stamps carry the GLOBAL arrival index (seq = i*N + rank), one grid fuse per step (MAX on stamps, SUM on a
       snapshot of the counters).  Weak scaling: every GPU ingests B packets per step.
  adv  SURVEY.md 8(d) D2's adversarial stream: uniform-random poses / yaw / distances (seed 1234 + rank), 2 bots --
       worst-case locality: the case where the raycast really is bound by memory, not by LDS.
--grid 8192 runs any of them on the 8192^2 grid (origin -204.8; 256 MiB of stamps + 512 MiB of counters).

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (child processes, started
before anything touches a GPU); under torchrun (RANK / WORLD_SIZE set) it is one rank.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline`, `cpu_baseline`, `parity_checked`.
Exit code != 0 if the GPU results of the last step differ from the oracle's on the same stream.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "distributed-multi-agent-slam-swarm-robotics-system_amd"

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0       # achievable float4-copy rate, same guide
N_CU = 256
PROFILE_ROUND = "r03"       # profiles/<round>/: the committed rocprofv3 summaries the replayed counter figures come from


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1 << 20, help="packets per step per GPU")
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c3", "adv"])
    ap.add_argument("--stream", default=None, choices=[None, "adversarial"], help="alias: --stream adversarial = --workload adv")
    ap.add_argument("--raycast-mode", type=int, default=0)
    ap.add_argument("--no-counts", action="store_true", help="tri-state stamps only (no hit/miss counters)")
    ap.add_argument("--ekf", type=int, default=1, help="run the per-bot EKF stage (1) or not (0)")
    ap.add_argument("--bots", type=int, default=0, help="bots per GPU (0: the workload's: c1/adv 2, c3 64)")
    ap.add_argument("--bots-per-graph", type=int, default=-1,
                    help="bots sharing one pose graph: -1 = the reference's deployment unit (2 bots per mapper process, i.e. "
                         "per pose graph), 0 = all bots of the GPU in one graph")
    ap.add_argument("--slam-mode", default="per_shard", choices=["per_shard", "replicated"],
                    help="N > 1: pose graphs per shard, or ONE pose graph over all bots, its chain replicated on every rank")
    ap.add_argument("--fuse", default="sparse", choices=["sparse", "allreduce", "direct"],
                    help="N > 1 grid fuse: sparse = only the 4 x 16-cell blocks a rank wrote since the last fuse travel (bitmap "
                         "all-gather + point-to-point exchange of the packed blocks + fold); allreduce = RCCL's dense all-reduce of "
                         "the whole map (the reference point); direct = dense point-to-point reduce-scatter + K3 fold + all-gather")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true", help="skip the K2 view / K3 fuse timings (N = 1 only anyway)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="packets of the same stream timed on the CPU (0: one whole step; the reference's closure\n"
                         "search is O(landmarks) per landmark packet, so the rate depends on the length)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank uses cuda:0 (with --backend gloo): rehearses the N > 1 code path on a one-GPU box; "
                         "the numbers mean nothing")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1,
                    help="with --spawn-selftest: this rank exits with code 3 before the rendezvous (the launcher must stop its siblings)")
    ap.add_argument("--spawn-selftest", action="store_true",
                    help="no GPU: the ranks only rendezvous (backend as given), all-reduce one number and rank 0 prints "
                         "{n_gpus: world}; used by the CPU test of the launcher")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes of a parent that never touches a GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and wait; returns the worst exit code.  The parent has not
    initialised any GPU, and it does not replace itself: it stays the parent and exits with the children's code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # poll: a rank that dies (bad arguments, a HIP error, a parity mismatch) while its siblings sit in a collective must not
    # leave the parent waiting for them -- first non-zero exit ends the others (terminate, then kill), and so does the limit
    deadline = time.time() + float(os.environ.get("QS_BENCH_LIMIT_S", "3000"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0] if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill(); p.wait()
            print(f"bench.py: rank exit codes {codes}: stopped the remaining ranks (exit {rc})", file=sys.stderr)
            return rc
        if all(c == 0 for c in codes):
            return 0
        time.sleep(0.2)


# ------------------------------------------------------------------------------------------------------
# workloads
def resolve(args, world):
    wl = "adv" if args.stream == "adversarial" else args.workload
    if wl == "auto":
        wl = "c1" if world == 1 else "c3"
    bots = args.bots or (64 if wl == "c3" else 2)
    bpg = args.bots_per_graph
    if bpg < 0:
        bpg = 2 if bots > 2 else 0
    if args.slam_mode == "replicated":
        bpg = 0                                   # one pose graph over every bot of every rank
    return wl, bots, bpg


def make_stream(pkg, replay, wl, bots, B, grid, rank, world, replicated):
    """This rank's B packets (uint8 [B, 42]).  replicated: agent ids are global (rank*bots + 1 ..)."""
    half = grid * 0.05 / 2
    if wl == "adv":
        return replay.adversarial_stream(B, seed=1234 + rank, lo=-(half - 2.4), hi=half - 2.4, max_agent=bots)
    session, _ = replay.telemetry_csv_to_packets()
    if wl == "c1" and world == 1:
        return replay.cycle_stream(session, B)
    if wl == "c1":                                # N ranks x configs[1]: this rank's two bots in their own room tile
        stream = replay.cycle_stream(session, B)
        rec = stream.view(pkg.protocol.PACKET_DTYPE).reshape(-1)
        rec["x"] = (rec["x"].astype(np.float64) + 8.0 * (rank % 8) - 28.0).astype(np.float32)
        rec["y"] = (rec["y"].astype(np.float64) + 8.0 * (rank // 8)).astype(np.float32)
        if replicated:
            rec["agent"] = rec["agent"] + rank * bots
        return stream
    # every bot replays its own generator run (seed 42 + lane; replay.multibot_lanes): SURVEY.md 8(d) D2
    return replay.multi_bot_stream(None, bots, B, tile0=rank * bots, agent0=(rank * bots + 1) if replicated else 1)


def oracle_for(orc, grid, bots, bpg, ekf, wl):
    half = grid * 0.05 / 2
    m = orc.OracleMapper(grid, 0.05, -half, -half, 0.0, max_agent=bots, bots_per_graph=bpg)
    if wl == "c3":
        pass                                       # (offset of bot 2 is 0.0 already: separation = 0)
    if ekf:
        m.enable_ekf(0.0107)
    return m


def _cpu_worker(job):
    """One oracle process over one stream (the all-cores baseline: independent mapper processes)."""
    stream, grid, bots, bpg, ekf, wl = job
    from oracle import oracle as orc
    m = oracle_for(orc, grid, bots, bpg, ekf, wl)
    t0 = time.perf_counter()
    m.feed_stream(stream, None, np.arange(len(stream)) * 0.25)
    return time.perf_counter() - t0


def cpu_baselines(stream, grid, sample, ekf, bots, bpg, wl):
    """The oracle (C restatement of the reference path) on the same stream: 1 thread; P independent mapper processes
    (how a CPU deployment of the reference scales: one mapper process per bot pair); and the pure-Python restatement
    (oracle/pymapper.py) on a bounded prefix.  Returns (json dict, the 1-thread oracle's final state)."""
    from oracle import oracle as orc
    n = min(sample, len(stream)) if sample > 0 else len(stream)
    times = np.arange(len(stream)) * 0.25
    m = oracle_for(orc, grid, bots, bpg, ekf, wl)
    t0 = time.perf_counter()
    m.feed_stream(stream[:n], None, times[:n])
    dt = time.perf_counter() - t0
    ncpu = os.cpu_count() or 1
    try:
        ncpu_avail = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu_avail = ncpu
    out = {"value": n / dt, "unit": "packets/s", "cores": 1, "kind": "port",
           "sample": f"{'one whole step:' if n == len(stream) else 'first'} {n} packets of the same stream, oracle/oracle.c "
                     f"(gcc -O2), {dt:.2f} s (the reference scans its whole landmark list per landmark packet, :292-326, so "
                     f"the rate falls with the stream length); host has {ncpu} logical CPUs ({ncpu_avail} usable)"}
    # all cores: P mapper processes, each over its own copy of the same sample (a pose graph is a sequential
    # recurrence: one mapper cannot use a second core; a CPU deployment scales by running more mappers)
    P = max(1, min(ncpu_avail, 16))
    try:
        import multiprocessing as mp
        ctx = mp.get_context("fork")               # no GPU has been initialised in this process yet
        t0 = time.perf_counter()
        with ctx.Pool(P) as pool:
            pool.map(_cpu_worker, [(stream[:n], grid, bots, bpg, ekf, wl)] * P)
        dtp = time.perf_counter() - t0
        out["all_cores"] = {"value": P * n / dtp, "unit": "packets/s", "cores": P,
                            "sample": f"{P} independent oracle processes (capped at the box's 16-core share), each {n} packets, {dtp:.2f} s"}
    except Exception as e:                         # a baseline must never take the bench down
        out["all_cores"] = {"value": None, "error": repr(e)}
    # the same port with a spatial index over the landmarks in place of the reference's list scan (identical closures:
    # tests/test_oracle_golden.py): what the GPU buys over a CPU that is given the GPU path's data structure
    try:
        mi = oracle_for(orc, grid, bots, bpg, ekf, wl)
        mi.use_index(True)
        t0 = time.perf_counter()
        mi.feed_stream(stream[:n], None, times[:n])
        dti = time.perf_counter() - t0
        same = all((mi.closures(g)[0] == m.closures(g)[0]).all() for g in range(m.n_graphs)) and bool((mi.grid == m.grid).all())
        out["indexed"] = {"value": n / dti, "unit": "packets/s", "cores": 1, "kind": "port",
                          "sample": f"the same {n} packets, oracle/oracle.c with its optional bucket index over the landmarks instead of the "
                                    f"reference's O(landmarks) list scan, {dti:.2f} s; NOT the reference's algorithm, the same results "
                                    f"(closures and grid identical to the list scan: {same})"}
        del mi
    except Exception as e:
        out["indexed"] = {"value": None, "error": repr(e)}
    # the single-process Python statement of the reference's loop (2-bot wire only: agent in {1, 2})
    if bots == 2 and bpg in (0, 2):
        from oracle.pymapper import PyMapper
        half = grid * 0.05 / 2
        npy = min(n, 1 << 17)
        pm = PyMapper(grid, 0.05, -half, -half, 0.0)
        t0 = time.perf_counter()
        pm.feed_stream(stream[:npy])
        dtpy = time.perf_counter() - t0
        out["python"] = {"value": npy / dtpy, "unit": "packets/s", "cores": 1, "kind": "port",
                         "sample": f"first {npy} packets, oracle/pymapper.py (CPython {sys.version_info.major}.{sys.version_info.minor}, the "
                                   f"reference's loop structure: per-cell numpy stores, list scans), {dtpy:.2f} s"}
    return out, (m if n == len(stream) else None)


# ------------------------------------------------------------------------------------------------------
def check_parity(m, o, n_graphs, world, cnt, torch, dist_mod, sm, dev):
    """GPU state of the last step against the oracle fed the same stream (this rank's, in per-shard mode; every rank's
    interleaved, in replicated mode).  Integer results bit-exact, floats to 1e-5 (north_star)."""
    bad = []
    for g in range(n_graphs):
        idx, corr = m.closures(g)
        oi, oc = o.closures(g)
        if idx.shape != oi.shape or not (idx == oi).all():
            bad.append(f"graph {g}: closure index pairs differ ({len(idx)} vs {len(oi)})")
        elif len(oi) and np.abs(corr - oc).max() > 1e-5:
            bad.append(f"graph {g}: closure corrections differ by {np.abs(corr - oc).max():.3g}")
        if m.slam_sizes(g)[1] != len(o.landmarks(g)[0]):
            bad.append(f"graph {g}: landmark count differs")
        if len(bad) > 4:
            break
    for b in range(1, m.max_agent + 1):
        if np.abs(m.drift(b) - o.drift(b)).max() > 1e-5:
            bad.append(f"bot {b}: drift differs")
            break
    if world == 1:
        if cnt["cells"] != o.n_cells_written or cnt["rays"] != o.n_rays:
            bad.append(f"cells/rays counters differ: {cnt['cells']}/{cnt['rays']} vs {o.n_cells_written}/{o.n_rays}")
        grid = m.grid_i8()
        og = o.grid
        if hashlib.sha256(grid.tobytes()).digest() != hashlib.sha256(og.tobytes()).digest():
            bad.append(f"grid differs in {(grid != og).sum()} cells")
        if m.cfg.enable_counts:
            h, mi = m.counts()
            if not ((h == o.hits).all() and (mi == o.misses).all()):
                bad.append("hit/miss counters differ")
        return bad
    # N > 1: the fused device grid against the same fuse of the ranks' oracle grids (collective on the oracle's arrays)
    import torch.distributed as dist
    ost = torch.from_numpy(o.stamps.astype(np.int64).astype(np.int32)).to(dev)
    dist.all_reduce(ost, op=dist.ReduceOp.MAX)
    stamps, _ = dist_mod.grid_tensors(m, dev)
    if not torch.equal(stamps, ost):
        bad.append(f"fused stamps differ from the fused oracle stamps in {int((stamps != ost).sum())} cells")
    if m.cfg.enable_counts:
        oc = torch.from_numpy(np.stack([o.misses, o.hits], axis=-1).copy()).to(dev)
        dist.all_reduce(oc, op=dist.ReduceOp.SUM)
        fz = sm.last_fused                    # the tensor aliasing the context's fused-counter snapshot
        if fz is None or not torch.equal(fz, oc):
            bad.append("fused counters differ from the sum of the oracles' counters")
    return bad


def micro_benches(pkg, torch, dev, m, grid):
    """K2 view and K3 fuse on their own (HIP events on the context's stream = torch's current stream)."""
    cells = grid * grid
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    view = torch.empty(cells, dtype=torch.int8, device=dev)
    s = timed(lambda: m.grid_i8_device(view.data_ptr()), 20)
    out["k2_view_i8"] = {"bytes": 5 * cells, "ms": s * 1e3, "gbs": 5 * cells / s / 1e9, "frac_of_peak": 5 * cells / s / 1e9 / HBM_PEAK_GBS,
                         "frac_of_copy": 5 * cells / s / 1e9 / HBM_COPY_GBS, "rule": "size^2 x (4 B stamp read + 1 B int8 written)",
                         "residency": ("Infinity-Cache resident: the stamp grid (%d MiB) is re-read from the 256 MB MALL on every repetition, so this "
                                       "is a cache rate, not an HBM rate" % (cells * 4 >> 20)) if cells * 4 <= (128 << 20) else
                                      "past the 256 MB Infinity Cache with its output: an HBM rate"}
    del view
    for G in (8, 64):
        st = [torch.randint(0, 1 << 30, (cells,), dtype=torch.int32, device=dev) for _ in range(G)]
        s = timed(lambda: m.fuse_buffers([t.data_ptr() for t in st], None), 5)
        nbytes = (G + 2) * cells * 4          # G sources + destination read + destination written
        out[f"k3_fuse_stamps_G{G}"] = {"bytes": nbytes, "ms": s * 1e3, "gbs": nbytes / s / 1e9,
                                       "frac_of_peak": nbytes / s / 1e9 / HBM_PEAK_GBS, "frac_of_copy": nbytes / s / 1e9 / HBM_COPY_GBS,
                                       "rule": "(G + 1) x size^2 x 4 B read + size^2 x 4 B written"}
        if m.cfg.enable_counts and G == 8:
            ct = [torch.randint(0, 1 << 20, (cells, 2), dtype=torch.int32, device=dev) for _ in range(G)]
            s = timed(lambda: m.fuse_buffers([t.data_ptr() for t in st], [t.data_ptr() for t in ct]), 5)
            nb2 = nbytes + (G + 2) * cells * 8
            out[f"k3_fuse_stamps+counts_G{G}"] = {"bytes": nb2, "ms": s * 1e3, "gbs": nb2 / s / 1e9, "frac_of_peak": nb2 / s / 1e9 / HBM_PEAK_GBS,
                                                  "frac_of_copy": nb2 / s / 1e9 / HBM_COPY_GBS}
            del ct
        del st
    return out


def secondary_64_bots(pkg, replay, torch, dev, side, args, G, counts, bpg=2, steps=5, warm=2):
    """configs[2] in the same run (N = 1, default workload only): 64 bots on this GPU, every bot its own generator run, `steps`
    timed steps, the last one compared with the oracle -- so that the 64-bot figures are the driver's, not only the builder's.
    bpg = 2: 32 pose graphs (one per bot pair, the reference's deployment unit: one mapper process per two bots);
    bpg = 0: ONE PoseGraphSLAM over all 64 bots, as the reference's class is written (node indices count every bot's poses,
    dual_bot_mapper.py:267-275; cross-bot matches, :294-309) -- one chain workgroup, its 13 owner waves share the 64 agents."""
    from oracle import oracle as orc
    B, half = args.batch, G * 0.05 / 2
    stream = replay.multi_bot_stream(None, 64, B)                 # bot i: the reference generator run with seed 42 + i
    d_stream = torch.from_numpy(stream).to(dev)
    d_time = torch.arange(B, dtype=torch.float64, device=dev) * 0.25
    m = pkg.QuasarMapper(G, 0.05, -half, -half, max_agent=64, bots_per_graph=bpg, enable_counts=counts, enable_ekf=bool(args.ekf),
                         device=dev.index, raycast_mode=args.raycast_mode)
    m.set_stream(side.cuda_stream)

    def step():
        m.reset()
        m.ingest_device(d_stream.data_ptr(), B, 42, 0, d_time.data_ptr(), seq0=0)

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    m.stage_times(reset=True); m.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.timing_enable(False)
    st = {k: (v[0] / max(v[1], 1)) for k, v in m.stage_times(reset=True).items() if v[1]}
    cnt = m.counters()
    o = orc.OracleMapper(G, 0.05, -half, -half, 0.0, max_agent=64, bots_per_graph=bpg)
    if args.ekf:
        o.enable_ekf(0.0107)
    t0 = time.perf_counter()
    o.feed_stream(stream, None, np.arange(B) * 0.25)
    cpu_s = time.perf_counter() - t0
    bad = check_parity(m, o, m.n_graphs, 1, cnt, torch, None, None, dev)
    for line in bad:
        print(f"PARITY MISMATCH (64 bots, {m.n_graphs} pose graphs): {line}", file=sys.stderr)
    n_graphs = m.n_graphs
    free_running = m.chain_form() in ("free", "free_posting")
    m.close()
    win = max(cnt["slam_windows"], 1)
    return {"workload": f"configs[2]: 64 bots (own generator runs, seeds 42..105) in {n_graphs} pose graph{'s' if n_graphs > 1 else ''}, own room "
                        f"tiles, {G}x{G} grid, {B} packets/step, same stages",
            "pose_graphs_per_gpu": n_graphs, "bots_per_graph": bpg or 64,
            "value": B * steps / el, "unit": "packets/s", "steps": steps, "warmup": warm, "ms_per_step": el / steps * 1e3,
            "stages_ms_per_step": st, "closures_per_step": cnt["closures"], "chain_form": "free-running" if free_running else "windowed",
            ("chain_committer_batches_per_step" if free_running else "slam_windows_per_step"): cnt["slam_windows"],
            ("chain_cycles_per_decision" if free_running else "chain_cycles_per_window"):
                (cnt["slam_cycles"] / n_graphs) / (max(cnt["closures"], 1) / (n_graphs * (bpg or 64))) if free_running else cnt["slam_cycles"] / win,
            "parity_checked": not bad,
            "cpu_baseline": {"value": B / cpu_s, "unit": "packets/s", "cores": 1, "kind": "port",
                             "sample": f"one whole step, oracle/oracle.c, {cpu_s:.1f} s"}}


def measured_copy_gbs(torch, dev):
    """Device-to-device copy bandwidth of this GPU in this run (bytes read + bytes written per second):
    the practical HBM ceiling next to the 8 TB/s spec peak."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.zero_(); b.copy_(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    ev0.record()
    for _ in range(reps):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    del a, b
    return 2.0 * n / (ms * 1e-3) / 1e9


def replayed_traffic(profile_dir, kernels):
    """HBM bytes per launch from committed rocprofv3 PMC summaries (separate --pmc passes of this command; FETCH_SIZE
    doubled per the gfx950 note of MI355X_MICROARCH.md section HBM, WRITE_SIZE as is; both in KiB).  NOT measured in
    this run: labelled as replayed wherever it is printed."""
    total, found = 0.0, False
    rnd = PROFILE_ROUND if os.path.isdir(os.path.join(ROOT, "profiles", PROFILE_ROUND, profile_dir)) else "r02"
    for fname, scale in (("pmc_FETCH_SIZE.csv", 2.0), ("pmc_WRITE_SIZE.csv", 1.0)):
        path = os.path.join(ROOT, "profiles", rnd, profile_dir, fname)
        if not os.path.exists(path):
            return None
        import csv
        for row in csv.DictReader(open(path, newline="")):
            if any(k in row["kernel"] for k in kernels):
                total += float(row["avg_KiB_per_dispatch"]) * 1024.0 * scale
                found = True
    return (total, rnd) if found else None


# ------------------------------------------------------------------------------------------------------
def selftest_rank(args):
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if rank == args.selftest_fail_rank:
        return 3
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "sum": float(t.item()), "selftest": True}))
    dist.destroy_process_group()
    return 0


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU", file=sys.stderr)
        return 2
    wl, bots, bpg = resolve(args, world)
    replicated = args.slam_mode == "replicated" and world > 1
    B, G = args.batch, args.grid
    half = G * 0.05 / 2
    counts = not args.no_counts

    pkg = importlib.import_module(PKG)
    replay = importlib.import_module(PKG + ".replay")
    distmod = importlib.import_module(PKG + ".dist")
    if replicated and bots * world > 255:
        print("bench.py: replicated pose graph needs globally unique agent ids: bots x ranks <= 255", file=sys.stderr)
        return 2
    stream = make_stream(pkg, replay, wl, bots, B, G, rank, world, replicated)
    max_agent = bots * world if replicated else bots

    # ---- CPU legs first: nothing below this block has touched a GPU yet, so worker processes may be forked ----------
    cpu, o_state = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, o_state = cpu_baselines(stream, G, args.cpu_sample, bool(args.ekf), bots, bpg, wl)

    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one explicit HIP stream for everything: the library's kernels, torch's copies and events, and the stream RCCL's
    # collectives order themselves against (torch's default stream has handle 0, which qs_set_stream reads as "own stream")
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)

    d_stream = torch.from_numpy(stream).to(dev)               # resident in HBM before timing
    d_time = torch.arange(B, dtype=torch.float64, device=dev) * 0.25
    torch.cuda.synchronize()

    m = pkg.QuasarMapper(G, 0.05, -half, -half, max_agent=max_agent, bots_per_graph=bpg,
                         enable_counts=counts, enable_ekf=bool(args.ekf), device=local_rank,
                         raycast_mode=args.raycast_mode, seq_stride=1 if replicated else world,
                         shard_bots=bots if replicated else 0, shard_rank=rank if replicated else 0)
    assert side.cuda_stream != 0
    m.set_stream(side.cuda_stream)
    sm = distmod.ShardedMapper(m, dev, rank, world, mode="replicated" if replicated else "per_shard", fuse=args.fuse)
    sm.last_fused = None

    def step(k):
        m.reset()
        sm.ingest(d_stream, d_time, seq_base=0)
        if world > 1:
            sm.last_fused = sm.fuse(counts=counts)

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    m.stage_times(reset=True)
    m.timing_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    m.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stages = m.stage_times(reset=True)
    cnt = m.counters()                       # counters of the last step (reset every step)
    st_ms = {k: (v[0] / max(v[1], 1)) for k, v in stages.items() if v[1]}

    # ---- parity of what was just timed (last step) against the oracle on the same stream -----------------------------
    parity, parity_note = None, None
    if not args.no_cpu_baseline:
        from oracle import oracle as orc
        if world == 1:
            o = o_state
            if o is None and rank == 0 and args.cpu_sample == 0:
                o = oracle_for(orc, G, bots, bpg, False, wl); o.feed_stream(stream)
        else:
            o = orc.OracleMapper(G, 0.05, -half, -half, 0.0, max_agent=max_agent, bots_per_graph=bpg)
            if replicated:
                gathered = [torch.empty_like(d_stream) for _ in range(world)]
                dist.all_gather(gathered, d_stream)
                full = torch.stack(gathered, dim=1).reshape(world * B, 42).cpu().numpy()
                o.set_owned(rank * bots + 1, (rank + 1) * bots)
                o.feed_stream(full)
            else:
                o.set_sequence(rank, world)
                o.feed_stream(stream)
        if o is not None:
            bad = check_parity(m, o, m.n_graphs, world, cnt, torch, distmod, sm, dev)
            ok = torch.tensor([0 if bad else 1], device=dev)
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            parity = bool(ok.item())
            parity_note = ("closure index pairs, landmark counts, drift per bot, "
                           + ("cells/rays counters, grid SHA-256, hit/miss counters" if world == 1 else
                              "fused stamps and fused counters of all ranks == the same fuse of the ranks' oracle grids")
                           + f" vs oracle/oracle.c over the same {len(stream) * (world if replicated else 1)} packets")
            for line in bad:
                print(f"[rank {rank}] PARITY MISMATCH: {line}", file=sys.stderr)

    # ---- roofline: the kernel with the largest share of the step ----------------------------------------------------
    per_cell = 8 + (8 if counts else 0)
    alg_bytes = 42 * cnt["datagrams"] + cnt["cells"] * per_cell     # SURVEY.md 8(d) D4, per launch (= per step)
    chain_ms, ray_ms = st_ms.get("slam_chain", 0.0), st_ms.get("raycast", 0.0)
    ray_alg_gbs = alg_bytes / (ray_ms * 1e-3) / 1e9 if ray_ms > 0 else 0.0
    prof_dir = {"c1": "c1_4096", "adv": "adv_4096", "c3": "c3_4096"}[wl] if G == 4096 else f"{wl}_{G}"
    _rt = replayed_traffic(prof_dir, ("qs_rays_kernel", "qs_table_scan_kernel", "qs_scatter_kernel", "qs_raster_kernel"))
    ray_traffic, prof_round = _rt if _rt else (None, PROFILE_ROUND)
    raycast_entry = {
        "kernel": "K1 raycast stage: qs_rays + qs_table_scan + qs_scatter + qs_raster", "bound": "hbm",
        "avg_launch_ms": ray_ms, "kernels_ms": {k: st_ms.get(k) for k in ("rc_rays", "rc_sort", "rc_raster")},
        "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_gbs": ray_alg_gbs, "frac_algorithmic": ray_alg_gbs / HBM_PEAK_GBS,
        "counter_bytes_per_launch": ray_traffic,
        "counter_gbs": (ray_traffic / (ray_ms * 1e-3) / 1e9) if (ray_traffic and ray_ms > 0) else None,
        "frac_counter": (ray_traffic / (ray_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (ray_traffic and ray_ms > 0) else None,
        "counter_source": f"replayed from profiles/{prof_round}/{prof_dir}/pmc_*.csv (rocprofv3 --pmc passes of this command), NOT measured in this run"
                          if ray_traffic else None}
    # which form of the chain kernel ran (csrc/slam.hip; qs_chain_form): free-running -- owner waves decide, a committer wave inserts
    # behind them, no per-window barrier -- or per-window (QS_CHAIN_AUTO picks it for streams whose queries mostly find nothing)
    agents_per_graph = bpg or max_agent
    free_running = m.chain_form() in ("free", "free_posting")
    if chain_ms >= ray_ms:
        dom_ms = chain_ms
        win = max(cnt["slam_windows"], 1)
        # sequential decisions of ONE agent's recurrence per launch (the agents of a graph decide side by side), and the kernel's
        # cycles per such decision; windowed form: its unit is the window (every role synchronises once per window)
        decisions = max(cnt["closures"], 1) / max(m.n_graphs * agents_per_graph, 1)
        cyc_unit = (cnt["slam_cycles"] / max(m.n_graphs, 1)) / decisions if free_running else cnt["slam_cycles"] / win
        roofline = {
            "bound": "latency",
            "kernel": ("qs_slam_chain_free_kernel" if free_running and agents_per_graph <= 13 else "qs_slam_chain_dyn_kernel" if free_running
                       else "qs_slam_chain_kernel") + " (K4 loop-closure recurrence)",
            "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": dom_ms, "share_of_step": dom_ms / (elapsed / args.steps * 1e3),
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_rule": "whole step's D4 bytes (42 B/packet + in-bounds cell writes x (8 B stamp RMW + 8 B counter RMW)) over the "
                                      "dominant kernel's duration: the step cannot finish before this kernel does",
            "nature": "latency-bound sequential recurrence, not a bandwidth kernel: one workgroup per pose graph" +
                      (" -- free-running form: one owner wave per agent runs that agent's closure decisions one after the other, "
                       "a committer wave inserts behind them; no barrier per window" if free_running else
                       " -- windowed form: every role synchronises once per window of < MIN_POSES_BETWEEN nodes"),
            "form": "free-running" if free_running else "windowed",
            "workgroups": m.n_graphs, "cus_occupied": min(m.n_graphs, N_CU), "cus_total": N_CU,
            "closures_per_launch": cnt["closures"], "sequential_decisions_per_agent": decisions if free_running else None,
            "cycles_per_decision" if free_running else "cycles_per_window": cyc_unit,
            "committer_batches_per_launch" if free_running else "windows_per_launch": cnt["slam_windows"],
            "note": "achieved / peak / frac keep the HBM form the contract asks for (the step's D4 bytes over this kernel's time against "
                    "8 TB/s); the kernel is a recurrence bound by the dependent chain between two decisions, priced in latency_floor"}
        try:
            lat = m.diag_latencies()
            # The shortest dependent chain one window's decision can be made of, each link at its latency MEASURED on this GPU by
            # one workgroup (csrc/diag.hip).  Counts: pose + nine bucket addresses = 5 fp64 + 6 integer dependent
            # VALU steps; distance test + select = 4 fp64 + 1 lane-to-scalar step; wave-wide minimum = 6 DPP steps + 1 readlane;
            # winner's fields = 2 lane-to-scalar steps; closure arithmetic = 3 fp64 steps.
            valu = 5 * lat["fma_f64"] + 6 * lat["dpp_step"] + 4 * lat["fma_f64"] + lat["readlane_step"] + 6 * lat["dpp_step"] + \
                lat["readlane_step"] + 2 * lat["readlane_step"] + 3 * lat["fma_f64"]
            floor = lat["barrier_5_waves"] + lat["lds_read"] + lat["l2_load"] + lat["lds_read"] + valu
            roofline["latency_floor"] = {
                "cycles_per_decision": floor, "achieved_cycles_per_decision": cyc_unit,
                "achieved_over_floor": cyc_unit / floor, "frac_of_floor": floor / cyc_unit,
                "floor_ms_per_launch": floor * (decisions if free_running else win / max(m.n_graphs, 1)) / (lat["clock_mhz"] * 1e3),
                "chain": "1 workgroup barrier (5 waves; the windowed form's: the free-running form has none, its hand-over is an LDS "
                         "word) + 1 LDS read (the events / the frontier) + 22 dependent VALU / cross-lane steps (pose -> nine bucket "
                         "addresses; distance test; wave-wide minimum; winner; closure) + 1 L2 round trip (node rows) + 1 LDS write "
                         "the next decision's consumer can see",
                "measured_cycles": lat,
                "source": "qs_diag_latencies (csrc/diag.hip), measured in this run; one decision per window is the minimum the "
                          "recurrence allows (dual_bot_mapper.py:292-326: a closure moves every later pose of its agent)"}
        except Exception as ex:                      # a diagnostic must never take the bench down
            roofline["latency_floor"] = {"error": repr(ex)}
    else:
        roofline = {"bound": "hbm", "kernel": raycast_entry["kernel"], "achieved": ray_alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ray_alg_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ray_ms,
                    "share_of_step": ray_ms / (elapsed / args.steps * 1e3), "algorithmic_bytes_per_launch": alg_bytes,
                    "algorithmic_bytes_rule": "42 B/packet + in-bounds cell writes x (8 B stamp RMW + 8 B counter RMW)",
                    "traffic_replayed": ray_traffic, "traffic_source": raycast_entry["counter_source"]}

    micro = None
    copy_gbs = 0.0
    if rank == 0 and world == 1 and not args.no_micro:
        copy_gbs = measured_copy_gbs(torch, dev)
        micro = micro_benches(pkg, torch, dev, m, G)
    rc = 0
    fuse_report = None
    if world > 1:
        # what ONE fuse moved (the last step's; every step is the same work): rank 0's figures and the maximum over the ranks
        fs = dict(sm.fuse_stats)
        dense = G * G * (4 + (8 if counts else 0))
        mx = torch.tensor([fs.get("payload_bytes", 0), fs.get("sent_bytes", 0), fs.get("received_bytes", 0)], dtype=torch.int64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        fuse_report = {"algorithm": args.fuse, "dense_map_bytes": dense,
                       "payload_bytes_per_gpu": int(mx[0]), "payload_frac_of_dense_map": int(mx[0]) / dense,
                       "sent_bytes_per_gpu": int(mx[1]), "received_bytes_per_gpu": int(mx[2]),
                       "blocks_own": fs.get("blocks_own"), "blocks_all_ranks": fs.get("blocks_all"), "block_bytes": fs.get("block_bytes"),
                       "note": ("payload = this rank's packed dirty blocks (4 x 16 cells: 64 stamps + 64 counter deltas), the bytes each "
                                "of its N-1 links carries; sent = payload x (N-1) + the bitmap all-gather; max over the ranks"
                                if args.fuse == "sparse" else
                                "dense: the whole stamp grid (+ the counter snapshot); sent = 2 (N-1)/N x that on a ring")}
    if rank == 0:
        wl_name = {"c1": "configs[1]: 2-bot stream", "adv": "adversarial uniform-random stream (SURVEY 8(d) D2), 2 bots",
                   "c3": ("configs[3]" if world > 1 else "configs[2] shape") + f": {bots} bots/GPU in their own room tiles"}[wl]
        out = {
            "metric": "QuasarPackets/sec into 4096^2 grid" if G == 4096 else f"QuasarPackets/sec into {G}^2 grid",
            "value": world * B * args.steps / elapsed,
            "unit": "packets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic: reference generator's 2-bot session cycled" if wl != "adv" else "synthetic: uniform-random adversarial stream",
            "config": {"workload": f"{wl_name}, {G}x{G} grid, res 0.05, {B} packets/step/GPU, fresh session per step, "
                                   f"decode+loop-closure+raycast{'+EKF' if args.ekf else ''}{'+grid fuse (' + args.fuse + ')' if world > 1 else ''}",
                       "batch": B, "grid": G, "counts": counts, "ekf": bool(args.ekf),
                       "raycast_mode": args.raycast_mode, "bots_per_gpu": bots, "bots_per_graph": bpg or max_agent,
                       "pose_graphs_per_gpu": m.n_graphs, "slam_mode": ("replicated (one pose graph over all bots)" if replicated else
                                                                       "per_shard (pose graphs per shard)") if world > 1 else "single mapper",
                       "sharding": f"by agent, {world} x {bots} bots", "fuse": args.fuse if world > 1 else None},
            "fuse": fuse_report,
            "stages_ms_per_step": st_ms,
            "counters_per_step": cnt,
            "roofline": roofline,
            "roofline_raycast": raycast_entry,
            "parity_checked": parity, "parity_scope": parity_note,
        }
        if world > 1:
            # The N = 1 default is configs[1] (2 bots); this line is configs[3] (64 bots per GPU).  For a scaling figure on ONE
            # workload, the same 64-bot workload at N = 1 -- replayed from the committed run, not measured now.
            ref = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"bench_{wl}{'' if G == 4096 else '_' + str(G)}.json")
            if not os.path.exists(ref):
                ref = os.path.join(ROOT, "profiles", "r02", f"bench_{wl}{'' if G == 4096 else '_' + str(G)}.json")
            try:
                r1 = json.load(open(ref))
                out["same_workload_n1"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                           "source": os.path.relpath(ref, ROOT) + " (python bench.py --workload " + wl + ", replayed)"}
            except Exception:
                out["same_workload_n1"] = None
        if micro is not None:
            out["roofline_streaming"] = micro
            out["copy_peak_measured_gbs"] = copy_gbs
        if world == 1 and wl == "c1" and micro is not None and not args.no_cpu_baseline:
            out["configs2_64_bots"] = secondary_64_bots(pkg, replay, torch, dev, side, args, G, counts, bpg=2, steps=20, warm=3)
            out["configs2_64_bots_one_graph"] = secondary_64_bots(pkg, replay, torch, dev, side, args, G, counts, bpg=0, steps=5, warm=1)
            if out["configs2_64_bots"]["parity_checked"] is False or out["configs2_64_bots_one_graph"]["parity_checked"] is False:
                parity = False
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if parity is False:
        rc = 1
    m.close()
    if world > 1:
        dist.destroy_process_group()
    return rc


def main():
    args = parse()
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.spawn_selftest:
        sys.exit(selftest_rank(args))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
