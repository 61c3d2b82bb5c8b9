"""Round 3: the sparse (dirty-block) fuse on the GPU -- N HIP contexts of one process play the N ranks
(dist.sparse_fuse_local: device-to-device copies in the collectives' place), and bench.py's own N-rank path with real
processes (gloo, all ranks on cuda:0).  The bar is the dense fuse's: every rank's stamps == cell-wise MAX of the ranks'
oracle stamps, every rank's fused counters == the SUM of the ranks' oracle counters (shared-grid semantics,
dual_bot_mapper.py:785), bit for bit, after every fuse of a session."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # before the HIP library: torch bundles its own HIP runtime, and whichever of the two is loaded first has to be torch's

from conftest import ROOT, load_pkg
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
FLOAT_TOL = 1e-5


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _mods(pkg):
    return importlib.import_module(pkg.__name__ + ".dist"), importlib.import_module(pkg.__name__ + ".replay")


def _device_grids(distmod, m, dev):
    import torch
    torch.cuda.synchronize()
    st, _ = distmod.grid_tensors(m, dev)
    fz = distmod.fused_counts_view(m, dev)
    return st.cpu().numpy().astype(np.uint32), (fz.cpu().numpy() if fz is not None else None)


def _check_all(distmod, mappers, oracles, dev, tag):
    stamps = np.maximum.reduce([o.stamps for o in oracles])
    hits = np.sum([o.hits for o in oracles], axis=0); misses = np.sum([o.misses for o in oracles], axis=0)
    for r, m in enumerate(mappers):
        st, fz = _device_grids(distmod, m, dev)
        assert (st == stamps).all(), f"{tag}: rank {r}: {(st != stamps).sum()} stamps differ from the fused oracle stamps"
        if fz is not None:
            assert (fz[..., 1] == hits).all() and (fz[..., 0] == misses).all(), f"{tag}: rank {r}: fused counters differ"
        h, mi = m.counts()                                    # the views read the fused counters after a sparse fuse
        assert (h == hits).all() and (mi == misses).all(), f"{tag}: rank {r}: counts() view"


@pytest.mark.parametrize("world,mode", [(2, 2), (3, 1), (8, 0)])
def test_sparse_fuse_contexts_equal_dense_fuse_of_the_oracles(pkg, world, mode):
    """per-shard sharding as bench.py does it (rank r: its own bots, arrival index i*N + r), three batches with NO reset, a
    fuse after each -- tiled raster merge (mode 2), direct kernel (mode 1), auto (mode 0: the last, 100-packet batch takes
    the direct kernel) --, plus rays through the object API (qs_update_rays) on rank 0 between two fuses, plus a fuse with
    nothing to move.  The moved blocks shrink to what the last batch touched."""
    import torch
    distmod, replay = _mods(pkg)
    dev = torch.device("cuda", 0)
    session, _ = replay.telemetry_csv_to_packets()
    G, half = 1024, 25.6
    geo = dict(pitch=5.0, tiles_per_row=8, origin=(-22.0, -20.0))       # rooms 5 m apart: neighbours share blocks
    bots, B = 4, 6100
    streams = [replay.multi_bot_stream(session, bots, B, tile0=r * bots, **geo) for r in range(world)]
    oracles = [orc.OracleMapper(G, 0.05, -half, -half, 0.0, max_agent=bots, bots_per_graph=2) for _ in range(world)]
    for r, o in enumerate(oracles):
        o.set_sequence(r, world)
    mappers = [pkg.QuasarMapper(G, 0.05, -half, -half, max_agent=bots, bots_per_graph=2, seq_stride=world, raycast_mode=mode)
               for _ in range(world)]
    try:
        for m in mappers:
            m.dirty_tracking(True)
        moved = []
        cuts = [0, 3000, 6000, 6100]
        for k in range(3):
            lo, hi = cuts[k], cuts[k + 1]
            for r, (m, o) in enumerate(zip(mappers, oracles)):
                m.ingest_array(streams[r][lo:hi], seq0=lo * world + r)
                o.feed_stream(streams[r][lo:hi])
            own, _ = mappers[0].dirty_blocks()
            n = distmod.sparse_fuse_local(mappers, dev)
            assert int(n[0]) == own > 0
            moved.append(int(n.sum()))
            _check_all(distmod, mappers, oracles, dev, f"batch {k}")
        assert moved[2] < moved[0] / 3, moved                            # 100 packets touch a few blocks, not the rooms again
        # object-API rays on rank 0, written after everything ingested so far (the oracle's update_rays keeps no ordinals of
        # its own: expected = the three rays, in order, over the fused map)
        rx = np.array([-20.0, -19.5, 3.0]); ry = np.array([-19.0, -18.5, 2.0])
        hx = rx + np.array([1.0, -0.7, 0.9]); hy = ry + np.array([0.3, 0.8, -1.1]); valid = np.array([1, 0, 1], np.uint8)
        mappers[0].update_rays(rx, ry, hx, hy, valid, seq0=6100 * world + 8)
        over = orc.OracleMapper(G, 0.05, -half, -half, 0.0)
        over.update_rays(rx, ry, hx, hy, valid)
        stamps = np.maximum.reduce([o.stamps for o in oracles])
        tri = np.where(stamps == 0, -1, np.where(stamps & 1, 100, 0)).astype(np.int8)
        tri = np.where(over.grid != -1, over.grid, tri)
        hits = np.sum([o.hits for o in oracles], axis=0) + over.hits; misses = np.sum([o.misses for o in oracles], axis=0) + over.misses
        for tag, expect in (("update_rays", (1, 40)), ("empty fuse", (0, 0))):     # second fuse: nothing written since, nothing moves
            n = distmod.sparse_fuse_local(mappers, dev)
            assert expect[0] <= int(n.sum()) <= expect[1], (tag, n)
            for r, m in enumerate(mappers):
                assert (m.grid_i8() == tri).all(), f"{tag}: rank {r}"
                h, mi = m.counts()
                assert (h == hits).all() and (mi == misses).all(), f"{tag}: rank {r}"
        # the pose graphs are the shards' own
        for m, o in zip(mappers, oracles):
            for g in range(m.n_graphs):
                assert (m.closures(g)[0] == o.closures(g)[0]).all()
    finally:
        for m in mappers:
            m.close()


def test_sparse_fuse_across_an_epoch_boundary_and_odd_width(pkg):
    """A grid whose width is not a multiple of the block width (68: the last block of a row is 4 cells wide), exact-trig
    edge rays on, and a stream that crosses the 2^28 stamp epoch: a shard with unfused writes is refused the batch that
    rebases; after the (sparse) fuse it goes through -- what dist.ShardedMapper does when qs_epoch_query says so.  The
    replicated-pose-graph shard (seq_stride 1, shard_bots > 0) is refused in the same way."""
    import torch
    distmod, replay = _mods(pkg)
    dev = torch.device("cuda", 0)
    G, res = 68, 0.05
    n = 1208
    P = pkg.protocol

    def stream(seed, agent=1):
        r = np.random.default_rng(seed)
        return P.pack_packets(np.full(n, agent), r.uniform(-1.5, 1.5, n), r.uniform(-1.5, 1.5, n), np.radians(r.integers(0, 24, n) * 15.0),
                              np.arange(n), np.zeros(n, int), np.round(r.uniform(0.02, 1.6, (n, 4)), 2), r.choice([0, 0, 0, 5], n))
    world = 2
    streams = [stream(100 + r) for r in range(world)]
    base = (1 << 28) - 2 - 2 * 808
    oracles = [orc.OracleMapper(G, res, -1.7, -1.7, 0.0, max_agent=1) for _ in range(world)]
    mappers = [pkg.QuasarMapper(G, res, -1.7, -1.7, max_agent=1, seq_stride=world) for _ in range(world)]

    def both(lo, hi):
        for r, (m, o) in enumerate(zip(mappers, oracles)):
            m.ingest_array(streams[r][lo:hi], seq0=base + world * lo + r)
            o.set_sequence(base + world * lo + r, world); o.feed_stream(streams[r][lo:hi])

    def check(tag):
        grid = np.maximum.reduce([o.stamps for o in oracles])
        tri = np.where(grid == 0, -1, np.where(grid & 1, 100, 0)).astype(np.int8)
        hits = np.sum([o.hits for o in oracles], axis=0); misses = np.sum([o.misses for o in oracles], axis=0)
        for m in mappers:
            assert (m.grid_i8() == tri).all(), tag
            h, mi = m.counts()
            assert (h == hits).all() and (mi == misses).all(), tag
    try:
        for m in mappers:
            m.dirty_tracking(True)
        for k, (lo, hi) in enumerate(((0, 400), (400, 800))):
            assert not any(m.epoch_would_rebase(hi - lo, base + world * lo + r) for r, m in enumerate(mappers))
            both(lo, hi)
            distmod.sparse_fuse_local(mappers, dev)
            check(f"batch {k}")
        both(800, 808)                                                    # eight more records each: unfused writes on both shards
        assert all(m.epoch_would_rebase(400, base + world * 808 + r) for r, m in enumerate(mappers))
        with pytest.raises(pkg.QuasarError, match="crosses a stamp epoch"):
            mappers[0].ingest_array(streams[0][808:1208], seq0=base + world * 808)
        distmod.sparse_fuse_local(mappers, dev)
        both(808, 1208)
        assert all(m.counters()["rebases"] == 1 for m in mappers)
        distmod.sparse_fuse_local(mappers, dev)
        check("after the rebase")
        assert sum(m.counters()["edge_rays"] for m in mappers) >= 0
    finally:
        for m in mappers:
            m.close()
    # replicated pose graph: seq_stride = 1, but the shards cast different agents' rays -- same guard
    inter = stream(7, agent=1); inter[1::2, 4] = 2
    with pkg.QuasarMapper(G, res, -1.7, -1.7, max_agent=2, shard_bots=1, shard_rank=0) as a:
        a.ingest_array(inter[:100], seq0=(1 << 28) - 500)
        with pytest.raises(pkg.QuasarError, match="crosses a stamp epoch"):
            a.ingest_array(inter[100:700], seq0=(1 << 28) - 400)
        a.mark_fused()
        a.ingest_array(inter[100:700], seq0=(1 << 28) - 400)
        assert a.counters()["rebases"] == 1


def _expect_on_device(oracles, dev):
    import torch
    stamps = np.maximum.reduce([o.stamps for o in oracles]).astype(np.int64).astype(np.int32)
    cnt = np.stack([np.sum([o.misses for o in oracles], axis=0), np.sum([o.hits for o in oracles], axis=0)], axis=-1).astype(np.int32)
    return torch.from_numpy(stamps).to(dev), torch.from_numpy(cnt).to(dev)


@pytest.mark.parametrize("grid", [4096, 8192], ids=["configs3_4096", "configs4_8192"])
def test_configs3_and_4_at_eight_ranks_on_one_gpu(pkg, grid):
    """BASELINE.json configs[3] / configs[4] at their own shape: 512 bots, 64 per rank, EIGHT ranks -- eight HIP contexts of
    this process (a one-GPU box admits six GPU processes, so the 8-rank case cannot be eight bench.py ranks), 4096^2 and
    8192^2, bot i of rank r in room tile r*64 + i, global arrival indices, EKF on, two batches without reset, a sparse fuse
    after each.  Every rank ends with the same map as the dense fuse of the eight oracles, and one fuse moves <= 5 % of the
    dense map's bytes per rank."""
    import torch
    distmod, replay = _mods(pkg)
    dev = torch.device("cuda", 0)
    session, _ = replay.telemetry_csv_to_packets()
    world, bots, B = 8, 64, 16384
    half = grid * 0.05 / 2
    geo = dict(tiles_per_row=25 if grid == 4096 else 51, origin=(-half + 4.4, -half + 4.4))
    streams = [replay.multi_bot_stream(session, bots, B, tile0=r * bots, **geo) for r in range(world)]
    oracles = [orc.OracleMapper(grid, 0.05, -half, -half, 0.0, max_agent=bots, bots_per_graph=2) for _ in range(world)]
    mappers = [pkg.QuasarMapper(grid, 0.05, -half, -half, max_agent=bots, bots_per_graph=2, seq_stride=world, enable_ekf=True)
               for _ in range(world)]
    try:
        for r, (m, o) in enumerate(zip(mappers, oracles)):
            m.dirty_tracking(True)
            o.set_sequence(r, world); o.enable_ekf(0.0107)
        d_time = np.arange(B) * 0.25
        for k, (lo, hi) in enumerate(((0, B // 2), (B // 2, B))):
            for r, (m, o) in enumerate(zip(mappers, oracles)):
                m.ingest_array(streams[r][lo:hi], recv_time=d_time[lo:hi], seq0=lo * world + r)
                o.feed_stream(streams[r][lo:hi], None, d_time[lo:hi])
            n = distmod.sparse_fuse_local(mappers, dev)
            bb = 768
            dense = grid * grid * 12
            assert int(n.max()) * bb <= 0.05 * dense, (n, int(n.max()) * bb / dense)
            est, ecnt = _expect_on_device(oracles, dev)
            for r, m in enumerate(mappers):
                st, _ = distmod.grid_tensors(m, dev)
                assert torch.equal(st, est), f"batch {k}: rank {r}: {int((st != est).sum())} stamps differ"
                assert torch.equal(distmod.fused_counts_view(m, dev), ecnt), f"batch {k}: rank {r}: fused counters differ"
            del est, ecnt
        for r, (m, o) in enumerate(zip(mappers, oracles)):                # the shards' own pose graphs, drift and filters
            for g in (0, 13, 31):
                idx, corr = m.closures(g); oi, oc = o.closures(g)
                assert (idx == oi).all() and (len(oi) == 0 or np.abs(corr - oc).max() < FLOAT_TOL)
            for b in (1, 17, 64):
                assert np.abs(m.drift(b) - o.drift(b)).max() < FLOAT_TOL
                x, _ = m.ekf_state(b); ox_, _ = o.ekf_state(b)
                assert np.abs(x - ox_).max() < 1e-6 * max(1.0, np.abs(ox_).max())
    finally:
        for m in mappers:
            m.close()


def test_replicated_pose_graph_at_eight_ranks_sparse_fuse(pkg):
    """The parity-keeping mode at eight ranks: ONE pose graph over 8 x 31 = 248 bots (agent_id is one byte), its chain
    replicated on every rank, rays cast by the owner only; sparse fuse; == one mapper fed the interleaved stream."""
    import torch
    distmod, replay = _mods(pkg)
    dev = torch.device("cuda", 0)
    session, _ = replay.telemetry_csv_to_packets()
    world, bots, B, grid = 8, 31, 2048, 4096
    half = grid * 0.05 / 2
    shards = [replay.multi_bot_stream(session, bots, B, tile0=r * bots, agent0=r * bots + 1) for r in range(world)]
    inter = np.empty((world * B, 42), np.uint8)
    for r in range(world):
        inter[r::world] = shards[r]
    ref = orc.OracleMapper(grid, 0.05, -half, -half, 0.0, max_agent=world * bots, bots_per_graph=0)
    ref.feed_stream(inter)
    mappers = [pkg.QuasarMapper(grid, 0.05, -half, -half, max_agent=world * bots, bots_per_graph=0, shard_bots=bots, shard_rank=r)
               for r in range(world)]
    try:
        for m in mappers:
            m.dirty_tracking(True)
            m.ingest_array(inter)
        distmod.sparse_fuse_local(mappers, dev)
        oi, oc = ref.closures(0)
        assert len(oi) > 3
        for r, m in enumerate(mappers):
            assert (m.grid_i8() == ref.grid).all(), f"rank {r}"
            h, mi = m.counts()
            assert (h == ref.hits).all() and (mi == ref.misses).all()
            idx, corr = m.closures(0)
            assert (idx == oi).all() and np.abs(corr - oc).max() < FLOAT_TOL
    finally:
        for m in mappers:
            m.close()


@pytest.mark.parametrize("extra", [["--gpus", "2"], ["--gpus", "2", "--fuse", "allreduce"], ["--gpus", "4"],
                                   ["--gpus", "4", "--slam-mode", "replicated", "--bots", "31"], ["--gpus", "2", "--grid", "8192"]],
                         ids=["2_sparse", "2_allreduce", "4_sparse", "4_replicated_sparse", "2_sparse_8192"])
def test_bench_ranks_rehearsed_on_one_gpu_sparse(extra):
    """bench.py's N > 1 path with real processes and the real HIP path on every rank (all on cuda:0, gloo): the default fuse
    is the sparse one; the bench's own parity check (fused stamps / counters of all ranks == the same fuse of the ranks'
    oracle grids) must hold and the line must say what the fuse moved."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--rehearse-on-one-gpu",
           "--batch", "32768", "--steps", "2", "--warmup", "1", "--no-micro"] + extra
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    n = int(extra[1])
    assert rec["n_gpus"] == n and rec["parity_checked"] is True
    f = rec["fuse"]
    if "--fuse" in extra:
        assert f["algorithm"] == "allreduce" and f["payload_bytes_per_gpu"] == f["dense_map_bytes"]
    else:
        assert f["algorithm"] == "sparse" and 0 < f["payload_frac_of_dense_map"] <= 0.05, f
        assert f["sent_bytes_per_gpu"] >= f["payload_bytes_per_gpu"] * (n - 1)


def test_sparse_fuse_over_rccl_behind_the_c_abi_single_rank(pkg):
    """qs_sparse_fuse_rccl (csrc/rccl_fuse.hip): the fuse with its exchanges on RCCL inside the library, for hosts without
    torch.  One GPU means one rank: this pins that RCCL loads on demand, a communicator comes up, the collective runs on
    the context's stream and the fused counters accumulate deltas over two fuses -- the multi-rank exchange itself is the
    protocol test_dist_cpu.py and the N-context tests pin, here with ncclSend / ncclRecv in the copies' place."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "session_512.npz"), allow_pickle=False)
    pk = g["datagrams"][:, :42]
    o = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0)
    with pkg.QuasarMapper(512, 0.05, -12.8, -12.8) as m:
        m.dirty_tracking(True)
        comm = m.rccl_comm_init(pkg.QuasarMapper.rccl_unique_id(), 1, 0)
        try:
            half = len(pk) // 2
            for lo, hi in ((0, half), (half, len(pk))):
                m.ingest_array(pk[lo:hi]); o.feed_stream(pk[lo:hi])
                own, _ = m.dirty_blocks()
                st = m.sparse_fuse_rccl(comm, 1, 0)
                assert st["blocks_own"] == own > 0 and st["payload_bytes"] == own * 768 and st["sent_bytes"] == 0
                h, mi = m.counts()                                   # the fused view: what the ranks' deltas add up to
                assert (h == o.hits).all() and (mi == o.misses).all() and (m.grid_i8() == o.grid).all()
            assert m.sparse_fuse_rccl(comm, 1, 0)["blocks_own"] == 0
        finally:
            m.rccl_comm_destroy(comm)
