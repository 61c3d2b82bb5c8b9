"""The N>1 path on CPU: two processes (gloo), stream sharded by agent, per-rank grids fused with the
same all-reduce rule the GPU path uses (MAX on stamps carrying the GLOBAL arrival index, SUM on
counts).  The fused grid must equal one mapper fed the interleaved stream (shared-grid semantics,
dual_bot_mapper.py:785).  The per-rank grids come from the CPU oracle here; on the GPU box the same
rule runs over RCCL on the device buffers (bench.py --gpus N)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, PKG_NAME


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        distmod = importlib.import_module(PKG_NAME + ".dist")
        P = importlib.import_module(PKG_NAME + ".protocol")
        g = np.load(os.path.join(GOLDEN, "session_512.npz"), allow_pickle=False)
        pk = g["datagrams"][:, :42]
        agents = pk[:, 4]
        # rank r owns bot r+1; on its own wire that bot is agent 1; global arrival index = position in the stream
        mine = np.nonzero(agents == rank + 1)[0]
        shard = pk[mine].copy(); shard[:, 4] = 1
        m = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=1)
        for i, d in zip(mine, shard):
            m.set_sequence(int(i), 1)
            m.feed(d.tobytes())
        stamps = torch.from_numpy(m.stamps.astype(np.int64).astype(np.int32))      # < 2^31 by construction
        counts = torch.from_numpy(np.stack([m.misses, m.hits], axis=-1).copy())
        distmod.allreduce_tensors(stamps, counts)
        fused = distmod.tri_state_from_stamps(stamps.numpy())
        # one mapper, two independent pose graphs (bots_per_graph = 1), interleaved stream
        ref = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=2, bots_per_graph=1)
        ref.feed_stream(pk)
        ok = bool((fused == ref.grid).all() and (counts[..., 1].numpy() == ref.hits).all()
                  and (counts[..., 0].numpy() == ref.misses).all()
                  and (stamps.numpy().astype(np.uint32) == ref.stamps).all())
        q.put((rank, ok, int((fused != ref.grid).sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_fuse_equals_single_mapper():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ndiff in res:
        assert ok, f"rank {rank}: fused grid differs from the single mapper in {ndiff} cells"


def test_rank_sequence_and_sharding_helpers():
    distmod = importlib.import_module(PKG_NAME + ".dist")
    assert distmod.shard_of_bot(0, 64) == (0, 1) and distmod.shard_of_bot(63, 64) == (0, 64)
    assert distmod.shard_of_bot(64, 64) == (1, 1) and distmod.shard_of_bot(511, 64) == (7, 64)
    s = distmod.rank_sequence(3, 8, 1000, 4)
    assert s.tolist() == [1003, 1011, 1019, 1027]
    st = np.array([[0, 2, 3], [10, 11, 0]], dtype=np.uint32)
    assert distmod.tri_state_from_stamps(st).tolist() == [[-1, 0, 100], [0, 100, -1]]


# ---- round 2: repeated fuses, the direct reduce-scatter + all-gather, the replicated pose graph, the launcher --------
def _shard_oracles(rank, world, pk, orc, owned=False, max_agent=2, bpg=1):
    agents = pk[:, 4]
    if owned:                       # replicated pose graph: every rank sees every packet, casts only its own agent's rays
        m = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=max_agent, bots_per_graph=0)
        m.set_owned(rank + 1, rank + 1)
        return m, None
    mine = np.nonzero(agents == rank + 1)[0]
    m = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=1)
    return m, mine


def _worker2(rank, world, port, q, what):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        distmod = importlib.import_module(PKG_NAME + ".dist")
        g = np.load(os.path.join(GOLDEN, "laps5_512.npz"), allow_pickle=False)
        pk = g["datagrams"][:, :42]
        ok, note = True, ""
        if what == "two_batches":
            # ingest, fuse, ingest, fuse with NO reset in between.  Stamps: MAX in place is idempotent.  Counters:
            # the collective sums a SNAPSHOT of each rank's local counters, so the second fuse does not add the first again.
            m, mine = _shard_oracles(rank, world, pk, orc)
            ref = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=2, bots_per_graph=1)
            half = len(pk) // 2
            stamps = torch.zeros((512, 512), dtype=torch.int32)
            for lo, hi in ((0, half), (half, len(pk))):
                for i in mine[(mine >= lo) & (mine < hi)]:
                    d = pk[i].copy(); d[4] = 1
                    m.set_sequence(int(i), 1); m.feed(d.tobytes())
                ref.feed_stream(pk[lo:hi])
                # the rank's stamp buffer holds max(previous global, own writes): exactly what the device buffer holds
                stamps = torch.maximum(stamps, torch.from_numpy(m.stamps.astype(np.int64).astype(np.int32)))
                snap = torch.from_numpy(np.stack([m.misses, m.hits], axis=-1).copy())       # snapshot of LOCAL counters
                distmod.allreduce_tensors(stamps, snap)
                fused = distmod.tri_state_from_stamps(stamps.numpy())
                ok = ok and bool((fused == ref.grid).all() and (snap[..., 1].numpy() == ref.hits).all()
                                 and (snap[..., 0].numpy() == ref.misses).all())
            note = "two batches"
        elif what == "direct":
            # reduce_scatter_allgather == all_reduce, for MAX and for SUM
            rng = np.random.default_rng(7 + rank)
            a = torch.from_numpy(rng.integers(0, 1 << 30, 512 * 512).astype(np.int32))
            c = torch.from_numpy(rng.integers(0, 1000, 512 * 512 * 2).astype(np.int32))
            a_ref, c_ref = a.clone(), c.clone()
            dist.all_reduce(a_ref, op=dist.ReduceOp.MAX); dist.all_reduce(c_ref, op=dist.ReduceOp.SUM)

            def fold_max(off, n, recv):
                a[off:off + n] = torch.maximum(a[off:off + n], recv.max(dim=0).values)

            def fold_sum(off, n, recv):
                c[off:off + n] += recv.sum(dim=0)

            distmod.reduce_scatter_allgather(a, fold_max)
            distmod.reduce_scatter_allgather(c, fold_sum)
            ok = bool(torch.equal(a, a_ref) and torch.equal(c, c_ref))
            note = "direct"
        elif what == "replicated":
            # ONE pose graph over both bots (the reference's PoseGraphSLAM, :275, :294-309), its chain replicated: each rank
            # ingests the all-gathered interleaved stream and casts only its own agent's rays; fused == one mapper.
            per = [pk[pk[:, 4] == r + 1] for r in range(world)]
            B = min(len(p) for p in per)
            mine = torch.from_numpy(per[rank][:B].copy())
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            full = torch.stack(gathered, dim=1).reshape(world * B, 42).numpy()       # round-robin interleave
            m, _ = _shard_oracles(rank, world, pk, orc, owned=True)
            m.feed_stream(full)
            stamps = torch.from_numpy(m.stamps.astype(np.int64).astype(np.int32))
            snap = torch.from_numpy(np.stack([m.misses, m.hits], axis=-1).copy())
            distmod.allreduce_tensors(stamps, snap)
            ref = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=2, bots_per_graph=0)
            ref.feed_stream(full)
            fused = distmod.tri_state_from_stamps(stamps.numpy())
            ci, cc = m.closures(0); ri, rc = ref.closures(0)
            ok = bool((fused == ref.grid).all() and (snap[..., 1].numpy() == ref.hits).all() and (ci == ri).all()
                      and len(ri) > 5 and np.abs(m.drift(1) - ref.drift(1)).max() == 0 and np.abs(m.drift(2) - ref.drift(2)).max() == 0)
            note = f"replicated, {len(ri)} closures"
        q.put((rank, ok, note))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("what", ["two_batches", "direct", "replicated"])
def test_two_rank_modes(what):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, q, what)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, note in res:
        assert ok, f"rank {rank}: {what} ({note}) differs from the single mapper"


def test_bench_launcher_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun) must start two ranks: the parent spawns them before it touches any GPU,
    gives each RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and n_gpus in the output is the world size the ranks saw."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-selftest", "--backend", "gloo"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["sum"] == 3.0


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launcher():
    """Under torchrun (RANK / WORLD_SIZE set) bench.py is ONE rank; `--gpus` must then equal the world size (round 1 parsed
    the flag and ignored it: `--gpus 8` printed n_gpus: 1)."""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr and out.stdout.strip() == ""
