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


def test_bench_launcher_stops_the_siblings_of_a_rank_that_dies():
    """a rank that exits non-zero while the others sit in a collective (here: in the rendezvous) must not leave the
    parent waiting for them -- it terminates them and returns the failing rank's code."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--spawn-selftest", "--backend", "gloo",
                          "--selftest-fail-rank", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert "stopped the remaining ranks" in out.stderr and time.time() - t0 < 120


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launcher():
    """Under torchrun (RANK / WORLD_SIZE set) bench.py is ONE rank; `--gpus` must then equal the world size (round 1 parsed
    the flag and ignored it: `--gpus 8` printed n_gpus: 1)."""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr and out.stdout.strip() == ""


# ---- round 3: the sparse (dirty-block) fuse -- the protocol of dist.sparse_fuse with a numpy adapter in the kernels' place ----
BW, BH = 16, 4        # QS_DIRTY_BLOCK_W / QS_DIRTY_BLOCK_H (checked against the header in test_abi_cpu.py)


class NumpySparseAdapter:
    """CPU stand-in for dist.MapperSparseAdapter with the SAME bitmap layout (rows of `pitch` words, one bit per 4 x 16 block)
    and payload layout (64 stamps + 64 counter deltas per block, ascending block index) as csrc/sparse_fuse.hip.  State =
    what a context holds: stamps (max of own writes and everything fused so far), own counters, counters as of the last
    fuse, fused counters."""

    def __init__(self, size):
        self.size = size
        self.bx, self.by = -(-size // BW), -(-size // BH)
        self.pitch = -(-self.bx // 32)
        self.stamps = np.zeros((size, size), np.int32)
        self.counts = np.zeros((size, size, 2), np.int32)          # [..., 0] misses, [..., 1] hits (the device's lo32 / hi32)
        self.sent = np.zeros_like(self.counts)
        self.fused = np.zeros_like(self.counts)
        self.dirty = np.zeros((self.by, self.pitch), np.uint32)

    def write(self, new_stamps, new_counts):
        """The oracle's state after a batch becomes the context's: blocks where a stamp or a counter changed are marked."""
        ch = (np.maximum(self.stamps, new_stamps) != self.stamps) | (new_counts != self.counts).any(axis=-1)
        self.stamps = np.maximum(self.stamps, new_stamps); self.counts = new_counts.copy()
        pad = np.zeros((self.by * BH, self.bx * BW), bool); pad[:self.size, :self.size] = ch
        blk = pad.reshape(self.by, BH, self.bx, BW).any(axis=(1, 3))
        for y, x in zip(*np.nonzero(blk)):
            self.dirty[y, x // 32] |= np.uint32(1 << (x % 32))

    def _cells(self, bid):
        y0, x0 = (bid // (self.pitch * 32)) * BH, (bid % (self.pitch * 32)) * BW
        return slice(y0, y0 + BH), slice(x0, min(x0 + BW, self.size)), min(BW, self.size - x0)

    def begin(self, world, rank):
        self.world, self.rank = world, rank
        self.bm = torch.zeros((world, self.by * self.pitch), dtype=torch.int32)
        self.bm[rank] = torch.from_numpy(self.dirty.reshape(-1).view(np.int32).copy())
        self.dirty[:] = 0
        return self.bm

    def plan(self, world):
        bits = self.bm.numpy().view(np.uint32)
        self.lists = [[w * 32 + b for w in np.nonzero(bits[s])[0] for b in range(32) if bits[s][w] >> b & 1] for s in range(world)]
        n = np.array([len(l) for l in self.lists], np.uint32)
        bb = 64 * 4 + 64 * 8
        off = np.concatenate([[0], np.cumsum(n.astype(np.int64) * bb)])
        buf = torch.zeros(int(off[-1]), dtype=torch.uint8)
        seg = buf.numpy()
        for k, bid in enumerate(self.lists[self.rank]):
            ys, xs, w = self._cells(bid)
            st = np.zeros((BH, BW), np.int32); st[:, :w] = self.stamps[ys, xs]
            d = np.zeros((BH, BW, 2), np.int32); d[:, :w] = self.counts[ys, xs] - self.sent[ys, xs]
            self.sent[ys, xs] = self.counts[ys, xs]
            o = int(off[self.rank]) + k * bb
            seg[o:o + 256] = st.reshape(-1).view(np.uint8)
            seg[o + 256:o + 768] = d.reshape(-1).view(np.uint8)
        self.off, self.buf, self.bb = off, buf, bb
        return n, off, (buf if len(buf) else None), bb

    def apply(self):
        seg = self.buf.numpy()
        for s in range(self.world):
            for k, bid in enumerate(self.lists[s]):
                ys, xs, w = self._cells(bid)
                o = int(self.off[s]) + k * self.bb
                st = seg[o:o + 256].view(np.int32).reshape(BH, BW)[:, :w]
                d = seg[o + 256:o + 768].view(np.int32).reshape(BH, BW, 2)[:, :w]
                if s != self.rank:
                    self.stamps[ys, xs] = np.maximum(self.stamps[ys, xs], st)
                self.fused[ys, xs] += d


def _worker_sparse(rank, world, port, q, size, pitch_m):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        distmod = importlib.import_module(PKG_NAME + ".dist")
        replay = importlib.import_module(PKG_NAME + ".replay")
        session, _ = replay.telemetry_csv_to_packets()
        half = size * 0.05 / 2
        geo = dict(pitch=pitch_m, tiles_per_row=2 * world, origin=(-half + 3.0, -1.0))
        n = 900                                                    # records per rank, three batches
        # rank r: bots 2r, 2r + 1 (agents 1, 2 on its own wire, one pose graph); record i of rank r is global record i*W + r
        shards = [replay.multi_bot_stream(session, 2, n, tile0=2 * r, **geo) for r in range(world)]
        mine = shards[rank]
        inter = np.empty((n * world, 42), np.uint8)
        for r in range(world):
            g = shards[r].copy(); g[:, 4] += 2 * r                 # globally unique agent ids for the one reference mapper
            inter[r::world] = g
        m = orc.OracleMapper(size, 0.05, -half, -half, 0.0, max_agent=2)
        m.set_sequence(rank, world)
        ref = orc.OracleMapper(size, 0.05, -half, -half, 0.0, max_agent=2 * world, bots_per_graph=2)
        ad = NumpySparseAdapter(size)
        ok, stats, moved = True, {}, []
        for k in range(3):
            lo, hi = k * n // 3, (k + 1) * n // 3
            m.feed_stream(mine[lo:hi]); ref.feed_stream(inter[lo * world:hi * world])
            ad.write(m.stamps.astype(np.int64).astype(np.int32), np.stack([m.misses, m.hits], axis=-1))
            distmod.sparse_fuse(ad, rank, world, stats=stats)
            moved.append(stats["blocks_own"])
            ok = ok and bool((ad.stamps.astype(np.uint32) == ref.stamps).all() and (ad.fused[..., 1] == ref.hits).all()
                             and (ad.fused[..., 0] == ref.misses).all())
            # an empty fuse right after: nothing travels, nothing changes
            distmod.sparse_fuse(ad, rank, world, stats=stats)
            ok = ok and stats["blocks_all"] == 0 and bool((ad.fused[..., 1] == ref.hits).all())
        dense = size * size * 12
        q.put((rank, ok, moved, stats.get("bitmap_bytes"), dense))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,pitch_m", [(2, 704, 8.0), (3, 520, 1.5), (4, 1024, 6.0)], ids=["2_apart", "3_overlapping_ragged_edge", "4_apart"])
def test_sparse_fuse_protocol_equals_single_mapper(world, size, pitch_m):
    """dist.sparse_fuse over gloo: bitmaps all-gathered, every rank's packed dirty blocks sent to every peer point to
    point, folded in.  After every batch the stamps and the fused counters of EVERY rank equal one mapper fed the
    interleaved stream -- rooms apart (disjoint blocks), rooms on top of each other (the same blocks from three ranks, a grid
    whose width is not a multiple of the block width) -- and later fuses move only what the new batch touched."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sparse, args=(r, world, port, q, size, pitch_m)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, moved, bitmap_bytes, dense in res:
        assert ok, f"rank {rank}: sparse-fused grid differs from the single mapper"
        assert moved[0] > 0 and moved[0] * 768 < 0.35 * dense, moved          # a room, not the map
