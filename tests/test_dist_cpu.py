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
