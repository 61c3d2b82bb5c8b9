"""Multi-GPU: shard telemetry by agent, fuse per-GPU grids with RCCL over xGMI.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  Every rank raycasts its
own bots' packets into a full-size local stamp grid; because a stamp carries the packet's GLOBAL
arrival index, the cell-wise MAX of the per-rank grids is bit-identical to one mapper fed the
interleaved stream (shared-grid semantics of dual_bot_mapper.py:785).  Hit/miss counters add.
torch is plumbing here: device memory views and the collectives, no arithmetic.

Two things differ between the ranks' streams and one mapper's, and each has a mode here:

* the pose graph.  PoseGraphSLAM is global across bots in the reference (node.index counts every bot's
  poses, :275; a landmark of one bot can close another's loop, :294-309).
  - mode "per_shard": every rank keeps the pose graph(s) of its own bots (a shard = one mapper instance; with the
    reference's 2-bot limit, "one pose graph per mapper process" is the deployment unit).  No exchange before the fuse.
  - mode "replicated": ONE pose graph over all bots.  The ranks all-gather the batch's raw datagrams (42 B each), every
    rank decodes and runs the loop-closure chain over the whole interleaved stream -- identical work, identical drifts
    everywhere -- and casts rays only for its own agents (qs_config.shard_bots / shard_rank).  Result = one mapper fed
    the interleaved stream, bit for bit; the chain does not scale with ranks (it is the reference's recurrence).
* the counters.  They are per-rank sums of the rank's own writes and are never all-reduced in place (a second
  all-reduce would add the peers' totals again): the collective sums a snapshot (qs_fused_counts).

Fuse algorithms: "sparse" (default) = only the 4 x 16-cell blocks a rank has written since its last fuse travel: the
ranks all-gather their dirty-block bitmaps (32 KiB each at 4096^2), every rank packs its blocks (64 stamps + 64 counter
deltas, 768 B) and sends the packed segment to each peer point to point -- all xGMI links at once -- and folds what it
receives (stamps MAX, counter deltas ADD); a shard by agent writes a few rooms, ~4 % of the map on configs[3], so ~8 MB
leaves a GPU per link instead of 192 MiB going round a ring.  "allreduce" = RCCL's dense all-reduce of the whole map
(ring/tree, its choice) -- the reference point; "direct" = dense reduce-scatter by point-to-point exchange + local fold
(the K3 kernel, qs_fuse_buffers_range) + all-gather: on a fully connected xGMI node every rank sends 1/N of the grid to
each of its N-1 peers at once, so all seven links carry traffic in both phases, where a ring is
bound by one link.
"""
import numpy as np


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can alias library-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr,
                                         "version": 2, "strides": None}


def grid_tensors(mapper, device):
    """(stamps int32 [size,size], counts int32 [size,size,2] or None) aliasing the context's
    device buffers.  Stamps stay below 2^31 by construction, so int32 MAX is exact."""
    import torch
    sp, sb, cp, cb = mapper.device_buffers()
    n = mapper.size
    stamps = torch.as_tensor(_DevArray(sp, (n, n), "<i4"), device=device)
    counts = torch.as_tensor(_DevArray(cp, (n, n, 2), "<i4"), device=device) if cp else None
    return stamps, counts


def fused_counts_tensor(mapper, device):
    """Snapshot the local counters (qs_fused_counts) -> int32 [size,size,2] aliasing the snapshot."""
    import torch
    p, _ = mapper.fused_counts()
    n = mapper.size
    return torch.as_tensor(_DevArray(p, (n, n, 2), "<i4"), device=device)


def allreduce_tensors(stamps, counts=None, group=None):
    """The fuse rule on plain tensors (any backend): latest stamp wins, counters add.  `counts` must be a SNAPSHOT
    of the rank's counters (or a tensor that is summed only once): the sum lands in place."""
    import torch.distributed as dist
    dist.all_reduce(stamps, op=dist.ReduceOp.MAX, group=group)
    if counts is not None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)


def reduce_scatter_allgather(t, fold, group=None):
    """Direct all-reduce of a contiguous tensor whose element count divides by the world size:
       1. every rank sends slice p of its tensor to rank p and receives its peers' copies of its own slice
          (N-1 sends + N-1 receives, all in flight together: one batch_isend_irecv);
       2. fold(offset, n, recv) folds the received copies (recv: [N-1, n]) into t.view(-1)[offset : offset + n];
       3. all-gather of the folded slices, in place.
    `fold` is the K3 kernel on the GPU (ShardedMapper) and a torch reduction in the CPU tests."""
    import torch
    import torch.distributed as dist
    W, r = dist.get_world_size(group), dist.get_rank(group)
    flat = t.view(-1)
    if W == 1:
        return
    assert flat.numel() % W == 0, "tensor size must divide by the world size"
    n = flat.numel() // W
    recv = torch.empty((W - 1, n), dtype=flat.dtype, device=flat.device)
    peers = [p for p in range(W) if p != r]
    ops = []
    for k, p in enumerate(peers):
        gp = dist.get_global_rank(group, p) if group is not None else p
        ops.append(dist.P2POp(dist.isend, flat[p * n:(p + 1) * n], gp, group))
        ops.append(dist.P2POp(dist.irecv, recv[k], gp, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    fold(r * n, n, recv)
    mine = flat[r * n:(r + 1) * n]
    if flat.is_cuda:
        dist.all_gather_into_tensor(flat, mine, group=group)          # in place: input is slice r of the output
    else:
        parts = [torch.empty_like(mine) for _ in range(W)]            # gloo has no in-place form
        dist.all_gather(parts, mine.clone(), group=group)
        for p in range(W):
            flat[p * n:(p + 1) * n] = parts[p]


def allreduce_grids(mapper, device, group=None, counts=True, algo="allreduce", sync=True):
    """Fuse every rank's grid into the global map (RCCL over xGMI).  Stamps: MAX in place on all ranks (idempotent:
    a later call, or a later local write, keeps the order).  Counters: the sum of the ranks' SNAPSHOTS lands in each
    context's fused buffer (mapper.counts_source(True) is set, so counts()/logodds() read the global map); the local
    counters are untouched and the call may be repeated after every batch.  Returns the fused counts tensor or None.
    `sync`: wait for the context's stream first (not needed when it is torch's current stream)."""
    import torch
    if sync:
        mapper.sync()
    stamps, local = grid_tensors(mapper, device)
    fused = fused_counts_tensor(mapper, device) if (counts and local is not None) else None
    if sync and fused is not None:
        mapper.sync()                 # the snapshot copy runs on the context's stream
    if algo == "direct":
        cells = mapper.size * mapper.size

        def fold_stamps(off, n, recv):
            base = recv.data_ptr()
            if sync:
                torch.cuda.current_stream().synchronize()      # the receives complete on torch's stream
            mapper.fuse_buffers_range([base + k * n * 4 for k in range(recv.shape[0])], None, off, n)
            if sync:
                mapper.sync()

        def fold_counts(off, n, recv):          # recv: int32 [N-1, 2*cells/N]; offsets in int32 units -> cells
            base = recv.data_ptr()
            if sync:
                torch.cuda.current_stream().synchronize()
            mapper.fuse_buffers_range(None, [base + k * n * 4 for k in range(recv.shape[0])], off // 2, n // 2,
                                      counts_into_fused=True)
            if sync:
                mapper.sync()

        import torch.distributed as dist
        W = dist.get_world_size(group)
        if cells % (4 * W) != 0:
            raise ValueError("direct fuse: size*size must divide by 4 * world size")
        reduce_scatter_allgather(stamps, fold_stamps, group)
        if fused is not None:
            reduce_scatter_allgather(fused, fold_counts, group)
    else:
        allreduce_tensors(stamps, fused, group)
    mapper.mark_fused()
    if fused is not None:
        mapper.counts_source(True)
    return fused


# ---- sparse fuse: dirty blocks only ------------------------------------------------------------------------------------
class MapperSparseAdapter:
    """The three device-side steps of a sparse fuse (include/quasar_slam.h: qs_sparse_fuse_*) as torch tensors aliasing the
    context's buffers.  The CPU tests drive sparse_fuse() with a numpy adapter of the same shape."""

    def __init__(self, mapper, device, same_stream=True):
        """same_stream: the context runs on torch's current stream (collectives and kernels are then ordered by the stream);
        otherwise every hand-over between the two streams is a host synchronisation."""
        self.m, self.device, self.same_stream = mapper, device, same_stream

    def _torch_sync(self):
        import torch
        if not self.same_stream and self.device.type == "cuda":
            torch.cuda.current_stream().synchronize()

    def begin(self, world, rank):
        import torch
        p, nb = self.m.sparse_fuse_begin(world, rank)
        if not self.same_stream:
            self.m.sync()
        return torch.as_tensor(_DevArray(p, (world, nb // 4), "<i4"), device=self.device)

    def plan(self, world):
        import torch
        self._torch_sync()                                      # the bitmaps have arrived
        n, off, p, bb = self.m.sparse_fuse_plan(world)          # waits for the context's stream: the host needs the counts
        if not self.same_stream:
            self.m.sync()                                       # this rank's segment is packed
        total = int(off[world])
        buf = torch.as_tensor(_DevArray(p, (total,), "|u1"), device=self.device) if total else None
        return n, off, buf, bb

    def apply(self):
        self._torch_sync()                                      # the peers' segments have arrived
        self.m.sparse_fuse_apply()


def all_gather_rows(t, rank, group=None):
    """t: [world, n]; row `rank` is this rank's: after the call every row holds its rank's."""
    import torch
    import torch.distributed as dist
    W = t.shape[0]
    if W == 1:
        return
    if t.is_cuda and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(t.view(-1), t[rank], group=group)       # in place: input is slice `rank` of the output
    else:
        parts = [torch.empty_like(t[rank]) for _ in range(W)]
        dist.all_gather(parts, t[rank].clone(), group=group)
        for p in range(W):
            if p != rank:
                t[p].copy_(parts[p])


def exchange_segments(buf, off, rank, world, group=None):
    """buf: flat byte tensor holding one segment per rank at [off[p], off[p+1]); this rank's is filled.  Sends it to every
    peer and receives every peer's into its place: N-1 sends + N-1 receives in one batch (RCCL runs them concurrently:
    point to point over xGMI, one link per peer).  Empty segments are skipped on both sides (every rank knows all sizes)."""
    import torch.distributed as dist
    mine = buf[int(off[rank]):int(off[rank + 1])]
    ops = []
    for p in range(world):
        if p == rank:
            continue
        gp = dist.get_global_rank(group, p) if group is not None else p
        if mine.numel():
            ops.append(dist.P2POp(dist.isend, mine, gp, group))
        if off[p + 1] > off[p]:
            ops.append(dist.P2POp(dist.irecv, buf[int(off[p]):int(off[p + 1])], gp, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def sparse_fuse(adapter, rank, world, group=None, stats=None):
    """One sparse fuse over the ranks of `group`.  stats (dict, optional) receives the bytes this rank moved."""
    bm = adapter.begin(world, rank)
    all_gather_rows(bm, rank, group)
    n, off, buf, bb = adapter.plan(world)
    if world > 1 and buf is not None:
        exchange_segments(buf, off, rank, world, group)
    adapter.apply()
    if stats is not None:
        own = int(off[rank + 1] - off[rank])
        stats.update(block_bytes=int(bb), blocks_own=int(n[rank]), blocks_all=int(n.sum()),
                     payload_bytes=own,                                   # this rank's packed blocks = what each of its links carries
                     sent_bytes=own * (world - 1) + (world - 1) * bm.shape[1] * 4,
                     received_bytes=int(off[world]) - own + (world - 1) * bm.shape[1] * 4,
                     bitmap_bytes=int(bm.shape[1] * 4))
    return n


def sparse_fuse_local(mappers, device):
    """The same protocol among N contexts of ONE process on one GPU (the contexts play the ranks; device-to-device copies
    play the collectives): configs[2]'s on-GPU merge of per-bot-group mappers, and how the tests rehearse N = 8."""
    import torch
    W = len(mappers)
    ads = [MapperSparseAdapter(m, device) for m in mappers]
    for m in mappers:
        m.sync()
    bms = [a.begin(W, r) for r, a in enumerate(ads)]
    for m in mappers:
        m.sync()
    for r in range(W):
        for p in range(W):
            if p != r:
                bms[r][p].copy_(bms[p][p])
    torch.cuda.synchronize()
    plans = [a.plan(W) for a in ads]
    for m in mappers:
        m.sync()
    for r in range(W):
        n, off, buf, _ = plans[r]
        for p in range(W):
            if p != r and off[p + 1] > off[p]:
                buf[int(off[p]):int(off[p + 1])].copy_(plans[p][2][int(off[p]):int(off[p + 1])])
    torch.cuda.synchronize()
    for a in ads:
        a.apply()
    for m in mappers:
        m.sync()
    return [pl[0] for pl in plans][0]


def fused_counts_view(mapper, device):
    """int32 [size, size, 2] aliasing the fused counters as they stand (after a sparse fuse: the sum over the ranks)."""
    import torch
    p, _ = mapper.fused_counts_buffer()
    if not p:
        return None
    n = mapper.size
    return torch.as_tensor(_DevArray(p, (n, n, 2), "<i4"), device=device)


class ShardedMapper:
    """One rank of an N-way deployment of the central mapper (one process per GPU).

    per_shard:   mapper created with seq_stride = world; record i of this rank's batch is global record
                 seq_base + i*world + rank (N equal streams interleaved round-robin).
    replicated:  mapper created with shard_bots / shard_rank, seq_stride = 1, max_agent = all bots (<= 255: agent_id is
                 one byte on the wire and the global pose graph needs globally unique ids); the batch is all-gathered
                 and every rank ingests the interleaved whole.
    """

    def __init__(self, mapper, device, rank, world, group=None, mode="per_shard", fuse="sparse", same_stream=True,
                 track_single_rank=False):
        if mode not in ("per_shard", "replicated"):
            raise ValueError("mode must be per_shard or replicated")
        if fuse not in ("sparse", "allreduce", "direct"):
            raise ValueError("fuse must be sparse, allreduce or direct")
        if mode == "per_shard" and world > 1 and mapper.cfg.seq_stride != world:
            raise ValueError("per_shard mode: create the mapper with seq_stride = world size")
        if mode == "replicated" and (mapper.cfg.seq_stride not in (0, 1) or (world > 1 and mapper.cfg.shard_bots <= 0)):
            raise ValueError("replicated mode: create the mapper with seq_stride = 1 and shard_bots > 0")
        self.m, self.device, self.rank, self.world, self.group = mapper, device, rank, world, group
        self.mode, self.fuse_algo, self.same_stream = mode, fuse, same_stream
        self._gather = None
        self.fuse_stats = {}
        self._sparse = None
        if fuse == "sparse" and (world > 1 or track_single_rank):
            mapper.dirty_tracking(True)           # every writer of the grid marks the blocks it touches from here on
            self._sparse = MapperSparseAdapter(mapper, device, same_stream)

    def ingest(self, d_pkts, d_time=None, seq_base=0):
        """d_pkts: uint8 [B, stride] device tensor (this rank's batch); d_time: float64 [B] or None;
        seq_base: global arrival index of record 0 of rank 0's batch.  The work is enqueued on the context's stream and the
        call returns without waiting for the GPU (exact-trig edge rays wait on the device until the map is next observed:
        a fuse, a grid read, sync)."""
        import torch
        import torch.distributed as dist
        B, stride = d_pkts.shape
        if self.mode == "per_shard" or self.world == 1:
            seq0 = seq_base + (self.rank if self.mode == "per_shard" else 0)
            if self.world > 1 and self.m.epoch_would_rebase(B, seq0):
                self.fuse(counts=False)          # the shards exchange their stamps before the grid is rebased
            self.m.ingest_device(d_pkts.data_ptr(), B, stride, 0, d_time.data_ptr() if d_time is not None else 0, seq0=seq0)
            return B
        W = self.world
        if self._gather is None or self._gather[0].shape != (W, B, stride):
            self._gather = (torch.empty((W, B, stride), dtype=torch.uint8, device=d_pkts.device),
                            torch.empty((W, B), dtype=torch.float64, device=d_pkts.device))
        g_p, g_t = self._gather
        dist.all_gather_into_tensor(g_p.view(-1), d_pkts.contiguous().view(-1), group=self.group)
        full = g_p.permute(1, 0, 2).contiguous().view(W * B, stride)       # round-robin interleave: record i*W + r
        t_ptr = 0
        if d_time is not None:
            dist.all_gather_into_tensor(g_t.view(-1), d_time.contiguous().view(-1), group=self.group)
            full_t = g_t.permute(1, 0).contiguous().view(-1)
            t_ptr = full_t.data_ptr()
            self._keep = (full, full_t)
        else:
            self._keep = (full,)
        if not self.same_stream:
            torch.cuda.current_stream().synchronize()
        if self.m.epoch_would_rebase(W * B, seq_base):
            self.fuse(counts=False)              # replicated shards own different agents' rays: exchange before the rebase too
        self.m.ingest_device(full.data_ptr(), W * B, stride, 0, t_ptr, seq0=seq_base)
        return W * B

    def fuse(self, counts=True):
        """Fuse the ranks' grids.  Returns the tensor aliasing the fused counters (or None).  The sparse fuse always
        carries the counter deltas of the blocks it moves (`counts` only matters to the dense algorithms)."""
        if self._sparse is not None:
            if not self.same_stream:
                self.m.sync()
            sparse_fuse(self._sparse, self.rank, self.world, self.group, stats=self.fuse_stats)
            return fused_counts_view(self.m, self.device) if self.m.cfg.enable_counts else None
        if self.world == 1:
            self.m.mark_fused()
            return None
        cells = self.m.size * self.m.size
        dense = cells * 4 + (cells * 8 if (counts and self.m.cfg.enable_counts) else 0)
        self.fuse_stats.update(payload_bytes=dense, sent_bytes=int(2 * (self.world - 1) / self.world * dense),
                               received_bytes=int(2 * (self.world - 1) / self.world * dense))
        return allreduce_grids(self.m, self.device, self.group, counts=counts, algo=self.fuse_algo,
                               sync=not self.same_stream)


def tri_state_from_stamps(stamps):
    """Host decode of a stamp grid to OccupancyGrid values (-1 / 0 / 100), for checks."""
    s = np.asarray(stamps).astype(np.int64)
    return np.where(s == 0, -1, np.where(s & 1, 100, 0)).astype(np.int8)


def shard_of_bot(bot_index, bots_per_gpu):
    """global bot index (0-based) -> (rank, agent_id on that rank's wire, 1-based)."""
    return bot_index // bots_per_gpu, bot_index % bots_per_gpu + 1


def rank_sequence(rank, world, step_base, n):
    """Global arrival indices of rank `rank`'s n records when `world` equal streams are
    interleaved round-robin: seq = step_base + i*world + rank."""
    return step_base + np.arange(n, dtype=np.uint64) * np.uint64(world) + np.uint64(rank)
