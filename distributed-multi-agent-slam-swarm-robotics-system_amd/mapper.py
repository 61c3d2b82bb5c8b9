"""Host-side mirror of the reference mapper's object API over the HIP library.

`QuasarMapper` is the batched equivalent of dual_bot_mapper.py's main() recv loop
(:815-945): it owns one GPU context and exposes look-alikes of the objects the reference's
renderer and timers read -- `OccupancyGrid` (:110-237), `PoseGraphSLAM` (:261-338),
`drift_correction`, `zone_boxes`.  All arithmetic happens in the HIP kernels; this file only
marshals buffers (numpy <-> C ABI).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import QsConfig, QuasarError, UINT64_MAX, check
from . import protocol as P


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class QuasarMapper:
    """One mapper instance on one GPU.  Defaults are the reference's constants."""

    def __init__(self, size=P.GRID_SIZE, resolution=P.GRID_RESOLUTION, origin_x=P.GRID_ORIGIN_X,
                 origin_y=P.GRID_ORIGIN_Y, separation=0.0, max_agent=2, bots_per_graph=0,
                 enable_counts=True, enable_ekf=False, device=0, raycast_mode=0,
                 ekf_metres_per_tick=0.0107, min_poses_between=P.MIN_POSES_BETWEEN,
                 closure_radius=P.CLOSURE_RADIUS, closure_correction=P.CLOSURE_CORRECTION,
                 seq_stride=1, shard_bots=0, shard_rank=0, exact_trig=True):
        self._L = _lib.load()
        cfg = QsConfig()
        check(None, self._L.qs_config_default(C.byref(cfg)), "qs_config_default")
        cfg.size, cfg.res, cfg.ox, cfg.oy = size, resolution, origin_x, origin_y
        cfg.separation = separation
        cfg.max_agent, cfg.bots_per_graph = max_agent, bots_per_graph
        cfg.enable_counts, cfg.enable_ekf = int(enable_counts), int(enable_ekf)
        cfg.device, cfg.raycast_mode = device, raycast_mode
        cfg.ekf_metres_per_tick = ekf_metres_per_tick
        cfg.min_poses_between = min_poses_between
        cfg.closure_radius, cfg.closure_correction = closure_radius, closure_correction
        cfg.seq_stride = seq_stride
        cfg.shard_bots, cfg.shard_rank = shard_bots, shard_rank
        cfg.exact_trig = int(bool(exact_trig))
        self.cfg = cfg
        self.size, self.res, self.ox, self.oy = size, resolution, origin_x, origin_y
        self.max_agent = max_agent
        h = C.c_void_p()
        rc = self._L.qs_create(C.byref(cfg), C.byref(h))      # validates every field
        if rc != 0:
            raise QuasarError(f"qs_create failed ({rc}): {self._L.qs_last_error(None).decode()}")
        self._h = h
        self.bots_per_graph = bots_per_graph if bots_per_graph > 0 else max_agent
        self.n_graphs = (max_agent + self.bots_per_graph - 1) // self.bots_per_graph
        self._last_n = 0
        self._map_version = 0          # bumped by everything that can change a cell: the look-alike's .grid cache key
        self.occ_grid = OccupancyGrid._attached(self)
        self.slam = PoseGraphSLAM._attached(self)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.qs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc, what):
        check(self._h, rc, what)

    def reset(self):
        self._chk(self._L.qs_reset(self._h), "qs_reset")
        self._map_version += 1

    def sync(self):
        self._chk(self._L.qs_sync(self._h), "qs_sync")

    def set_stream(self, hip_stream_ptr):
        self._chk(self._L.qs_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "qs_set_stream")

    def set_bot_offset(self, bot, off_x):
        self._chk(self._L.qs_set_bot_offset(self._h, bot, off_x), "qs_set_bot_offset")

    # -- ingest: dual_bot_mapper.py:826-919 ---------------------------------------------------
    def ingest(self, datagrams, recv_time=None, seq0=None):
        """datagrams: list of bytes objects (any lengths), in arrival order."""
        if len(datagrams) == 0:
            return 0
        if all(len(d) == P.PACKET_SIZE for d in datagrams):
            buf = np.frombuffer(b"".join(datagrams), dtype=np.uint8).reshape(-1, P.PACKET_SIZE)
            return self.ingest_array(buf, None, recv_time, seq0)
        buf, lens = P.pack_datagrams(datagrams)
        return self.ingest_array(buf, lens, recv_time, seq0)

    def ingest_array(self, buf, lengths=None, recv_time=None, seq0=None):
        """buf: uint8 [n, stride] host array; lengths: uint16 [n] or None."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.ndim != 2:
            raise ValueError("buf must be [n, stride]")
        n, stride = buf.shape
        lens = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.uint16)
        t = None if recv_time is None else np.ascontiguousarray(recv_time, dtype=np.float64)
        if lens is not None and len(lens) != n:
            raise ValueError("lengths must have one entry per record")
        if t is not None and len(t) != n:
            raise ValueError("recv_time must have one entry per record")
        self._chk(self._L.qs_ingest(self._h, _ptr(buf), n, stride, _ptr(lens), _ptr(t),
                                    UINT64_MAX if seq0 is None else int(seq0)), "qs_ingest")
        self._map_version += 1
        self._last_n = n
        return n

    def ingest_device(self, d_pkts, n, stride, d_lens=0, d_time=0, seq0=None):
        """Device-resident input (raw device addresses as ints); asynchronous."""
        self._chk(self._L.qs_ingest_device(self._h, C.c_void_p(d_pkts), n, stride,
                                           C.c_void_p(d_lens) if d_lens else None,
                                           C.c_void_p(d_time) if d_time else None,
                                           UINT64_MAX if seq0 is None else int(seq0)), "qs_ingest_device")
        self._map_version += 1
        self._last_n = n

    def last_batch(self):
        """(accepted uint8 [n], pose float64 [n,3]) of the last ingest (rx, ry, ryaw; :850-857)."""
        n = self._last_n
        acc = np.zeros(n, dtype=np.uint8)
        pose = np.zeros((n, 3), dtype=np.float64)
        self._chk(self._L.qs_last_batch(self._h, _ptr(acc), _ptr(pose), n), "qs_last_batch")
        return acc, pose

    def last_hits(self):
        """(xy float64 [n,4,2], valid uint8 [n,4]) of the last ingest: point_clouds appends (:892)."""
        n = self._last_n
        xy = np.zeros((n, 4, 2), dtype=np.float64)
        valid = np.zeros((n, 4), dtype=np.uint8)
        self._chk(self._L.qs_last_hits(self._h, _ptr(xy), _ptr(valid), n), "qs_last_hits")
        return xy, valid

    # -- grid ---------------------------------------------------------------------------------
    def grid_i8(self):
        out = np.empty((self.size, self.size), dtype=np.int8)
        self._chk(self._L.qs_grid_i8(self._h, _ptr(out)), "qs_grid_i8")
        return out

    def grid_i8_device(self, d_out):
        self._chk(self._L.qs_grid_i8_device(self._h, C.c_void_p(d_out)), "qs_grid_i8_device")

    def counts(self):
        hits = np.empty((self.size, self.size), dtype=np.int32)
        misses = np.empty((self.size, self.size), dtype=np.int32)
        self._chk(self._L.qs_grid_counts(self._h, _ptr(hits), _ptr(misses)), "qs_grid_counts")
        return hits, misses

    def logodds(self, l_occ=0.85, l_free=0.4, lmin=-2.0, lmax=3.5):
        out = np.empty((self.size, self.size), dtype=np.float32)
        self._chk(self._L.qs_grid_logodds(self._h, l_occ, l_free, lmin, lmax, _ptr(out)), "qs_grid_logodds")
        return out

    def update_rays(self, rx, ry, hx, hy, valid, seq0=None):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (rx, ry, hx, hy)]
        v = np.ascontiguousarray(valid, dtype=np.uint8)
        self._chk(self._L.qs_update_rays(self._h, *[_ptr(x) for x in a], _ptr(v), len(v),
                                         UINT64_MAX if seq0 is None else int(seq0)), "qs_update_rays")
        self._map_version += 1

    def world_to_grid(self, w, axis=0):
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.empty(len(w), dtype=np.int64)
        self._chk(self._L.qs_world_to_grid(self._h, _ptr(w), len(w), axis, _ptr(out)), "qs_world_to_grid")
        return out

    def device_buffers(self):
        """(stamps_ptr, stamps_bytes, counts_ptr, counts_bytes) raw device addresses.  Whoever gets them may write the
        grid (a collective): the look-alike's cached .grid is dropped."""
        self._map_version += 1
        sp, cp = C.c_void_p(), C.c_void_p()
        sb, cb = C.c_size_t(), C.c_size_t()
        self._chk(self._L.qs_device_buffers(self._h, C.byref(sp), C.byref(sb), C.byref(cp), C.byref(cb)),
                  "qs_device_buffers")
        return sp.value, sb.value, cp.value or 0, cb.value

    # -- SLAM -----------------------------------------------------------------------------------
    def slam_sizes(self, graph=0):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self._L.qs_slam_sizes(self._h, graph, C.byref(a), C.byref(b), C.byref(c)), "qs_slam_sizes")
        return a.value, b.value, c.value

    def closure_agents(self, graph=0):
        """agent_id of each closure's closing node (self.nodes[node_idx].agent_id in the reference, :335)."""
        n = self.slam_sizes(graph)[2]
        out = np.zeros(n, dtype=np.uint8)
        if n:
            self._chk(self._L.qs_slam_closure_agents(self._h, graph, _ptr(out), n), "qs_slam_closure_agents")
        return out

    def closures(self, graph=0):
        n = self.slam_sizes(graph)[2]
        idx = np.zeros((n, 2), dtype=np.int64)
        corr = np.zeros((n, 2), dtype=np.float64)
        if n:
            self._chk(self._L.qs_slam_closures(self._h, graph, _ptr(idx), _ptr(corr), n), "qs_slam_closures")
        return idx, corr

    def landmarks(self, graph=0):
        n = self.slam_sizes(graph)[1]
        xy = np.zeros((n, 2), dtype=np.float64)
        ti = np.zeros((n, 2), dtype=np.int64)
        if n:
            self._chk(self._L.qs_slam_landmarks(self._h, graph, _ptr(xy), _ptr(ti), n), "qs_slam_landmarks")
        return xy, ti

    def slam_add_poses(self, x, y, agent, landmark):
        """Batched PoseGraphSLAM.add_pose on poses as given -> (closed uint8 [n], corr float64 [n, 2])."""
        xs = np.ascontiguousarray(x, dtype=np.float64); ys = np.ascontiguousarray(y, dtype=np.float64)
        ag = np.ascontiguousarray(agent, dtype=np.uint8); lm = np.ascontiguousarray(landmark, dtype=np.uint8)
        n = len(xs)
        closed = np.zeros(n, dtype=np.uint8); corr = np.zeros((n, 2), dtype=np.float64)
        self._chk(self._L.qs_slam_add_poses(self._h, _ptr(xs), _ptr(ys), _ptr(ag), _ptr(lm), n, _ptr(closed), _ptr(corr)),
                  "qs_slam_add_poses")
        return closed, corr

    def drift(self, bot):
        out = np.zeros(2, dtype=np.float64)
        self._chk(self._L.qs_drift(self._h, bot, _ptr(out)), "qs_drift")
        return out

    @property
    def drift_correction(self):
        """dict bot -> (dx, dy), as main()'s drift_correction (:782)."""
        return {b: tuple(self.drift(b)) for b in range(1, self.max_agent + 1)}

    # -- ZONE -------------------------------------------------------------------------------------
    def zone(self, bot):
        out = np.zeros(4, dtype=np.float64)
        valid = C.c_int32()
        self._chk(self._L.qs_zone(self._h, bot, _ptr(out), C.byref(valid)), "qs_zone")
        return tuple(out) if valid.value else None

    def zone_packet(self, bot, online=True) -> bytes:
        out = np.zeros(P.ZONE_SIZE, dtype=np.uint8)
        self._chk(self._L.qs_zone_packet(self._h, bot, int(online), _ptr(out)), "qs_zone_packet")
        return out.tobytes()

    @property
    def zone_boxes(self):
        return {b: self.zone(b) for b in range(1, self.max_agent + 1)}

    # -- merge ------------------------------------------------------------------------------------
    def fuse(self, others):
        arr = (C.c_void_p * len(others))(*[o._h for o in others])
        self._chk(self._L.qs_fuse(self._h, arr, len(others)), "qs_fuse")
        self._map_version += 1

    def fuse_buffers(self, stamp_ptrs, count_ptrs=None):
        n = len(stamp_ptrs)
        sp = (C.c_void_p * n)(*stamp_ptrs)
        cp = (C.c_void_p * n)(*count_ptrs) if count_ptrs else None
        self._chk(self._L.qs_fuse_buffers(self._h, sp, cp, n), "qs_fuse_buffers")
        self._map_version += 1

    def fuse_buffers_range(self, stamp_ptrs, count_ptrs, cell_offset, n_cells, counts_into_fused=False):
        """Fold peers' copies of cells [cell_offset, cell_offset + n_cells) into this grid (raw device addresses of the
        first cell of the range; either list may be None)."""
        n = len(stamp_ptrs if stamp_ptrs else count_ptrs)
        sp = (C.c_void_p * n)(*stamp_ptrs) if stamp_ptrs else None
        cp = (C.c_void_p * n)(*count_ptrs) if count_ptrs else None
        self._chk(self._L.qs_fuse_buffers_range(self._h, sp, cp, n, cell_offset, n_cells, int(counts_into_fused)),
                  "qs_fuse_buffers_range")
        self._map_version += 1

    def fused_counts(self):
        """Snapshot the local counters into the context's second buffer -> (device address, bytes); a collective sums
        that buffer over the ranks (never the local counters themselves)."""
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self._L.qs_fused_counts(self._h, C.byref(p), C.byref(b)), "qs_fused_counts")
        return p.value, b.value

    # -- the same fuse with RCCL behind the C ABI (hosts without torch; csrc/rccl_fuse.hip) -------------------------------
    @staticmethod
    def rccl_unique_id():
        out = np.zeros(128, dtype=np.uint8)
        check(None, _lib.load().qs_rccl_unique_id(_ptr(out)), "qs_rccl_unique_id")
        return out

    def rccl_comm_init(self, unique_id, world, rank):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        comm = C.c_void_p()
        self._chk(self._L.qs_rccl_comm_init(self._h, _ptr(uid), world, rank, C.byref(comm)), "qs_rccl_comm_init")
        return comm

    def rccl_comm_destroy(self, comm):
        self._chk(self._L.qs_rccl_comm_destroy(comm), "qs_rccl_comm_destroy")

    def sparse_fuse_rccl(self, comm, world, rank):
        """One sparse fuse over RCCL -> dict(blocks_own, payload_bytes, sent_bytes, received_bytes)."""
        st = np.zeros(4, dtype=np.uint64)
        self._chk(self._L.qs_sparse_fuse_rccl(self._h, comm, world, rank, _ptr(st)), "qs_sparse_fuse_rccl")
        self._map_version += 1
        return dict(zip(("blocks_own", "payload_bytes", "sent_bytes", "received_bytes"), (int(v) for v in st)))

    def fused_counts_buffer(self):
        """(device address, bytes) of the fused counters as they stand (no snapshot); address 0 before the first fuse."""
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self._L.qs_fused_counts_buffer(self._h, C.byref(p), C.byref(b)), "qs_fused_counts_buffer")
        return p.value or 0, b.value

    def counts_source(self, fused):
        self._chk(self._L.qs_counts_source(self._h, int(bool(fused))), "qs_counts_source")

    def epoch_would_rebase(self, n, seq0=None):
        w = C.c_int32()
        self._chk(self._L.qs_epoch_query(self._h, UINT64_MAX if seq0 is None else int(seq0), n, C.byref(w)), "qs_epoch_query")
        return bool(w.value)

    def mark_fused(self):
        """Record that the shards have exchanged their stamps.  Whoever fuses into the device buffers from outside (a
        collective on dist.grid_tensors) calls this afterwards: it also drops the look-alike's cached .grid."""
        self._chk(self._L.qs_mark_fused(self._h), "qs_mark_fused")
        self._map_version += 1

    # -- sparse fuse: only the blocks written since the last fuse travel (include/quasar_slam.h) -----------------------
    def dirty_tracking(self, on=True):
        self._chk(self._L.qs_dirty_tracking(self._h, int(bool(on))), "qs_dirty_tracking")

    def dirty_blocks(self):
        """(blocks marked since the last sparse fuse, cells per block)."""
        n, cells = C.c_size_t(), C.c_size_t()
        self._chk(self._L.qs_dirty_blocks(self._h, C.byref(n), C.byref(cells)), "qs_dirty_blocks")
        return n.value, cells.value

    def sparse_fuse_begin(self, world, rank):
        """-> (device address of the [world][bitmap_bytes] bitmap array, bitmap_bytes); slot `rank` holds this rank's."""
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self._L.qs_sparse_fuse_begin(self._h, world, rank, C.byref(p), C.byref(b)), "qs_sparse_fuse_begin")
        return p.value, b.value

    def sparse_fuse_plan(self, world):
        """-> (n_blocks uint32 [world], offsets [world + 1] bytes, payload device address, block_bytes)."""
        n = np.zeros(world, dtype=np.uint32)
        off = np.zeros(world + 1, dtype=np.uintp)
        p, bb = C.c_void_p(), C.c_size_t()
        self._chk(self._L.qs_sparse_fuse_plan(self._h, _ptr(n), _ptr(off), C.byref(p), C.byref(bb)), "qs_sparse_fuse_plan")
        return n, off.astype(np.int64), p.value or 0, bb.value

    def sparse_fuse_apply(self):
        self._chk(self._L.qs_sparse_fuse_apply(self._h), "qs_sparse_fuse_apply")
        self._map_version += 1

    def grid_to_pcd(self, grid, res, ox, oy):
        """MapMerger.grid_to_pcd (map_merger.py:64-85) -> float64 [n,2] (x, y)."""
        grid = np.ascontiguousarray(grid, dtype=np.int8)
        h, w = grid.shape
        n = C.c_size_t()
        self._chk(self._L.qs_grid_to_pcd(self._h, _ptr(grid), h, w, res, ox, oy, None, 0, C.byref(n)),
                  "qs_grid_to_pcd")
        xy = np.zeros((n.value, 2), dtype=np.float64)
        if n.value:
            self._chk(self._L.qs_grid_to_pcd(self._h, _ptr(grid), h, w, res, ox, oy, _ptr(xy), n.value,
                                             C.byref(n)), "qs_grid_to_pcd")
        return xy

    def rasterise(self, xy, res):
        """MapMerger.publish_global_map (map_merger.py:87-127) -> (int8 grid, (min_x, min_y))."""
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        dims = np.zeros(2, dtype=np.int32)
        origin = np.zeros(2, dtype=np.float64)
        self._chk(self._L.qs_rasterise(self._h, _ptr(xy), len(xy), res, _ptr(dims), _ptr(origin), None),
                  "qs_rasterise")
        if len(xy) == 0:
            return None, None
        grid = np.empty((int(dims[0]), int(dims[1])), dtype=np.int8)
        self._chk(self._L.qs_rasterise(self._h, _ptr(xy), len(xy), res, _ptr(dims), _ptr(origin), _ptr(grid)),
                  "qs_rasterise")
        return grid, origin

    # -- ICP / voxel down-sample: map_merger.py:45-60 (Open3D semantics, parity unpinned) ---------
    def icp(self, src_xy, dst_xy, max_dist=1.0, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        """registration_icp(source, target, max_dist, I, PointToPoint, max_iteration) on planar clouds.
        Returns (T 3x3, fitness, inlier_rmse, iterations)."""
        a = np.ascontiguousarray(src_xy, dtype=np.float64); b = np.ascontiguousarray(dst_xy, dtype=np.float64)
        T = np.zeros(9, dtype=np.float64)
        fit, rm, it = C.c_double(), C.c_double(), C.c_int32()
        self._chk(self._L.qs_icp(self._h, _ptr(a), len(a), _ptr(b), len(b), max_dist, max_iter, rel_fitness, rel_rmse,
                                 _ptr(T), C.byref(fit), C.byref(rm), C.byref(it)), "qs_icp")
        return T.reshape(3, 3), fit.value, rm.value, it.value

    def nn_search(self, src_xy, dst_xy, max_dist=1.0, mode=0):
        """Nearest target of every source point (the correspondence step of registration_icp): (corr int32 [n], d2 float64 [n],
        (search_ms, prep_ms)).  mode 0 auto, 1 scalar fp64, 2 MFMA-screened; results are identical."""
        a = np.ascontiguousarray(src_xy, dtype=np.float64); b = np.ascontiguousarray(dst_xy, dtype=np.float64)
        corr = np.empty(len(a), dtype=np.int32); d2 = np.empty(len(a), dtype=np.float64)
        ms = np.zeros(2, dtype=np.float32)
        self._chk(self._L.qs_nn_search(self._h, _ptr(a), len(a), _ptr(b), len(b), max_dist, mode, _ptr(corr), _ptr(d2), _ptr(ms)),
                  "qs_nn_search")
        return corr, d2, (float(ms[0]), float(ms[1]))

    CHAIN_FORMS = {"auto": 0, "free": 1, "window": 2, "free_posting": 3}

    def set_chain_form(self, form):
        """Which device form of the loop-closure chain runs: "auto" (default), "free" (free-running), "free_posting" (free-running,
        the owner waves post their landmarks' poses for each other), "window" (one barrier per window).  Same closures,
        landmarks and drifts whichever (dual_bot_mapper.py:292-326)."""
        self._chk(self._L.qs_set_chain_form(self._h, self.CHAIN_FORMS[form]), "qs_set_chain_form")

    def chain_form(self):
        """The form the last ingest used: "free", "free_posting" or "window"."""
        return {1: "free", 2: "window", 3: "free_posting"}[self._L.qs_chain_form(self._h)]

    def mfma_f64_rate(self):
        """Measured dense fp64 MFMA rate of this GPU in TFLOP/s (diagnostic)."""
        v = C.c_double()
        self._chk(self._L.qs_diag_mfma_f64_rate(self._h, C.byref(v)), "qs_diag_mfma_f64_rate")
        return v.value

    def diag_latencies(self):
        """Measured latencies (shader-clock cycles) of the primitives a loop-closure decision chains together, by one
        workgroup on this GPU: dict with l2_load, l1_load, lds_read, fma_f64, dpp_step, readlane_step, barrier_16_waves,
        barrier_5_waves, clock_mhz."""
        out = np.zeros(9, dtype=np.float64)
        self._chk(self._L.qs_diag_latencies(self._h, _ptr(out)), "qs_diag_latencies")
        names = ("l2_load", "l1_load", "lds_read", "fma_f64", "dpp_step", "readlane_step", "barrier_16_waves", "barrier_5_waves", "clock_mhz")
        return dict(zip(names, (float(v) for v in out)))

    def voxel_downsample(self, xy, voxel):
        a = np.ascontiguousarray(xy, dtype=np.float64)
        n = C.c_size_t()
        self._chk(self._L.qs_voxel_downsample(self._h, _ptr(a), len(a), voxel, None, 0, C.byref(n)), "qs_voxel_downsample")
        out = np.zeros((n.value, 2), dtype=np.float64)
        if n.value:
            self._chk(self._L.qs_voxel_downsample(self._h, _ptr(a), len(a), voxel, _ptr(out), n.value, C.byref(n)),
                      "qs_voxel_downsample")
        return out

    # -- frontiers: dual_bot_mapper.py:181-237, :948-956 ----------------------------------------
    def frontier_cells(self):
        """OccupancyGrid.get_frontiers() -> int32 [n, 2] (gx, gy), row-major order."""
        n = C.c_size_t()
        self._chk(self._L.qs_frontier_cells(self._h, None, 0, C.byref(n)), "qs_frontier_cells")
        xy = np.zeros((n.value, 2), dtype=np.int32)
        if n.value:
            self._chk(self._L.qs_frontier_cells(self._h, _ptr(xy), n.value, C.byref(n)), "qs_frontier_cells")
        return xy

    def frontier_clusters(self, min_cluster=3):
        """cluster_frontiers() as int64 [k, 5]: size, first_gx, first_gy, sum_gx, sum_gy, in the
        reference's cluster order."""
        n = C.c_size_t()
        self._chk(self._L.qs_frontier_clusters(self._h, min_cluster, None, 0, C.byref(n)), "qs_frontier_clusters")
        st = np.zeros((n.value, 5), dtype=np.int64)
        if n.value:
            self._chk(self._L.qs_frontier_clusters(self._h, min_cluster, _ptr(st), n.value, C.byref(n)),
                      "qs_frontier_clusters")
        return st

    def frontier_members(self):
        """int32 [n, 3]: every frontier cell (gx, gy) in row-major order with the linear index of the first cell of its
        4-connected cluster (clusters labelled on the device)."""
        n = C.c_size_t()
        self._chk(self._L.qs_frontier_members(self._h, None, 0, C.byref(n)), "qs_frontier_members")
        out = np.zeros((n.value, 3), dtype=np.int32)
        if n.value:
            self._chk(self._L.qs_frontier_members(self._h, _ptr(out), n.value, C.byref(n)), "qs_frontier_members")
        return out

    def frontier_centroids(self, min_cluster=3):
        """[cluster_centroid_world(c) for c in clusters] (:955): mean cell index by true division,
        then grid_to_world (:127-131, cell centre)."""
        out = []
        for size, _, _, sx, sy in self.frontier_clusters(min_cluster).tolist():
            ax, ay = sx / size, sy / size
            out.append((self.ox + (ax + 0.5) * self.res, self.oy + (ay + 0.5) * self.res))
        return out

    # -- EKF --------------------------------------------------------------------------------------
    def ekf_init(self, bot, t, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self._chk(self._L.qs_ekf_init(self._h, bot, t, _ptr(x0)), "qs_ekf_init")

    def ekf_step(self, bots, omega_m, t, z_v=None, z_omega=None):
        b = np.ascontiguousarray(bots, dtype=np.int32)
        om = np.ascontiguousarray(omega_m, dtype=np.float64)
        tt = np.ascontiguousarray(t, dtype=np.float64)
        upd = z_v is not None
        zv = np.ascontiguousarray(z_v, dtype=np.float64) if upd else None
        zo = np.ascontiguousarray(z_omega, dtype=np.float64) if upd else None
        self._chk(self._L.qs_ekf_step(self._h, _ptr(b), _ptr(om), _ptr(tt), _ptr(zv), _ptr(zo), len(b), int(upd)),
                  "qs_ekf_step")

    def ekf_state(self, bot):
        x = np.zeros(6, dtype=np.float64)
        Pm = np.zeros((6, 6), dtype=np.float64)
        self._chk(self._L.qs_ekf_state(self._h, bot, _ptr(x), _ptr(Pm)), "qs_ekf_state")
        return x, Pm

    # -- counters / timing ------------------------------------------------------------------------
    def counters(self):
        out = np.zeros(len(_lib.QS_CNT_NAMES), dtype=np.uint64)
        self._chk(self._L.qs_counters(self._h, _ptr(out)), "qs_counters")
        return dict(zip(_lib.QS_CNT_NAMES, (int(v) for v in out)))

    def timing_enable(self, on=True):
        self._chk(self._L.qs_timing_enable(self._h, int(on)), "qs_timing_enable")

    def stage_times(self, reset=True):
        ms = np.zeros(len(_lib.QS_STAGE_NAMES), dtype=np.float64)
        ln = np.zeros(len(_lib.QS_STAGE_NAMES), dtype=np.uint64)
        self._chk(self._L.qs_stage_times(self._h, _ptr(ms), _ptr(ln), int(reset)), "qs_stage_times")
        return {k: (float(m), int(c)) for k, m, c in zip(_lib.QS_STAGE_NAMES, ms, ln)}


class OccupancyGrid:
    """Look-alike of dual_bot_mapper.py::OccupancyGrid (:110-237) backed by the device grid.
    `.grid` is synchronised from the GPU on access; attribute names match what MapRenderer
    reads (:494-512)."""

    def __init__(self, size=P.GRID_SIZE, resolution=P.GRID_RESOLUTION,
                 origin_x=P.GRID_ORIGIN_X, origin_y=P.GRID_ORIGIN_Y, device=0):
        m = QuasarMapper(size, resolution, origin_x, origin_y, device=device)
        self._m = m
        self.size, self.res, self.ox, self.oy = size, resolution, origin_x, origin_y

    @classmethod
    def _attached(cls, mapper):
        g = cls.__new__(cls)
        g._m = mapper
        g.size, g.res, g.ox, g.oy = mapper.size, mapper.res, mapper.ox, mapper.oy
        return g

    @property
    def grid(self):
        """np.int8 [size, size], indexed [gy, gx] (:119).  Downloaded from the GPU once per change of the map, not per
        access: the renderer reads occ_grid.grid[gy, gx] cell by cell (:505-516)."""
        v = self._m._map_version
        if getattr(self, "_grid_cache", None) is None or self._grid_cache[0] != v:
            self._grid_cache = (v, self._m.grid_i8())
            self.downloads = getattr(self, "downloads", 0) + 1
        return self._grid_cache[1]

    def world_to_grid(self, wx, wy):                      # :121-125
        gx = int((wx - self.ox) / self.res)
        gy = int((wy - self.oy) / self.res)
        return gx, gy

    def grid_to_world(self, gx, gy):                      # :127-131
        return self.ox + (gx + 0.5) * self.res, self.oy + (gy + 0.5) * self.res

    def in_bounds(self, gx, gy):                          # :133-134
        return 0 <= gx < self.size and 0 <= gy < self.size

    def update_ray(self, robot_x, robot_y, hit_x, hit_y, hit_valid):   # :136-156
        self._m.update_rays([robot_x], [robot_y], [hit_x], [hit_y], [1 if hit_valid else 0])

    def update_rays(self, robot_x, robot_y, hit_x, hit_y, hit_valid):
        self._m.update_rays(robot_x, robot_y, hit_x, hit_y, hit_valid)

    def get_frontiers(self):                              # :181-196
        return [tuple(c) for c in self._m.frontier_cells().tolist()]

    def cluster_frontiers(self, frontier_cells=None, min_cluster=P.FRONTIER_MIN_CLUSTER):     # :198-231
        """Clusters of 4-connected frontier cells with at least FRONTIER_MIN_CLUSTER members, in the reference's cluster
        order (by first cell, row-major), each a list of (gx, gy).  The labelling runs on the device over the current
        grid; `frontier_cells`, if given, must be get_frontiers() of that grid (as main() passes it, :951-952).  Inside a
        cluster the cells come in row-major order, not in the reference's BFS visiting order (nothing reads that order:
        the reference only takes len() and the coordinate sums, :233-237)."""
        mem = self._m.frontier_members()
        if frontier_cells is not None and len(frontier_cells) != len(mem):
            raise ValueError("cluster_frontiers: frontier_cells is not get_frontiers() of the current grid")
        if len(mem) == 0:
            return []
        order = np.argsort(mem[:, 2], kind="stable")                 # by cluster (= by first cell), row-major inside
        roots, start = np.unique(mem[order, 2], return_index=True)
        bounds = list(start) + [len(mem)]
        out = []
        for k in range(len(roots)):
            cells = mem[order[bounds[k]:bounds[k + 1]], :2]
            if len(cells) >= min_cluster:
                out.append([(int(x), int(y)) for x, y in cells])
        return out

    def cluster_centroid_world(self, cluster):                       # :233-237
        avg_x = sum(c[0] for c in cluster) / len(cluster)
        avg_y = sum(c[1] for c in cluster) / len(cluster)
        return self.grid_to_world(avg_x, avg_y)

    def frontier_centroids(self, min_cluster=P.FRONTIER_MIN_CLUSTER):    # :951-956
        return self._m.frontier_centroids(min_cluster)


class _Node:
    __slots__ = ("index", "agent_id")

    def __init__(self, index, agent_id):
        self.index, self.agent_id = index, agent_id


class _NodeList:
    def __init__(self, mapper, graph):
        self._n = mapper.slam_sizes(graph)[0]
        idx, _ = mapper.closures(graph)
        self._agent = dict(zip((int(v) for v in idx[:, 1]), (int(a) for a in mapper.closure_agents(graph))))

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._agent:
            raise KeyError(f"node {i}: only the closing nodes of closures keep their agent_id on the host side")
        return _Node(i, self._agent[i])


class PoseGraphSLAM:
    """Read view of the device pose graph with the reference's attribute names (:267-271)."""

    @classmethod
    def _attached(cls, mapper, graph=0):
        s = cls.__new__(cls)
        s._m, s._g = mapper, graph
        return s

    @property
    def closures(self):
        idx, corr = self._m.closures(self._g)
        return [(int(i[0]), int(i[1]), float(c[0]), float(c[1])) for i, c in zip(idx, corr)]

    @property
    def landmarks(self):
        xy, ti = self._m.landmarks(self._g)
        return [(float(p[0]), float(p[1]), int(t[0]), int(t[1])) for p, t in zip(xy, ti)]

    @property
    def n_nodes(self):
        return self._m.slam_sizes(self._g)[0]

    @property
    def nodes(self):
        """self.nodes (:268) as far as the reference reads it: len(nodes) (:275) and nodes[node_idx].agent_id for the
        closing node of a closure (:335).  The poses themselves stay on the device (qs_last_batch returns a batch's)."""
        return _NodeList(self._m, self._g)

    def add_pose(self, x, y, yaw, agent_id, landmark_type, timestamp=0.0):
        """dual_bot_mapper.py:273-290: returns (closure_detected, correction_dx, correction_dy).  The
        pose is used as given (the caller applies its own drift correction first, :855-857)."""
        closed, corr = self._m.slam_add_poses([x], [y], [agent_id], [landmark_type])
        return bool(closed[0]), float(corr[0, 0]), float(corr[0, 1])

    def get_correction_for_agent(self, agent_id):
        """:328-338: the sum of this agent's closure corrections, i.e. its drift correction."""
        return tuple(float(v) for v in self._m.drift(agent_id))
