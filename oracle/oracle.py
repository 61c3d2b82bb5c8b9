"""ctypes front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  See oracle.c for the parity status
(mapper path pinned by tests/golden; EKF and map_merger rasterise "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()                      # (re)compiles only when oracle.c is newer than the library
        L = C.CDLL(_SO)
        vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_long, C.c_double
        L.qso_create.restype = vp
        L.qso_create.argtypes = [i32, f64, f64, f64, f64, i32, i32]
        L.qso_destroy.argtypes = [vp]
        L.qso_set_offset.argtypes = [vp, i32, f64]
        L.qso_set_closure_params.argtypes = [vp, f64, i64, f64]
        L.qso_set_owned.argtypes = [vp, i32, i32]
        L.qso_use_index.argtypes = [vp, i32]
        L.qso_feed.restype = i32
        L.qso_feed.argtypes = [vp, vp, i32]
        L.qso_feed_stream.restype = i64
        L.qso_feed_stream.argtypes = [vp, vp, i64, i64, vp]
        L.qso_feed_stream_t.restype = i64
        L.qso_feed_stream_t.argtypes = [vp, vp, i64, i64, vp, vp]
        L.qso_enable_ekf.argtypes = [vp, f64]
        L.qso_ekf_state.restype = vp
        L.qso_ekf_state.argtypes = [vp, i32]
        L.qso_update_rays.argtypes = [vp, vp, vp, vp, vp, vp, i64]
        L.qso_world_to_grid.argtypes = [vp, vp, i64, i32, vp]
        L.qso_bresenham.restype = i64
        L.qso_bresenham.argtypes = [i64, i64, i64, i64, vp, i64]
        L.qso_set_sequence.argtypes = [vp, C.c_uint64, C.c_uint64]
        for name in ("qso_grid", "qso_stamps", "qso_hits", "qso_misses", "qso_poses", "qso_pose_agents",
                     "qso_hit_points", "qso_hit_agent_sensor"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        for name in ("qso_n_nodes", "qso_n_landmarks", "qso_n_closures"):
            getattr(L, name).restype = i64
            getattr(L, name).argtypes = [vp, i32]
        for name in ("qso_n_poses", "qso_n_hits", "qso_n_rays", "qso_n_cells_written"):
            getattr(L, name).restype = i64
            getattr(L, name).argtypes = [vp]
        L.qso_drift.argtypes = [vp, i32, vp]
        L.qso_closures.argtypes = [vp, i32, vp, vp]
        L.qso_landmarks.argtypes = [vp, i32, vp, vp]
        L.qso_zone.restype = i32
        L.qso_zone.argtypes = [vp, i32, vp]
        L.qso_zone_packet.argtypes = [vp, i32, i32, vp]
        L.qso_logodds.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        L.qso_grid_to_pcd.restype = i64
        L.qso_grid_to_pcd.argtypes = [vp, i32, i32, f64, f64, f64, vp, i64]
        L.qso_rasterise.restype = i32
        L.qso_rasterise.argtypes = [vp, i64, f64, vp, vp, vp]
        L.qso_ekf_init.argtypes = [vp, f64, vp]
        L.qso_ekf_predict.argtypes = [vp, f64, f64]
        L.qso_ekf_update.argtypes = [vp, f64, f64]
        L.qso_ekf_packet.argtypes = [vp, vp, f64, f64, f64, f64, f64, f64]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class OracleMapper:
    """The reference mapper's per-packet path (dual_bot_mapper.py:826-945) on the CPU."""

    def __init__(self, size=200, res=0.05, ox=-5.0, oy=-5.0, separation=0.0,
                 max_agent=2, bots_per_graph=0):
        self.size, self.res, self.ox, self.oy = size, res, ox, oy
        self.max_agent = max_agent
        self.bots_per_graph = bots_per_graph if bots_per_graph > 0 else max_agent
        self.n_graphs = (max_agent + self.bots_per_graph - 1) // self.bots_per_graph
        self._h = lib().qso_create(size, res, ox, oy, separation, max_agent, bots_per_graph)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().qso_destroy(self._h)
            self._h = None

    def set_offset(self, bot, off_x):
        lib().qso_set_offset(self._h, bot, off_x)

    def set_owned(self, lo, hi):
        """One shard of a replicated-pose-graph deployment: add_pose for every packet, rays/zones/EKF for agents lo..hi."""
        lib().qso_set_owned(self._h, lo, hi)

    def use_index(self, on=True):
        """Closure search through a spatial index over self.landmarks instead of the reference's list scan (:294): the same
        closures, not the reference's algorithm -- for the `indexed` leg of the CPU baseline.  Before the first landmark."""
        lib().qso_use_index(self._h, int(bool(on)))

    def set_closure_params(self, radius=0.6, min_between=30, correction=0.5):
        """Other values of CLOSURE_RADIUS / MIN_POSES_BETWEEN / CLOSURE_CORRECTION (:99-101); before the first packet."""
        lib().qso_set_closure_params(self._h, radius, min_between, correction)

    def feed(self, datagram: bytes) -> int:
        b = np.frombuffer(datagram, dtype=np.uint8) if len(datagram) else np.zeros(1, np.uint8)
        return lib().qso_feed(self._h, _ptr(b), len(datagram))

    def feed_stream(self, buf, lengths=None, times=None) -> int:
        """buf: uint8 [n, stride]; lengths: uint16 [n] or None (all == stride); times: float64 [n]."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        n, stride = buf.shape
        lp = None
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint16)
            lp = _ptr(lengths)
        if times is not None:
            times = np.ascontiguousarray(times, dtype=np.float64)
            return lib().qso_feed_stream_t(self._h, _ptr(buf), n, stride, lp, _ptr(times))
        return lib().qso_feed_stream(self._h, _ptr(buf), n, stride, lp)

    def enable_ekf(self, metres_per_tick=0.0107):
        lib().qso_enable_ekf(self._h, metres_per_tick)

    def ekf_state(self, bot):
        f = _view(lib().qso_ekf_state(self._h, bot), (44,), np.float64)
        return f[:6], f[6:42].reshape(6, 6)

    def update_rays(self, rx, ry, hx, hy, valid):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (rx, ry, hx, hy)]
        v = np.ascontiguousarray(valid, dtype=np.uint8)
        lib().qso_update_rays(self._h, *[_ptr(x) for x in a], _ptr(v), len(v))

    def world_to_grid(self, w, axis=0):
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.empty(len(w), dtype=np.int64)
        lib().qso_world_to_grid(self._h, _ptr(w), len(w), axis, _ptr(out))
        return out

    @property
    def grid(self):
        return _view(lib().qso_grid(self._h), (self.size, self.size), np.int8)

    @property
    def stamps(self):
        return _view(lib().qso_stamps(self._h), (self.size, self.size), np.uint32)

    def set_sequence(self, seq0, stride=1):
        lib().qso_set_sequence(self._h, seq0, stride)

    @property
    def hits(self):
        return _view(lib().qso_hits(self._h), (self.size, self.size), np.int32)

    @property
    def misses(self):
        return _view(lib().qso_misses(self._h), (self.size, self.size), np.int32)

    def logodds(self, l_occ=0.85, l_free=0.4, lmin=-2.0, lmax=3.5):
        out = np.empty((self.size, self.size), dtype=np.float32)
        lib().qso_logodds(self._h, l_occ, l_free, lmin, lmax, _ptr(out))
        return out

    @property
    def poses(self):
        return _view(lib().qso_poses(self._h), (lib().qso_n_poses(self._h), 3), np.float64)

    @property
    def pose_agents(self):
        return _view(lib().qso_pose_agents(self._h), (lib().qso_n_poses(self._h),), np.int32)

    @property
    def hit_points(self):
        return _view(lib().qso_hit_points(self._h), (lib().qso_n_hits(self._h), 2), np.float64)

    @property
    def hit_agent_sensor(self):
        return _view(lib().qso_hit_agent_sensor(self._h), (lib().qso_n_hits(self._h),), np.int32)

    @property
    def n_rays(self):
        return lib().qso_n_rays(self._h)

    @property
    def n_cells_written(self):
        return lib().qso_n_cells_written(self._h)

    def n_nodes(self, graph=0):
        return lib().qso_n_nodes(self._h, graph)

    def drift(self, bot):
        out = np.zeros(2)
        lib().qso_drift(self._h, bot, _ptr(out))
        return out

    def closures(self, graph=0):
        n = lib().qso_n_closures(self._h, graph)
        idx = np.zeros((n, 2), dtype=np.int64)
        corr = np.zeros((n, 2), dtype=np.float64)
        if n:
            lib().qso_closures(self._h, graph, _ptr(idx), _ptr(corr))
        return idx, corr

    def landmarks(self, graph=0):
        n = lib().qso_n_landmarks(self._h, graph)
        xy = np.zeros((n, 2), dtype=np.float64)
        ti = np.zeros((n, 2), dtype=np.int64)
        if n:
            lib().qso_landmarks(self._h, graph, _ptr(xy), _ptr(ti))
        return xy, ti

    def zone(self, bot):
        out = np.zeros(4)
        ok = lib().qso_zone(self._h, bot, _ptr(out))
        return out if ok else None

    def zone_packet(self, bot, online=True) -> bytes:
        out = np.zeros(20, dtype=np.uint8)
        lib().qso_zone_packet(self._h, bot, int(online), _ptr(out))
        return out.tobytes()


def bresenham(x0, y0, x1, y1):
    cap = max(abs(x1 - x0), abs(y1 - y0)) + 1
    out = np.zeros((cap, 2), dtype=np.int64)
    n = lib().qso_bresenham(x0, y0, x1, y1, _ptr(out), cap)
    assert n == cap
    return out


def grid_to_pcd(data, res, ox, oy):
    data = np.ascontiguousarray(data, dtype=np.int8)
    h, w = data.shape
    n = lib().qso_grid_to_pcd(_ptr(data), h, w, res, ox, oy, None, 0)
    xy = np.zeros((n, 2), dtype=np.float64)
    if n:
        lib().qso_grid_to_pcd(_ptr(data), h, w, res, ox, oy, _ptr(xy), n)
    return xy


def rasterise(xy, res):
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    dims = np.zeros(2, dtype=np.int32)
    origin = np.zeros(2, dtype=np.float64)
    if not lib().qso_rasterise(_ptr(xy), len(xy), res, _ptr(dims), _ptr(origin), None):
        return None, None
    grid = np.empty((int(dims[0]), int(dims[1])), dtype=np.int8)
    lib().qso_rasterise(_ptr(xy), len(xy), res, _ptr(dims), _ptr(origin), _ptr(grid))
    return grid, origin


class OracleEKF:
    """Firmware EKF (AgentFirmware_Bot1/ekf.cpp) + the build-defined telemetry wiring."""

    STRIDE = 44

    def __init__(self, n_bots):
        self.f = np.zeros((n_bots, self.STRIDE), dtype=np.float64)
        self.prev = np.zeros((n_bots, 4), dtype=np.float64)

    def init(self, b, t, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        lib().qso_ekf_init(_ptr(self.f[b]), t, _ptr(x0))

    def predict(self, b, omega_m, t):
        lib().qso_ekf_predict(_ptr(self.f[b]), omega_m, t)

    def update(self, b, z_v, z_omega):
        lib().qso_ekf_update(_ptr(self.f[b]), z_v, z_omega)

    def packet(self, b, t, x, y, yaw, enc, metres_per_tick):
        lib().qso_ekf_packet(_ptr(self.f[b]), _ptr(self.prev[b]), t, x, y, yaw, enc, metres_per_tick)

    def state(self, b):
        return self.f[b, :6].copy()

    def cov(self, b):
        return self.f[b, 6:42].reshape(6, 6).copy()


def frontier_cells(grid):
    """OccupancyGrid.get_frontiers (dual_bot_mapper.py:181-196) -> int32 [n, 2] (gx, gy)."""
    L = lib()
    L.qso_frontier_cells.restype = C.c_long
    L.qso_frontier_cells.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    grid = np.ascontiguousarray(grid, dtype=np.int8)
    size = grid.shape[0]
    n = L.qso_frontier_cells(_ptr(grid), size, None, 0)
    xy = np.zeros((n, 2), dtype=np.int32)
    if n:
        L.qso_frontier_cells(_ptr(grid), size, _ptr(xy), n)
    return xy


def frontier_clusters(cells_xy, size, min_cluster=3):
    """cluster_frontiers (:198-231) -> int64 [k, 5]: size, first_x, first_y, sum_x, sum_y."""
    L = lib()
    L.qso_frontier_clusters.restype = C.c_long
    L.qso_frontier_clusters.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long]
    xy = np.ascontiguousarray(cells_xy, dtype=np.int32)
    k = L.qso_frontier_clusters(_ptr(xy), len(xy), size, min_cluster, None, 0)
    st = np.zeros((k, 5), dtype=np.int64)
    if k:
        L.qso_frontier_clusters(_ptr(xy), len(xy), size, min_cluster, _ptr(st), k)
    return st


def cluster_centroids_world(stats, res, ox, oy):
    """cluster_centroid_world (:233-237): mean index (true division) -> grid_to_world (:127-131)."""
    out = []
    for size, _, _, sx, sy in stats.tolist():
        ax, ay = sx / size, sy / size
        out.append((ox + (ax + 0.5) * res, oy + (ay + 0.5) * res))
    return np.array(out, dtype=np.float64).reshape(-1, 2)


# ---- ICP / voxel down-sample restated from Open3D's published algorithm (map_merger.py:45-60).
# PARITY UNPINNED: Open3D is not importable; these follow the same steps as csrc/icp.hip in numpy.
def icp_planar(src, dst, max_dist=1.0, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    src = np.asarray(src, dtype=np.float64).copy(); dst = np.asarray(dst, dtype=np.float64)
    T = np.eye(3)

    def evaluate(p):
        d2 = ((p[:, None, :] - dst[None, :, :]) ** 2).sum(-1)
        j = d2.argmin(1)                      # ties -> lowest index
        dmin = d2[np.arange(len(p)), j]
        ok = dmin < max_dist * max_dist
        n = int(ok.sum())
        return ok, j, n / len(p), (np.sqrt(dmin[ok].sum() / n) if n else 0.0)

    ok, j, fit, rm = evaluate(src)
    it = 0
    while it < max_iter:
        U = np.eye(3)
        if ok.any():
            a, b = src[ok], dst[j[ok]]
            am, bm = a.mean(0), b.mean(0)
            ac, bc = a - am, b - bm
            theta = np.arctan2((ac[:, 0] * bc[:, 1] - ac[:, 1] * bc[:, 0]).sum(), (ac[:, 0] * bc[:, 0] + ac[:, 1] * bc[:, 1]).sum())
            c, s = np.cos(theta), np.sin(theta)
            R = np.array([[c, -s], [s, c]])
            U[:2, :2] = R; U[:2, 2] = bm - R @ am
        T = U @ T
        src = src @ U[:2, :2].T + U[:2, 2]
        bfit, brm = fit, rm
        ok, j, fit, rm = evaluate(src)
        it += 1
        if abs(bfit - fit) < rel_fitness and abs(brm - rm) < rel_rmse:
            break
    return T, fit, rm, it


def voxel_downsample(xy, voxel):
    xy = np.asarray(xy, dtype=np.float64)
    if len(xy) == 0:
        return xy.reshape(0, 2)
    mn = xy.min(0) - voxel * 0.5
    v = np.floor((xy - mn) / voxel).astype(np.int64)
    key = v[:, 1] * (1 << 32) + v[:, 0]
    order = np.argsort(key, kind="stable")
    out, p = [], 0
    while p < len(xy):
        q = p
        while q < len(xy) and key[order[q]] == key[order[p]]:
            q += 1
        sel = order[p:q]
        sx = sy = 0.0
        for i in sel:
            sx += xy[i, 0]; sy += xy[i, 1]
        out.append((sx / (q - p), sy / (q - p)))
        p = q
    return np.array(out, dtype=np.float64).reshape(-1, 2)
