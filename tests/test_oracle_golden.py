"""The CPU oracle (oracle/oracle.c) against the fixtures the reference itself produced
(tests/golden/make_golden.py imported the reference's Python in the build container).
Integers bit-exact; floats to 1e-12 here (the oracle and CPython share libm)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN

SCENARIOS = ["session_200", "session_512", "session_4096", "session_sep_512", "laps5_512",
             "session_fine_1024", "mixed_200", "adversarial_512", "adversarial_dense_200"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def run_oracle(g):
    size, res, ox, oy, sep = g["cfg"]
    m = orc.OracleMapper(int(size), res, ox, oy, sep)
    acc = m.feed_stream(g["datagrams"], g["lengths"])
    return m, acc


def test_kat_sizes_and_zone():
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    assert kat["sizes"] == {"v2": 42, "v1": 41, "zone": 20, "target": 12}
    m = orc.OracleMapper()
    # simulation_tools/send_test_zone.py:9-10 known answer
    assert struct.pack("<4sffff", b"ZONE", 1.0, 2.0, 3.0, 4.0).hex() == kat["zone_1234_hex"]
    assert m.zone_packet(1, online=False) == struct.pack("<4sffff", b"ZONE", 999.0, 999.0, -999.0, -999.0)
    assert m.zone(1) is None


def test_world_to_grid_kat():
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for key, cfg in (("world_to_grid_200", (200, 0.05, -5.0, -5.0)),
                     ("world_to_grid_4096", (4096, 0.05, -102.4, -102.4))):
        m = orc.OracleMapper(*cfg)
        w = np.array([p[0] for p in kat[key]])
        want = np.array([p[1] for p in kat[key]])
        assert (m.world_to_grid(w, 0) == want).all()
        assert (m.world_to_grid(w, 1) == want).all()


def test_bresenham_exhaustive():
    t = load("bresenham_d40")
    D = int(t["D"][0])
    starts, cells = t["starts"], t["cells"]
    i = 0
    for dy in range(-D, D + 1):
        for dx in range(-D, D + 1):
            want = cells[starts[i]:starts[i + 1]]
            got = orc.bresenham(0, 0, dx, dy)
            assert got.shape == want.shape and (got == want).all(), (dx, dy)
            i += 1
    # translation invariance
    assert (orc.bresenham(7, -3, 12, -1) - [7, -3] == orc.bresenham(0, 0, 5, 2)).all()


def test_bresenham_major_axis_property():
    """The raster kernel walks rays in major/minor form (csrc/raycast_tiled.hip): k = max(dx, dy) free
    cells then the end cell, ONE minor-axis test per cell (E < (dmaj + 1) >> 1).  That restatement of
    dual_bot_mapper.py:158-179 is checked here against the oracle's walk for every offset the tile
    path can see (|d| < 64) and beyond."""
    D = 96
    dys, dxs = np.meshgrid(np.arange(-D, D + 1), np.arange(-D, D + 1), indexing="ij")
    dxs, dys = dxs.ravel(), dys.ravel()
    adx, ady = np.abs(dxs), np.abs(dys)
    sx, sy = np.where(dxs > 0, 1, -1), np.where(dys > 0, 1, -1)
    xmaj = adx >= ady
    dmaj, dmin = np.where(xmaj, adx, ady), np.where(xmaj, ady, adx)
    E, H = dmaj - dmin, (dmaj + 1) >> 1
    x, y = np.zeros_like(dxs), np.zeros_like(dys)
    walk = np.zeros((dxs.size, D + 1, 2), dtype=np.int64)
    for it in range(D + 1):
        live = it <= dmaj
        walk[live, it, 0], walk[live, it, 1] = x[live], y[live]
        minor = E < H
        E = E + np.where(minor, dmaj - dmin, -dmin)
        x = x + np.where(xmaj | minor, sx, 0)
        y = y + np.where(~xmaj | minor, sy, 0)
    for i in range(0, dxs.size, 7):          # every 7th offset: 5.3 k oracle walks
        want = orc.bresenham(0, 0, int(dxs[i]), int(dys[i]))
        n = int(dmaj[i]) + 1
        assert want.shape[0] == n and (walk[i, :n] == want).all(), (dxs[i], dys[i])
        assert (want[-1] == [dxs[i], dys[i]]).all()
        major = want[:, 0] if xmaj[i] else want[:, 1]
        assert (np.abs(np.diff(major)) == 1).all() or n == 1


def test_update_ray_cases():
    t = load("update_ray_cases")
    size, res, ox, oy = t["cfg"]
    rays = t["rays"]
    for i, (rx, ry, hx, hy, v) in enumerate(rays):
        m = orc.OracleMapper(int(size), res, ox, oy)
        m.update_rays([rx], [ry], [hx], [hy], [int(v)])
        assert (m.grid == t["grids"][i]).all(), i
    m = orc.OracleMapper(int(size), res, ox, oy)
    m.update_rays(rays[:, 0], rays[:, 1], rays[:, 2], rays[:, 3], rays[:, 4].astype(np.uint8))
    assert (m.grid == t["grid_sequential"]).all()


@pytest.mark.parametrize("name", SCENARIOS)
def test_scenario(name):
    g = load(name)
    m, acc = run_oracle(g)
    assert acc == int(g["accepted"].sum())
    grid = m.grid
    assert hashlib.sha256(grid.tobytes()).digest() == g["grid_sha256"].tobytes()
    assert [(grid == 0).sum(), (grid == 100).sum(), (grid == -1).sum()] == g["grid_counts"].tolist()
    if "grid" in g.files:
        assert (grid == g["grid"]).all()
    yx = g["grid_known_yx"]
    assert (grid[yx[:, 0], yx[:, 1]] == g["grid_known_val"]).all()
    # poses after offset + drift, per accepted packet
    assert (m.pose_agents == g["pose_agent"]).all()
    np.testing.assert_allclose(m.poses, g["pose_xyyaw"], rtol=0, atol=1e-12)
    # SLAM
    assert m.n_nodes(0) == int(g["n_nodes"][0])
    idx, corr = m.closures(0)
    assert (idx == g["closures_idx"]).all()
    np.testing.assert_allclose(corr, g["closures_corr"], rtol=0, atol=1e-12)
    xy, ti = m.landmarks(0)
    assert (ti == g["landmarks_type_idx"]).all()
    np.testing.assert_allclose(xy, g["landmarks_xy"], rtol=0, atol=1e-12)
    for b in (1, 2):
        np.testing.assert_allclose(m.drift(b), g["drift"][b - 1], rtol=0, atol=1e-12)
        z = m.zone(b)
        if np.isnan(g[f"zone_bot{b}"]).any():
            assert z is None
        else:
            np.testing.assert_allclose(z, g[f"zone_bot{b}"], rtol=0, atol=1e-12)
        assert m.zone_packet(b) == g[f"zone_bytes_bot{b}"].tobytes()
        sel = (m.hit_agent_sensor // 4) == b
        # reference stores hits per sensor list; compare as per-sensor sequences
        for s, k in enumerate(("front", "left", "back", "right")):
            got = m.hit_points[m.hit_agent_sensor == b * 4 + s]
            np.testing.assert_allclose(got, g[f"hits_bot{b}_{k}"], rtol=0, atol=1e-12)
        assert sel.sum() == len(g[f"hits_bot{b}"])


def test_session_kat_values():
    """SURVEY.md section 8(c) C4 known answers."""
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))["session"]
    g = load("session_512")
    assert hashlib.sha256(g["datagrams"][:, :42].tobytes()).hexdigest() == kat["packets_sha256"]
    assert kat["packets_sha256"].startswith("5269993c83e79c0d")
    assert kat["grid512_sha256"].startswith("cc477243d5f007ab")
    m, acc = run_oracle(g)
    assert acc == 687 and m.n_rays == 2748 and len(m.hit_points) == 1041
    idx, corr = m.closures(0)
    assert len(idx) == 10 and idx[0].tolist() == [236, 267] and idx[-1].tolist() == [6, 645]
    assert corr[0].tolist() == [-0.23625004291534424, 0.004499971866607666]
    assert m.zone_packet(1).hex() == "5a4f4e4561328dbfb615cfbf957370401ceb2240"
    # hit/miss counters (build extension) are consistent with the tri-state grid
    hits, misses = m.hits, m.misses
    assert ((hits + misses > 0) == (m.grid != -1)).all()
    assert hits.sum() + misses.sum() == m.n_cells_written


@pytest.mark.parametrize("name", ["session_200", "session_512", "laps5_512", "adversarial_dense_200", "mixed_200"])
def test_frontiers_vs_reference(name):
    """N1: get_frontiers / cluster_frontiers / cluster_centroid_world (dual_bot_mapper.py:181-237)."""
    fr = load("frontiers")
    g = load(name)
    size, res, ox, oy, _ = g["cfg"]
    cells = orc.frontier_cells(g["grid"])
    assert (cells == fr[name + "_cells"]).all()
    st = orc.frontier_clusters(cells, int(size))
    assert (st[:, 0] == fr[name + "_sizes"]).all()
    assert (st[:, 1:3] == fr[name + "_first"]).all()
    assert (st[:, 3:5] == fr[name + "_sums"]).all()
    cents = orc.cluster_centroids_world(st, res, ox, oy)
    assert (cents == fr[name + "_centroids"]).all()          # bit-exact: same integer sums, same fp64 ops


@pytest.mark.parametrize("name", ["session_200", "session_sep_512", "laps5_512", "mixed_200", "adversarial_dense_200"])
def test_pure_python_restatement_equals_reference_fixtures(name):
    """oracle/pymapper.py -- the single-process Python baseline of bench.py and a second, independent statement
    of the path -- against the fixtures the reference produced, and against oracle.c."""
    import hashlib
    from oracle.pymapper import PyMapper
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    size, res, ox, oy, sep = g["cfg"]
    pm = PyMapper(int(size), res, ox, oy, sep)
    pm.feed_stream(g["datagrams"], g["lengths"])
    assert hashlib.sha256(pm.grid.tobytes()).digest() == g["grid_sha256"].tobytes()
    assert pm.accepted == int(g["accepted"].sum()) and pm.n_nodes == int(g["n_nodes"][0])
    ci = np.array([(a, b) for a, b, _, _ in pm.closures], dtype=np.int64).reshape(-1, 2)
    cc = np.array([(c, d) for _, _, c, d in pm.closures]).reshape(-1, 2)
    assert (ci == g["closures_idx"]).all() and (len(cc) == 0 or np.abs(cc - g["closures_corr"]).max() < 1e-12)
    o = orc.OracleMapper(int(size), res, ox, oy, sep)
    o.feed_stream(g["datagrams"], g["lengths"])
    assert (pm.grid == o.grid).all()
    for b in (1, 2):
        assert np.abs(np.array(pm.drift[b]) - o.drift(b)).max() == 0.0


def test_indexed_closure_search_equals_the_list_scan():
    """The optional spatial index of the CPU restatement (the `indexed` leg of bench.py's cpu_baseline) finds the reference's
    first match: closures, landmarks, drift and grid are those of the list scan (:294-318) on the session, five laps, a long
    cycled stream, 40 bots in one pose graph, other closure constants, and an adversarial stream."""
    import importlib
    from conftest import PKG_NAME
    replay = importlib.import_module(PKG_NAME + ".replay")
    session, _ = replay.telemetry_csv_to_packets()
    cases = [("session", session, 2, 0, None), ("cycled", replay.cycle_stream(session, 30000), 2, 0, None),
             ("40 bots, one graph", replay.multi_bot_stream(None, 40, 20000, pitch=1.0, origin=(-8.0, -8.0), tiles_per_row=5), 40, 0, None),
             ("13 bots, graphs of 2", replay.multi_bot_stream(None, 13, 12000, pitch=8.0, origin=(-10.0, -10.0), tiles_per_row=4), 13, 2, (0.3, 5, 0.25)),
             ("no cool-down", replay.cycle_stream(session, 8000), 2, 0, (0.6, 0, 1.0)),
             ("adversarial", replay.adversarial_stream(20000, seed=5, lo=-10, hi=10), 2, 0, (1.0, 31, 0.5))]
    for name, stream, bots, bpg, params in cases:
        out = []
        for ix in (False, True):
            o = orc.OracleMapper(512, 0.05, -12.8, -12.8, 0.0, max_agent=bots, bots_per_graph=bpg)
            if params:
                o.set_closure_params(*params)
            o.use_index(ix)
            o.feed_stream(stream)
            out.append(o)
        a, b = out
        n_cl = 0
        for g in range(a.n_graphs):
            ia, ca = a.closures(g); ib, cb = b.closures(g)
            assert ia.shape == ib.shape and (ia == ib).all() and (ca == cb).all(), name
            la, ta = a.landmarks(g); lb, tb = b.landmarks(g)
            assert (ta == tb).all() and (la == lb).all(), name
            n_cl += len(ia)
        assert n_cl > 0, name
        assert (a.grid == b.grid).all() and all((a.drift(k) == b.drift(k)).all() for k in range(1, bots + 1)), name
