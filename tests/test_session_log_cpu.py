"""N4 session writers: files in the reference's schemas (dual_bot_mapper.py:733-735, :867-893,
:1009-1040), readable the way simulation_tools/playback_dual_session.py:58-105 reads them.  Uses the
reference-produced golden (poses, hits, closures) as the mapper's outputs."""
import csv
import importlib
import math
import os

import numpy as np

from conftest import GOLDEN, PKG_NAME


def test_session_files_roundtrip(tmp_path):
    sl = importlib.import_module(PKG_NAME + ".session_log")
    g = np.load(os.path.join(GOLDEN, "session_512.npz"))
    n = len(g["datagrams"])
    acc = g["accepted"]
    pose = np.full((n, 3), np.nan); pose[acc == 1] = g["pose_xyyaw"]
    # rebuild per-record hit slots from the per-sensor golden lists
    hxy = np.zeros((n, 4, 2)); hv = np.zeros((n, 4), dtype=np.uint8)
    cursor = {(b, s): 0 for b in (1, 2) for s in range(4)}
    names = ("front", "left", "back", "right")
    for i in range(n):
        d = g["datagrams"][i]
        b = int(d[4])
        dist = d[25:41].view("<f4")
        for s in range(4):
            if 0.05 < float(dist[s]) <= 1.2:
                hxy[i, s] = g[f"hits_bot{b}_{names[s]}"][cursor[(b, s)]]; hv[i, s] = 1; cursor[(b, s)] += 1
    log = sl.SessionLog(str(tmp_path))
    log.log_batch(g["datagrams"][:300], acc[:300], pose[:300], hxy[:300], hv[:300], g["recv_time"][:300])
    log.log_batch(g["datagrams"][300:], acc[300:], pose[300:], hxy[300:], hv[300:], g["recv_time"][300:])
    closures = [(int(a), int(b), float(c), float(d)) for (a, b), (c, d) in zip(g["closures_idx"], g["closures_corr"])]
    log.close(closures)
    # read back exactly like playback_dual_session.load_session (:72-85, :93-100)
    rows = list(csv.DictReader(open(tmp_path / "telemetry.csv")))
    assert len(rows) == 687 and list(rows[0].keys()) == sl.TELEMETRY_HEADER
    r0 = rows[0]
    assert int(r0["agent"]) == int(g["datagrams"][0][4]) and abs(float(r0["x"]) - g["pose_xyyaw"][0][0]) < 5e-5
    assert abs(math.radians(float(r0["yaw_deg"])) - g["pose_xyyaw"][0][2]) < 1e-3
    assert all(float(r["front_cm"]) >= 0 for r in rows) and {int(r["landmark"]) for r in rows} == {0, 2, 3, 5}
    pts = list(csv.DictReader(open(tmp_path / "pointcloud.csv")))
    assert len(pts) == 1041 and set(p["sensor"] for p in pts) == set(names)
    merged = np.loadtxt(tmp_path / "pointcloud_merged.csv", delimiter=",", skiprows=1)
    assert merged.shape == (1041, 2)
    b1 = np.loadtxt(tmp_path / "pointcloud_bot1.csv", delimiter=",", skiprows=1)
    np.testing.assert_allclose(b1, g["hits_bot1"], rtol=0, atol=1e-12)     # grouped by sensor, as the reference
    cl = list(csv.reader(open(tmp_path / "slam_closures.csv")))
    assert cl[0] == ["node_i", "node_j", "corr_dx", "corr_dy"] and len(cl) == 11
    assert cl[1] == ["236", "267", "-0.2363", "0.0045"]


def test_v1_datagram_in_a_reused_slot_logs_no_landmark(tmp_path):
    """MissionControl reuses its 48-byte slots; a 41-byte v1 datagram written over an earlier 42-byte one
    leaves that one's landmark byte in the slot's tail.  The reference forces LM_NONE for v1 packets (:832-836)."""
    import csv
    import importlib
    from conftest import PKG_NAME
    P = importlib.import_module(PKG_NAME + ".protocol")
    SL = importlib.import_module(PKG_NAME + ".session_log")
    buf = np.zeros((2, 48), dtype=np.uint8)
    v2 = P.pack_packets([1], [0.5], [0.25], [0.0], [7], [9], np.array([[0.3, 0.4, 0.5, 0.6]]), [5])[0]
    buf[0, :42] = v2
    buf[1, :42] = v2                      # the slot's previous tenant: landmark 5 at byte 41 ...
    lengths = np.array([42, 41], dtype=np.uint16)     # ... under a v1 datagram of 41 bytes
    log = SL.SessionLog(str(tmp_path), max_agent=2)
    acc = np.ones(2, dtype=np.uint8); pose = np.zeros((2, 3)); hxy = np.zeros((2, 4, 2)); hv = np.zeros((2, 4), dtype=np.uint8)
    log.log_batch(buf, acc, pose, hxy, hv, np.array([0.0, 0.1]), lengths=lengths)
    log.close()
    rows = list(csv.DictReader(open(tmp_path / "telemetry.csv")))
    assert [int(r["landmark"]) for r in rows] == [5, 0]
