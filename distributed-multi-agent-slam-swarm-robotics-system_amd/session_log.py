"""Session / log writers: the files dual_bot_mapper.py writes, so the
reference's playback / plotting tools (simulation_tools/playback_dual_session.py:58-105) can read
a session mapped on the GPU unchanged.  Host-side only.

  telemetry.csv          header :733-734, one row per accepted packet :867-874 (pose AFTER offset and
                         drift, yaw in degrees, distances in cm)
  pointcloud.csv         header :735, one row per valid hit :893
  pointcloud_merged.csv  :1011-1020   all hits, grouped by bot then sensor (np.savetxt, header "x,y")
  pointcloud_bot{b}.csv  :1023-1031
  slam_closures.csv      :1034-1038   node_i, node_j, corr_dx, corr_dy (4 decimals)
"""
import csv
import math
import os

import numpy as np

from . import protocol as P

TELEMETRY_HEADER = ["time", "agent", "x", "y", "yaw_deg", "encoder", "v2v",
                    "front_cm", "left_cm", "back_cm", "right_cm", "landmark"]
POINTS_HEADER = ["time", "agent", "sensor", "x", "y"]


class SessionLog:
    def __init__(self, directory, max_agent=2):
        os.makedirs(directory, exist_ok=True)
        self.dir = directory
        self.max_agent = max_agent
        self._f_telem = open(os.path.join(directory, "telemetry.csv"), "w", newline="")
        self._f_points = open(os.path.join(directory, "pointcloud.csv"), "w", newline="")
        self._w_telem = csv.writer(self._f_telem)
        self._w_points = csv.writer(self._f_points)
        self._w_telem.writerow(TELEMETRY_HEADER)                     # :733-734
        self._w_points.writerow(POINTS_HEADER)                       # :735
        # point_clouds[bot][sensor] in arrival order (:764-767), kept for the exit-time files
        self._clouds = {b: {s: [] for s in P.SENSOR_NAMES} for b in range(1, max_agent + 1)}

    def log_batch(self, buf, accepted, pose, hits_xy, hits_valid, recv_time, lengths=None):
        """One ingested batch: buf uint8 [n, stride] datagrams, accepted / pose from
        QuasarMapper.last_batch(), hits from last_hits(), recv_time float [n]; lengths: per-datagram lengths (None: every
        datagram fills its slot).  A 41-byte v1 datagram has no landmark byte: LM_NONE (:832-836), whatever an earlier,
        longer datagram left in the slot's tail."""
        n = len(accepted)
        for i in range(n):
            if not accepted[i]:
                continue
            rec = np.frombuffer(buf[i, :42].tobytes() if buf.shape[1] >= 42 else buf[i].tobytes() + b"\0", dtype=P.PACKET_DTYPE)[0]
            agent = int(rec["agent"])
            ln = buf.shape[1] if lengths is None else int(lengths[i])
            lm = int(rec["lm"]) if ln >= 42 else 0
            now = float(recv_time[i])
            rx, ry, ryaw = pose[i]
            self._w_telem.writerow([                                  # :867-874
                f"{now:.3f}", agent, f"{rx:.4f}", f"{ry:.4f}", f"{math.degrees(ryaw):.2f}",
                int(rec["enc"]), int(rec["v2v"]),
                f"{float(rec['front']) * 100:.1f}", f"{float(rec['left']) * 100:.1f}",
                f"{float(rec['back']) * 100:.1f}", f"{float(rec['right']) * 100:.1f}", lm])
            for s, name in enumerate(P.SENSOR_NAMES):
                if hits_valid[i, s]:
                    wx, wy = hits_xy[i, s]
                    self._clouds[agent][name].append((wx, wy))        # :892
                    self._w_points.writerow([f"{now:.3f}", agent, name, f"{wx:.4f}", f"{wy:.4f}"])   # :893
        self._f_telem.flush()                                         # :875
        self._f_points.flush()                                        # :905

    def close(self, closures=()):
        """The `finally:` block of main() (:1009-1043)."""
        all_pts = [p for b in self._clouds for s in P.SENSOR_NAMES for p in self._clouds[b][s]]
        if all_pts:                                                   # :1011-1020
            np.savetxt(os.path.join(self.dir, "pointcloud_merged.csv"), np.array(all_pts), delimiter=",",
                       header="x,y", comments="")
        for b in self._clouds:                                        # :1023-1031
            pts = [p for s in P.SENSOR_NAMES for p in self._clouds[b][s]]
            if pts:
                np.savetxt(os.path.join(self.dir, f"pointcloud_bot{b}.csv"), np.array(pts), delimiter=",",
                           header="x,y", comments="")
        with open(os.path.join(self.dir, "slam_closures.csv"), "w", newline="") as f:   # :1034-1038
            w = csv.writer(f)
            w.writerow(["node_i", "node_j", "corr_dx", "corr_dy"])
            for lm_idx, node_idx, cdx, cdy in closures:
                w.writerow([lm_idx, node_idx, f"{cdx:.4f}", f"{cdy:.4f}"])
        self._f_telem.close()
        self._f_points.close()
