"""Wire formats of the mapper path (host side, no GPU involved).

Mirrors server_nodes/dual_bot_mapper.py:40-54 (struct formats), :57-99 (constants) and the
packet producer struct of AgentFirmware_Bot1/AgentFirmware_Bot1.ino:172-185.
"""
import math
import struct

import numpy as np

# QuasarPacket v2 / v1 from the bots, ZONE / TARG to the bots  (dual_bot_mapper.py:41-54)
PACKET_FMT = "<4sBfffiIffffB"
PACKET_SIZE = struct.calcsize(PACKET_FMT)          # 42
PACKET_FMT_V1 = "<4sBfffiIffff"
PACKET_SIZE_V1 = struct.calcsize(PACKET_FMT_V1)    # 41
ZONE_FMT = "<4sffff"
ZONE_SIZE = struct.calcsize(ZONE_FMT)              # 20
TARGET_FMT = "<4sff"
TARGET_SIZE = struct.calcsize(TARGET_FMT)          # 12

MAX_DIST_M = 1.20
MIN_DIST_M = 0.05
SENSOR_ANGLES_RAD = {"front": 0.0, "left": math.pi / 2, "back": math.pi, "right": -math.pi / 2}
SENSOR_NAMES = ("front", "left", "back", "right")

LM_NONE, LM_CORNER_L, LM_CORNER_R, LM_CORRIDOR, LM_DEAD_END, LM_OPEN = range(6)
LANDMARK_NAMES = {LM_NONE: "NONE", LM_CORNER_L: "CORNER_L", LM_CORNER_R: "CORNER_R",
                  LM_CORRIDOR: "CORRIDOR", LM_DEAD_END: "DEAD_END", LM_OPEN: "OPEN"}

HEARTBEAT_TIMEOUT = 5.0
ZONE_UPDATE_INTERVAL = 2.0

GRID_RESOLUTION = 0.05
GRID_SIZE = 200
GRID_ORIGIN_X = -5.0
GRID_ORIGIN_Y = -5.0
CELL_UNKNOWN, CELL_FREE, CELL_OCCUPIED = -1, 0, 100

CLOSURE_RADIUS = 0.60
MIN_POSES_BETWEEN = 30
CLOSURE_CORRECTION = 0.5
FRONTIER_MIN_CLUSTER = 3
FRONTIER_SEPARATION = 1.0
TARGET_INTERVAL = 3.0

# numpy view of the packed 42-byte record (same field order as PACKET_FMT)
PACKET_DTYPE = np.dtype([("magic", "S4"), ("agent", "u1"), ("x", "<f4"), ("y", "<f4"), ("yaw", "<f4"),
                         ("enc", "<i4"), ("v2v", "<u4"), ("front", "<f4"), ("left", "<f4"),
                         ("back", "<f4"), ("right", "<f4"), ("lm", "u1")])
assert PACKET_DTYPE.itemsize == PACKET_SIZE


def pack_packet(agent, x, y, yaw, enc, v2v, front, left, back, right, landmark=LM_NONE) -> bytes:
    return struct.pack(PACKET_FMT, b"QSRL", agent, x, y, yaw, enc, v2v, front, left, back, right, landmark)


def pack_packets(agent, x, y, yaw, enc, v2v, dist4, landmark) -> np.ndarray:
    """Vectorised packer: arrays -> uint8 [n, 42]."""
    n = len(agent)
    rec = np.zeros(n, dtype=PACKET_DTYPE)
    rec["magic"] = b"QSRL"
    rec["agent"], rec["x"], rec["y"], rec["yaw"] = agent, x, y, yaw
    rec["enc"], rec["v2v"] = enc, v2v
    d = np.asarray(dist4, dtype=np.float32)
    rec["front"], rec["left"], rec["back"], rec["right"] = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    rec["lm"] = landmark
    return rec.view(np.uint8).reshape(n, PACKET_SIZE)


def pack_datagrams(datagrams):
    """Variable-length datagrams -> (uint8 [n, 48] zero padded, uint16 lengths); datagrams
    longer than 48 bytes are recorded with their true length and dropped by the decoder."""
    n = len(datagrams)
    buf = np.zeros((n, 48), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, d in enumerate(datagrams):
        m = min(len(d), 48)
        buf[i, :m] = np.frombuffer(d[:m], dtype=np.uint8)
        lens[i] = min(len(d), 65535)
    return buf, lens


def zone_packet(box) -> bytes:
    """send_zone_to_bot's payload (dual_bot_mapper.py:675-684)."""
    if box is None:
        return struct.pack(ZONE_FMT, b"ZONE", 999.0, 999.0, -999.0, -999.0)
    return struct.pack(ZONE_FMT, b"ZONE", box[0], box[1], box[2], box[3])


def compute_bounding_box(points_x, points_y):
    """dual_bot_mapper.py:702-706."""
    if len(points_x) == 0:
        return None
    return (min(points_x), min(points_y), max(points_x), max(points_y))


# ---- legacy Quasar-Lite v0 packet of the ROS bridge (server_nodes/udp_bridge.py:25-38) ----------
# magic, agent_id, x, y, yaw, scan_count, 181 servo-sweep ranges; command back: 'CMD1', linear_x, angular_z
PACKET_FMT_V0 = "<4sBfffH181f"
PACKET_SIZE_V0 = struct.calcsize(PACKET_FMT_V0)      # 743
CMD_FMT = "<4sff"
PACKET_DTYPE_V0 = np.dtype([("magic", "S4"), ("agent", "u1"), ("x", "<f4"), ("y", "<f4"), ("yaw", "<f4"),
                            ("scan_count", "<u2"), ("ranges", "<f4", (181,))])
assert PACKET_DTYPE_V0.itemsize == PACKET_SIZE_V0


def unpack_v0(data: bytes):
    """udp_bridge.py:53-75: None unless the size and magic match; else (agent, x, y, yaw, ranges[181])."""
    if len(data) != PACKET_SIZE_V0:
        return None
    rec = np.frombuffer(data, dtype=PACKET_DTYPE_V0)[0]
    if rec["magic"] != b"QSRL":
        return None
    return int(rec["agent"]), float(rec["x"]), float(rec["y"]), float(rec["yaw"]), rec["ranges"].copy()


def pack_cmd(linear_x, angular_z) -> bytes:
    """udp_bridge.py:140-146."""
    return struct.pack(CMD_FMT, b"CMD1", linear_x, angular_z)
