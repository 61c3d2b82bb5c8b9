"""Replay harness: telemetry CSV -> QuasarPacket bytes, and the synthetic streams of
BASELINE.json's configs.

The reference has no replay-into-mapper tool (simulation_tools/playback_dual_session.py renders
the CSVs itself), so this is the build's own: it turns a `telemetry.csv` written by
simulation_tools/generate_fake_dual_session.py (schema: :369-370 / dual_bot_mapper.py:733-734)
back into the 42-byte datagrams the bots would have sent (AgentFirmware_Bot1.ino:172-185).
Host-side numpy only.
"""
import csv
import os

import numpy as np

from . import protocol as P

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
# the reference generator's seed-42 session (tests/golden/make_golden.py wrote it; byte-identical to the golden fixture)
SESSION_CSV = os.path.join(DATA, "session_telemetry.csv")
# 64 lanes, lane i = the bot-1 (even i) / bot-2 (odd i) packets of the reference generator run with seed 42 + i
# (tests/golden/make_multibot_sessions.py, build container only)
MULTIBOT_NPZ = os.path.join(DATA, "multibot_sessions.npz")


def telemetry_csv_to_packets(path=SESSION_CSV):
    """Rows in FILE order -> (uint8 [n,42], recv_time float64 [n]).  Field conversions follow the
    firmware's units: yaw degrees -> radians, centimetres -> metres, all stored as f32."""
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    col = lambda k, t: np.array([t(r[k]) for r in rows])
    pk = P.pack_packets(
        col("agent", int), col("x", float), col("y", float), np.radians(col("yaw_deg", float)),
        col("encoder", int), col("v2v", int),
        np.stack([col("front_cm", float) / 100.0, col("left_cm", float) / 100.0,
                  col("back_cm", float) / 100.0, col("right_cm", float) / 100.0], axis=1),
        col("landmark", int))
    return pk, col("time", float)


def cycle_stream(pkts, n):
    """config C2: the 2-bot session repeated (per-lap world offset 0) up to n packets."""
    reps = -(-n // len(pkts))
    return np.ascontiguousarray(np.tile(pkts, (reps, 1))[:n])


def multibot_lanes(path=MULTIBOT_NPZ):
    """The per-bot sessions of SURVEY.md 8(d) D2: list of 64 record arrays, lane i generated with seed 42 + i."""
    z = np.load(path, allow_pickle=False)
    rec = z["packets"].view(P.PACKET_DTYPE).reshape(-1)
    st = z["lane_start"]
    return [rec[st[i]:st[i + 1]] for i in range(len(st) - 1)]


def multi_bot_stream(pkts, n_bots, n, pitch=8.0, tiles_per_row=25, origin=(-98.0, -98.0), lap_shift=37, tile0=0,
                     agent0=1):
    """configs C3/C4 (build-defined; the reference has no >2-bot generator): bot i (agent id
    agent0+i) lives in its own room tile (lattice position tile0+i: rank r of a sharded deployment passes
    tile0 = r*n_bots, so the 512 bots of configs[3] occupy 512 different tiles of the 25 x 25 lattice) on a
    `pitch`-metre lattice; streams are interleaved round-robin, n packets in total.
    pkts = None: bot i replays ITS OWN session -- lane (tile0 + i) mod 64 of multibot_lanes(): the reference generator run
    with seed 42 + lane, bot-1 waypoints for even lanes, bot-2 for odd ones, own sensor noise and odometry drift (beyond
    64 bots the lanes repeat, in different tiles).  pkts = a 2-bot session: every bot replays that session's bot-1 (even i)
    or bot-2 (odd i) packets (two noise realisations in all; rounds 1-2, kept for the tests that pin it).  Either way bot i
    starts `lap_shift*(tile0+i)` packets into its lap."""
    if pkts is None:
        all_lanes = multibot_lanes()
        lanes = None
    else:
        rec = pkts.view(P.PACKET_DTYPE).reshape(-1)
        lanes = [rec[rec["agent"] == 1], rec[rec["agent"] == 2]]
    per_bot = -(-n // n_bots)
    out = np.zeros((per_bot, n_bots), dtype=P.PACKET_DTYPE)
    for i in range(n_bots):
        t = tile0 + i
        src = lanes[i & 1] if lanes is not None else all_lanes[t % len(all_lanes)]
        idx = (np.arange(per_bot) + lap_shift * t) % len(src)
        r = src[idx].copy()
        tx = origin[0] + pitch * (t % tiles_per_row)
        ty = origin[1] + pitch * (t // tiles_per_row)
        r["x"] = (r["x"].astype(np.float64) + tx).astype(np.float32)
        r["y"] = (r["y"].astype(np.float64) + ty).astype(np.float32)
        r["agent"] = agent0 + i
        out[:, i] = r
    flat = out.reshape(-1)[:n]
    return np.ascontiguousarray(flat.view(np.uint8).reshape(n, P.PACKET_SIZE))


def adversarial_stream(n, seed=1234, lo=-100.0, hi=100.0, max_agent=2):
    """Uniform-random poses, yaw, distances; landmark types drawn from the session histogram
    {0:555, 5:128, 3:3, 2:1}: the worst-case-locality stream of SURVEY.md 8(d) D2."""
    rng = np.random.default_rng(seed)
    lm = rng.choice(np.array([0, 5, 3, 2], dtype=np.uint8), size=n, p=np.array([555, 128, 3, 1]) / 687.0)
    return P.pack_packets(rng.integers(1, max_agent + 1, n), rng.uniform(lo, hi, n), rng.uniform(lo, hi, n),
                          rng.uniform(-np.pi, np.pi, n), np.arange(n), rng.integers(0, 1000, n),
                          rng.uniform(0.02, 2.5, (n, 4)), lm)
