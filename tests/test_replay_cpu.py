"""Host-side replay harness: CSV -> QuasarPacket bytes reproduces the reference-derived stream."""
import hashlib
import importlib
import json
import os

import numpy as np

from conftest import GOLDEN, load_pkg


def test_csv_to_packets_matches_the_golden_stream():
    pkg = load_pkg()
    replay = importlib.import_module(pkg.__name__ + ".replay")
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))["session"]
    pk, t = replay.telemetry_csv_to_packets()
    assert pk.shape == (687, 42) and len(t) == 687
    assert hashlib.sha256(pk.tobytes()).hexdigest() == kat["packets_sha256"]
    g = np.load(os.path.join(GOLDEN, "session_512.npz"))
    assert (g["datagrams"][:, :42] == pk).all()
    np.testing.assert_allclose(t, g["recv_time"], rtol=0, atol=0)


def test_synthetic_streams_shape_and_determinism():
    pkg = load_pkg()
    replay = importlib.import_module(pkg.__name__ + ".replay")
    P = pkg.protocol
    pk, _ = replay.telemetry_csv_to_packets()
    c = replay.cycle_stream(pk, 2000)
    assert c.shape == (2000, 42) and (c[687:1374] == pk).all()
    m = replay.multi_bot_stream(pk, 64, 6400)
    rec = m.view(P.PACKET_DTYPE).reshape(-1)
    assert rec["agent"].min() == 1 and rec["agent"].max() == 64 and (rec["magic"] == b"QSRL").all()
    assert np.abs(rec["x"]).max() < 102.4 and np.abs(rec["y"]).max() < 102.4
    assert (replay.multi_bot_stream(pk, 64, 6400) == m).all()
    a = replay.adversarial_stream(1000)
    assert a.shape == (1000, 42) and (replay.adversarial_stream(1000) == a).all()
