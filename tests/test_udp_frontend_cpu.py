"""N2 live UDP front-end on CPU: real sockets on localhost, a stub in place of the HIP mapper.
Checks the host logic mirrored from dual_bot_mapper.py:745-753, :805-812, :815-848, :922-945."""
import importlib
import socket
import struct
import time

import numpy as np

from conftest import PKG_NAME


class StubMapper:
    def __init__(self):
        self.batches = []
        self._acc = None

    def ingest_array(self, buf, lens, times):
        self.batches.append((buf.copy(), lens.copy(), times.copy()))
        ok = (lens == 42) | (lens == 41)
        magic = (buf[:, 0] == ord("Q")) & (buf[:, 1] == ord("S")) & (buf[:, 2] == ord("R")) & (buf[:, 3] == ord("L"))
        agent = (buf[:, 4] >= 1) & (buf[:, 4] <= 2)
        self._acc = (ok & magic & agent).astype(np.uint8)

    def last_batch(self):
        return self._acc, None

    def zone(self, bot):
        return (float(bot), 2.0, 3.0, 4.0)

    def zone_packet(self, bot, online=True):
        return struct.pack("<4sffff", b"ZONE", *(self.zone(bot) if online else (999.0, 999.0, -999.0, -999.0)))


def test_poll_heartbeat_and_zone_cadence():
    fe = importlib.import_module(PKG_NAME + ".udp_frontend")
    P = importlib.import_module(PKG_NAME + ".protocol")
    srv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    srv.bind(("127.0.0.1", 0))
    port = srv.getsockname()[1]
    mc = fe.MissionControl(StubMapper(), sock=srv)
    bot1 = socket.socket(socket.AF_INET, socket.SOCK_DGRAM); bot1.bind(("127.0.0.1", 0))
    bot2 = socket.socket(socket.AF_INET, socket.SOCK_DGRAM); bot2.bind(("127.0.0.1", 0))
    # the reference always answers on fixed ports (:759); aim them at our fake bots for the test
    mc.bot_ports = {1: bot1.getsockname()[1], 2: bot2.getsockname()[1]}
    p1 = P.pack_packet(1, 0.1, 0.2, 0.3, 1, 2, 0.5, 0.6, 0.7, 0.8, 5)
    p2 = P.pack_packet(2, 1.1, 1.2, 1.3, 1, 2, 0.5, 0.6, 0.7, 0.8, 0)
    for d in (p1, p2, p1[:41], b"junk", p1 + b"x" * 20, b"QSRX" + p1[4:], p1[:4] + b"\x07" + p1[5:]):
        bot1.sendto(d, ("127.0.0.1", port))
    bot2.sendto(p2, ("127.0.0.1", port))
    time.sleep(0.05)
    t0 = 1000.0
    assert mc.poll(now=t0) == 8
    buf, lens, times = mc.mapper.batches[0]
    assert buf.shape == (8, 48) and sorted(lens.tolist()) == sorted([42, 42, 41, 4, 65535, 42, 42, 42])
    assert (times == t0).all()
    assert mc.pkt_counts == {1: 2, 2: 2} and mc.online == {1: True, 2: True}
    assert mc.bot_addrs[1] == ("127.0.0.1", bot1.getsockname()[1])
    assert mc.poll(now=t0) == 0
    # zone cadence: nothing before 2 s, then each bot gets the OTHER bot's box (:922-941)
    mc.last_zone_send = t0
    assert mc.zone_tick(now=t0 + 1.9) == {}
    sent = mc.zone_tick(now=t0 + 2.1)
    assert struct.unpack("<4sffff", sent[1])[1] == 2.0 and struct.unpack("<4sffff", sent[2])[1] == 1.0
    bot1.settimeout(1.0); bot2.settimeout(1.0)
    assert bot1.recv(64) == sent[1] and bot2.recv(64) == sent[2]
    # heartbeat: 5 s of silence -> offline -> partner's zone is lifted (:805-812, :942-945)
    assert mc.heartbeat(now=t0 + 4.9) == []
    bot2.sendto(p2, ("127.0.0.1", port)); time.sleep(0.05)
    mc.poll(now=t0 + 4.0)
    assert mc.heartbeat(now=t0 + 5.5) == [1] and mc.online == {1: False, 2: True}
    sent = mc.zone_tick(now=t0 + 5.6)
    assert struct.unpack("<4sffff", sent[2])[1:] == (999.0, 999.0, -999.0, -999.0)     # bot 2 told: no zone
    assert struct.unpack("<4sffff", sent[1])[1] == 2.0                                  # bot 1 (offline) still addressed
    # a returning bot is online again on its next packet (:860-864)
    bot1.sendto(p1, ("127.0.0.1", port)); time.sleep(0.05)
    mc.poll(now=t0 + 6.0)
    assert mc.online[1] is True
    assert mc.other_of(1) == 2 and mc.other_of(2) == 1 and mc.other_of(3) == 4
    for s in (bot1, bot2):
        s.close()
    mc.close()
