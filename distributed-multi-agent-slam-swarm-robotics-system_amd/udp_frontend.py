"""Live UDP front-end: the reference's main() loop without the renderer.

Mirrors server_nodes/dual_bot_mapper.py:
  socket            :745-753   UDP, SO_REUSEADDR, bind 0.0.0.0:port, non-blocking
  receive           :815-848   drain datagrams, remember each bot's source IP, reply port 8888 for
                               bot 1 / 8889 for bot 2 (:759, :846), per-bot packet counts
  per-packet body   :826-919   handed to the GPU as ONE batch per poll (QuasarMapper.ingest_array)
  heartbeat         :805-812   5 s of silence -> offline; any accepted packet -> online (:860-864)
  zone timer        :922-945   every > 2 s: each bot gets the OTHER bot's bounding box, or the lift
                               box (999, 999, -999, -999) when the other is offline
Differences: the reference throttles itself to 20 packets per 30 fps frame (:816, :474); here a
poll drains the socket (up to max_batch datagrams).  Host-side Python only; the mapper can be any
object with ingest_array / last_batch / zone_packet (tests use a stub, production the HIP mapper).
"""
import socket
import time

import numpy as np

from . import protocol as P

SLOT = 48      # bytes per datagram slot handed to qs_ingest (42-byte packets, room for oversize marks)


class MissionControl:
    def __init__(self, mapper, port=8888, bind_addr="0.0.0.0", max_batch=65536, sock=None, max_agent=2):
        self.mapper = mapper
        self.max_agent = max_agent
        self.max_batch = max_batch
        if sock is None:
            sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)                   # :745
            sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)                # :746
            sock.bind((bind_addr, port))                                              # :748
        sock.setblocking(False)                                                       # :753
        self.sock = sock
        bots = range(1, max_agent + 1)
        self.bot_addrs = {b: None for b in bots}                                      # :758
        self.bot_ports = {b: 8887 + b for b in bots}                                  # :759  (1 -> 8888, 2 -> 8889)
        self.last_packet_time = {b: 0.0 for b in bots}                                # :760
        self.pkt_counts = {b: 0 for b in bots}                                        # :761
        self.online = {b: False for b in bots}
        self.seen = {b: False for b in bots}
        self.zone_boxes = {b: None for b in bots}                                     # :773
        self.last_zone_send = time.time()                                             # :788
        self._buf = np.zeros((max_batch, SLOT), dtype=np.uint8)
        self._lens = np.zeros(max_batch, dtype=np.uint16)
        self._times = np.zeros(max_batch, dtype=np.float64)
        self._addrs = [None] * max_batch
        self.datagrams = 0

    # ---- :815-848 + :826-919 -----------------------------------------------------------------
    def poll(self, now=None):
        """Drain the socket into one batch and ingest it.  Returns the number of datagrams."""
        now = time.time() if now is None else now
        n = 0
        view = memoryview(self._buf).cast("B")
        while n < self.max_batch:
            try:
                nbytes, addr = self.sock.recvfrom_into(view[n * SLOT:(n + 1) * SLOT], SLOT)   # :818
            except BlockingIOError:
                break
            except OSError:
                break
            self._lens[n] = nbytes if nbytes < SLOT else 65535     # a datagram that fills the slot may be longer: drop it
            self._times[n] = now
            self._addrs[n] = addr
            n += 1
        if n == 0:
            return 0
        self.datagrams += n
        self.mapper.ingest_array(self._buf[:n], self._lens[:n], self._times[:n])
        accepted, _ = self.mapper.last_batch()
        agents = self._buf[:n, 4]
        for i in np.nonzero(accepted)[0]:
            a = int(agents[i])
            self.bot_addrs[a] = (self._addrs[i][0], self.bot_ports[a])                # :846
            self.last_packet_time[a] = now                                            # :847
            self.pkt_counts[a] += 1                                                   # :848
            self.online[a] = True                                                     # :860-864
            self.seen[a] = True
        return n

    # ---- :805-812 --------------------------------------------------------------------------------
    def heartbeat(self, now=None):
        now = time.time() if now is None else now
        went_offline = []
        for b in self.online:
            if self.seen[b] and self.last_packet_time[b] > 0 and now - self.last_packet_time[b] > P.HEARTBEAT_TIMEOUT:
                if self.online[b]:
                    self.online[b] = False
                    went_offline.append(b)
        return went_offline

    def other_of(self, bot_id):
        """The reference pairs bot 1 with bot 2 (:926); larger swarms are paired (1,2), (3,4), ..."""
        return bot_id + 1 if bot_id % 2 == 1 else bot_id - 1

    # ---- :922-945 --------------------------------------------------------------------------------
    def zone_tick(self, now=None, force=False):
        """Send every bot its partner's zone if the 2 s interval has elapsed.  Returns {bot: datagram}."""
        now = time.time() if now is None else now
        if not force and not (now - self.last_zone_send > P.ZONE_UPDATE_INTERVAL):    # :922
            return {}
        self.last_zone_send = now
        sent = {}
        for bot_id in self.online:
            other = self.other_of(bot_id)
            if other not in self.online:
                continue
            online = self.online.get(other, False)                                     # :929
            pkt = self.mapper.zone_packet(other, online=online)                       # :940-941 / :944-945
            self.zone_boxes[other] = self.mapper.zone(other) if online else None
            if self.bot_addrs[bot_id] is not None:                                    # send_zone_to_bot :677-678
                try:
                    self.sock.sendto(pkt, self.bot_addrs[bot_id])
                except OSError:
                    pass
            sent[bot_id] = pkt
        return sent

    def step(self, now=None):
        """One iteration of the reference's while-loop body (without events and rendering)."""
        now = time.time() if now is None else now
        self.heartbeat(now)
        n = self.poll(now)
        self.zone_tick(now)
        return n

    def run(self, duration, idle_sleep=0.001):
        end = time.time() + duration
        while time.time() < end:
            if self.step() == 0:
                time.sleep(idle_sleep)

    def close(self):
        self.sock.close()
