"""Single-process pure-Python restatement of the reference's per-packet path.

TEST INFRASTRUCTURE ONLY (like oracle.c): imported by tests/ and by bench.py's `cpu_baseline` leg, never by
the product package.  Purpose: (1) the "single-process Python baseline" of SURVEY.md 8(d) D5 -- the
reference's own .py files do not travel to the GPU box, so the interpreter cost of its loop structure is
measured on this restatement (same data structures: an np.int8 grid written cell by cell, Python lists for
the Bresenham cells, nodes, landmarks and closures); (2) a second, independent statement of the path that
oracle.c is checked against on the CPU (tests/test_oracle_golden.py).

Follows /root/reference/server_nodes/dual_bot_mapper.py:
  unpack + filters :826-843, offset/drift :850-857, rays + trust filter :882-903, update_ray :136-156,
  _bresenham :158-179, world_to_grid :121-125, add_pose/_check_closure :273-326, closure use :908-914.
Parity status: pinned through the same reference-generated fixtures as oracle.c (tests/golden).
"""
import math
import struct

import numpy as np

V2 = struct.Struct("<4sBfffiIffffB")      # :41
V1 = struct.Struct("<4sBfffiIffff")       # :45
SENSOR_OFFSETS = (0.0, math.pi / 2, math.pi, -math.pi / 2)     # front, left, back, right  :61-66
MIN_D, MAX_D = 0.05, 1.20                 # :57-58
RADIUS, GAP, DAMP = 0.60, 30, 0.5         # :97-99


class PyMapper:
    def __init__(self, size=200, res=0.05, ox=-5.0, oy=-5.0, separation=0.0):
        self.size, self.res, self.ox, self.oy, self.separation = size, res, ox, oy, separation
        self.grid = np.full((size, size), -1, dtype=np.int8)            # :119
        self.n_nodes = 0
        self.landmarks = []                # (x, y, type, node index)   :269
        self.closures = []                 # (lm index, node index, dx, dy)   :270
        self.last_closure = {1: -GAP, 2: -GAP}                          # :271
        self.drift = {1: [0.0, 0.0], 2: [0.0, 0.0]}                     # :782
        self.accepted = 0

    # -- OccupancyGrid -------------------------------------------------------------------------
    def _cell(self, wx, wy):                                            # :121-125
        return int((wx - self.ox) / self.res), int((wy - self.oy) / self.res)

    @staticmethod
    def _line(x0, y0, x1, y1):                                          # :158-179
        out = []
        dx, dy = abs(x1 - x0), abs(y1 - y0)
        sx = 1 if x0 < x1 else -1
        sy = 1 if y0 < y1 else -1
        err = dx - dy
        while True:
            out.append((x0, y0))
            if x0 == x1 and y0 == y1:
                return out
            e2 = 2 * err
            if e2 > -dy:
                err -= dy
                x0 += sx
            if e2 < dx:
                err += dx
                y0 += sy

    def update_ray(self, rx, ry, hx, hy, valid):                        # :136-156
        n = self.size
        cells = self._line(*self._cell(rx, ry), *self._cell(hx, hy))
        g = self.grid
        for gx, gy in cells[:-1]:
            if 0 <= gx < n and 0 <= gy < n:
                g[gy, gx] = 0
        if valid:
            gx, gy = cells[-1]
            if 0 <= gx < n and 0 <= gy < n:
                g[gy, gx] = 100

    # -- PoseGraphSLAM ---------------------------------------------------------------------------
    def add_pose(self, x, y, agent, lm):                                # :273-326
        idx = self.n_nodes
        self.n_nodes += 1
        if lm == 0:
            return None
        found = None
        if idx - self.last_closure.get(agent, -999) >= GAP:            # (loop-invariant test of :303-304 hoisted)
            for lx, ly, lt, li in self.landmarks:
                if lt != lm or idx - li < GAP:
                    continue
                if math.sqrt((x - lx) ** 2 + (y - ly) ** 2) < RADIUS:
                    found = ((lx - x) * DAMP, (ly - y) * DAMP)
                    self.closures.append((li, idx, found[0], found[1]))
                    self.last_closure[agent] = idx
                    break
        self.landmarks.append((x, y, lm, idx))
        return found

    # -- recv-loop body :826-919 -------------------------------------------------------------------
    def feed(self, data):
        if len(data) == V2.size:
            magic, agent, rx, ry, ryaw, _enc, _v2v, d0, d1, d2, d3, lm = V2.unpack(data)
        elif len(data) == V1.size:
            magic, agent, rx, ry, ryaw, _enc, _v2v, d0, d1, d2, d3 = V1.unpack(data)
            lm = 0
        else:
            return False
        if magic != b"QSRL" or agent not in (1, 2):
            return False
        if not (math.isfinite(rx) and math.isfinite(ry) and math.isfinite(ryaw)):
            return False                    # the reference would raise in int(); the build drops the packet
        if agent == 2:
            rx += self.separation
        dr = self.drift[agent]
        rx += dr[0]
        ry += dr[1]
        for off, dist in zip(SENSOR_OFFSETS, (d0, d1, d2, d3)):
            a = ryaw + off
            if MIN_D < dist <= MAX_D:
                self.update_ray(rx, ry, rx + dist * math.cos(a), ry + dist * math.sin(a), True)
            else:
                r = min(dist, MAX_D) if dist > MIN_D else MAX_D
                self.update_ray(rx, ry, rx + r * math.cos(a), ry + r * math.sin(a), False)
        corr = self.add_pose(rx, ry, agent, lm)
        if corr is not None:
            dr[0] += corr[0]
            dr[1] += corr[1]
        self.accepted += 1
        return True

    def feed_stream(self, buf, lengths=None):
        """buf: uint8 [n, stride] array; lengths: per-record datagram lengths or None (all == stride)."""
        raw = np.ascontiguousarray(buf, dtype=np.uint8)
        stride = raw.shape[1]
        mem = raw.tobytes()
        for i in range(raw.shape[0]):
            ln = stride if lengths is None else int(lengths[i])
            self.feed(mem[i * stride:i * stride + ln])
        return self.accepted
