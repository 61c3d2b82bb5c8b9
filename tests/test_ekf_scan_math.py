"""CPU check of the algebra behind csrc/ekf_scan.hip (the EKF over a batch, parallel in time).

The kernel file cuts a bot's filter steps into chunks: (E1) one Kalman-scan element per chunk for the
linear sub-state (theta, v, omega, bias), (E2) the bot's state pushed through the chunk elements, (E3)
each chunk re-run from its start state while accumulating the coefficients of the (x, y) rows, (E4)
a fold of those coefficients and of the heading-wrap count.  This file restates exactly those four
steps in numpy and compares the result with the sequential filter of oracle/oracle.c (which follows
AgentFirmware_Bot1/ekf.cpp:5-92) on streams with varying, zero and negative time steps."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

QS = np.diag([0.01, 0.1, 0.1, 0.001])          # ekf.cpp:11, (theta, v, omega, bias)
QP = 0.01 * np.eye(2)                          # (x, y)
R = 0.05 * np.eye(2)                           # ekf.cpp:12
H = np.zeros((2, 4)); H[0, 1] = 1; H[1, 2] = 1
E = H.T
I4 = np.eye(4)
PI = np.pi


def step_model(dt, om):
    F = np.eye(4); F[0, 3] = -dt; F[2, 2] = 0; F[2, 3] = -1
    return F, np.array([om * dt, 0, om, 0.0])


def extend(e, dt, om, z, pred):
    """es_extend: element of a run followed by one more step."""
    A1, b1, C1, h1, J1 = e
    if pred:
        (F, u), q = step_model(dt, om), np.array([0.01, 0.1, 0.1, 0.001])
    else:
        F, u, q = np.eye(4), np.zeros(4), np.zeros(4)
    d0, d1 = q[1] + 0.05, q[2] + 0.05
    k0, k1 = q[1] / d0, q[2] / d1
    Phi = F[1:3, :]
    G1 = C1 @ Phi.T
    Psi = Phi @ A1
    Shi = np.linalg.inv(np.diag([d0, d1]) + Phi @ G1)
    rho = z - (F @ b1 + u)[1:3]
    Kt = G1 @ Shi
    A2 = np.diag([1, 1 - k0, 1 - k1, 1.0]) @ F
    b2 = u.copy(); b2[1] += k0 * (z[0] - u[1]); b2[2] += k1 * (z[1] - u[2])
    C2 = np.diag([q[0], (1 - k0) * q[1], (1 - k1) * q[2], q[3]])
    return (A2 @ (A1 - Kt @ Psi), A2 @ (b1 + Kt @ rho) + b2, A2 @ (C1 - Kt @ G1.T) @ A2.T + C2,
            h1 + Psi.T @ Shi @ rho, J1 + Psi.T @ Shi @ Psi)


def apply(e, m, P):
    """es_apply: a filter state through an element."""
    A2, b2, C2, h2, J2 = e
    X = np.linalg.inv(I4 + P @ J2)
    return A2 @ X @ (m + P @ h2) + b2, A2 @ X @ P @ A2.T + C2


def wiring(t, yaw, enc, mpt):
    """(dt, omega_m, v_enc, predict?) of every step record; the first record initialises the filter."""
    steps, last_t = [], t[0]
    for k in range(1, len(t)):
        dtp = t[k] - t[k - 1]
        if not dtp > 0:
            continue
        dyaw = yaw[k] - yaw[k - 1]
        if dyaw > PI: dyaw -= 2 * PI
        elif dyaw < -PI: dyaw += 2 * PI
        inv = 1.0 / dtp
        dt = t[k] - last_t
        pred = dt > 0
        if pred: last_t = t[k]
        steps.append((dt, dyaw * inv, (enc[k] - enc[k - 1]) * mpt * inv, pred))
    return steps


def scan_filter(t, px, py, yaw, enc, mpt, chunk):
    steps = wiring(t, yaw, enc, mpt)
    chunks = [steps[i:i + chunk] for i in range(0, len(steps), chunk)]
    # E1
    aggs = []
    for ch in chunks:
        e = (np.eye(4), np.zeros(4), np.zeros((4, 4)), np.zeros(4), np.zeros((4, 4)))
        for dt, om, ve, pred in ch:
            e = extend(e, dt, om, np.array([ve, om]), pred)
        aggs.append(e)
    # E2
    s, A = np.array([yaw[0], 0, 0, 0.0]), np.eye(4)
    starts = []
    for e in aggs:
        starts.append((s.copy(), A.copy()))
        s, A = apply(e, s, A)
    s_fin, A_fin = s, A
    # E3
    agg2 = []
    for ch, (s, A) in zip(chunks, starts):
        L = np.eye(4); N = np.zeros((2, 4)); m = np.zeros(4); q = np.zeros(2)
        Wc = np.zeros((4, 4)); Uc = np.zeros((4, 2)); Vc = np.zeros((2, 2))
        wc = int(np.rint(s[0] / (2 * PI)))
        wn = [wc - 1, wc, wc + 1]
        for dt, om, ve, pred in ch:
            z = np.array([ve, om]); th, v = s[0], s[1]
            if pred:
                F, u = step_model(dt, om); c, sn = np.cos(th), np.sin(th)
                G = np.array([[-v * sn * dt, c * dt, 0, 0], [v * c * dt, sn * dt, 0, 0]])
                q0 = v * np.array([c, sn]) * dt
                Ap = F @ A @ F.T + QS; sp = F @ s + u; Qpp = QP
                for i in range(3):
                    w = sp[0] - 2 * PI * wn[i]
                    if w > PI: wn[i] += 1
                    elif w < -PI: wn[i] -= 1
            else:
                F = np.eye(4); G = np.zeros((2, 4)); q0 = np.zeros(2); Ap = A; sp = s; Qpp = np.zeros((2, 2))
            Si = np.linalg.inv(H @ Ap @ H.T + R); y = z - H @ sp
            T = F.T @ E; M = F.T @ (I4 - E @ Si @ E.T @ Ap); W = T @ Si @ T.T; GA = G @ A; mv = T @ Si @ y
            U = G.T - W @ GA.T; V = G @ A @ G.T + Qpp - GA @ W @ GA.T
            q = q + N @ mv + q0 + GA @ mv; m = m + L @ mv
            Vc = Vc - N @ W @ N.T + N @ U + (N @ U).T + V
            Uc = Uc + L @ (U - W @ N.T)
            Wc = Wc + L @ W @ L.T
            N = (N + GA) @ M; L = L @ M
            Ks = Ap @ E @ Si; s = sp + Ks @ y; A = Ap - Ks @ E.T @ Ap
        agg2.append((L, N, m, q, Wc, Uc, Vc, wc, wn))
    # E4
    p = np.array([px[0], py[0]]); B = np.zeros((2, 4)); D = np.eye(2); nw = 0
    for L, N, m, q, Wc, Uc, Vc, wc, wn in agg2:
        p = p + B @ m + q
        D = D - B @ Wc @ B.T + B @ Uc + (B @ Uc).T + Vc
        B = B @ L + N
        assert 0 <= nw - wc + 1 <= 2
        nw = wn[nw - wc + 1]
    x = np.concatenate([p, [s_fin[0] - 2 * PI * nw], s_fin[1:]])
    return x, np.block([[D, B], [B.T, A_fin]]), nw


@pytest.mark.parametrize("n,chunk,dts,probs,seed", [
    (4000, 64, [1.0, 0.5, 0.05, 0.0, 2.0], [.6, .2, .1, .05, .05], 1),
    (12000, 256, [1.0], [1.0], 2),
    (6000, 128, [0.05, 0.0, -0.05, 0.1], [.7, .1, .1, .1], 3),
    (3000, 1024, [0.25], [1.0], 4),
])
def test_chunked_filter_equals_sequential(n, chunk, dts, probs, seed):
    rng = np.random.default_rng(seed)
    t = np.cumsum(rng.choice(dts, size=n, p=probs)) + 10.0
    yaw = np.cumsum(rng.normal(0, 0.2, n)) + 0.3
    yaw = (yaw + PI) % (2 * PI) - PI
    enc = np.cumsum(rng.integers(0, 6, n)).astype(float)
    px, py = rng.normal(0, 1, n), rng.normal(0, 1, n)
    mpt = 0.0107
    ek = orc.OracleEKF(1)
    for k in range(n):
        ek.packet(0, t[k], px[k], py[k], yaw[k], enc[k], mpt)
    x, P, nw = scan_filter(t, px, py, yaw, enc, mpt, chunk)
    x_ref, P_ref = ek.state(0), ek.cov(0)
    assert np.abs(x - x_ref).max() < 1e-9 * max(1.0, np.abs(x_ref).max())
    assert np.abs(P - P_ref).max() < 1e-9 * np.abs(P_ref).max()
