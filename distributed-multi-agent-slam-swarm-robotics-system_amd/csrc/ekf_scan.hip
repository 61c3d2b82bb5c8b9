// ekf_scan.hip -- K5, batch form: the firmware's 6-state EKF over a whole batch, parallel in time.
// Semantics: AgentFirmware_Bot1/ekf.cpp:5-92 on the build-defined telemetry wiring of ekf.hip.
//
// ekf.hip walks a bot's packets one filter step at a time: a strict recurrence, ~0.75 us per step
// on a lone wave, 0.39 s for the 2-bot 1 M-packet batch -- 1000x the rest of the ingest.  This file
// computes the same filter with the recurrence cut into chunks of 128..1024 steps that are processed
// concurrently.  What makes that possible is the structure of this particular filter:
//
//   state  [x, y | theta, v, omega, bias];   measurement = [v, omega]
//
// (1) The sub-state s = (theta, v, omega, bias) is an exactly LINEAR Gaussian system
//         s' = F s + u,  F = [[1,0,0,-dt],[0,1,0,0],[0,0,0,-1],[0,0,0,1]],  u = (w_m dt, 0, w_m, 0)
//     (ekf.cpp:33-45 and the Jacobian rows :47-60) observed through H = rows (v, omega).  Kalman steps
//     of a linear system compose associatively (Sarkka & Garcia-Fernandez, "Temporal parallelization
//     of Bayesian smoothers", IEEE TAC 2021): a run of steps is an element (A, b, C, eta, J) that maps
//     the filter state at its start to the state at its end.  E1 builds one element per chunk, E2
//     pushes the bot's state through the chunk elements (one 4x4 solve per chunk), which gives every
//     chunk its exact start state.
// (2) Given theta and v along the trajectory, the position p = (x, y), its cross-covariance
//     B = Cov(p, s) and D = Cov(p, p) obey recurrences that are affine (p, B) and quadratic (D) in the
//     value of B at the chunk start, with coefficients that only involve quantities of (1).  E3
//     re-runs each chunk's 4x4 filter from its start state and accumulates those coefficients; E4
//     folds them over the chunks.
// (3) The heading wrap of ekf.cpp:40-41 (one +-2 pi per predict) never feeds back into the
//     filter (sin, cos and the covariance do not see it), so theta is carried unwrapped and the
//     number of wraps is replayed: every chunk evaluates the wrap rule for the three wrap counts its
//     start can have, E4 picks.
//
// The arithmetic is a re-association of the sequential filter's, so results agree with it to
// rounding (1e-13 relative on the CPU restatement of this file's algebra, tests/test_ekf_scan_math.py;
// north_star's bar for floats is 1e-5), not bit for bit.  Batches below ES_MIN_BATCH keep the
// serial kernel of ekf.hip (lower latency for a handful of packets).
#include "qs_internal.h"

#define ES_CHUNK_MIN 128          // filter steps per chunk: chosen per launch so that a bot's share of the
#define ES_CHUNK_MAX 1024         // batch is ~256 chunks (E2 / E4 walk a bot's chunks one after the other)
#define ES_PI 3.14159265358979323846
#define ES_TWO_PI 6.28318530717958647692

// per accepted record, bot-major, arrival order
struct EsRec { double t, om, ve; unsigned int kind, pad; };   // kind 0 init, 1 step, 2 nothing

// per chunk
struct EsAgg1 { double A[16], b[4], C[16], h[4], J[16]; };
struct EsStart { double s[4], A[16]; };
struct EsAgg2 { double L[16], N[8], m[4], q[2], W[16], U[8], V[4]; int wrap_c, wrap_out[3]; double last_out; };

struct EsWs {
    unsigned int *count;         // [256]   accepted records per bot
    unsigned int *base;          // [257]   exclusive prefix of count
    unsigned int chunk;          //         filter steps per chunk of this launch
    unsigned int *chunk_base;    // [257]   exclusive prefix of ceil(count / chunk)
    unsigned int *idx;           // [n]     packet index of every accepted record, bot-major
    unsigned int *tile_off;      // [257][n_tiles] records of a bot in a tile of ES_TILE packets -> where they start in idx
    EsRec *rec;                  // [n]
    unsigned long long *cmax;    // [chunks] max t over the chunk's init / step records (ordered key), 0 = none
    double *last_in;             // [chunks] filter's last predict time at chunk start
    EsAgg1 *agg1;                // [chunks]
    EsStart *start;              // [chunks]
    EsAgg2 *agg2;                // [chunks]
    double *fin;                 // [256][20] (s, A) after the batch
};

__device__ inline void es_sincos(double x, double *sn, double *cs)
{
    if (!(fabs(x) < 1.0e5)) { *sn = sin(x); *cs = cos(x); return; }
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = x - n * 1.57079632673412561417e+00;
    r = r - n * 6.07710050630396597660e-11;
    r = r - n * 2.02226624879595063154e-21;
    const double z = r * r;
    const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10,
                      -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                      8.33333333332248946124e-03);
    const double s = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                      -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                      2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c = 1.0 - (0.5 * z - z * (z * pc));
    const int q = (int)n & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

// ---- index: accepted records per bot, bot-major order (a stable partition by bot) ----------------
// one workgroup per tile of ES_TILE packets: the tile's records per bot -> tile_cnt[bot][tile], and the
// per-bot totals
#define ES_TILE 16384
#define ES_CMP_BLOCK 1024
#define ES_CMP_PER (ES_TILE / ES_CMP_BLOCK)
__global__ void __launch_bounds__(ES_CMP_BLOCK)
es_count_kernel(size_t n, QsBatch b, int max_agent, EsWs ws, unsigned int n_tiles)
{
    __shared__ unsigned int s_cnt[QS_MAX_AGENT + 1];
    for (int a = threadIdx.x; a <= QS_MAX_AGENT; a += ES_CMP_BLOCK) s_cnt[a] = 0;
    __syncthreads();
    const size_t i0 = (size_t)blockIdx.x * ES_TILE;
    for (int k = 0; k < ES_CMP_PER; k++) {
        const size_t i = i0 + (size_t)k * ES_CMP_BLOCK + threadIdx.x;
        if (i < n) {
            const int a = b.agent[i];
            if (b.map_ok[i] && a >= 1 && a <= max_agent) atomicAdd(&s_cnt[a], 1u);
        }
    }
    __syncthreads();
    for (int a = threadIdx.x; a <= QS_MAX_AGENT; a += ES_CMP_BLOCK) {
        const unsigned int v = s_cnt[a];
        ws.tile_off[(size_t)a * n_tiles + blockIdx.x] = v;
        if (v) atomicAdd(&ws.count[a], v);
    }
}

__global__ void __launch_bounds__(256)
es_plan_kernel(EsWs ws, int max_agent)
{
    __shared__ unsigned int s_a[256], s_c[256];
    const int a = threadIdx.x;
    const unsigned int cnt = (a >= 1 && a <= max_agent) ? ws.count[a] : 0u;
    const unsigned int chk = (cnt + ws.chunk - 1) / ws.chunk;
    s_a[a] = cnt; s_c[a] = chk;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned int va = 0, vc = 0;
        if (a >= off) { va = s_a[a - off]; vc = s_c[a - off]; }
        __syncthreads();
        s_a[a] += va; s_c[a] += vc;
        __syncthreads();
    }
    ws.base[a] = s_a[a] - cnt; ws.chunk_base[a] = s_c[a] - chk;
    if (a == 255) { ws.base[256] = s_a[255]; ws.chunk_base[256] = s_c[255]; }
}

// one wave per bot: where each tile's records of the bot start (its base + the tiles before)
__global__ void __launch_bounds__(QS_WAVE)
es_tile_scan_kernel(EsWs ws, unsigned int n_tiles)
{
    const int bot = blockIdx.x + 1, lane = threadIdx.x;
    unsigned int run = ws.base[bot];
    unsigned int *row = ws.tile_off + (size_t)bot * n_tiles;
    for (unsigned int t0 = 0; t0 < n_tiles; t0 += QS_WAVE) {
        const unsigned int t = t0 + lane;
        const unsigned int v = t < n_tiles ? row[t] : 0u;
        unsigned int inc = v;
        #pragma unroll
        for (int off = 1; off < QS_WAVE; off <<= 1) { const unsigned int u = __shfl_up(inc, off); if (lane >= off) inc += u; }
        if (t < n_tiles) row[t] = run + inc - v;
        run += __shfl(inc, QS_WAVE - 1);
    }
}

// one workgroup per (tile, bot): stable compaction of the bot's accepted packet indices of that tile
__global__ void __launch_bounds__(ES_CMP_BLOCK)
es_compact_kernel(size_t n, QsBatch b, EsWs ws, unsigned int n_tiles)
{
    __shared__ unsigned int s_wave[ES_CMP_BLOCK / QS_WAVE];
    const int bot = blockIdx.y + 1;
    const unsigned int tile = blockIdx.x;
    const unsigned int base = ws.tile_off[(size_t)bot * n_tiles + tile];
    const unsigned int next = tile + 1 < n_tiles ? ws.tile_off[(size_t)bot * n_tiles + tile + 1] : ws.base[bot + 1];
    if (next == base) return;                                  // none of the bot's records in this tile (uniform)
    const int tid = threadIdx.x, lane = tid & (QS_WAVE - 1), wave = tid >> 6;
    const size_t i0 = (size_t)tile * ES_TILE + (size_t)tid * ES_CMP_PER;
    unsigned int mask = 0;
    #pragma unroll
    for (int k = 0; k < ES_CMP_PER; k++) {
        const size_t i = i0 + k;
        if (i < n && b.map_ok[i] && b.agent[i] == bot) mask |= 1u << k;
    }
    const unsigned int cnt = (unsigned int)__builtin_popcount(mask);
    unsigned int inc = cnt;
    #pragma unroll
    for (int off = 1; off < QS_WAVE; off <<= 1) { const unsigned int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    if (lane == QS_WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    unsigned int before = 0;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    unsigned int pos = base + before + inc - cnt;
    #pragma unroll
    for (int k = 0; k < ES_CMP_PER; k++)
        if (mask & (1u << k)) ws.idx[pos++] = (unsigned int)(i0 + k);
}

__device__ inline int es_bot_of(const unsigned int *__restrict__ prefix, unsigned int v)
{
    // last bot a with prefix[a] <= v (prefix has 257 entries, non-decreasing)
    int lo = 0, hi = 256;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (prefix[mid] <= v) lo = mid; else hi = mid; }
    return lo;
}

// one thread per accepted record: the telemetry wiring of ekf.hip (inv_dt, w_m, v_enc), which only
// needs the bot's previous record
__global__ void __launch_bounds__(256)
es_wire_kernel(size_t n, QsBatch b, const double *__restrict__ recv_time, double t_nominal0, EsWs ws,
               const double *__restrict__ prev, double metres_per_tick)
{
    const unsigned int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ws.base[256]) return;
    const int bot = es_bot_of(ws.base, j);
    const unsigned int j0 = ws.base[bot];
    const unsigned int i = ws.idx[j];
    const double t = recv_time ? recv_time[i] : t_nominal0 + (double)i;
    const double yaw = b.yaw[i], enc = (double)b.enc[i];
    double tp, yawp, encp;
    bool have_prev = true;
    if (j > j0) {
        const unsigned int ip = ws.idx[j - 1];
        tp = recv_time ? recv_time[ip] : t_nominal0 + (double)ip;
        yawp = b.yaw[ip]; encp = (double)b.enc[ip];
    } else {
        tp = prev[4 * bot]; yawp = prev[4 * bot + 1]; encp = prev[4 * bot + 2];
        have_prev = prev[4 * bot + 3] != 0.0;
    }
    EsRec r; r.t = t; r.om = 0.0; r.ve = 0.0; r.kind = 2; r.pad = 0;
    if (!have_prev) r.kind = 0;
    else {
        const double dtp = t - tp;
        if (dtp > 0) {
            double dyaw = yaw - yawp;
            if (dyaw > ES_PI) dyaw -= 2 * ES_PI;
            else if (dyaw < -ES_PI) dyaw += 2 * ES_PI;
            const double inv_dt = 1.0 / dtp;
            r.om = dyaw * inv_dt;
            r.ve = (enc - encp) * metres_per_tick * inv_dt;
            r.kind = 1;
        }
    }
    ws.rec[j] = r;
    if (r.kind != 2) atomicMax(&ws.cmax[ws.chunk_base[bot] + (j - j0) / ws.chunk], qs_ord_from_double(t));
}

// ---- E0: last predict time at every chunk start (a prefix max per bot) ---------------------------
__global__ void __launch_bounds__(64)
es_last_kernel(EsWs ws, const double *__restrict__ ekf, const double *__restrict__ prev, int max_agent)
{
    const int bot = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (bot > max_agent || ws.count[bot] == 0) return;
    // a bot that has been seen carries its filter's last_t; a new one starts at its first record
    double last = prev[4 * bot + 3] != 0.0 ? ekf[(size_t)bot * 44 + 42] : -__builtin_inf();
    for (unsigned int c = ws.chunk_base[bot]; c < ws.chunk_base[bot + 1]; c++) {
        ws.last_in[c] = last;
        const unsigned long long k = ws.cmax[c];
        if (k != 0) { const double m = qs_double_from_ord(k); last = m > last ? m : last; }
    }
}

// ---- E1: one Kalman-scan element per chunk --------------------------------------------------------
// Extending the element (A, b, C, eta, J) of a run by one step (F, u, Q, z) is the ordinary Kalman
// step applied to the affine family the element describes:
//   Psi = H F A,  G = C (H F)^T,  S^ = H Q H^T + R + (H F) G,  rho = z - H (F b + u),  K = G S^^-1
//   A <- Dk F (A - K Psi);  b <- Dk F (b + K rho) + b2;  C <- Dk F (C - K G^T) F^T Dk + C2
//   eta <- eta + Psi^T S^^-1 rho;  J <- J + Psi^T S^^-1 Psi
// with Dk = I - K2 H, b2, C2 the step's own gain terms (K2 = Q H^T (H Q H^T + R)^-1, diagonal here).
struct EsStep { double dt, om, z0, z1; bool pred; };

__device__ inline void es_extend(EsAgg1 &e, const EsStep &st)
{
    const bool pr = st.pred;
    const double dt = st.dt, om = st.om;
    const double q0 = pr ? 0.01 : 0.0, q1 = pr ? 0.1 : 0.0, q2 = pr ? 0.1 : 0.0, q3 = pr ? 0.001 : 0.0;   // ekf.cpp:11 (theta, v, omega, bias)
    const double d0 = q1 + 0.05, d1 = q2 + 0.05;                                                       // ekf.cpp:12
    const double k0 = q1 / d0, k1 = q2 / d1;
    double g0[4], g1[4], p0[4], p1[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        g0[i] = e.C[4 * i + 1]; g1[i] = pr ? -e.C[4 * i + 3] : e.C[4 * i + 2];
        p0[i] = e.A[4 + i];     p1[i] = pr ? -e.A[12 + i] : e.A[8 + i];
    }
    const double s00 = d0 + g0[1], s01 = g1[1], s10 = pr ? -g0[3] : g0[2], s11 = d1 + (pr ? -g1[3] : g1[2]);
    const double idet = 1.0 / (s00 * s11 - s01 * s10);
    const double i00 = s11 * idet, i01 = -s01 * idet, i10 = -s10 * idet, i11 = s00 * idet;
    // F b + u
    const double fb1 = e.b[1], fb2 = pr ? om - e.b[3] : e.b[2];
    const double r0 = st.z0 - fb1, r1 = st.z1 - fb2;
    const double w0 = i00 * r0 + i01 * r1, w1 = i10 * r0 + i11 * r1;           // S^^-1 rho
    double K0[4], K1[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) { K0[i] = g0[i] * i00 + g1[i] * i10; K1[i] = g0[i] * i01 + g1[i] * i11; }
    // eta, J (use Psi before A changes)
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        e.h[i] += p0[i] * w0 + p1[i] * w1;
        const double t0 = i00 * p0[i] + i10 * p1[i], t1 = i01 * p0[i] + i11 * p1[i];   // (Psi^T S^^-1)[i][.]
        #pragma unroll
        for (int j = 0; j < 4; j++) e.J[4 * i + j] += t0 * p0[j] + t1 * p1[j];
    }
    // A' = A - K Psi, b' = b + K rho, C' = C - K G^T
    double A[16], C[16], bb[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        bb[i] = e.b[i] + K0[i] * r0 + K1[i] * r1;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            A[4 * i + j] = e.A[4 * i + j] - (K0[i] * p0[j] + K1[i] * p1[j]);
            C[4 * i + j] = e.C[4 * i + j] - (K0[i] * g0[j] + K1[i] * g1[j]);
        }
    }
    if (pr) {
        // rows: F M
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            const double a3 = A[12 + j], c3 = C[12 + j];
            A[j] -= dt * a3; A[8 + j] = -a3;
            C[j] -= dt * c3; C[8 + j] = -c3;
        }
        bb[0] = bb[0] - dt * bb[3] + om * dt; bb[2] = om - bb[3];
        // columns: C F^T
        #pragma unroll
        for (int i = 0; i < 4; i++) { const double c3 = C[4 * i + 3]; C[4 * i] -= dt * c3; C[4 * i + 2] = -c3; }
    }
    // Dk, b2, C2
    const double e1 = 1.0 - k0, e2 = 1.0 - k1;
    #pragma unroll
    for (int j = 0; j < 4; j++) { A[4 + j] *= e1; A[8 + j] *= e2; C[4 + j] *= e1; C[8 + j] *= e2; }
    #pragma unroll
    for (int i = 0; i < 4; i++) { C[4 * i + 1] *= e1; C[4 * i + 2] *= e2; }
    C[0] += q0; C[5] += e1 * q1; C[10] += e2 * q2; C[15] += q3;
    // b = Dk (F b') + u + K2 (z - H u): the u part is already in bb (rows of F b' + u), so
    // Dk applies to (F b' + u) - u ... written out: b_i = dk_i (F b')_i + u_i + k_i (z_i - u_i)
    {
        const double u1 = 0.0, u2 = pr ? om : 0.0;
        const double fb1n = bb[1] - u1, fb2n = bb[2] - u2;       // (F b')_1, (F b')_2
        bb[1] = e1 * fb1n + u1 + k0 * (st.z0 - u1);
        bb[2] = e2 * fb2n + u2 + k1 * (st.z1 - u2);
    }
    #pragma unroll
    for (int i = 0; i < 16; i++) { e.A[i] = A[i]; e.C[i] = C[i]; }
    #pragma unroll
    for (int i = 0; i < 4; i++) e.b[i] = bb[i];
}

__device__ inline bool es_next_step(const EsRec &r, double &last, EsStep &st)
{
    if (r.kind == 0) { last = r.t; return false; }               // init: handled as the prior
    if (r.kind != 1) return false;
    st.dt = r.t - last; st.pred = st.dt > 0;                     // `if (dt <= 0) return;` ekf.cpp:28-29 (update still runs)
    if (st.pred) last = r.t;
    st.om = r.om; st.z0 = r.ve; st.z1 = r.om;
    return true;
}

__global__ void __launch_bounds__(QS_WAVE)
es_agg1_kernel(EsWs ws)
{
    const unsigned int c = blockIdx.x * QS_WAVE + threadIdx.x;
    if (c >= ws.chunk_base[256]) return;
    const int bot = es_bot_of(ws.chunk_base, c);
    const unsigned int j0 = ws.base[bot] + (c - ws.chunk_base[bot]) * ws.chunk;
    const unsigned int j1 = min(j0 + ws.chunk, ws.base[bot + 1]);
    EsAgg1 e;
    #pragma unroll
    for (int i = 0; i < 16; i++) { e.A[i] = (i % 5 == 0) ? 1.0 : 0.0; e.C[i] = 0.0; e.J[i] = 0.0; }
    #pragma unroll
    for (int i = 0; i < 4; i++) { e.b[i] = 0.0; e.h[i] = 0.0; }
    double last = ws.last_in[c];
    EsRec r = ws.rec[j0];
    for (unsigned int j = j0; j < j1; j++) {
        const EsRec cur = r;
        if (j + 1 < j1) r = ws.rec[j + 1];
        EsStep st;
        if (es_next_step(cur, last, st)) es_extend(e, st);
    }
    ws.agg1[c] = e;
}

// ---- E2: push the bot's (s, A) through the chunk elements -> start state of every chunk ----------
// (m, P) o (A, b, C, eta, J):  X = (I + P J)^-1;  m' = A X (m + P eta) + b;  P' = A X P A^T + C
__device__ inline void es_apply(const EsAgg1 &e, double s[4], double P[16])
{
    double M[4][9];          // [ I + P J | m + P eta | P ]
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            double a = (i == j) ? 1.0 : 0.0;
            #pragma unroll
            for (int k = 0; k < 4; k++) a += P[4 * i + k] * e.J[4 * k + j];
            M[i][j] = a; M[i][5 + j] = P[4 * i + j];
        }
        double v = s[i];
        #pragma unroll
        for (int k = 0; k < 4; k++) v += P[4 * i + k] * e.h[k];
        M[i][4] = v;
    }
    // Gauss-Jordan with partial pivoting (I + P J has eigenvalues >= 1 but is not symmetric)
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        #pragma unroll
        for (int i = k + 1; i < 4; i++) {
            const bool sw = fabs(M[i][k]) > fabs(M[k][k]);
            #pragma unroll
            for (int j = 0; j < 9; j++) { const double a = M[k][j], bb = M[i][j]; M[k][j] = sw ? bb : a; M[i][j] = sw ? a : bb; }
        }
        const double ip = 1.0 / M[k][k];
        #pragma unroll
        for (int j = 0; j < 9; j++) M[k][j] *= ip;
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i == k) continue;
            const double f = M[i][k];
            #pragma unroll
            for (int j = 0; j < 9; j++) M[i][j] -= f * M[k][j];
        }
    }
    double T[16];            // A X P
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        double v = e.b[i];
        #pragma unroll
        for (int k = 0; k < 4; k++) v += e.A[4 * i + k] * M[k][4];
        s[i] = v;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            double a = 0.0;
            #pragma unroll
            for (int k = 0; k < 4; k++) a += e.A[4 * i + k] * M[k][5 + j];
            T[4 * i + j] = a;
        }
    }
    #pragma unroll
    for (int i = 0; i < 4; i++)
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            double a = e.C[4 * i + j];
            #pragma unroll
            for (int k = 0; k < 4; k++) a += T[4 * i + k] * e.A[4 * j + k];
            P[4 * i + j] = a;
        }
}

// prior of a bot for this batch: its stored state, or EKF::EKF + init at its first record (ekf.cpp:5-19)
__device__ inline void es_prior(const EsWs &ws, const QsBatch &b, const double *ekf, const double *prev, int bot,
                                double x[6], double P[36])
{
    if (prev[4 * bot + 3] != 0.0) {
        #pragma unroll
        for (int i = 0; i < 6; i++) x[i] = ekf[(size_t)bot * 44 + i];
        #pragma unroll
        for (int i = 0; i < 36; i++) P[i] = ekf[(size_t)bot * 44 + 6 + i];
    } else {
        const unsigned int i0 = ws.idx[ws.base[bot]];
        x[0] = b.px[i0]; x[1] = b.py[i0]; x[2] = b.yaw[i0]; x[3] = 0.0; x[4] = 0.0; x[5] = 0.0;
        #pragma unroll
        for (int i = 0; i < 36; i++) P[i] = (i % 7 == 0) ? 1.0 : 0.0;
    }
}

typedef unsigned long long __attribute__((may_alias)) es_word;      // a chunk's element / coefficients, moved a word per lane

// One WAVE per bot.  The walk over the bot's chunks is a strict recurrence, a 4 x 4 solve per chunk; what it must not do is
// fetch the chunk's element (56 doubles) a lane per bot -- 64 lanes x 56 uncoalesced loads kept the address unit busy for
// ~4 k cycles per chunk, more than the arithmetic.  Here the wave reads the element a lane per double (one coalesced request,
// asked for a chunk ahead), passes it through LDS, and every lane does the same arithmetic on it (same operations, same order
// as the lane-per-bot form: same doubles); lane 0 writes.
__global__ void __launch_bounds__(QS_WAVE)
es_apply_kernel(EsWs ws, QsBatch b, const double *__restrict__ ekf, const double *__restrict__ prev, int max_agent)
{
    constexpr int NW = (int)(sizeof(EsAgg1) / 8);
    static_assert(sizeof(EsAgg1) % 8 == 0 && sizeof(EsAgg1) / 8 <= QS_WAVE, "element = at most one word per lane");
    __shared__ EsAgg1 el[2];
    const int bot = blockIdx.x + 1, lane = threadIdx.x;
    if (bot > max_agent || ws.count[bot] == 0) return;
    double x[6], P[36], s[4], A[16];
    es_prior(ws, b, ekf, prev, bot, x, P);
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        s[i] = x[2 + i];
        #pragma unroll
        for (int j = 0; j < 4; j++) A[4 * i + j] = P[6 * (2 + i) + 2 + j];
    }
    const unsigned int c0 = ws.chunk_base[bot], c1 = ws.chunk_base[bot + 1];
    es_word nxt = (c0 < c1 && lane < NW) ? ((const es_word *)(ws.agg1 + c0))[lane] : 0ull;
    int par = 0;
    for (unsigned int c = c0; c < c1; c++, par ^= 1) {
        if (lane < NW) ((es_word *)&el[par])[lane] = nxt;
        if (c + 1 < c1 && lane < NW) nxt = ((const es_word *)(ws.agg1 + c + 1))[lane];
        if (lane == 0) {
            EsStart st;
            #pragma unroll
            for (int i = 0; i < 4; i++) st.s[i] = s[i];
            #pragma unroll
            for (int i = 0; i < 16; i++) st.A[i] = A[i];
            ws.start[c] = st;
        }
        const EsAgg1 e = el[par];
        es_apply(e, s, A);
    }
    if (lane == 0) {
        #pragma unroll
        for (int i = 0; i < 4; i++) ws.fin[(size_t)bot * 20 + i] = s[i];
        #pragma unroll
        for (int i = 0; i < 16; i++) ws.fin[(size_t)bot * 20 + 4 + i] = A[i];
    }
}

// ---- E3: per chunk, from its start state: the 4x4 filter, and the coefficients of (p, B, D) ------
// With B_in the value of B at the chunk start, B before step k is B_in L + N, and
//   p_out = p_in + B_in m + q
//   D_out = D_in - B_in W B_in^T + B_in U + (B_in U)^T + V
//   B_out = B_in L + N
// Per step (G = d(p')/d(s), T = F^T H^T, S = H A' H^T + R, y the innovation, A the covariance of s
// before the predict, A' after it):
//   mv = T S^-1 y;  M = F^T (I - H^T S^-1 H A');  Wk = T S^-1 T^T;  Uk = G^T - Wk (G A)^T
//   Vk = G A G^T + Qp - (G A) Wk (G A)^T
//   q += N mv + v (cos, sin) dt + G A mv;   m += L mv
//   V += -N Wk N^T + N Uk + (N Uk)^T + Vk;  U += L (Uk - Wk N^T);  W += L Wk L^T
//   N <- (N + G A) M;  L <- L M
__global__ void __launch_bounds__(QS_WAVE)
es_agg2_kernel(EsWs ws)
{
    const unsigned int c = blockIdx.x * QS_WAVE + threadIdx.x;
    if (c >= ws.chunk_base[256]) return;
    const int bot = es_bot_of(ws.chunk_base, c);
    const unsigned int j0 = ws.base[bot] + (c - ws.chunk_base[bot]) * ws.chunk;
    const unsigned int j1 = min(j0 + ws.chunk, ws.base[bot + 1]);
    double s[4], A[16];
    {
        const EsStart st = ws.start[c];
        #pragma unroll
        for (int i = 0; i < 4; i++) s[i] = st.s[i];
        #pragma unroll
        for (int i = 0; i < 16; i++) A[i] = st.A[i];
    }
    double L[16], N[8], m[4], q[2], W[16], U[8], V[4];
    #pragma unroll
    for (int i = 0; i < 16; i++) { L[i] = (i % 5 == 0) ? 1.0 : 0.0; W[i] = 0.0; }
    #pragma unroll
    for (int i = 0; i < 8; i++) { N[i] = 0.0; U[i] = 0.0; }
    #pragma unroll
    for (int i = 0; i < 4; i++) { m[i] = 0.0; V[i] = 0.0; }
    q[0] = 0.0; q[1] = 0.0;
    const int wc = (int)rint(s[0] * (1.0 / ES_TWO_PI));
    int wn0 = wc - 1, wn1 = wc, wn2 = wc + 1;
    double last = ws.last_in[c];
    EsRec r = ws.rec[j0];
    for (unsigned int j = j0; j < j1; j++) {
        const EsRec cur = r;
        if (j + 1 < j1) r = ws.rec[j + 1];
        EsStep st;
        if (!es_next_step(cur, last, st)) continue;
        const bool pr = st.pred;
        const double dt = st.dt, om = st.om;
        // ---- predict of s and A; G, G A, G A G^T ----
        double sp[4], Ap[16], GA[8], GAG[4], G0[2], G1[2], q0[2];
        if (pr) {
            double sn, cs;
            es_sincos(s[0], &sn, &cs);                                        // ekf.cpp:36-37 use the old theta
            const double v = s[1];
            G0[0] = -v * sn * dt; G0[1] = v * cs * dt;                        // d p' / d theta   ekf.cpp:50-53
            G1[0] = cs * dt;      G1[1] = sn * dt;                            // d p' / d v
            q0[0] = v * cs * dt;  q0[1] = v * sn * dt;                        // ekf.cpp:36-37
            sp[0] = s[0] - dt * s[3] + om * dt; sp[1] = s[1]; sp[2] = om - s[3]; sp[3] = s[3];   // :33-45
            // A' = F A F^T + Qs
            double T4[16];
            #pragma unroll
            for (int j2 = 0; j2 < 4; j2++) {
                T4[j2] = A[j2] - dt * A[12 + j2]; T4[4 + j2] = A[4 + j2]; T4[8 + j2] = -A[12 + j2]; T4[12 + j2] = A[12 + j2];
            }
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                Ap[4 * i] = T4[4 * i] - dt * T4[4 * i + 3]; Ap[4 * i + 1] = T4[4 * i + 1];
                Ap[4 * i + 2] = -T4[4 * i + 3]; Ap[4 * i + 3] = T4[4 * i + 3];
            }
            Ap[0] += 0.01; Ap[5] += 0.1; Ap[10] += 0.1; Ap[15] += 0.001;      // ekf.cpp:11
            #pragma unroll
            for (int rr = 0; rr < 2; rr++)
                #pragma unroll
                for (int j2 = 0; j2 < 4; j2++) GA[4 * rr + j2] = G0[rr] * A[j2] + G1[rr] * A[4 + j2];
            #pragma unroll
            for (int rr = 0; rr < 2; rr++)
                #pragma unroll
                for (int cc = 0; cc < 2; cc++) GAG[2 * rr + cc] = GA[4 * rr] * G0[cc] + GA[4 * rr + 1] * G1[cc];
            // wrap rule of ekf.cpp:40-41 for the three possible wrap counts at the chunk start
            const double a = sp[0];
            { const double w = a - ES_TWO_PI * wn0; if (w > ES_PI) wn0++; else if (w < -ES_PI) wn0--; }
            { const double w = a - ES_TWO_PI * wn1; if (w > ES_PI) wn1++; else if (w < -ES_PI) wn1--; }
            { const double w = a - ES_TWO_PI * wn2; if (w > ES_PI) wn2++; else if (w < -ES_PI) wn2--; }
        } else {
            G0[0] = G0[1] = G1[0] = G1[1] = q0[0] = q0[1] = 0.0;
            #pragma unroll
            for (int i = 0; i < 4; i++) sp[i] = s[i];
            #pragma unroll
            for (int i = 0; i < 16; i++) Ap[i] = A[i];
            #pragma unroll
            for (int i = 0; i < 8; i++) GA[i] = 0.0;
            #pragma unroll
            for (int i = 0; i < 4; i++) GAG[i] = 0.0;
        }
        // ---- S^-1, innovation ----
        const double s00 = Ap[5] + 0.05, s01 = Ap[6], s10 = Ap[9], s11 = Ap[10] + 0.05;      // ekf.cpp:76
        const double idet = 1.0 / (s00 * s11 - s01 * s10);
        const double i00 = s11 * idet, i01 = -s01 * idet, i10 = -s10 * idet, i11 = s00 * idet;
        const double y0 = st.z0 - sp[1], y1 = st.z1 - sp[2];
        const double w0 = i00 * y0 + i01 * y1, w1 = i10 * y0 + i11 * y1;     // S^-1 y
        // X T = [X[:,1], f(X)] with f = -col 3 (predict) or col 2
        #define ES_TCOL(X, row, stride) (pr ? -(X)[(row) * (stride) + 3] : (X)[(row) * (stride) + 2])
        double mv[4];
        mv[0] = 0.0; mv[1] = w0; mv[2] = pr ? 0.0 : w1; mv[3] = pr ? -w1 : 0.0;
        double R2[8];                                                        // S^-1 A'[{1,2}, :]
        #pragma unroll
        for (int j2 = 0; j2 < 4; j2++) {
            R2[j2] = i00 * Ap[4 + j2] + i01 * Ap[8 + j2];
            R2[4 + j2] = i10 * Ap[4 + j2] + i11 * Ap[8 + j2];
        }
        double NT[4], GAT[4], LT[8];
        #pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            NT[2 * rr] = N[4 * rr + 1];   NT[2 * rr + 1] = ES_TCOL(N, rr, 4);
            GAT[2 * rr] = GA[4 * rr + 1]; GAT[2 * rr + 1] = ES_TCOL(GA, rr, 4);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) { LT[2 * i] = L[4 * i + 1]; LT[2 * i + 1] = ES_TCOL(L, i, 4); }
        // q, m
        #pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            double a = q0[rr];
            #pragma unroll
            for (int k = 0; k < 4; k++) a += (N[4 * rr + k] + GA[4 * rr + k]) * mv[k];
            q[rr] += a;
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            double a = 0.0;
            #pragma unroll
            for (int k = 0; k < 4; k++) a += L[4 * i + k] * mv[k];
            m[i] += a;
        }
        // Si * X^T helpers: SiNT = S^-1 NT^T (2x2), SiGAT, and their sum
        double SiNT[4], SiGAT[4];
        #pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            SiNT[cc] = i00 * NT[2 * cc] + i01 * NT[2 * cc + 1];      SiNT[2 + cc] = i10 * NT[2 * cc] + i11 * NT[2 * cc + 1];
            SiGAT[cc] = i00 * GAT[2 * cc] + i01 * GAT[2 * cc + 1];   SiGAT[2 + cc] = i10 * GAT[2 * cc] + i11 * GAT[2 * cc + 1];
        }
        // V += -NT Si NT^T + NU + NU^T + Vk,  NU = N G^T - NT Si GAT^T,  Vk = GAG + Qp - GAT Si GAT^T
        {
            double NU[4], VK[4], NN[4];
            #pragma unroll
            for (int rr = 0; rr < 2; rr++)
                #pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    NU[2 * rr + cc] = N[4 * rr] * G0[cc] + N[4 * rr + 1] * G1[cc]
                                      - (NT[2 * rr] * SiGAT[cc] + NT[2 * rr + 1] * SiGAT[2 + cc]);
                    NN[2 * rr + cc] = NT[2 * rr] * SiNT[cc] + NT[2 * rr + 1] * SiNT[2 + cc];
                    VK[2 * rr + cc] = GAG[2 * rr + cc] - (GAT[2 * rr] * SiGAT[cc] + GAT[2 * rr + 1] * SiGAT[2 + cc]);
                }
            if (pr) { VK[0] += 0.01; VK[3] += 0.01; }                        // ekf.cpp:11 (x, y)
            V[0] += -NN[0] + NU[0] + NU[0] + VK[0];
            V[1] += -NN[1] + NU[1] + NU[2] + VK[1];
            V[2] += -NN[2] + NU[2] + NU[1] + VK[2];
            V[3] += -NN[3] + NU[3] + NU[3] + VK[3];
        }
        // U += L G^T - LT Si (GAT + NT)^T ;  W += LT Si LT^T
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const double l0 = LT[2 * i], l1 = LT[2 * i + 1];
            #pragma unroll
            for (int cc = 0; cc < 2; cc++)
                U[2 * i + cc] += L[4 * i] * G0[cc] + L[4 * i + 1] * G1[cc]
                                 - (l0 * (SiGAT[cc] + SiNT[cc]) + l1 * (SiGAT[2 + cc] + SiNT[2 + cc]));
            const double t0 = l0 * i00 + l1 * i10, t1 = l0 * i01 + l1 * i11;
            #pragma unroll
            for (int j2 = 0; j2 < 4; j2++) W[4 * i + j2] += t0 * LT[2 * j2] + t1 * LT[2 * j2 + 1];
        }
        // N <- (N + GA) M,  L <- L M,  X M = X F^T - (X T) R2
        #pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            double X[4];
            #pragma unroll
            for (int k = 0; k < 4; k++) X[k] = N[4 * rr + k] + GA[4 * rr + k];
            const double xt0 = NT[2 * rr] + GAT[2 * rr], xt1 = NT[2 * rr + 1] + GAT[2 * rr + 1];
            double XF[4];
            if (pr) { XF[0] = X[0] - dt * X[3]; XF[1] = X[1]; XF[2] = -X[3]; XF[3] = X[3]; }
            else { XF[0] = X[0]; XF[1] = X[1]; XF[2] = X[2]; XF[3] = X[3]; }
            #pragma unroll
            for (int k = 0; k < 4; k++) N[4 * rr + k] = XF[k] - (xt0 * R2[k] + xt1 * R2[4 + k]);
        }
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const double x0 = L[4 * i], x1 = L[4 * i + 1], x2 = L[4 * i + 2], x3 = L[4 * i + 3];
            const double xt0 = LT[2 * i], xt1 = LT[2 * i + 1];
            double XF[4];
            if (pr) { XF[0] = x0 - dt * x3; XF[1] = x1; XF[2] = -x3; XF[3] = x3; }
            else { XF[0] = x0; XF[1] = x1; XF[2] = x2; XF[3] = x3; }
            #pragma unroll
            for (int k = 0; k < 4; k++) L[4 * i + k] = XF[k] - (xt0 * R2[k] + xt1 * R2[4 + k]);
        }
        #undef ES_TCOL
        // ---- update of s and A   ekf.cpp:70-92 ----
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const double a1 = Ap[4 * i + 1], a2 = Ap[4 * i + 2];
            const double k0 = a1 * i00 + a2 * i10, k1 = a1 * i01 + a2 * i11;
            s[i] = sp[i] + (k0 * y0 + k1 * y1);
            #pragma unroll
            for (int j2 = 0; j2 < 4; j2++) A[4 * i + j2] = Ap[4 * i + j2] - (k0 * Ap[4 + j2] + k1 * Ap[8 + j2]);
        }
    }
    EsAgg2 o;
    #pragma unroll
    for (int i = 0; i < 16; i++) { o.L[i] = L[i]; o.W[i] = W[i]; }
    #pragma unroll
    for (int i = 0; i < 8; i++) { o.N[i] = N[i]; o.U[i] = U[i]; }
    #pragma unroll
    for (int i = 0; i < 4; i++) { o.m[i] = m[i]; o.V[i] = V[i]; }
    o.q[0] = q[0]; o.q[1] = q[1];
    o.wrap_c = wc; o.wrap_out[0] = wn0; o.wrap_out[1] = wn1; o.wrap_out[2] = wn2;
    o.last_out = last;
    ws.agg2[c] = o;
}

// ---- E4: fold (p, B, D) and the wrap count over the chunks, write the bot's state back ------------
// (one wave per bot, the chunk's coefficients read a lane per word and a chunk ahead: as es_apply_kernel)
__global__ void __launch_bounds__(QS_WAVE)
es_fold_kernel(EsWs ws, QsBatch b, const double *__restrict__ recv_time, double t_nominal0,
                               double *__restrict__ ekf, double *__restrict__ prev, int max_agent,
                               unsigned long long *__restrict__ counters)
{
    constexpr int NW = (int)(sizeof(EsAgg2) / 8);
    static_assert(sizeof(EsAgg2) % 8 == 0 && sizeof(EsAgg2) / 8 <= QS_WAVE, "coefficients = at most one word per lane");
    __shared__ EsAgg2 el[2];
    const int bot = blockIdx.x + 1, lane = threadIdx.x;
    if (bot > max_agent || ws.count[bot] == 0) return;
    double x[6], P[36];
    es_prior(ws, b, ekf, prev, bot, x, P);
    double p[2] = {x[0], x[1]}, B[8], D[4];
    #pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        #pragma unroll
        for (int k = 0; k < 4; k++) B[4 * rr + k] = P[6 * rr + 2 + k];
        D[2 * rr] = P[6 * rr]; D[2 * rr + 1] = P[6 * rr + 1];
    }
    int nw = 0;
    double last = prev[4 * bot + 3] != 0.0 ? ekf[(size_t)bot * 44 + 42] : -__builtin_inf();
    const unsigned int c0 = ws.chunk_base[bot], c1 = ws.chunk_base[bot + 1];
    es_word nxt = (c0 < c1 && lane < NW) ? ((const es_word *)(ws.agg2 + c0))[lane] : 0ull;
    int par = 0;
    for (unsigned int c = c0; c < c1; c++, par ^= 1) {
        if (lane < NW) ((es_word *)&el[par])[lane] = nxt;
        if (c + 1 < c1 && lane < NW) nxt = ((const es_word *)(ws.agg2 + c + 1))[lane];
        const EsAgg2 a = el[par];
        double BU[4], BW[8];
        #pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            double acc = a.q[rr];
            #pragma unroll
            for (int k = 0; k < 4; k++) acc += B[4 * rr + k] * a.m[k];
            p[rr] += acc;
            #pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                double u = 0.0;
                #pragma unroll
                for (int k = 0; k < 4; k++) u += B[4 * rr + k] * a.U[2 * k + cc];
                BU[2 * rr + cc] = u;
            }
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                double w = 0.0;
                #pragma unroll
                for (int k = 0; k < 4; k++) w += B[4 * rr + k] * a.W[4 * k + j];
                BW[4 * rr + j] = w;
            }
        }
        #pragma unroll
        for (int rr = 0; rr < 2; rr++)
            #pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                double bwb = 0.0;
                #pragma unroll
                for (int k = 0; k < 4; k++) bwb += BW[4 * rr + k] * B[4 * cc + k];
                D[2 * rr + cc] += -bwb + BU[2 * rr + cc] + BU[2 * cc + rr] + a.V[2 * rr + cc];
            }
        double Bn[8];
        #pragma unroll
        for (int rr = 0; rr < 2; rr++)
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                double acc = a.N[4 * rr + j];
                #pragma unroll
                for (int k = 0; k < 4; k++) acc += B[4 * rr + k] * a.L[4 * k + j];
                Bn[4 * rr + j] = acc;
            }
        #pragma unroll
        for (int i = 0; i < 8; i++) B[i] = Bn[i];
        // wrap count: the chunk evaluated the rule for wrap_c - 1, wrap_c, wrap_c + 1
        int k = nw - a.wrap_c + 1;
        if (k < 0 || k > 2) { if (lane == 0) atomicAdd(&counters[QS_CNT_EKF_WRAP_CLAMP], 1ull); k = k < 0 ? 0 : 2; }
        nw = k == 0 ? a.wrap_out[0] : k == 1 ? a.wrap_out[1] : a.wrap_out[2];      // (an index would send the whole struct to scratch)
        last = a.last_out;
    }
    if (lane != 0) return;
    const double *fin = ws.fin + (size_t)bot * 20;
    double *f = ekf + (size_t)bot * 44;
    f[0] = p[0]; f[1] = p[1]; f[2] = fin[0] - ES_TWO_PI * (double)nw; f[3] = fin[1]; f[4] = fin[2]; f[5] = fin[3];
    #pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        f[6 + 6 * rr] = D[2 * rr]; f[6 + 6 * rr + 1] = D[2 * rr + 1];
        #pragma unroll
        for (int k = 0; k < 4; k++) { f[6 + 6 * rr + 2 + k] = B[4 * rr + k]; f[6 + 6 * (2 + k) + rr] = B[4 * rr + k]; }
    }
    #pragma unroll
    for (int i = 0; i < 4; i++)
        #pragma unroll
        for (int j = 0; j < 4; j++) f[6 + 6 * (2 + i) + 2 + j] = fin[4 + 4 * i + j];
    f[42] = last; f[43] = 1.0;
    const unsigned int il = ws.idx[ws.base[bot + 1] - 1];
    prev[4 * bot] = recv_time ? recv_time[il] : t_nominal0 + (double)il;
    prev[4 * bot + 1] = b.yaw[il]; prev[4 * bot + 2] = (double)b.enc[il]; prev[4 * bot + 3] = 1.0;
}

// ---- host side ------------------------------------------------------------------------------------
static inline size_t es_align(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t es_tiles(size_t n) { return (n + 16384 - 1) / 16384 + 1; }

static size_t es_max_chunks(const qs_ctx *c, size_t n) { return n / ES_CHUNK_MIN + (size_t)c->cfg.max_agent + 2; }

static unsigned int es_chunk_for(const qs_ctx *c, size_t n)
{
    // ~256 chunks for a bot with an even share of the batch
    size_t want = n / ((size_t)c->cfg.max_agent * 256);
    unsigned int ch = ES_CHUNK_MIN;
    while (ch < ES_CHUNK_MAX && ch < want) ch <<= 1;
    return ch;
}

size_t qs_ekf_scan_workspace_bytes(const qs_ctx *c, size_t n)
{
    const size_t ch = es_max_chunks(c, n);
    return es_align(256 * 4) + 2 * es_align(257 * 4) + es_align(n * 4) + es_align(257 * es_tiles(n) * 4) + es_align(n * sizeof(EsRec)) + es_align(ch * 8) +
           es_align(ch * 8) + es_align(ch * sizeof(EsAgg1)) + es_align(ch * sizeof(EsStart)) + es_align(ch * sizeof(EsAgg2)) +
           es_align(256 * 20 * 8);
}

hipError_t qs_launch_ekf_scan(qs_ctx *c, size_t n, const double *d_time, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const size_t need = qs_ekf_scan_workspace_bytes(c, c->cap_batch);
    if (need > c->ekf_ws_bytes) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        if (c->d_ekf_ws) { hipFree(c->d_ekf_ws); c->d_ekf_ws = nullptr; c->ekf_ws_bytes = 0; }
        e = hipMalloc(&c->d_ekf_ws, need);
        if (e != hipSuccess) return e;
        c->ekf_ws_bytes = need;
    }
    const size_t cap = c->cap_batch, ch = es_max_chunks(c, cap);
    EsWs ws;
    ws.chunk = es_chunk_for(c, n);
    char *p = (char *)c->d_ekf_ws;
    ws.count = (unsigned int *)p; p += es_align(256 * 4);
    ws.cmax = (unsigned long long *)p; p += es_align(ch * 8);           // adjacent to count: one memset clears both
    ws.base = (unsigned int *)p; p += es_align(257 * 4);
    ws.chunk_base = (unsigned int *)p; p += es_align(257 * 4);
    ws.idx = (unsigned int *)p; p += es_align(cap * 4);
    ws.tile_off = (unsigned int *)p; p += es_align(257 * es_tiles(cap) * 4);
    ws.rec = (EsRec *)p; p += es_align(cap * sizeof(EsRec));
    ws.last_in = (double *)p; p += es_align(ch * 8);
    ws.agg1 = (EsAgg1 *)p; p += es_align(ch * sizeof(EsAgg1));
    ws.start = (EsStart *)p; p += es_align(ch * sizeof(EsStart));
    ws.agg2 = (EsAgg2 *)p; p += es_align(ch * sizeof(EsAgg2));
    ws.fin = (double *)p;
    hipError_t e = hipMemsetAsync(ws.count, 0, es_align(256 * 4) + es_align(ch * 8), st);
    if (e != hipSuccess) return e;
    const int ma = c->cfg.max_agent;
    const double t0 = (double)c->next_seq;
    const unsigned int chunks = (unsigned int)(n / ws.chunk + (size_t)c->cfg.max_agent + 2);
    const unsigned int n_tiles = (unsigned int)((n + ES_TILE - 1) / ES_TILE);
    hipLaunchKernelGGL(es_count_kernel, dim3(n_tiles), dim3(ES_CMP_BLOCK), 0, st, n, c->b, ma, ws, n_tiles);
    hipLaunchKernelGGL(es_plan_kernel, dim3(1), dim3(256), 0, st, ws, ma);
    hipLaunchKernelGGL(es_tile_scan_kernel, dim3(ma), dim3(QS_WAVE), 0, st, ws, n_tiles);
    hipLaunchKernelGGL(es_compact_kernel, dim3(n_tiles, ma), dim3(ES_CMP_BLOCK), 0, st, n, c->b, ws, n_tiles);
    hipLaunchKernelGGL(es_wire_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, st, n, c->b, d_time, t0, ws,
                       c->d_ekf_prev, c->cfg.ekf_metres_per_tick);
    hipLaunchKernelGGL(es_last_kernel, dim3((ma + 63) / 64), dim3(64), 0, st, ws, c->d_ekf, c->d_ekf_prev, ma);
    hipLaunchKernelGGL(es_agg1_kernel, dim3((chunks + QS_WAVE - 1) / QS_WAVE), dim3(QS_WAVE), 0, st, ws);
    hipLaunchKernelGGL(es_apply_kernel, dim3(ma), dim3(QS_WAVE), 0, st, ws, c->b, c->d_ekf, c->d_ekf_prev, ma);
    hipLaunchKernelGGL(es_agg2_kernel, dim3((chunks + QS_WAVE - 1) / QS_WAVE), dim3(QS_WAVE), 0, st, ws);
    hipLaunchKernelGGL(es_fold_kernel, dim3(ma), dim3(QS_WAVE), 0, st, ws, c->b, d_time, t0, c->d_ekf, c->d_ekf_prev, ma,
                       c->d_counters);
    return hipGetLastError();
}
