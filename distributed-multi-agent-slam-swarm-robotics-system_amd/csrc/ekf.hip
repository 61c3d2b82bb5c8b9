// ekf.hip -- K5: the firmware's 6-state EKF, batched per bot.
// Semantics: AgentFirmware_Bot1/ekf.cpp:5-92 (state layout ekf.h:38-44):
//   x = [x, y, theta, v, omega, bias_omega], P0 = I, Q = diag(.01,.01,.01,.1,.1,.001),
//   R = diag(.05,.05); predict() on the gyro rate, update() on the encoder [v, omega].
// The Jacobian is the identity plus seven entries, so J P J^T and (I - K H) P are evaluated
// row/column-sparse; every kept term is summed in the same (ascending-k) order as a dense
// triple loop, so the result equals the dense product exactly (the dropped terms are +-0).
//
// The mapper never uses the EKF pose (dual_bot_mapper.py:829-831 takes the packet pose; the
// firmware itself runs the filter "for internal state estimation", AgentFirmware_Bot1.ino:
// 697-707), so this stage is telemetry: it must not feed the raycast.
#include "qs_internal.h"

#define EKF_STRIDE 44   // x[6], P[36], last_time, initialized
#define EKF_PI 3.14159265358979323846

// sin/cos for the filter's heading (|theta| stays near [-pi, pi]): Cody-Waite reduction by pi/2 in
// three pieces and the classic degree-13/14 minimax kernels; ~35 instructions instead of two
// general-range library calls on the critical path of a strictly serial recurrence.  (The raycast
// keeps the library's sin/cos: there the value decides a cell index.)
__device__ inline void ekf_sincos(double x, double *sn, double *cs)
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

struct EkfState { double x[6]; double P[36]; double last_t; double init; };

__device__ inline void ekf_load(const double *f, EkfState &s)
{
    #pragma unroll
    for (int i = 0; i < 6; i++) s.x[i] = f[i];
    #pragma unroll
    for (int i = 0; i < 36; i++) s.P[i] = f[6 + i];
    s.last_t = f[42]; s.init = f[43];
}
__device__ inline void ekf_store(double *f, const EkfState &s)
{
    #pragma unroll
    for (int i = 0; i < 6; i++) f[i] = s.x[i];
    #pragma unroll
    for (int i = 0; i < 36; i++) f[6 + i] = s.P[i];
    f[42] = s.last_t; f[43] = s.init;
}
__device__ inline void ekf_init(EkfState &s, double t, const double x0[6])   // ekf.cpp:5-19
{
    #pragma unroll
    for (int i = 0; i < 6; i++) s.x[i] = x0[i];
    #pragma unroll
    for (int r = 0; r < 6; r++)
        #pragma unroll
        for (int c = 0; c < 6; c++) s.P[6 * r + c] = (r == c) ? 1.0 : 0.0;
    s.last_t = t; s.init = 1.0;
}

__device__ inline void ekf_predict(EkfState &s, double omega_measured, double t)   // ekf.cpp:26-68
{
    if (s.init == 0.0) return;
    const double dt = t - s.last_t;
    if (!(dt > 0)) return;                                // `if (dt <= 0) return;` (NaN: skip too)
    s.last_t = t;
    const double theta = s.x[2], v = s.x[3], bias = s.x[5];
    const double omega_c = omega_measured - bias;
    double theta_new = theta + omega_c * dt;
    if (theta_new > EKF_PI) theta_new -= 2 * EKF_PI;
    else if (theta_new < -EKF_PI) theta_new += 2 * EKF_PI;
    double ct, st;
    ekf_sincos(theta, &st, &ct);
    const double x_new = s.x[0] + v * ct * dt;
    const double y_new = s.x[1] + v * st * dt;
    s.x[0] = x_new; s.x[1] = y_new; s.x[2] = theta_new; s.x[4] = omega_c;
    const double j02 = -v * st * dt, j03 = ct * dt, j12 = v * ct * dt, j13 = st * dt, j25 = -dt;
    double JP[36];
    #pragma unroll
    for (int c = 0; c < 6; c++) {
        JP[0 * 6 + c] = (s.P[0 * 6 + c] + j02 * s.P[2 * 6 + c]) + j03 * s.P[3 * 6 + c];
        JP[1 * 6 + c] = (s.P[1 * 6 + c] + j12 * s.P[2 * 6 + c]) + j13 * s.P[3 * 6 + c];
        JP[2 * 6 + c] = s.P[2 * 6 + c] + j25 * s.P[5 * 6 + c];
        JP[3 * 6 + c] = s.P[3 * 6 + c];
        JP[4 * 6 + c] = -1.0 * s.P[5 * 6 + c];            // J(omega,omega) = 0, J(omega,bias) = -1
        JP[5 * 6 + c] = s.P[5 * 6 + c];
    }
    const double Q[6] = {0.01, 0.01, 0.01, 0.1, 0.1, 0.001};                            // ekf.cpp:11
    #pragma unroll
    for (int r = 0; r < 6; r++) {
        const double a0 = (JP[6 * r + 0] + JP[6 * r + 2] * j02) + JP[6 * r + 3] * j03;
        const double a1 = (JP[6 * r + 1] + JP[6 * r + 2] * j12) + JP[6 * r + 3] * j13;
        const double a2 = JP[6 * r + 2] + JP[6 * r + 5] * j25;
        const double a3 = JP[6 * r + 3];
        const double a4 = JP[6 * r + 5] * -1.0;
        const double a5 = JP[6 * r + 5];
        s.P[6 * r + 0] = a0 + (r == 0 ? Q[0] : 0.0);
        s.P[6 * r + 1] = a1 + (r == 1 ? Q[1] : 0.0);
        s.P[6 * r + 2] = a2 + (r == 2 ? Q[2] : 0.0);
        s.P[6 * r + 3] = a3 + (r == 3 ? Q[3] : 0.0);
        s.P[6 * r + 4] = a4 + (r == 4 ? Q[4] : 0.0);
        s.P[6 * r + 5] = a5 + (r == 5 ? Q[5] : 0.0);
    }
}

__device__ inline void ekf_update(EkfState &s, double z_v, double z_omega)   // ekf.cpp:70-92
{
    if (s.init == 0.0) return;
    const double R0 = 0.05, R1 = 0.05;                                                  // ekf.cpp:12
    const double y0 = z_v - s.x[3], y1 = z_omega - s.x[4];
    const double s00 = s.P[3 * 6 + 3] + R0, s01 = s.P[3 * 6 + 4];
    const double s10 = s.P[4 * 6 + 3], s11 = s.P[4 * 6 + 4] + R1;
    const double det = s00 * s11 - s01 * s10;
    const double invdet = 1.0 / det;
    const double i00 = s11 * invdet, i01 = -s01 * invdet, i10 = -s10 * invdet, i11 = s00 * invdet;
    double K0[6], K1[6];
    #pragma unroll
    for (int r = 0; r < 6; r++) {
        const double p3 = s.P[6 * r + 3], p4 = s.P[6 * r + 4];
        K0[r] = p3 * i00 + p4 * i10;
        K1[r] = p3 * i01 + p4 * i11;
    }
    #pragma unroll
    for (int r = 0; r < 6; r++) s.x[r] = s.x[r] + (K0[r] * y0 + K1[r] * y1);
    double P3[6], P4[6];
    #pragma unroll
    for (int c = 0; c < 6; c++) { P3[c] = s.P[3 * 6 + c]; P4[c] = s.P[4 * 6 + c]; }
    #pragma unroll
    for (int c = 0; c < 6; c++) {
        // (I - K H) P, rows in ascending k: k = r term, then k = 3, then k = 4 (r < 3);
        // k = 3, k = 4, then k = 5 (r = 5); rows 3 and 4 carry (1 - K) on their own column.
        const double n0 = (s.P[0 * 6 + c] + (-K0[0]) * P3[c]) + (-K1[0]) * P4[c];
        const double n1 = (s.P[1 * 6 + c] + (-K0[1]) * P3[c]) + (-K1[1]) * P4[c];
        const double n2 = (s.P[2 * 6 + c] + (-K0[2]) * P3[c]) + (-K1[2]) * P4[c];
        const double n3 = (1.0 - K0[3]) * P3[c] + (-K1[3]) * P4[c];
        const double n4 = (-K0[4]) * P3[c] + (1.0 - K1[4]) * P4[c];
        const double n5 = ((-K0[5]) * P3[c] + (-K1[5]) * P4[c]) + s.P[5 * 6 + c];
        s.P[0 * 6 + c] = n0; s.P[1 * 6 + c] = n1; s.P[2 * 6 + c] = n2;
        s.P[3 * 6 + c] = n3; s.P[4 * 6 + c] = n4; s.P[5 * 6 + c] = n5;
    }
}

// ---- ingest wiring (build-defined; modelled on esp32_firmware/src/main.cpp:176-188) --------
// Per accepted packet of a bot, in arrival order: the first packet initialises the filter at the
// packet pose; later packets (dt = t - t_prev > 0) derive inv_dt = 1/dt,
// omega_m = wrap(yaw - yaw_prev) * inv_dt, v_enc = (enc - enc_prev) * metres_per_tick * inv_dt and run
// predict(omega_m, t); update(v_enc, omega_m).
//
// The filter is a strict recurrence per bot and a lone wave issues about one instruction every
// 4-8 cycles, so the serial path is kept short: one wave per bot; everything that does not depend
// on the filter state (the wiring above, with its division) is done SIMD across the 64 records of
// a chunk before the serial loop; the 6x6 covariance is spread over 36 lanes (lane l = 6r + c holds
// P[r][c] and a copy of x[r]) so J P J^T and (I - K H) P cost eight 64-bit lane permutes per step
// instead of ~450 fp64 instructions on one lane.  Every element is evaluated with the operand order
// of ekf_predict / ekf_update above (terms that are absent there appear here as + 0.0 * x), so the
// two forms agree bit for bit.
#define EKF_BLOCK 256

__device__ inline double ekf_rl(double v, int src_lane)      // wave-uniform read of one lane
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}
__device__ inline double ekf_perm(double v, int byte_index)   // per-lane read of another lane (index * 4)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_index, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_index, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(EKF_BLOCK)
qs_ekf_ingest_kernel(size_t n, QsBatch b, const double *__restrict__ recv_time, double t_nominal0,
                     double *__restrict__ ekf, double *__restrict__ prev, int max_agent, double metres_per_tick)
{
    const int lane = threadIdx.x & 63;
    const int bot = blockIdx.x * (EKF_BLOCK / QS_WAVE) + (threadIdx.x >> 6) + 1;
    if (bot > max_agent) return;
    const int l = lane < 36 ? lane : 0;
    const int r = l / 6, c = l % 6;
    double *f = ekf + (size_t)bot * EKF_STRIDE;
    double P = f[6 + l], xr = f[r];
    double last_t = f[42], init = f[43];
    double pt = prev[4 * bot], pyaw = prev[4 * bot + 1], penc = prev[4 * bot + 2], seen = prev[4 * bot + 3];
    const double Qd[6] = {0.01, 0.01, 0.01, 0.1, 0.1, 0.001};                          // ekf.cpp:11
    const double qdiag = (r == c) ? Qd[r] : 0.0;
    const double R0 = 0.05, R1 = 0.05;                                                  // ekf.cpp:12
    // lane constants of the permutes (byte indices) and of the term selection
    const int col_a = 4 * ((r <= 1) ? 12 + c : (r == 2 || r == 4) ? 30 + c : l);   // J P: rows 2 / 5 of my column
    const int col_b = 4 * ((r <= 1) ? 18 + c : l);                                  //      row 3 of my column
    const int row_a = 4 * ((c <= 1) ? 6 * r + 2 : (c == 2 || c == 4) ? 6 * r + 5 : l);   // (JP) J^T: cols 2 / 5 of my row
    const int row_b = 4 * ((c <= 1) ? 6 * r + 3 : l);                                     //           col 3 of my row
    const int row_3 = 4 * (6 * r + 3), row_4 = 4 * (6 * r + 4), col_3 = 4 * (18 + c), col_4 = 4 * (24 + c);

    // The stream is scanned 4 x 64 records per iteration; the accept / agent bytes of the NEXT group
    // are requested before the current group is processed, so the scan never waits on them.
    #define EKF_GROUP 4
    unsigned char acc_n[EKF_GROUP], ag_n[EKF_GROUP];
    #pragma unroll
    for (int q = 0; q < EKF_GROUP; q++) {
        const size_t i = (size_t)q * QS_WAVE + lane;
        acc_n[q] = i < n ? b.map_ok[i] : 0; ag_n[q] = i < n ? b.agent[i] : 0;
    }
    for (size_t gbase = 0; gbase < n; gbase += EKF_GROUP * QS_WAVE) {
        unsigned char acc_c[EKF_GROUP], ag_c[EKF_GROUP];
        #pragma unroll
        for (int q = 0; q < EKF_GROUP; q++) { acc_c[q] = acc_n[q]; ag_c[q] = ag_n[q]; }
        #pragma unroll
        for (int q = 0; q < EKF_GROUP; q++) {
            const size_t i = gbase + (size_t)(EKF_GROUP + q) * QS_WAVE + lane;
            acc_n[q] = i < n ? b.map_ok[i] : 0; ag_n[q] = i < n ? b.agent[i] : 0;
        }
      #pragma unroll
      for (int sub = 0; sub < EKF_GROUP; sub++) {
        const size_t base = gbase + (size_t)sub * QS_WAVE;
        const size_t i = base + lane;
        const bool mine = i < n && acc_c[sub] && ag_c[sub] == bot;
        unsigned long long m = __ballot(mine);
        if (!m) continue;
        // ---- SIMD prologue: the chunk's fields (one coalesced load per array) and the wiring -------
        double t_l = 0, px_l = 0, py_l = 0, yaw_l = 0, enc_l = 0;
        if (mine) {
            t_l = recv_time ? recv_time[i] : t_nominal0 + (double)i;
            px_l = b.px[i]; py_l = b.py[i]; yaw_l = b.yaw[i]; enc_l = (double)b.enc[i];
        }
        const unsigned long long below = m & ((1ull << lane) - 1);
        const int pl = below ? 63 - __clzll((long long)below) : lane;          // previous record of this bot
        double tp = __shfl(t_l, pl), yawp = __shfl(yaw_l, pl), encp = __shfl(enc_l, pl);
        const bool chained = below != 0;
        if (!chained) { tp = pt; yawp = pyaw; encp = penc; }
        int kind_l = 2;                                                        // 0 init, 1 step, 2 nothing
        double om_l = 0, ve_l = 0;
        if (mine) {
            if (!chained && seen == 0.0) kind_l = 0;
            else {
                const double dtp = t_l - tp;
                if (dtp > 0) {
                    double dyaw = yaw_l - yawp;
                    if (dyaw > EKF_PI) dyaw -= 2 * EKF_PI;
                    else if (dyaw < -EKF_PI) dyaw += 2 * EKF_PI;
                    const double inv_dt = 1.0 / dtp;
                    om_l = dyaw * inv_dt;
                    ve_l = (enc_l - encp) * metres_per_tick * inv_dt;
                    kind_l = 1;
                }
            }
        }
        {   // carry the last record of this bot into the next chunk
            const int hi = 63 - __clzll((long long)m);
            pt = ekf_rl(t_l, hi); pyaw = ekf_rl(yaw_l, hi); penc = ekf_rl(enc_l, hi); seen = 1.0;
        }
        // ---- serial part: one filter step per record, in arrival order ------------------------------
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int kind = __builtin_amdgcn_readlane(kind_l, j);
            if (kind == 0) {                                          // EKF::EKF + init  ekf.cpp:5-19
                const double x0[6] = {ekf_rl(px_l, j), ekf_rl(py_l, j), ekf_rl(yaw_l, j), 0, 0, 0};
                xr = x0[r]; P = (r == c) ? 1.0 : 0.0; last_t = ekf_rl(t_l, j); init = 1.0;
            } else if (kind == 1 && init != 0.0) {
                // the predict-phase permutes only need last step's P: issue them first so that their
                // LDS-pipeline latency hides behind the readlanes / sincos below
                const double pa = ekf_perm(P, col_a), pb = ekf_perm(P, col_b);
                const double t = ekf_rl(t_l, j), omega_m = ekf_rl(om_l, j), v_enc = ekf_rl(ve_l, j);
                // ---- predict  ekf.cpp:26-68 ----
                const double dt = t - last_t;
                if (dt > 0) {
                    last_t = t;
                    const double theta = ekf_rl(xr, 12), v = ekf_rl(xr, 18), bias = ekf_rl(xr, 30);
                    const double omega_c = omega_m - bias;
                    double theta_new = theta + omega_c * dt;
                    if (theta_new > EKF_PI) theta_new -= 2 * EKF_PI;
                    else if (theta_new < -EKF_PI) theta_new += 2 * EKF_PI;
                    double ct, st;
                    ekf_sincos(theta, &st, &ct);
                    const double vct = v * ct, vst = v * st;
                    const double j02 = -v * st * dt, j03 = ct * dt, j12 = vct * dt, j13 = st * dt, j25 = -dt;
                    if (r == 0) xr = xr + vct * dt;
                    else if (r == 1) xr = xr + vst * dt;
                    else if (r == 2) xr = theta_new;
                    else if (r == 4) xr = omega_c;
                    // J P: (base + k1 * P[ra][c]) + k2 * P[3][c]
                    const double k1 = (r == 0) ? j02 : (r == 1) ? j12 : (r == 2) ? j25 : (r == 4) ? -1.0 : 0.0;
                    const double k2 = (r == 0) ? j03 : (r == 1) ? j13 : 0.0;
                    const double JP = (((r == 4) ? 0.0 : P) + k1 * pa) + k2 * pb;
                    // (J P) J^T: (base + JP[r][ca] * k1') + JP[r][3] * k2'
                    const double h1 = (c == 0) ? j02 : (c == 1) ? j12 : (c == 2) ? j25 : (c == 4) ? -1.0 : 0.0;
                    const double h2 = (c == 0) ? j03 : (c == 1) ? j13 : 0.0;
                    const double M = (((c == 4) ? 0.0 : JP) + ekf_perm(JP, row_a) * h1) + ekf_perm(JP, row_b) * h2;
                    P = M + qdiag;
                }
                // ---- update  ekf.cpp:70-92 ----
                const double pr3 = ekf_perm(P, row_3), pr4 = ekf_perm(P, row_4);
                const double p3c = ekf_perm(P, col_3), p4c = ekf_perm(P, col_4);
                const double y0 = v_enc - ekf_rl(xr, 18), y1 = omega_m - ekf_rl(xr, 24);
                const double s00 = ekf_rl(P, 21) + R0, s01 = ekf_rl(P, 22);
                const double s10 = ekf_rl(P, 27), s11 = ekf_rl(P, 28) + R1;
                const double det = s00 * s11 - s01 * s10;
                const double invdet = 1.0 / det;
                const double i00 = s11 * invdet, i01 = -s01 * invdet, i10 = -s10 * invdet, i11 = s00 * invdet;
                const double K0 = pr3 * i00 + pr4 * i10, K1 = pr3 * i01 + pr4 * i11;
                xr = xr + (K0 * y0 + K1 * y1);
                const double A = (r == 3) ? 1.0 - K0 : -K0, B = (r == 4) ? 1.0 - K1 : -K1;
                const double tt = (((r < 3) ? P : 0.0) + A * p3c) + B * p4c;
                P = (r == 5) ? tt + P : tt;
            }
        }
      }
    }
    if (lane < 36) {
        f[6 + lane] = P;
        if (c == 0) f[r] = xr;
    }
    if (lane == 0) {
        f[42] = last_t; f[43] = init;
        prev[4 * bot] = pt; prev[4 * bot + 1] = pyaw; prev[4 * bot + 2] = penc; prev[4 * bot + 3] = seen;
    }
}

hipError_t qs_launch_ekf_ingest(qs_ctx *c, size_t n, const double *d_time, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int waves = EKF_BLOCK / QS_WAVE;
    hipLaunchKernelGGL(qs_ekf_ingest_kernel, dim3((c->cfg.max_agent + waves - 1) / waves), dim3(EKF_BLOCK), 0,
                       st, n, c->b, d_time, (double)c->next_seq, c->d_ekf, c->d_ekf_prev, c->cfg.max_agent,
                       c->cfg.ekf_metres_per_tick);
    return hipGetLastError();
}

// ---- batched object API: one predict (+ update) per listed bot -----------------------------
__global__ void qs_ekf_step_kernel(const int *__restrict__ bots, const double *__restrict__ omega,
                                   const double *__restrict__ t, const double *__restrict__ zv,
                                   const double *__restrict__ zo, size_t n, int do_update, int max_agent,
                                   double *__restrict__ ekf)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int bot = bots[k];
    if (bot < 1 || bot > max_agent) return;
    EkfState s;
    ekf_load(ekf + (size_t)bot * EKF_STRIDE, s);
    ekf_predict(s, omega[k], t[k]);
    if (do_update) ekf_update(s, zv[k], zo[k]);
    ekf_store(ekf + (size_t)bot * EKF_STRIDE, s);
}

hipError_t qs_launch_ekf_step(qs_ctx *c, const int *d_bots, const double *d_omega, const double *d_t,
                              const double *d_zv, const double *d_zo, size_t n, int do_update)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_ekf_step_kernel, dim3((unsigned int)((n + 63) / 64)), dim3(64), 0, c->stream, d_bots,
                       d_omega, d_t, d_zv, d_zo, n, do_update, c->cfg.max_agent, c->d_ekf);
    return hipGetLastError();
}
