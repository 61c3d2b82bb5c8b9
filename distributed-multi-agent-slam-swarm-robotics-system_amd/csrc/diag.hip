// diag.hip -- measured latencies of the primitives a loop-closure decision is made of (qs_diag_latencies).
//
// qs_slam_chain_kernel (slam.hip) is a recurrence: decision k + 1 needs the drift decision k left behind.  What bounds it is
// not bandwidth but the DEPENDENT chain between two decisions: a workgroup barrier, an LDS round trip for the window, the
// arithmetic that turns a pose into nine bucket addresses, one L2 round trip for the node rows, the distance tests and the
// wave-wide minimum, the closure arithmetic, and an LDS write the next window can see.  bench.py prices that chain with the
// numbers measured here (`roofline.latency_floor`), so that "cycles per window" has something to be compared with.
// One workgroup, as the chain kernel runs: everything is timed with s_memtime on the wave that does it.
#include "qs_internal.h"

#define DG_CHASE 4096            // pointer-chase steps per measurement
#define DG_REP 2048              // repetitions of the register-only chains

__global__ void __launch_bounds__(1024)
qs_diag_lat_kernel(const unsigned int *__restrict__ chase_l2, const unsigned int *__restrict__ chase_l1, double *__restrict__ out)
{
    __shared__ unsigned int s_chase[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < 1024; t += 1024) s_chase[t] = (unsigned int)((t * 389 + 77) & 1023);
    __syncthreads();
    if (wave == 0) {
        // ---- dependent global loads: every address comes out of the previous load ----
        for (int pass = 0; pass < 2; pass++) {
            const unsigned int *p = pass ? chase_l1 : chase_l2;
            // every lane walks the same chain -- one cache line per step, the cheapest VECTOR load there is.  z is 0 in every lane
            // but only at run time (table entries are < 2^31): without it the compiler proves the address uniform and issues
            // scalar loads, which is not the path the node rows take
            const unsigned int z = chase_l1[lane] >> 31;
            unsigned int k = z;
            for (int i = 0; i < (pass ? 4096 : 64); i++) k = p[k] + z;      // warm: TLB, and the small buffer into L1
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < DG_CHASE; i++) k = p[k] + z;
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (k == 0xffffffffu) out[15] = 1.0;          // (keeps the chain alive)
            if (lane == 0) out[pass] = (double)(t1 - t0) / DG_CHASE;
        }
        {   // ---- dependent LDS reads ----
            unsigned int k = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < DG_CHASE; i++) k = s_chase[k];
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (k == 0xffffffffu) out[15] = 1.0;
            if (lane == 0) out[2] = (double)(t1 - t0) / DG_CHASE;
        }
        {   // ---- dependent fp64 arithmetic (v_fma_f64 feeding itself): what a lone wave pays per dependent VALU instruction ----
            double a = 1.0 + lane * 1e-9, b = 0.999999;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            #pragma unroll 16
            for (int i = 0; i < DG_REP; i++) a = __builtin_fma(a, b, 1e-12);
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (a == 123.456) out[15] = 1.0;
            if (lane == 0) out[3] = (double)(t1 - t0) / DG_REP;
        }
        {   // ---- dependent cross-lane steps (DPP row_shr feeding itself), as in the wave-wide minimum ----
            unsigned int v = (unsigned int)lane * 2654435761u;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            #pragma unroll 16
            for (int i = 0; i < DG_REP; i++) v = min(v + 1u, (unsigned int)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x111, 0xf, 0xf, false));
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (v == 0x12345u) out[15] = 1.0;
            if (lane == 0) out[4] = (double)(t1 - t0) / DG_REP / 2.0;      // two dependent instructions per step (add, dpp-min)
        }
        {   // ---- v_readlane -> SGPR -> VALU round trip (a winner's fields are read this way) ----
            int v = lane;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            #pragma unroll 16
            for (int i = 0; i < DG_REP; i++) v = __builtin_amdgcn_readlane(v, 5) + lane;
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (v == 0x12345) out[15] = 1.0;
            if (lane == 0) out[5] = (double)(t1 - t0) / DG_REP / 2.0;
        }
    }
    __syncthreads();
    // ---- workgroup barrier with LDS fences (slam.hip: lds_barrier), 16 waves then 5 waves arriving together ----
    for (int pass = 0; pass < 2; pass++) {
        const int n_waves = pass ? 5 : 16;
        if (wave < n_waves) {
            // (waves that do not take part must not be counted by s_barrier: they have to be gone -- so the 5-wave pass runs last
            // and the others leave first)
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 1024; i++) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (tid == 0) out[6 + pass] = (double)(t1 - t0) / 1024;
        }
        if (pass == 0 && wave >= 5) return;
    }
    if (tid == 0) {
        // shader clock against the 100 MHz reference counter
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0;
        while (r1 - r0 < 20000) r1 = __builtin_amdgcn_s_memrealtime();          // 200 us
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[8] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0;                   // MHz
    }
}

hipError_t qs_launch_diag_latencies(qs_ctx *c, const unsigned int *d_chase_l2, const unsigned int *d_chase_l1, double *d_out)
{
    hipLaunchKernelGGL(qs_diag_lat_kernel, dim3(1), dim3(1024), 0, c->stream, d_chase_l2, d_chase_l1, d_out);
    return hipGetLastError();
}
