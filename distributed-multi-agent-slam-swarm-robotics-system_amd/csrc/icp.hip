// icp.hip -- point-to-point ICP and voxel down-sampling for the map merger.
// Call sites in the reference: server_nodes/map_merger.py:45-60
//     registration_icp(local, global, 1.0, I4, TransformationEstimationPointToPoint(),
//                      ICPConvergenceCriteria(max_iteration=30));  fitness < 0.6 -> reject;
//     local.transform(T); global += local; global.voxel_down_sample(voxel_size=res)
// The arithmetic lives in Open3D (absent here, no version pinned, no reference test) => PARITY
// UNPINNED.  Restated from Open3D's published algorithm:
//   correspondences  : for every source point its nearest target point if closer than the
//                      threshold (hybrid search, max_nn = 1); fitness = #corr / #source,
//                      inlier_rmse = sqrt(sum d^2 / #corr);
//   update           : Umeyama without scaling on the correspondences; for planar clouds (z = 0,
//                      as grid_to_pcd produces) that is the closed-form 2-D Kabsch rotation
//                      theta = atan2(sum(ax*by - ay*bx), sum(ax*bx + ay*by)) on demeaned pairs;
//   loop             : at most max_iteration updates; stop when |d fitness| < 1e-6 and |d rmse| < 1e-6.
// Nearest neighbours are exact brute force in fp64 (targets staged through LDS tiles; ties -> lowest
// target index); sums are two-level and fixed-order, so results are reproducible run to run.
#include "qs_internal.h"

#define ICP_BLOCK 256

// ---- nearest target of every source point --------------------------------------------------------
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_nn_kernel(const double2 *__restrict__ src, size_t n_src, const double2 *__restrict__ dst, size_t n_dst,
                 double max_d2, int *__restrict__ corr, double *__restrict__ d2_out)
{
    __shared__ double2 s_t[ICP_BLOCK];
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    const double2 p = i < n_src ? src[i] : make_double2(0, 0);
    double best = INFINITY;
    int best_j = -1;
    for (size_t base = 0; base < n_dst; base += ICP_BLOCK) {
        const size_t j = base + threadIdx.x;
        s_t[threadIdx.x] = j < n_dst ? dst[j] : make_double2(INFINITY, INFINITY);
        __syncthreads();
        const int lim = (int)((n_dst - base < ICP_BLOCK) ? n_dst - base : ICP_BLOCK);
        for (int k = 0; k < lim; k++) {
            const double dx = p.x - s_t[k].x, dy = p.y - s_t[k].y;
            const double d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; best_j = (int)(base + k); }
        }
        __syncthreads();
    }
    if (i < n_src) {
        const bool ok = best_j >= 0 && best < max_d2;
        corr[i] = ok ? best_j : -1;
        d2_out[i] = ok ? best : 0.0;
    }
}

// ---- the same search with the distance matrix on the matrix cores ------------------------------------------
// The source x target squared-distance matrix is the one dense contraction of this code base:
//     d2(i, j) = |s_i|^2 + ( |t_j|^2 - 2 s_i . t_j )  =  |s_i|^2 + [sx, sy, 1, 0] . [-2 tx, -2 ty, |t|^2, 0]^T,
// K = 4 exactly: one v_mfma_f64_16x16x4_f64 gives the bracket S(i, j) for 16 sources x 16 targets.  The matrix value
// is a SCREEN, never the decision: it differs from the reference expression (dx*dx + dy*dy on the raw coordinates) by
// rounding -- coordinates are centred to keep the cancellation small -- so every source row keeps a threshold: the lowest
// exact squared distance it has seen, as an S, plus a margin (2^-40 (|s'|^2 + max |t'|^2), ~40 times the worst-case
// difference between S + |s'|^2 and the reference expression); an element at or under that threshold is re-evaluated
// with the reference expression in fp64 and competes on (d2, target index).  The true nearest target always passes
// (its S is within the margin of every other S of its row), so the result is the scalar kernel's, bit for bit, ties
// included.
// What makes it fast is keeping the screen tight and the re-evaluation off the matrix pipe's critical path:
//   * one wave owns NNM_ROWT row tiles of 16 sources; a B fragment (16 targets) feeds NNM_ROWT MFMAs;
//   * two accumulator sets: tile n + 1's products are on the matrix pipe while tile n's are screened on the VALU;
//   * target chunks are visited in a strided order (a map's points come in raster order: in stream order the threshold
//     would crawl towards every source row and most tiles would take the re-evaluation path), and after every chunk the
//     16 lanes of a source row share their lowest threshold: after the first chunks only genuine near-ties pass;
//   * the re-evaluation reads raw coordinates from LDS (targets: staged with the operands; sources: 1 KiB per wave),
//     so it is ~100 VALU cycles under MFMAs already in flight, not a global round trip.
#define NNM_WAVES 2
#define NNM_ROWT 4                           // 64 sources per wave
#define NNM_CHUNK 512                        // targets per LDS stage: 5 planes x 4 KiB (4 workgroups per CU: the VGPR limit)
typedef double qs_d4 __attribute__((ext_vector_type(4)));

// planes[k][j], k = 0..2: -2 tx', -2 ty', |t'|^2 for target j (centred); padding columns: |t'|^2 = +inf
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_prep_kernel(const double2 *__restrict__ dst, size_t n_dst, size_t n_pad, double cx, double cy, double *__restrict__ planes)
{
    const size_t j = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (j >= n_pad) return;
    double a = 0, b = 0, c = INFINITY;
    if (j < n_dst) {
        const double tx = dst[j].x - cx, ty = dst[j].y - cy;
        a = -2.0 * tx; b = -2.0 * ty; c = tx * tx + ty * ty;
    }
    planes[j] = a; planes[n_pad + j] = b; planes[2 * n_pad + j] = c;
}

struct NnmState { double best[NNM_ROWT][4]; int best_j[NNM_ROWT][4]; };
// The screen costs nothing but three integer ORs and one compare per row tile: the row's threshold rides in the contraction.
// K = 4 has a free slot -- A[row][3] = -thr(row), B[3][col] = 1 -- so the MFMA delivers D = S - thr(row) and "S <= thr" is
// the SIGN of D: the four high words of a tile's results ORed together, one v_cmp_lt_i32 (the fp64 MFMA runs on the fp64
// vector datapath, and sixteen v_cmp_le_f64 per tile were a third of the time; 32-bit integer ops are not).  thr(row) is the
// lowest EXACT squared distance the row has seen, minus the row's centred norm, plus the margin -- kept in LDS, lowered with
// ds_min_f64 by whoever finds a closer target, reloaded into the A operand of the row's k = 3 lane: every column class of a
// row tightens with every find of any of them, at once.
// A negative D (sign bit set; a NaN with its sign set takes the path too and decides nothing) is re-evaluated exactly --
// raw coordinates from LDS, the reference expression -- and competes on (d2, target index).
__device__ inline void nnm_recheck(const qs_d4 (&d)[NNM_ROWT], const unsigned long long (&mk)[NNM_ROWT], NnmState &st, double (&a_op)[NNM_ROWT],
                                   int j, double cx, double cy, double t2max, double tx, double ty, const double2 *s_src_lk,
                                   double *s_thr_w, int lc, int lk)
{
    #pragma unroll
    for (int t = 0; t < NNM_ROWT; t++) {
        if (mk[t] == 0) continue;                                              // (uniform)
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            if (__double_as_longlong(d[t][r]) < 0) {
                const double2 p = s_src_lk[16 * t + 4 * r];
                const double dx = p.x - tx, dy = p.y - ty;                     // the reference expression
                const double d2 = dx * dx + dy * dy;
                if (d2 < st.best[t][r] || (d2 == st.best[t][r] && j < st.best_j[t][r])) { st.best[t][r] = d2; st.best_j[t][r] = j; }
                const double ux = p.x - cx, uy = p.y - cy;
                const double u2 = ux * ux + uy * uy;
                // S = d2 - |s'|^2 up to rounding; the margin (2^-40 of the operands' scale, ~40 x that rounding) on top
                const double nt = (d2 - u2) + (u2 + t2max) * 0x1p-40;
                if (nt < INFINITY) __hip_atomic_fetch_min(&s_thr_w[16 * t + 4 * r + lk], nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // the wave's LDS operations complete in order: the reload below sees the minima above
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double th = __hip_atomic_load(&s_thr_w[16 * t + lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lk == 3) a_op[t] = -th;
    }
}
// One step: the NNM_ROWT MFMAs of the NEXT target tile (operand `bop`) are issued one by one, each followed by the screen of
// one row tile of the CURRENT tile's products: is any of its four results negative?
#define NNM_STEP(dnew, dold, bop, mk, anym)                                                                    \
    _Pragma("unroll") for (int t = 0; t < NNM_ROWT; t++) {                                                      \
        dnew[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[t], bop, zero, 0, 0, 0);                            \
        const int h_ = (int)(__double_as_longlong(dold[t][0]) >> 32) | (int)(__double_as_longlong(dold[t][1]) >> 32) |   \
                       (int)(__double_as_longlong(dold[t][2]) >> 32) | (int)(__double_as_longlong(dold[t][3]) >> 32);    \
        mk[t] = __ballot(h_ < 0); anym |= mk[t];                                                                \
    }

// Work item = (pair of 64-source groups, part of the targets): a launch is n_groups x n_parts workgroups, dealt to the CUs as
// slots free up, so that the chip stays full whatever n_src is (one item per wave over ALL targets left 1 563 waves on 2 048
// wave slots at 10^5 points: a quarter of the machine idle behind the SIMDs that had two).  A part starts from the thresholds
// the parts before it left in thr_seed (HBM, one per source, lowered with a global fp64 atomic min at the end of every part:
// workgroups are dispatched in blockIdx order -- all groups' part 0, then part 1 ... -- so a part usually starts warm; any
// order is correct, a seed is only ever an exact distance some target really has).  Every part writes the best (d2, target)
// it found per source; qs_icp_nn_merge_kernel takes the lexicographic minimum over the parts.
__global__ void __launch_bounds__(NNM_WAVES * QS_WAVE)
qs_icp_nn_mfma_kernel(const double2 *__restrict__ src, size_t n_src, const double2 *__restrict__ dst, size_t n_dst,
                      const double *__restrict__ planes, size_t n_pad, double cx, double cy, double t2max,
                      unsigned int n_groups, unsigned int chunks_per_part, double *__restrict__ thr_seed,
                      int *__restrict__ part_j, double *__restrict__ part_d2)
{
    __shared__ double s_b[3][NNM_CHUNK];                 // operand planes of the chunk
    __shared__ double s_tx[NNM_CHUNK], s_ty[NNM_CHUNK];  // its raw coordinates (re-evaluation)
    __shared__ double2 s_src[NNM_WAVES][16 * NNM_ROWT];  // the wave's raw source coordinates
    __shared__ double s_thr[NNM_WAVES][16 * NNM_ROWT];   // the rows' thresholds
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    const unsigned int group = blockIdx.x % n_groups, part = blockIdx.x / n_groups;
    const size_t row0 = ((size_t)group * NNM_WAVES + wave) * (16 * NNM_ROWT);

    double a_op[NNM_ROWT];                       // A fragment: A[row = lane & 15][k = lane >> 4]; k = 3: minus the row's threshold
    NnmState st;                                 // per result slot: row = row0 + 16 t + (lane >> 4) + 4 r, column class lane & 15
    #pragma unroll
    for (int t = 0; t < NNM_ROWT; t++) {
        const size_t ra = row0 + 16 * t + lc;
        const double2 pa = ra < n_src ? src[ra] : make_double2(cx, cy);
        const double seed = ra < n_src ? __hip_atomic_load(&thr_seed[ra], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFINITY;
        if (lk == 0) { s_src[wave][16 * t + lc] = pa; s_thr[wave][16 * t + lc] = seed; }
        const double ux = pa.x - cx, uy = pa.y - cy;
        a_op[t] = lk == 0 ? ux : (lk == 1 ? uy : (lk == 2 ? 1.0 : -seed));
        #pragma unroll
        for (int r = 0; r < 4; r++) { st.best[t][r] = INFINITY; st.best_j[t][r] = 0x7fffffff; }
    }
    const qs_d4 zero = {0.0, 0.0, 0.0, 0.0};
    const double2 *const s_src_lk = &s_src[wave][lk];
    double *const s_thr_w = s_thr[wave];
    const int pk = lk < 3 ? lk : 0;              // lanes of k = 3 carry B = 1 (the threshold's multiplier); they read plane 0 and drop it

    // this part's chunks, visited in steps of ~0.618 of their number (coprime: every chunk exactly once; a map's points come
    // in raster order: in stream order the thresholds would crawl)
    const unsigned int n_chunks_all = (unsigned int)((n_pad + NNM_CHUNK - 1) / NNM_CHUNK);
    const unsigned int c_lo = part * chunks_per_part;
    const unsigned int n_chunks = c_lo < n_chunks_all ? min(chunks_per_part, n_chunks_all - c_lo) : 0u;
    unsigned int chunk_step = (unsigned int)(0.6180339887 * n_chunks);
    if (chunk_step < 1) chunk_step = 1;
    for (;;) { unsigned int a = chunk_step, b = n_chunks; while (b) { const unsigned int t = a % b; a = b; b = t; } if (a <= 1) break; chunk_step++; }
    unsigned int ci = 0;
    for (unsigned int it = 0; it < n_chunks; it++, ci = (ci + chunk_step) % n_chunks) {
        const size_t base = (size_t)(c_lo + ci) * NNM_CHUNK;
        const size_t cnt = (n_pad - base < NNM_CHUNK) ? n_pad - base : NNM_CHUNK;      // a multiple of 16
        __syncthreads();
        for (size_t j = tid; j < cnt; j += NNM_WAVES * QS_WAVE) {
            s_b[0][j] = planes[base + j]; s_b[1][j] = planes[n_pad + base + j]; s_b[2][j] = planes[2 * n_pad + base + j];
            const double2 tp = base + j < n_dst ? dst[base + j] : make_double2(INFINITY, INFINITY);
            s_tx[j] = tp.x; s_ty[j] = tp.y;
        }
        __syncthreads();
        const int tiles = (int)(cnt / 16);
        // two accumulator sets (no copies), B operands read from LDS one step ahead of their MFMAs
        qs_d4 dA[NNM_ROWT], dB[NNM_ROWT];
        double b_cur = s_b[pk][lc];
        b_cur = lk < 3 ? b_cur : 1.0;
        double b_nxt = tiles > 1 ? s_b[pk][16 + lc] : 0.0;
        #pragma unroll
        for (int t = 0; t < NNM_ROWT; t++) dA[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[t], b_cur, zero, 0, 0, 0);
        for (int tile = 0; tile < tiles; tile += 2) {
            // dA = tile, issue tile + 1 into dB
            b_cur = lk < 3 ? b_nxt : 1.0;
            if (tile + 2 < tiles) b_nxt = s_b[pk][16 * (tile + 2) + lc];
            unsigned long long mk[NNM_ROWT], anym = 0;        // (past the last tile the MFMAs run on a stale operand; nobody looks)
            const bool i1 = tile + 1 < tiles;
            NNM_STEP(dB, dA, b_cur, mk, anym)
            if (anym) nnm_recheck(dA, mk, st, a_op, (int)(base + 16 * (size_t)tile + lc), cx, cy, t2max, s_tx[16 * tile + lc], s_ty[16 * tile + lc],
                                  s_src_lk, s_thr_w, lc, lk);
            if (!i1) break;
            // dB = tile + 1, issue tile + 2 into dA
            b_cur = lk < 3 ? b_nxt : 1.0;
            if (tile + 3 < tiles) b_nxt = s_b[pk][16 * (tile + 3) + lc];
            anym = 0;
            NNM_STEP(dA, dB, b_cur, mk, anym)
            if (anym) nnm_recheck(dB, mk, st, a_op, (int)(base + 16 * (size_t)(tile + 1) + lc), cx, cy, t2max, s_tx[16 * (tile + 1) + lc],
                                  s_ty[16 * (tile + 1) + lc], s_src_lk, s_thr_w, lc, lk);
        }
    }
    // the 16 lanes that hold one source row: lexicographic minimum of (d2, j)
    #pragma unroll
    for (int t = 0; t < NNM_ROWT; t++)
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            double b = st.best[t][r]; int bj = st.best_j[t][r];
            #pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const double ob = __shfl_xor(b, off); const int oj = __shfl_xor(bj, off);
                const bool take = ob < b || (ob == b && oj < bj);
                b = take ? ob : b; bj = take ? oj : bj;
            }
            const size_t rr = row0 + lk + 16 * t + 4 * r;
            if (lc == 0 && rr < n_src) { part_j[(size_t)part * n_src + rr] = bj; part_d2[(size_t)part * n_src + rr] = b; }
        }
    // what this part learned, for the parts that start after it
    #pragma unroll
    for (int t = 0; t < NNM_ROWT; t++) {
        const size_t ra = row0 + 16 * t + lc;
        if (lk == 0 && ra < n_src) {
            const double th = s_thr_w[16 * t + lc];
            if (th < INFINITY) __hip_atomic_fetch_min(&thr_seed[ra], th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// lexicographic minimum of (d2, target index) over the parts -> corr / d2 (Open3D: a correspondence only under max_dist)
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_nn_merge_kernel(size_t n_src, unsigned int n_parts, const int *__restrict__ part_j, const double *__restrict__ part_d2,
                       double max_d2, int *__restrict__ corr, double *__restrict__ d2_out)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i >= n_src) return;
    double b = INFINITY; int bj = 0x7fffffff;
    for (unsigned int p = 0; p < n_parts; p++) {
        const double d = part_d2[(size_t)p * n_src + i]; const int j = part_j[(size_t)p * n_src + i];
        if (d < b || (d == b && j < bj)) { b = d; bj = j; }
    }
    const bool ok = bj != 0x7fffffff && b < max_d2;
    corr[i] = ok ? bj : -1;
    d2_out[i] = ok ? b : 0.0;
}
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_fill_inf_kernel(double *__restrict__ v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i < n) v[i] = INFINITY;
}

// ---- fixed-order two-level sums ---------------------------------------------------------------------
// pass 0: per block {n, sum d2, sum ax, sum ay, sum bx, sum by}; pass 1 (means known):
// {sum (ax-am)(bx-bm) + (ay..)(by..), sum (ax-am)(by-bm) - (ay-am)(bx-bm)}
#define ICP_NSUM 6
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_sums_kernel(const double2 *__restrict__ src, size_t n_src, const double2 *__restrict__ dst,
                   const int *__restrict__ corr, const double *__restrict__ d2, int pass,
                   double amx, double amy, double bmx, double bmy, double *__restrict__ partial)
{
    __shared__ double s[ICP_NSUM][ICP_BLOCK];
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    double v[ICP_NSUM] = {0, 0, 0, 0, 0, 0};
    if (i < n_src && corr[i] >= 0) {
        const double2 a = src[i], b = dst[corr[i]];
        if (pass == 0) { v[0] = 1.0; v[1] = d2[i]; v[2] = a.x; v[3] = a.y; v[4] = b.x; v[5] = b.y; }
        else {
            const double ax = a.x - amx, ay = a.y - amy, bx = b.x - bmx, by = b.y - bmy;
            v[0] = ax * bx + ay * by;
            v[1] = ax * by - ay * bx;
        }
    }
    #pragma unroll
    for (int q = 0; q < ICP_NSUM; q++) s[q][threadIdx.x] = v[q];
    __syncthreads();
    for (int off = ICP_BLOCK / 2; off > 0; off >>= 1) {       // fixed tree order
        if (threadIdx.x < off) {
            #pragma unroll
            for (int q = 0; q < ICP_NSUM; q++) s[q][threadIdx.x] += s[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < ICP_NSUM) partial[(size_t)blockIdx.x * ICP_NSUM + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void qs_icp_final_kernel(const double *__restrict__ partial, size_t n_blocks, double *__restrict__ out)
{
    if (threadIdx.x < ICP_NSUM) {
        double acc = 0.0;
        for (size_t b = 0; b < n_blocks; b++) acc += partial[b * ICP_NSUM + threadIdx.x];   // fixed order
        out[threadIdx.x] = acc;
    }
}

// p <- R p + t  (PointCloud::Transform with a planar rigid transform)
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_transform_kernel(double2 *__restrict__ pts, size_t n, double c, double s, double tx, double ty)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i < n) {
        const double2 p = pts[i];
        pts[i] = make_double2(c * p.x - s * p.y + tx, s * p.x + c * p.y + ty);
    }
}

// ---- voxel down-sampling (PointCloud::VoxelDownSample): points are averaged per voxel; the voxel
// of p is floor((p - (min_bound - voxel/2)) / voxel).  Output is in ascending voxel order (Open3D's
// order is a hash-map's: unspecified). -----------------------------------------------------------------
__global__ void __launch_bounds__(ICP_BLOCK)
qs_voxel_key_kernel(const double2 *__restrict__ pts, size_t n, double minx, double miny, double voxel,
                    unsigned long long *__restrict__ keys)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i < n) {
        const long long vx = (long long)floor((pts[i].x - minx) / voxel), vy = (long long)floor((pts[i].y - miny) / voxel);
        keys[i] = ((unsigned long long)(vy & 0xffffffffll) << 32) | (unsigned long long)(vx & 0xffffffffll);
    }
}

hipError_t qs_launch_icp_nn(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, size_t n_dst,
                            double max_d2, int *corr, double *d2)
{
    hipLaunchKernelGGL(qs_icp_nn_kernel, dim3((unsigned int)((n_src + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, src, n_src, dst, n_dst, max_d2, corr, d2);
    return hipGetLastError();
}

// ---- diagnostic: the chip's fp64 MFMA issue rate (what the search above is priced against) -----------------
// every wave issues `iters` x 4 independent v_mfma_f64_16x16x4_f64 back to back; operands in registers
__global__ void __launch_bounds__(256)
qs_mfma_f64_rate_kernel(int iters, double *__restrict__ sink)
{
    qs_d4 acc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = qs_d4{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    double v = 0;
    #pragma unroll
    for (int q = 0; q < 4; q++) v += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (v == 123.456) sink[0] = v;          // keeps the chain alive
}
hipError_t qs_launch_mfma_f64_rate(qs_ctx *c, int blocks, int iters, double *sink)
{
    hipLaunchKernelGGL(qs_mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, c->stream, iters, sink);
    return hipGetLastError();
}

hipError_t qs_launch_icp_prep(qs_ctx *c, const double2 *dst, size_t n_dst, size_t n_pad, double cx, double cy, double *planes)
{
    hipLaunchKernelGGL(qs_icp_prep_kernel, dim3((unsigned int)((n_pad + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0, c->stream,
                       dst, n_dst, n_pad, cx, cy, planes);
    return hipGetLastError();
}

// How the targets are cut into parts: enough workgroups to keep every slot busy through the tail (>= 6 per slot of the 1 024 a
// launch holds: 4 workgroups per CU), no part shorter than 8 chunks.
void qs_icp_nn_plan(size_t n_src, size_t n_pad, unsigned int *n_groups, unsigned int *n_parts, unsigned int *chunks_per_part)
{
    const size_t rows_per_wg = (size_t)NNM_WAVES * 16 * NNM_ROWT;
    const unsigned int groups = (unsigned int)((n_src + rows_per_wg - 1) / rows_per_wg);
    const unsigned int n_chunks = (unsigned int)((n_pad + NNM_CHUNK - 1) / NNM_CHUNK);
    unsigned int parts = (6u * 1024u + groups - 1) / groups;
    const unsigned int max_parts = n_chunks / 8 > 0 ? n_chunks / 8 : 1;
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    const unsigned int cpp = (n_chunks + parts - 1) / parts;
    *n_groups = groups; *chunks_per_part = cpp; *n_parts = (n_chunks + cpp - 1) / cpp;
}

hipError_t qs_launch_icp_nn_mfma(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, size_t n_dst,
                                 const double *planes, size_t n_pad, double cx, double cy, double t2max, double max_d2,
                                 int *corr, double *d2, int *part_j, double *part_d2, double *thr_seed)
{
    unsigned int groups, parts, cpp;
    qs_icp_nn_plan(n_src, n_pad, &groups, &parts, &cpp);
    hipLaunchKernelGGL(qs_icp_fill_inf_kernel, dim3((unsigned int)((n_src + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0, c->stream, thr_seed, n_src);
    hipLaunchKernelGGL(qs_icp_nn_mfma_kernel, dim3(groups * parts), dim3(NNM_WAVES * QS_WAVE),
                       0, c->stream, src, n_src, dst, n_dst, planes, n_pad, cx, cy, t2max, groups, cpp, thr_seed, part_j, part_d2);
    hipLaunchKernelGGL(qs_icp_nn_merge_kernel, dim3((unsigned int)((n_src + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0, c->stream,
                       n_src, parts, part_j, part_d2, max_d2, corr, d2);
    return hipGetLastError();
}

hipError_t qs_launch_icp_sums(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, const int *corr,
                              const double *d2, int pass, const double means[4], double *partial, double *out6)
{
    const size_t nb = (n_src + ICP_BLOCK - 1) / ICP_BLOCK;
    hipLaunchKernelGGL(qs_icp_sums_kernel, dim3((unsigned int)nb), dim3(ICP_BLOCK), 0, c->stream, src, n_src, dst, corr, d2,
                       pass, means[0], means[1], means[2], means[3], partial);
    hipLaunchKernelGGL(qs_icp_final_kernel, dim3(1), dim3(64), 0, c->stream, partial, nb, out6);
    return hipGetLastError();
}

hipError_t qs_launch_icp_transform(qs_ctx *c, double2 *pts, size_t n, double cs, double sn, double tx, double ty)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_icp_transform_kernel, dim3((unsigned int)((n + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, pts, n, cs, sn, tx, ty);
    return hipGetLastError();
}

hipError_t qs_launch_voxel_keys(qs_ctx *c, const double2 *pts, size_t n, double minx, double miny, double voxel,
                                unsigned long long *keys)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_voxel_key_kernel, dim3((unsigned int)((n + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, pts, n, minx, miny, voxel, keys);
    return hipGetLastError();
}
