// slam.hip -- K4: pose-graph landmark loop closure and drift correction.
// Semantics: PoseGraphSLAM.add_pose / _check_closure (server_nodes/dual_bot_mapper.py:273-326)
// and the drift application around it (:855-857, :908-914).
//
// The reference is a sequential recurrence: a closure at node i changes the pose of every
// later packet of that agent.  Two facts make it batchable without changing any result:
//   (1) a landmark stored at node j can only be matched by a node i >= j + MIN_POSES_BETWEEN
//       (:300), and after a closure an agent cannot close again for MIN_POSES_BETWEEN nodes
//       (:304); hence inside a window of W <= MIN_POSES_BETWEEN consecutive nodes no event can
//       see a landmark of the same window and each agent closes at most once;
//   (2) landmarks are appended in node order (:288), so "idx - lm_idx >= MIN" selects a
//       prefix of the list and the first match in list order is the lowest matching slot.
// One workgroup per pose graph walks its packets in arrival order, window by window; inside a
// window every eligible landmark event scans the landmark list with the whole workgroup
// (lowest matching slot = first match in insertion order), then one lane commits the closures
// in node order and re-poses the closing agent's later packets of the window.
#include "qs_internal.h"

#define SLAM_BLOCK 256
#define SLAM_WAVES (SLAM_BLOCK / QS_WAVE)
#define LL_MAX 0x7fffffffffffffffll

__global__ void __launch_bounds__(SLAM_BLOCK)
qs_slam_kernel(QsGraphDev *__restrict__ graphs, int bots_per_graph, int max_agent, size_t n,
               QsBatch b, double *__restrict__ drift, long long *__restrict__ last_closure,
               int win, int min_between, double r2thr, double corr,
               unsigned long long *__restrict__ counters)
{
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QsGraphDev G = graphs[g];
    const int bot0 = g * bots_per_graph + 1;
    const int nb = min(bots_per_graph, max_agent - bot0 + 1);

    __shared__ double s_drift[QS_MAX_AGENT + 1][2];
    __shared__ long long s_last[QS_MAX_AGENT + 1];
    __shared__ int s_list[SLAM_BLOCK];
    __shared__ int s_wave_cnt[SLAM_WAVES];
    __shared__ long long s_red[SLAM_WAVES];
    __shared__ int w_pkt[QS_WIN_MAX], w_agent[QS_WIN_MAX], w_type[QS_WIN_MAX], w_elig[QS_WIN_MAX];
    __shared__ long long w_idx[QS_WIN_MAX], w_match[QS_WIN_MAX];
    __shared__ double w_x[QS_WIN_MAX], w_y[QS_WIN_MAX];
    __shared__ long long s_ncls;
    __shared__ int s_add;

    for (int t = tid; t < nb; t += SLAM_BLOCK) {
        s_drift[t][0] = drift[2 * (bot0 + t)];
        s_drift[t][1] = drift[2 * (bot0 + t) + 1];
        s_last[t] = last_closure[bot0 + t];
    }
    if (tid == 0) s_ncls = G.n_cls;
    long long n_nodes = G.n_nodes, n_lms = G.n_lms;
    __syncthreads();

    for (size_t base = 0; base < n; base += SLAM_BLOCK) {
        // this graph's accepted packets of the chunk, compacted in arrival order
        const size_t i = base + tid;
        bool mine = false;
        if (i < n && b.accept[i]) mine = ((int)b.agent[i] - 1) / bots_per_graph == g;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) s_wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = 0, cnt = 0;
        #pragma unroll
        for (int w = 0; w < SLAM_WAVES; w++) { if (w < wave) off += s_wave_cnt[w]; cnt += s_wave_cnt[w]; }
        if (mine) s_list[off + __popcll(m & ((1ull << lane) - 1))] = tid;
        __syncthreads();

        for (int w0 = 0; w0 < cnt; w0 += win) {
            const int wn = min(win, cnt - w0);
            if (tid < wn) {
                const size_t p = base + s_list[w0 + tid];
                const int a = (int)b.agent[p] - bot0, t = b.lm[p];
                const long long idx = n_nodes + w0 + tid;             // len(self.nodes)  :275
                w_pkt[tid] = s_list[w0 + tid];
                w_agent[tid] = a; w_type[tid] = t; w_idx[tid] = idx;
                w_x[tid] = b.px[p] + s_drift[a][0];                   // rx += cdx  :856
                w_y[tid] = b.py[p] + s_drift[a][1];                   // ry += cdy  :857
                w_elig[tid] = (t != 0) && (idx - s_last[a] >= min_between);   // :283, :304
                w_match[tid] = -1;
            }
            __syncthreads();

            for (int j = 0; j < wn; j++) {
                if (!w_elig[j]) continue;                              // uniform
                const long long limit = w_idx[j] - min_between;       // :300  idx - lm_idx >= MIN
                const int type = w_type[j];
                const double qx = w_x[j], qy = w_y[j];
                long long found = -1;
                for (long long c0 = 0; c0 < n_lms; c0 += SLAM_BLOCK) {
                    const long long k = c0 + tid;
                    bool cand = false;
                    if (k < n_lms && G.lm_idx[k] <= limit && G.lm_type[k] == type) {   // :296, :300
                        const double dx = qx - G.lm_x[k], dy = qy - G.lm_y[k];
                        cand = (dx * dx + dy * dy) < r2thr;            // sqrt(..) < RADIUS  :308-309
                    }
                    const unsigned long long cm = __ballot(cand);
                    if (lane == 0) s_red[wave] = cm ? (c0 + wave * QS_WAVE + __ffsll((long long)cm) - 1) : LL_MAX;
                    const long long lastk = (c0 + SLAM_BLOCK - 1 < n_lms - 1) ? c0 + SLAM_BLOCK - 1 : n_lms - 1;
                    const bool beyond = G.lm_idx[lastk] > limit;       // list is ascending in idx
                    __syncthreads();
                    long long first = s_red[0];
                    #pragma unroll
                    for (int w = 1; w < SLAM_WAVES; w++) first = s_red[w] < first ? s_red[w] : first;
                    __syncthreads();
                    if (first != LL_MAX) { found = first; break; }
                    if (beyond) break;
                }
                if (tid == 0) w_match[j] = found;
            }
            __syncthreads();

            if (tid == 0) {
                // commit closures in node order  (:309-324, :910-914)
                long long ncls = s_ncls;
                for (int j = 0; j < wn; j++) {
                    if (!w_elig[j] || w_match[j] < 0) continue;
                    const int a = w_agent[j];
                    if (w_idx[j] - s_last[a] < min_between) continue;  // an earlier closure of this window
                    const long long mslot = w_match[j];
                    const double ex = G.lm_x[mslot] - w_x[j], ey = G.lm_y[mslot] - w_y[j];   // :311-312
                    const double cdx = ex * corr, cdy = ey * corr;                           // :314-315
                    if (ncls < G.cap_cls) {
                        G.cl_lm_idx[ncls] = G.lm_idx[mslot]; G.cl_node_idx[ncls] = w_idx[j];  // :317
                        G.cl_dx[ncls] = cdx; G.cl_dy[ncls] = cdy;
                    }
                    ncls++;
                    s_last[a] = w_idx[j];                                                     // :318
                    s_drift[a][0] = s_drift[a][0] + cdx;                                      // :911-914
                    s_drift[a][1] = s_drift[a][1] + cdy;
                    for (int j2 = j + 1; j2 < wn; j2++)
                        if (w_agent[j2] == a) {
                            const size_t p2 = base + w_pkt[j2];
                            w_x[j2] = b.px[p2] + s_drift[a][0];
                            w_y[j2] = b.py[p2] + s_drift[a][1];
                        }
                }
                s_ncls = ncls;
            }
            __syncthreads();

            if (wave == 0) {
                const bool act = tid < wn;
                const bool ev = act && w_type[tid] != 0;
                const unsigned long long em = __ballot(ev);
                if (act) {
                    const size_t p = base + w_pkt[tid];
                    b.rx[p] = w_x[tid];
                    b.ry[p] = w_y[tid];
                }
                if (ev) {                                              // self.landmarks.append  :288
                    const long long slot = n_lms + __popcll(em & ((1ull << lane) - 1));
                    if (slot < G.cap_lms) {
                        G.lm_x[slot] = w_x[tid]; G.lm_y[slot] = w_y[tid];
                        G.lm_idx[slot] = w_idx[tid]; G.lm_type[slot] = (unsigned char)w_type[tid];
                    }
                }
                if (lane == 0) s_add = __popcll(em);
            }
            __syncthreads();
            n_lms += s_add;
        }
        n_nodes += cnt;
        __syncthreads();
    }

    for (int t = tid; t < nb; t += SLAM_BLOCK) {
        drift[2 * (bot0 + t)] = s_drift[t][0];
        drift[2 * (bot0 + t) + 1] = s_drift[t][1];
        last_closure[bot0 + t] = s_last[t];
    }
    if (tid == 0) {
        atomicAdd(&counters[QS_CNT_CLOSURES], (unsigned long long)(s_ncls - G.n_cls));
        atomicAdd(&counters[QS_CNT_LANDMARKS], (unsigned long long)(n_lms - G.n_lms));
        graphs[g].n_nodes = n_nodes;
        graphs[g].n_lms = n_lms;
        graphs[g].n_cls = s_ncls;
    }
}

hipError_t qs_launch_slam(qs_ctx *c, size_t n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_slam_kernel, dim3(c->n_graphs), dim3(SLAM_BLOCK), 0, c->stream, c->d_graphs,
                       c->bots_per_graph, c->cfg.max_agent, n, c->b, c->d_drift, c->d_last_closure,
                       c->win, c->cfg.min_poses_between, c->r2_threshold, c->cfg.closure_correction,
                       c->d_counters);
    return hipGetLastError();
}
