// slam.hip -- K4: pose-graph landmark loop closure and drift correction.
// Semantics: PoseGraphSLAM.add_pose / _check_closure (server_nodes/dual_bot_mapper.py:273-326)
// and the drift application around it (:855-857, :908-914).
//
// The reference is a sequential recurrence: a closure at node i changes the pose of every later
// packet of that agent, and with it the landmarks that agent stores and the matches it finds.
// Only the closures are sequential, so the stage is split:
//
//   index  (parallel)  node index of every accepted record inside its pose graph (a stable
//                      partition by graph) and the compacted list of landmark events per graph;
//   chain  (one workgroup per graph) walks ONLY the landmark events, in node order, in windows;
//   pose   (parallel)  every record's pose = raw + drift of its bot at that node, looked up in
//                      the bot's closure list of the batch.
//
// Two facts keep the chain exact while letting a window be handled at once:
//   (1) a landmark stored at node j can only be matched by a node i >= j + MIN_POSES_BETWEEN
//       (:300), and after a closure an agent cannot close again for MIN_POSES_BETWEEN nodes
//       (:304); so inside a window spanning < MIN_POSES_BETWEEN nodes no event can see a
//       landmark of the same window, every event is evaluated with the drift at window start,
//       and each agent closes at most once: at its first eligible event that has a match;
//   (2) landmarks are appended in node order (:288), so "idx - lm_idx >= MIN" selects a prefix
//       and the first match in list order is the LOWEST node index among the matches.
// The match search uses the spatial index of QsGraphDev: buckets of edge >= CLOSURE_RADIUS per
// landmark type (a hash table over the bucket cells: any pose has its bucket), entries in insertion
// order; the 3x3 buckets around a query contain every landmark within the radius, and the minimum
// node index over their first matches is the reference's first match.  Landmarks of a type the
// directory has no slab for (type > 5) live in a side list that every query also scans.
#include "qs_internal.h"
#include <cstdlib>
#include <cstring>

#define LL_MAX 0x7fffffffffffffffll
#define IDX_BLOCK QS_SLAM_IDX_BLOCK
#define IDX_WAVES (IDX_BLOCK / QS_WAVE)

int qs_slam_blocks(size_t n) { return (int)((n + IDX_BLOCK - 1) / IDX_BLOCK); }

// ---- index pass 1: per block, accepted records and landmark events of every graph ------------
__global__ void __launch_bounds__(IDX_BLOCK)
qs_slam_count_kernel(size_t n, QsBatch b, QsSlamBatch sb, int bots_per_graph, int n_graphs)
{
    __shared__ unsigned int s_acc[QS_MAX_AGENT + 1], s_ev[QS_MAX_AGENT + 1];
    const int tid = threadIdx.x;
    for (int t = tid; t < n_graphs; t += IDX_BLOCK) { s_acc[t] = 0; s_ev[t] = 0; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * IDX_BLOCK + tid;
    if (i < n && b.accept[i]) {
        const int g = ((int)b.agent[i] - 1) / bots_per_graph;
        atomicAdd(&s_acc[g], 1u);
        if (b.lm[i]) atomicAdd(&s_ev[g], 1u);
    }
    __syncthreads();
    for (int t = tid; t < n_graphs; t += IDX_BLOCK) {
        sb.blk_acc[(size_t)t * sb.n_blocks + blockIdx.x] = s_acc[t];
        sb.blk_ev[(size_t)t * sb.n_blocks + blockIdx.x] = s_ev[t];
    }
}

// ---- index pass 2: per graph, exclusive scan of its block counts -------------------------------
__global__ void __launch_bounds__(256)
qs_slam_blockscan_kernel(QsSlamBatch sb)
{
    __shared__ unsigned int s_a[256], s_e[256];
    const int g = blockIdx.x, tid = threadIdx.x;
    unsigned int *acc = sb.blk_acc + (size_t)g * sb.n_blocks, *ev = sb.blk_ev + (size_t)g * sb.n_blocks;
    const int per = (sb.n_blocks + 255) / 256;
    const int lo = min(tid * per, sb.n_blocks), hi = min(lo + per, sb.n_blocks);
    unsigned int a = 0, e = 0;
    for (int k = lo; k < hi; k++) { a += acc[k]; e += ev[k]; }
    s_a[tid] = a; s_e[tid] = e;
    __syncthreads();
    if (tid == 0) {
        unsigned int ra = 0, re = 0;
        for (int t = 0; t < 256; t++) { const unsigned int va = s_a[t], ve = s_e[t]; s_a[t] = ra; s_e[t] = re; ra += va; re += ve; }
        sb.acc_total[g] = ra;
        sb.ev_base[g + 1] = re;          // totals for now; made a prefix by the next kernel
    }
    __syncthreads();
    unsigned int ra = s_a[tid], re = s_e[tid];
    for (int k = lo; k < hi; k++) { const unsigned int va = acc[k], ve = ev[k]; acc[k] = ra; ev[k] = re; ra += va; re += ve; }
}

// ---- index pass 2b: prefix over graphs (event ranges) and over bots (closure regions) ----------
__global__ void qs_slam_prefix_kernel(QsSlamBatch sb, int n_graphs, int max_agent, const double *__restrict__ drift)
{
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        sb.ev_base[0] = 0;
        for (int g = 0; g < n_graphs; g++) { run += sb.ev_base[g + 1]; sb.ev_base[g + 1] = run; }
    }
    if (threadIdx.x == 64) {
        unsigned int run = 0;
        for (int a = 0; a <= max_agent + 1; a++) { const unsigned int v = a <= max_agent ? sb.agent_ev[a] : 0; sb.agent_ev[a] = run; run += v; }
    }
    for (int t = threadIdx.x; t <= max_agent; t += blockDim.x) {
        sb.acl_cnt[t] = 0;
        sb.drift_start[2 * t] = drift[2 * t];
        sb.drift_start[2 * t + 1] = drift[2 * t + 1];
    }
}

// ---- index pass 3: node index per record, event records per graph (stable in arrival order) ----
__global__ void __launch_bounds__(IDX_BLOCK)
qs_slam_index_kernel(size_t n, QsBatch b, QsSlamBatch sb, const QsGraphDev *__restrict__ graphs,
                     int bots_per_graph, int n_graphs)
{
    extern __shared__ unsigned int s_w[];          // [IDX_WAVES][n_graphs][2] per-wave counts
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < IDX_WAVES * n_graphs * 2; t += IDX_BLOCK) s_w[t] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * IDX_BLOCK + tid;
    const bool acc = i < n && b.accept[i];
    int g = -1, agent = 0, lmk = 0;
    if (acc) { agent = b.agent[i]; g = (agent - 1) / bots_per_graph; lmk = b.lm[i]; }
    // ranks inside the wave, per graph, in lane (= arrival) order
    unsigned int rank = 0, rank_ev = 0;
    unsigned long long remaining = __ballot(acc);
    while (remaining) {
        const int leader = __ffsll((long long)remaining) - 1;
        const int k = __shfl(g, leader);
        const unsigned long long grp = __ballot(acc && g == k);
        const unsigned long long egrp = __ballot(acc && g == k && lmk != 0);
        if (acc && g == k) {
            rank = __popcll(grp & ((1ull << lane) - 1));
            rank_ev = __popcll(egrp & ((1ull << lane) - 1));
            if (lane == leader) {
                s_w[(wave * n_graphs + k) * 2] = __popcll(grp);
                s_w[(wave * n_graphs + k) * 2 + 1] = __popcll(egrp);
            }
        }
        remaining &= ~grp;
    }
    __syncthreads();
    if (acc) {
        unsigned int off = 0, off_ev = 0;
        for (int w = 0; w < wave; w++) { off += s_w[(w * n_graphs + g) * 2]; off_ev += s_w[(w * n_graphs + g) * 2 + 1]; }
        const long long node = graphs[g].n_nodes + sb.blk_acc[(size_t)g * sb.n_blocks + blockIdx.x] + off + rank;   // len(self.nodes)  :275
        sb.node[i] = node;
        if (lmk) {
            const size_t e = (size_t)sb.ev_base[g] + sb.blk_ev[(size_t)g * sb.n_blocks + blockIdx.x] + off_ev + rank_ev;
            sb.ev_node[e] = node;
            sb.ev_agent[e] = (unsigned char)(agent - 1 - g * bots_per_graph);
            sb.ev_type[e] = (unsigned char)lmk;
            sb.ev_px[e] = b.px[i];
            sb.ev_py[e] = b.py[i];
        }
    } else if (i < n) {
        sb.node[i] = -1;
    }
}

#ifdef QS_CHAIN_PROF3
#define CH_TRACE_WINDOWS 16384
__device__ unsigned long long g_chain_trace[CH_TRACE_WINDOWS * 16];
extern "C" int qs_debug_chain_trace(unsigned long long *out, size_t n_windows)
{
    if (n_windows > CH_TRACE_WINDOWS) n_windows = CH_TRACE_WINDOWS;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_trace), n_windows * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// ---- the chain: one workgroup (CH_WAVES waves) per pose graph ------------------------------------
// The graph's landmark events are walked in windows of < MIN_POSES_BETWEEN nodes: a query never sees a
// landmark of its own window (:300), and an agent closes at most once per window (:304), so the
// queries of one window are independent of each other.  One role per wave, one barrier per window: see
// qs_slam_chain_kernel.  A lone wave issues roughly one instruction per 4-8 cycles and the whole batch is
// ONE chain of closure decisions (each shapes the landmarks the next may match), so what counts is the
// number of instructions and LDS / memory round trips between one decision and the next.
#define CH_WAVES 16
#define CH_PILE_NODES 64             // a bucket chain of more pool nodes than this is a pile (see qs_slam_chain_kernel, DENSE)
#define CH_THREADS (CH_WAVES * QS_WAVE)
// (waves go to the four SIMDs round robin: with <= 2 owners the insert wave has SIMD 3 to itself, the light
// fetch wave shares SIMD 2 with owner 2 -- the other way round costs 2 %)
#define CH_INS (CH_WAVES - 1)       // the wave that moves a committed window into the HBM index
#define CH_FETCH (CH_WAVES - 2)     // the wave that fetches the events
#define CH_AGW (CH_WAVES - 3)       // query waves 1 .. CH_AGW: with at most CH_AGW agents in the graph (ONE) agent a belongs to wave 1 + a for
                                    // good; with more, the fetch wave deals every window's agents to the waves (chain_fetch)

// barrier that orders LDS traffic only (the hand-offs between the roles go through LDS; a full
// __syncthreads would also wait for every outstanding global store to be acknowledged)
__device__ inline void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Bucket cell along one axis.  Only consistency between insert and query matters (not the
// reference's arithmetic): the edge exceeds the closure radius by 1e-9 relative, so two points
// closer than the radius are never two cells apart whatever the rounding of this product.
__device__ inline int bucket_coord(double v, double b0, double inv_cell)
{
    const double f = floor((v - b0) * inv_cell);
    // one conversion instruction (f64 -> i64 is a sequence); monotone, saturating at +-1e9 cells
    return (fabs(f) < 1.0e9) ? (int)f : (f > 0 ? 1000000000 : -1000000000);
}

__device__ inline long long rl64(long long v, int src_lane)      // wave-uniform read of one lane
{
    int lo = (int)(v & 0xffffffffll), hi = (int)(v >> 32);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// The directory is a hash table over the bucket cells (cx, cy), one slab per landmark type: any pose has
// its bucket, inside the configured world or not, and a drifting graph does not pile up at a border.  Two
// cells that share a table entry share a chain (still in node order); the distance test sorts them out.
__device__ inline unsigned int bucket_hash(int cx, int cy, unsigned int hmask)
{
    unsigned int h = ((unsigned int)cx * 0x9E3779B1u) ^ ((unsigned int)cy * 0x85EBCA77u);
    h ^= h >> 15;
    return h & hmask;
}
// cell of a pose; false: a landmark type the directory has no slab for (side list)
__device__ inline bool bucket_cell(double x, double y, int type, const QsBucketGeom &bg, int &cx, int &cy)
{
    cx = bucket_coord(x, bg.bx0, bg.inv_cell); cy = bucket_coord(y, bg.by0, bg.inv_cell);
    return type >= 1 && type <= QS_NTYPES;
}
__device__ inline long long bucket_key(int type, int cx, int cy, const QsBucketGeom &bg)
{
    return (long long)(type - 1) * ((long long)bg.hmask + 1) + bucket_hash(cx, cy, bg.hmask);
}


// The graph's pointers come out of a struct in memory, so the compiler has to treat them as FLAT
// (could be LDS): a flat load counts on the LDS counter too, and every wait for an LDS read would
// also wait for the node rows in flight.  The query waves read the index through global pointers.
#define QS_GLOBAL __attribute__((address_space(1)))
typedef const QS_GLOBAL QsLmNode *QsNodeG;
typedef const QS_GLOBAL unsigned int *QsU32G;

// Landmark log and bucket index: `k` events of one graph, in node order, one per lane (`inw` lanes, rank =
// the lane's position among them).  Appends to the reference's insertion-ordered log and to the bucket
// chains; events of one bucket are appended in lane (= node) order.
__device__ inline void chain_insert_lanes(const QsGraphDev &G, bool inw, int rank, long long idx, long long kb, double x, double y,
                                          int type, int k, int lane, long long &n_lms, long long &n_misc, unsigned int &pool, bool &pile)
{
    // (global address space: the graph's pointers come out of a struct in memory and would be FLAT otherwise)
    QS_GLOBAL double *const lm_x = (QS_GLOBAL double *)G.lm_x, *const lm_y = (QS_GLOBAL double *)G.lm_y;
    QS_GLOBAL long long *const lm_idx = (QS_GLOBAL long long *)G.lm_idx;
    QS_GLOBAL unsigned char *const lm_type = (QS_GLOBAL unsigned char *)G.lm_type;
    QS_GLOBAL unsigned int *const misc = (QS_GLOBAL unsigned int *)G.misc, *const nd_next = (QS_GLOBAL unsigned int *)G.nd_next;
    QS_GLOBAL QsDirEntry *const dir = (QS_GLOBAL QsDirEntry *)G.dir;
    QS_GLOBAL QsLmNode *const nodes = (QS_GLOBAL QsLmNode *)G.nodes;
    // self.landmarks.append((x, y, landmark_type, idx))  :288
    const long long log_slot = n_lms + rank;
    if (inw && log_slot < G.cap_lms) {
        lm_x[log_slot] = x; lm_y[log_slot] = y; lm_idx[log_slot] = idx; lm_type[log_slot] = (unsigned char)type;
    }
    const bool inb = inw && kb >= 0;                              // the centre bucket exists
    const unsigned int key = inb ? (unsigned int)kb : 0xffffffffu;   // (directory entries number well below 2^32)
    const bool is_misc = inw && !inb;
    {
        const unsigned long long mm = __ballot(is_misc);
        if (is_misc && log_slot < G.cap_lms) misc[n_misc + __popcll(mm & ((1ull << lane) - 1))] = (unsigned int)log_slot;
        n_misc += __popcll(mm);
    }
    QsDirEntry de = {0, 0, 0, 0};
    if (inb) { de.head = dir[kb].head; de.tail = dir[kb].tail; de.tail_cnt = dir[kb].tail_cnt; de.pad = dir[kb].pad; }
    // the events of one bucket: position among them (lane = node order), how many, who goes first -- the only
    // part that walks the distinct buckets one by one; everything after is lane-parallel
    unsigned int grank = 0, gsize = 0;
    int ldr = lane;
    for (unsigned long long rem = __ballot(inb); rem;) {
        const int ld = __ffsll((long long)rem) - 1;
        const unsigned int kk = (unsigned int)__builtin_amdgcn_readlane((int)key, ld);
        const unsigned long long grp = __ballot(inb && key == kk);
        if (inb && key == kk) { grank = (unsigned int)__popcll(grp & ((1ull << lane) - 1)); gsize = (unsigned int)__popcll(grp); ldr = ld; }
        rem &= ~grp;
    }
    // an empty bucket's tail is its first node, the one that belongs to the directory entry
    const unsigned int tail = de.head ? de.tail : 1u + key;
    const unsigned int tc = de.head ? de.tail_cnt : 0u;
    const unsigned int total = tc + gsize;
    const unsigned int nn = (inb && total > QS_NODE_CAP) ? (total - QS_NODE_CAP + QS_NODE_CAP - 1) / QS_NODE_CAP : 0;   // new pool nodes
    // pool nodes are handed out bucket by bucket, in the order of the buckets' first events
    const unsigned int mine = (inb && ldr == lane) ? nn : 0u;
    // (a bucket needs a new pool node once in seven landmarks: most windows need none, and skip the scan)
    const bool any_new = __ballot(mine != 0) != 0;
    unsigned int incl = mine;
    if (any_new) {
        #pragma unroll
        for (int off = 1; off < QS_WAVE; off <<= 1) { const unsigned int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    }
    const unsigned int base = any_new ? __shfl(pool + incl - mine, ldr) : pool;
    if (inb && (long long)base + nn <= G.node_cap) {
        const unsigned int p = tc + grank;
        const unsigned int nd = p < QS_NODE_CAP ? tail : base + (p - QS_NODE_CAP) / QS_NODE_CAP;
        const unsigned int sl = p < QS_NODE_CAP ? p : (p - QS_NODE_CAP) % QS_NODE_CAP;
        QS_GLOBAL QsLmNode *np = nodes + nd;
        np->idx[sl] = idx; np->x[sl] = x; np->y[sl] = y;
        if (ldr == lane) {
            QsDirEntry upd;
            upd.head = 1u + key; upd.tail = tail; upd.tail_cnt = total; upd.pad = de.head ? de.pad + nn : nn;     // pad: pool nodes of this chain
            if (nn) {
                for (unsigned int q = 0; q + 1 < nn; q++) nd_next[base + q] = base + q + 1;
                nd_next[tail] = base;
                upd.tail = base + nn - 1;
                upd.tail_cnt = total - QS_NODE_CAP * nn;
            }
            dir[key].head = upd.head; dir[key].tail = upd.tail; dir[key].tail_cnt = upd.tail_cnt; dir[key].pad = upd.pad;
        }
    }
    if (any_new) {
        pool += __shfl(incl, QS_WAVE - 1);
        // the directory entry's spare word counts the chain's pool nodes: a chain past CH_PILE_NODES is a pile
        pile = pile || __ballot(inb && ldr == lane && nn && de.pad + nn > CH_PILE_NODES) != 0;
    }
    n_lms += k;
}

// the window committed last (LDS arrays i_*): run by wave CH_INS while the next window's queries are in
// flight.  Those queries read the index for everything older and the LDS arrays for this window, so
// nothing waits for these stores; they are complete (vmcnt) before the barrier that ends the phase,
// i.e. before the window after next looks for them in HBM.
__device__ inline void chain_insert_window(const QsGraphDev &G, const QsBucketGeom &bg, const long long *i_idx,
                                           const double *i_x, const double *i_y, const int *i_type, int k, int lane,
                                           long long &n_lms, long long &n_misc, unsigned int &pool, bool &pile)
{
    const long long idx = lane < 32 ? i_idx[lane] : LL_MAX;
    const double x = lane < 32 ? i_x[lane] : 0, y = lane < 32 ? i_y[lane] : 0;
    const int type = lane < k ? i_type[lane] : 0;
    int cx, cy;
    const long long kb = bucket_cell(x, y, type, bg, cx, cy) ? bucket_key(type, cx, cy, bg) : -1;   // -1: side list
    chain_insert_lanes(G, lane < k, lane, idx, kb, x, y, type, k, lane, n_lms, n_misc, pool, pile);
}

// wave-uniform read of one lane of a double
__device__ inline double rlf64(double v, int src_lane) { return __longlong_as_double(rl64(__double_as_longlong(v), src_lane)); }

// minimum over the wave of an unsigned value, by DPP (no LDS, no loop over lanes): an inclusive min-scan
// inside each row of 16 lanes, then lane 15 of rows 0/2 into rows 1/3, then lane 31 into rows 2/3
__device__ inline unsigned int wave_min_u32(unsigned int v)
{
#define QS_DPP_MIN(ctrl, rows) v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, ctrl, rows, 0xf, false))
    QS_DPP_MIN(0x111, 0xf);     // row_shr:1
    QS_DPP_MIN(0x112, 0xf);     // row_shr:2
    QS_DPP_MIN(0x114, 0xf);     // row_shr:4
    QS_DPP_MIN(0x118, 0xf);     // row_shr:8
    QS_DPP_MIN(0x142, 0xa);     // row_bcast:15
    QS_DPP_MIN(0x143, 0xc);     // row_bcast:31
#undef QS_DPP_MIN
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum over the wave of a non-negative 64-bit value: high words first, then the low words of the
// lanes that hold the lowest high word
__device__ inline long long wave_min_nonneg_i64(long long v)
{
    const unsigned int hi = (unsigned int)((unsigned long long)v >> 32), lo = (unsigned int)v;
    const unsigned int mhi = wave_min_u32(hi);
    const unsigned int mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
    return (long long)(((unsigned long long)mhi << 32) | mlo);
}

// the window at the head of the events: the next events whose node index is < first + win (a
// contiguous prefix of the lanes).  Every role computes it for itself from the same LDS arrays.
// The window itself (which of the next events have a node index < first + win: a contiguous prefix; how many) depends on
// the events' node indices alone, so the fetch wave works it out one phase AHEAD, off the decisions' critical path: an
// event outside the window carries agent -1 in n_a, and the window's size is one LDS word (s_wk).  The roles only read.
struct ChWindow { long long v_idx; double px, py; int v_a, type, k; bool v_inw; };
// (every LDS word is read unconditionally by lanes 0..31 and masked afterwards: ONE round trip for the phase's inputs)
__device__ inline ChWindow chain_window(const long long *nidx, const int *na, const int *wk, bool active, int lane,
                                        const int *ntype = nullptr, const double *npx = nullptr, const double *npy = nullptr)
{
    ChWindow w;
    const int l = lane & 31;
    const int a = na[l], k = *wk;
    const long long idx = nidx ? nidx[l] : 0;
    const int ty = ntype ? ntype[l] : 0;
    const double x = npx ? npx[l] : 0, y = npy ? npy[l] : 0;
    w.v_inw = active && lane < 32 && a >= 0;
    w.v_idx = w.v_inw ? idx : LL_MAX;
    w.v_a = w.v_inw ? a : 0;
    w.type = w.v_inw ? ty : 0;
    w.px = w.v_inw ? x : 0; w.py = w.v_inw ? y : 0;
    w.k = active ? k : 0;
    return w;
}
// fetch wave: 32 events from position q0 on, laid out as the window that starts there
// DYN (graphs with more agents than owner waves): the fetch wave also deals the window's agents to the owner waves -- the r-th
// distinct agent of the window (in order of its first event) goes to wave 1 + r mod n_ow -- so that a window's queries spread
// over the owners whatever the agents' numbers are: nown[l] = the owner wave of event l, inwin[a] = the owner wave of agent a
// while it has events in the window in this buffer (0 otherwise; the marks of the window that used the buffer before are
// taken back first).
template <bool DYN>
__device__ inline void chain_fetch(const QsSlamBatch &sb, unsigned int q0, unsigned int e1, int lane, int win, long long *nidx, int *na,
                                   int *ntype, double *npx, double *npy, int *wk, int *nown = nullptr, int *inwin = nullptr,
                                   int n_ow = 1, bool clear_old = false)
{
    const unsigned int q = q0 + lane;
    long long f_idx = LL_MAX; int f_a = 0, f_type = 0; double f_px = 0, f_py = 0;
    const bool have = lane < 32 && q < e1;
    if (have) { f_idx = sb.ev_node[q]; f_a = sb.ev_agent[q]; f_type = sb.ev_type[q]; f_px = sb.ev_px[q]; f_py = sb.ev_py[q]; }
    const long long first = rl64(f_idx, 0);
    const bool inw = have && f_idx - first < win;
    const int k = __popcll(__ballot(inw));
    if (DYN) {
        int ow = 0, r = 0;
        for (unsigned long long rem = __ballot(inw); rem; r = (r + 1 == n_ow) ? 0 : r + 1) {
            const int ld = __ffsll((long long)rem) - 1;
            const int aa = __builtin_amdgcn_readlane(f_a, ld);
            const unsigned long long grp = __ballot(inw && f_a == aa);
            if (inw && f_a == aa) ow = 1 + r;
            rem &= ~grp;
        }
        if (clear_old && lane < 32) { const int oa = na[lane]; if (oa >= 0) inwin[oa] = 0; }
        if (inw) inwin[f_a] = ow;                        // (after the line above in program order: LDS keeps a wave's stores in order)
        if (lane < 32) nown[lane] = ow;
    }
    if (lane < 32) { nidx[lane] = f_idx; na[lane] = inw ? f_a : -1; ntype[lane] = f_type; npx[lane] = f_px; npy[lane] = f_py; }
    if (lane == 0) *wk = k;
}

// One workgroup per pose graph, one role per wave, one phase per window, one barrier per phase.  Each role
// runs its OWN loop over the phases (the same count in every role: it follows from the events alone), so
// that a role's loop keeps only that role's pointers and constants in registers.  In phase V, side by side:
//   wave 0          writes the closure records of window V - 1 in node order (each closing event's pose under
//                   the drift at window start is its own arithmetic), then lays out window V's landmarks:
//                   node index and type (LDS)
//   waves 1..CH_AGW each agent's owner finds the agent's first eligible event of window V, poses it, and
//                   scans the index for it -- the index holds everything up to window V - 2, window V - 1's
//                   landmarks (final poses) are in LDS; then the agent's next event, until a match: only an
//                   agent's FIRST eligible event with a match closes the loop (:304-318).  The owner applies
//                   the closure itself (the arithmetic is wave 0's, on the same operands) to its agent's
//                   state, and gives the agent's events in the window their final pose (LDS)
//   wave CH_INS     moves window V - 1's landmarks into the index
//   wave CH_FETCH   fetches the events after window V.
// Nothing inside a phase waits for another wave: every LDS word has one writer per phase and its readers
// come a barrier later.
// Waves without a role leave at once (a barrier counts the waves still running).
// ONE: at most CH_AGW bots per graph, i.e. one agent per owner (its state in lane 0): the agent -> owner /
// lane arithmetic folds away.
// DENSE: the variant for graphs that have grown a PILE -- a bucket chain of more than CH_PILE_NODES pool nodes.  A query whose
// point is next to such a bucket but out of reach of its landmarks, and whose own first match is younger than the pile, walks
// the whole chain: a dependent node read per 7 entries, the reference's O(L) scan only slower (tests/k4_adversarial.py: 5.8 ms
// per query at 10^5 entries, 3.6 x the CPU's list scan).  In this variant a query that is still walking after
// max(8, L / 512) node rounds scans the insertion-ordered landmark LOG instead, 128 entries per coalesced round -- the
// reference's own loop (:294), a wave wide (0.46 ms per query on the same pile).  The insert wave raises a flag when a chain
// passes CH_PILE_NODES; the host reads it with the edge-ray count at the end of an ingest and launches this variant from then
// on.  The plain variant stays as it is: the round counter and the scan cost every query ~100 cycles (3 %) when compiled in.
template <bool ONE, bool DENSE>
__global__ void __launch_bounds__(CH_THREADS)
qs_slam_chain_kernel(QsGraphDev *__restrict__ graphs, QsSlamBatch sb, QsBucketGeom bg, int bots_per_graph,
                     int max_agent, int win, int min_between, double r2thr, double corr,
                     double *__restrict__ drift, long long *__restrict__ last_closure,
                     unsigned long long *__restrict__ counters, int raw_pose, unsigned int *__restrict__ pile_flag)
{
    // raw_pose: poses are used as given (PoseGraphSLAM.add_pose object API: the caller has already
    // applied its drift correction, dual_bot_mapper.py:855-857 precede :908)
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QsGraphDev *const Gp = graphs + g;
    const int bot0 = g * bots_per_graph + 1;
    const int nb = min(bots_per_graph, max_agent - bot0 + 1);

    // drift and last closure of every agent as they stand at the START of a window, by window parity:
    // during window V the owners read/write their registers and publish the state after V into
    // [(V + 1) & 1], while wave 0 reads [V & 1] (commit of V - 1, prepare of V)
    __shared__ double s_dx[2][QS_MAX_AGENT + 1], s_dy[2][QS_MAX_AGENT + 1];
    __shared__ long long s_lastc[2][QS_MAX_AGENT + 1];
    __shared__ unsigned int s_acnt[QS_MAX_AGENT + 1];
    __shared__ long long w_ridx[2][32];         // query results of a window: matched landmark (node index, pose)
    __shared__ double w_rx[2][32], w_ry[2][32];
    __shared__ long long n_idx[2][32];          // events of the current window / the one after it (by parity)
    __shared__ double n_px[2][32], n_py[2][32];
    __shared__ int n_a[2][32], n_type[2][32];
    // A window's landmarks -- node index, type (wave 0), pose (each agent's owner) -- in a ring of three slots: window V is
    // final at the end of phase V, is moved into the index by wave CH_INS during phase V + 1 (its stores may still be in
    // flight during phase V + 2), and is read from here by the queries of windows V + 1 and V + 2; slot V % 3 is free again
    // in phase V + 3.
    __shared__ long long i_idx[3][32];
    __shared__ double i_x[3][32], i_y[3][32];
    __shared__ int i_type[3][32];
    __shared__ int s_ik[3];
    __shared__ int s_wk[2];                           // size of the window whose events are in n_*[parity]
    // !ONE (more agents than owner waves): owner wave of every event / of every agent with events in the window (chain_fetch)
    __shared__ int n_own[ONE ? 1 : 2][ONE ? 1 : 32];
    __shared__ int s_inwin[ONE ? 1 : 2][ONE ? 1 : QS_MAX_AGENT + 1];
    const int n_ow = min(CH_AGW, nb);               // owner waves in use
    __shared__ long long s_nmisc;
    __shared__ long long s_nlms;                      // DENSE: landmark-log entries whose stores are complete

    for (int t = tid; t < nb; t += CH_THREADS) {
        s_dx[0][t] = drift[2 * (bot0 + t)];
        s_dy[0][t] = drift[2 * (bot0 + t) + 1];
        s_lastc[0][t] = last_closure[bot0 + t];
        s_acnt[t] = sb.agent_ev[bot0 + t];             // where the agent's next closure record goes
    }
    if (tid == 0) { s_nmisc = Gp->n_misc; s_nlms = Gp->n_lms; s_ik[0] = 0; s_ik[1] = 0; s_ik[2] = 0; }
    if (!ONE) {
        for (int t = tid; t < 2 * (QS_MAX_AGENT + 1); t += CH_THREADS) (&s_inwin[0][0])[t] = 0;
        if (tid < 64) { (&n_own[0][0])[tid] = 0; (&n_a[0][0])[tid] = -1; }
        __syncthreads();                             // (n_a is read back by the first fetches into each buffer)
    }
    if (tid < 96) {
        const int h = tid >> 5, t = tid & 31;
        i_idx[h][t] = LL_MAX; i_x[h][t] = 0; i_y[h][t] = 0; i_type[h][t] = 0;
        if (h < 2) w_ridx[h][t] = LL_MAX;
    }

    const unsigned int e0 = sb.ev_base[g], e1 = sb.ev_base[g + 1];

    if (wave == CH_FETCH) {
        if (ONE) chain_fetch<false>(sb, e0, e1, lane, win, n_idx[0], n_a[0], n_type[0], n_px[0], n_py[0], &s_wk[0]);
        else chain_fetch<true>(sb, e0, e1, lane, win, n_idx[0], n_a[0], n_type[0], n_px[0], n_py[0], &s_wk[0], n_own[0], s_inwin[0], n_ow, true);
    }
    __syncthreads();

    unsigned int e = e0;
    int par = 0, ring = 0;                       // window parity (events, states, results), window number mod 3 (landmark slots)
    bool have_prev = false;
#define CH_R1 (ring == 0 ? 2 : ring - 1)       // slot of window V - 1
#define CH_R2 (ring == 2 ? 0 : ring + 1)       // slot of window V - 2
// what every role does at the end of a phase
#ifdef QS_CHAIN_PROF3
// per-window busy time of every role of graph 0 (first CH_TRACE_WINDOWS phases): [phase][wave] cycles from the barrier to the
// role's arrival at the next one; tools/chain_trace.py reads it through qs_debug_chain_trace
#define CH_P3_DECL unsigned long long p3_busy = 0, p3_t = __builtin_amdgcn_s_memtime(); unsigned int p3_w = 0
#define CH_PHASE_END(active_, k_)  { const unsigned long long p3_d = __builtin_amdgcn_s_memtime() - p3_t; p3_busy += p3_d; \
    if (g == 0 && lane == 0 && p3_w < CH_TRACE_WINDOWS) g_chain_trace[p3_w * 16 + wave] = p3_d; p3_w++; } \
    lds_barrier(); p3_t = __builtin_amdgcn_s_memtime(); e += (k_); have_prev = (active_); if (active_) { par ^= 1; ring = ring == 2 ? 0 : ring + 1; }
#define CH_P3_REPORT(slot_) if (lane == 0) atomicAdd(&counters[slot_], p3_busy)
#else
#define CH_P3_DECL
#define CH_PHASE_END(active_, k_)  lds_barrier(); e += (k_); have_prev = (active_); if (active_) { par ^= 1; ring = ring == 2 ? 0 : ring + 1; }
#define CH_P3_REPORT(slot_)
#endif

    if (wave == 0) {
        // =================================== wave 0: commit + prepare ===================================
        long long n_cls = Gp->n_cls;
        const long long n_cls0 = n_cls, cap_cls = Gp->cap_cls;
        long long *const cl_lm_idx = Gp->cl_lm_idx, *const cl_node_idx = Gp->cl_node_idx;
        double *const cl_dx = Gp->cl_dx, *const cl_dy = Gp->cl_dy;
        unsigned char *const cl_agent = Gp->cl_agent;
        unsigned long long st_windows = 0, st_a = 0, st_b = 0;
        const unsigned long long t0_cyc = __builtin_amdgcn_s_memtime(), t0_real = __builtin_amdgcn_s_memrealtime();
        // the window it prepared (its closure records are written one phase later)
        long long idx = LL_MAX;
        int a = 0;
        double x = 0, y = 0;
        bool inw = false;
        CH_P3_DECL;
        for (;;) {
            const bool active = e < e1;
            if (!active && !have_prev) break;
            const unsigned long long ta0 = __builtin_amdgcn_s_memtime();
            const ChWindow W = chain_window(n_idx[par], n_a[par], &s_wk[par], active, lane, n_type[par], n_px[par], n_py[par]);
            if (have_prev) {
                // ---- closure records of window V - 1, in node order ----
                const long long m_idx = lane < 32 ? w_ridx[par ^ 1][lane] : LL_MAX;
                const double m_x = lane < 32 ? w_rx[par ^ 1][lane] : 0, m_y = lane < 32 ? w_ry[par ^ 1][lane] : 0;
                // the query phase stops an agent at its first match, so a result marks exactly the closing event
                const bool closes = inw && m_idx != LL_MAX;
                const unsigned long long cmask = __ballot(closes);
                if (closes) {
                    const double ex = m_x - x, ey = m_y - y;                                   // :311-312
                    const double cdx = ex * corr, cdy = ey * corr;                             // :314-315
                    const long long slot = n_cls + __popcll(cmask & ((1ull << lane) - 1));
                    if (slot < cap_cls) {
                        cl_lm_idx[slot] = m_idx; cl_node_idx[slot] = idx;                      // :317
                        cl_dx[slot] = cdx; cl_dy[slot] = cdy;
                        cl_agent[slot] = (unsigned char)(bot0 + a);
                    }
                    const unsigned int pos = s_acnt[a];
                    sb.acl_node[pos] = idx; sb.acl_dx[pos] = s_dx[par][a]; sb.acl_dy[pos] = s_dy[par][a];   // :911-914
                    s_acnt[a] = pos + 1;
                }
                n_cls += __popcll(cmask);
                if (lane < 32) w_ridx[par ^ 1][lane] = LL_MAX;          // free for the window after this one
                st_a += __builtin_amdgcn_s_memtime() - ta0;
            }
            if (active) {
                // ---- window V: the events' poses with the drift at window start (what a closing event is matched
                // at); node index and type laid out as the window's landmarks: self.landmarks.append(...)  :288 ----
                idx = W.v_idx; a = W.v_a;
                inw = W.v_inw;
                if (!inw) a = 0;
                x = raw_pose ? W.px : W.px + s_dx[par][a];              // rx += cdx  :856
                y = raw_pose ? W.py : W.py + s_dy[par][a];              // ry += cdy  :857
                if (lane < 32) { i_idx[ring][lane] = inw ? idx : LL_MAX; i_type[ring][lane] = inw ? W.type : 0; }
                if (lane == 0) s_ik[ring] = W.k;
                st_windows++;
            } else {
                inw = false;
            }
            CH_PHASE_END(active, W.k);
            st_b += __builtin_amdgcn_s_memtime() - ta0;
        }
        if (lane == 0) {
            atomicAdd(&counters[QS_CNT_CLOSURES], (unsigned long long)(n_cls - n_cls0));
            if (pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAINW_HIT - QS_FLAG_PILE, (unsigned int)(n_cls - n_cls0));
            atomicAdd(&counters[QS_CNT_SLAM_WINDOWS], st_windows);
#if !defined(QS_CHAIN_PROF) && !defined(QS_CHAIN_PROF2) && !defined(QS_CHAIN_PROF3)
            atomicAdd(&counters[QS_CNT_SLAM_CYC_A], st_a); atomicAdd(&counters[QS_CNT_SLAM_CYC_B], st_b);
#endif
            CH_P3_REPORT(QS_CNT_SLAM_CYC_A);
            atomicAdd(&counters[QS_CNT_SLAM_CYCLES], __builtin_amdgcn_s_memtime() - t0_cyc);
            atomicAdd(&counters[QS_CNT_SLAM_REALTIME], __builtin_amdgcn_s_memrealtime() - t0_real);
            Gp->n_nodes = Gp->n_nodes + sb.acc_total[g];
            Gp->n_cls = n_cls;
        }
        for (int t = lane; t < nb; t += QS_WAVE) sb.acl_cnt[bot0 + t] = s_acnt[t] - sb.agent_ev[bot0 + t];
    } else if (wave == CH_FETCH) {
        // =================================== wave CH_FETCH: event fetch ===================================
        CH_P3_DECL;
        for (;;) {
            const bool active = e < e1;
            if (!active && !have_prev) break;
            const int k = active ? s_wk[par] : 0;
            if (!ONE && active) {
                // the states of the agents WITHOUT events in window V go over to the next window unchanged (those with events are
                // written by the owner wave they were dealt to): one writer per word
                for (int a = lane; a < nb; a += QS_WAVE)
                    if (s_inwin[par][a] == 0) { s_dx[par ^ 1][a] = s_dx[par][a]; s_dy[par ^ 1][a] = s_dy[par][a]; s_lastc[par ^ 1][a] = s_lastc[par][a]; }
            }
            if (active) {
                if (ONE) chain_fetch<false>(sb, e + k, e1, lane, win, n_idx[par ^ 1], n_a[par ^ 1], n_type[par ^ 1], n_px[par ^ 1], n_py[par ^ 1], &s_wk[par ^ 1]);
                else chain_fetch<true>(sb, e + k, e1, lane, win, n_idx[par ^ 1], n_a[par ^ 1], n_type[par ^ 1], n_px[par ^ 1], n_py[par ^ 1], &s_wk[par ^ 1],
                                       n_own[par ^ 1], s_inwin[par ^ 1], n_ow, true);
            }
            CH_PHASE_END(active, k);
        }
        CH_P3_REPORT(QS_CNT_SLAM_CYC_B);
    } else if (wave == CH_INS) {
        // =================================== wave CH_INS: index insert ===================================
        const QsGraphDev G = *Gp;
        long long n_lms = G.n_lms, n_misc = G.n_misc;
        unsigned int pool = G.nodes_used;
        bool pile = false;
        CH_P3_DECL;
        for (;;) {
            const bool active = e < e1;
            if (!active && !have_prev) break;
            const int k = active ? s_wk[par] : 0;
            // the stores of the insert one phase ago have had that whole phase: they are done by now (the queries of THIS
            // phase still see that window in LDS, the next phase's only in the index), and the side list's new length shows
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0 && s_nmisc != n_misc) s_nmisc = n_misc;
            if (DENSE && lane == 0) s_nlms = n_lms;
            if (have_prev) {
                const int r1 = CH_R1;
                if (s_ik[r1] > 0)
                    chain_insert_window(G, bg, i_idx[r1], i_x[r1], i_y[r1], i_type[r1], s_ik[r1], lane, n_lms, n_misc, pool, pile);
            }
            CH_PHASE_END(active, k);
        }
        CH_P3_REPORT(QS_CNT_SLAM_CYC_C);
        if (lane == 0) {
            atomicAdd(&counters[QS_CNT_LANDMARKS], (unsigned long long)(n_lms - G.n_lms));
            Gp->n_lms = n_lms;
            Gp->n_misc = n_misc;
            Gp->nodes_used = pool;
            if (pile && pile_flag) *pile_flag = 1u;
        }
    } else if (wave >= 1 && wave <= n_ow) {
        // =================================== query waves ===================================
        // ONE: the wave owns agent wave - 1 for good; its drift and last closure live in the wave's registers (every lane) -- the
        // authoritative copy, nothing else writes them.
        // !ONE: the agents' states live in LDS, double-buffered by window parity; an agent with events in the window belongs to
        // the owner wave the fetch wave dealt it to (chain_fetch), which reads its state at window start and writes its state
        // after the window; the states of the agents WITHOUT events in the window are carried over by the fetch wave.  Every
        // state word has exactly one writer per phase, and its readers come a barrier later.  An owner without an event in the
        // window goes straight to the barrier: with 64 agents in one graph the workgroup is bound by instruction issue on its
        // four SIMDs (16 waves), not by any one wave's latency.
        __builtin_amdgcn_s_setprio(3);             // the decisions are the critical path: ahead of the helper waves on a shared SIMD
        const int own = (wave - 1) + n_ow * lane;  // (!ONE: the agents whose final state this wave writes back at the end)
        double c_dx = 0, c_dy = 0;
        long long c_last = 0;
        unsigned int st_nomatch = 0;               // eligible queries that found nothing (what the free-running form would wait on)
        if (ONE) { c_dx = s_dx[0][wave - 1]; c_dy = s_dy[0][wave - 1]; c_last = s_lastc[0][wave - 1]; }     // one agent: every lane holds its state
        const QsNodeG g_nodes = (QsNodeG)Gp->nodes;
        const QsU32G g_next = (QsU32G)Gp->nd_next;
        // lane = (bucket of the 3x3 neighbourhood, entry of that bucket's current 7-entry node): a node
        // scan is three coalesced row loads (idx, x, y of 9 nodes)
        const int nbk = lane / QS_NODE_CAP, se = lane % QS_NODE_CAP;
        const int last_lane = min(nbk * QS_NODE_CAP + QS_NODE_CAP - 1, 63);
        const int nb_dx = (nbk % 3) - 1, nb_dy = (nbk / 3) - 1;        // the lane's neighbour of the 3x3
        unsigned long long st_misc = 0;            // side-list scans (rare path; the tests read it)
#ifdef QS_CHAIN_STATS
        unsigned long long st_rounds = 0, st_iters = 0;
#define CH_STAT(x) (x)++
#else
#define CH_STAT(x) do { } while (0)
#endif
#ifdef QS_CHAIN_PROF
        unsigned long long pq_a = 0, pq_b = 0, pq_c = 0;
#endif
        CH_P3_DECL;
#ifdef QS_CHAIN_PROF2
        unsigned long long p2_head = 0, p2_setup = 0, p2_scan = 0, p2_post = 0, p2_pub = 0, p2_bar = 0, p2_t4 = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
            const bool active = e < e1;
            if (!active && !have_prev) break;
#ifdef QS_CHAIN_PROF2
            const unsigned long long p2_t0 = __builtin_amdgcn_s_memtime();
            p2_bar += p2_t0 - p2_t4;
#endif
            int ev_own = 0;
            if (!ONE) {
                ev_own = (active && lane < 32) ? n_own[par][lane] : 0;
                if (__ballot(ev_own == wave) == 0) {               // none of the window's agents is this wave's
                    const int k0 = active ? s_wk[par] : 0;
                    CH_PHASE_END(active, k0);
                    continue;
                }
            }
            // window V - 1 is not in the index yet: its landmarks (final poses) are in LDS, in node order.  Read in
            // the same LDS round trip as the events, looked at in the shadow of a query's first node loads.
            const int lsl = lane < 32 ? CH_R2 : CH_R1;                     // lanes 0..31: window V - 2, lanes 32..63: window V - 1 (node order)
            const long long li = i_idx[lsl][lane & 31]; const double lx = i_x[lsl][lane & 31], ly = i_y[lsl][lane & 31]; const int lt = i_type[lsl][lane & 31];
            const long long nm = s_nmisc;
            const long long nl = DENSE ? s_nlms : 0;
            const int dense_after = (int)((nl >> 9) > 8 ? ((nl >> 9) < 100000 ? (nl >> 9) : 100000) : 8);     // DENSE: node rounds before a query scans the log
            const ChWindow W = chain_window(n_idx[par], n_a[par], &s_wk[par], active, lane, n_type[par], n_px[par], n_py[par]);
            const double o_dx = c_dx, o_dy = c_dy;                                  // ONE: drift at window start
            // !ONE: the state at window start of the lane's agent (any lane: the reads are unconditional, one round trip)
            double st_dx = 0, st_dy = 0;
            long long st_last = 0;
            if (!ONE) { st_dx = s_dx[par][W.v_a]; st_dy = s_dy[par][W.v_a]; st_last = s_lastc[par][W.v_a]; }
            const bool ownlane = W.v_inw && (ONE ? W.v_a + 1 : ev_own) == wave;
            // the lane's event may close if its agent is past its cool-down (:304); an agent that finds a match in
            // this window takes its later events off the list, so the state at window start decides for all of them
            const long long lane_last = ONE ? c_last : st_last;
            const bool may_close = ownlane && W.v_idx - lane_last >= min_between;
            // !ONE: the state after the window, as far as it is known now: unchanged (a closure below overwrites its agent's)
            double nw_dx = st_dx, nw_dy = st_dy;
            long long nw_last = st_last;
            if (!ONE && ownlane) { s_dx[par ^ 1][W.v_a] = st_dx; s_dy[par ^ 1][W.v_a] = st_dy; s_lastc[par ^ 1][W.v_a] = st_last; }
#ifdef QS_CHAIN_PROF2
            const unsigned long long p2_t1 = __builtin_amdgcn_s_memtime();
            p2_head += p2_t1 - p2_t0;
#endif
            for (unsigned long long qrem = __ballot(may_close); qrem; qrem &= qrem - 1) {
#if defined(QS_CHAIN_PROF) || defined(QS_CHAIN_PROF2)
                const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
                const int src = __ffsll((long long)qrem) - 1;
                const int qa = __builtin_amdgcn_readlane(W.v_a, src);
                const long long qidx = rl64(W.v_idx, src);
                const double odx = ONE ? c_dx : rlf64(st_dx, src), ody = ONE ? c_dy : rlf64(st_dy, src);   // agent qa's drift at window start
                const double spx = rlf64(W.px, src), spy = rlf64(W.py, src);
                const double qx = raw_pose ? spx : spx + odx;                         // rx += cdx  :856
                const double qy = raw_pose ? spy : spy + ody;                         // ry += cdy  :857
                const int qtype = __builtin_amdgcn_readlane(W.type, src);
                const long long limit = qidx - min_between;                        // :300
                int qcx, qcy;
                const bool indexed = bucket_cell(qx, qy, qtype, bg, qcx, qcy);
                unsigned int node = 0;
                if (indexed && lane < 9 * QS_NODE_CAP) node = 1u + (unsigned int)bucket_key(qtype, qcx + nb_dx, qcy + nb_dy, bg);   // the entry's own first node
                long long best = LL_MAX, gbest = LL_MAX;
                double bx = 0, by = 0;
                CH_STAT(st_rounds);
                long long l_idx = LL_MAX;
                double l_x = 0, l_y = 0;
#if defined(QS_CHAIN_PROF) || defined(QS_CHAIN_PROF2)
                const unsigned long long tq1 = __builtin_amdgcn_s_memtime();
#endif
                int rounds = 0;
                bool dense = false;
                for (bool first_scan = true;; first_scan = false) {
                    const bool anyn = __ballot(node != 0) != 0;
                    long long id = LL_MAX;
                    double nx = 0, ny = 0;
                    unsigned int nxt = 0;
                    long long lastid = LL_MAX;                         // the node's last entry: same 64-byte row as `id`,
                    if (node) {                                         // read directly instead of a cross-lane shuffle
                        const QsNodeG nd = g_nodes + node;
                        id = nd->idx[se]; lastid = nd->idx[QS_NODE_CAP - 1]; nx = nd->x[se]; ny = nd->y[se]; nxt = g_next[node];
                    }
                    if (first_scan) {
                        const double dx = qx - lx, dy = qy - ly;
                        const bool cand = li <= limit && lt == qtype && dx * dx + dy * dy < r2thr;    // (li = LL_MAX: no landmark)
                        const unsigned long long cm = __ballot(cand);
                        if (cm) {
                            const int w = __ffsll((long long)cm) - 1;
                            l_idx = rl64(li, w); l_x = rlf64(lx, w); l_y = rlf64(ly, w);
                        }
                    }
                    if (!anyn) break;
                    if (DENSE && rounds++ == dense_after) { dense = true; break; }
                    CH_STAT(st_iters);
                    const bool inlim = node != 0 && id <= limit;      // empty slots read as a huge index
                    bool newhit = false;
                    if (inlim && best == LL_MAX) {
                        const double dx = qx - nx, dy = qy - ny;
                        if (dx * dx + dy * dy < r2thr) { best = id; bx = nx; by = ny; newhit = true; }   // :308-309
                    }
                    if (__ballot(newhit)) {
                        // node indices are >= 0, and those that match are <= limit: below 2^32 the low words decide
                        const long long v = (unsigned long long)limit >> 32 ? wave_min_nonneg_i64(newhit ? best : LL_MAX)
                                                                            : (long long)wave_min_u32(newhit ? (unsigned int)best : 0xffffffffu);
                        gbest = v < gbest ? v : gbest;
                    }
                    const unsigned long long hitm = __ballot(best != LL_MAX);
                    const unsigned long long limm = __ballot(inlim);
                    // a bucket's chain goes on only if none of its entries matched, its node was full
                    // and within the limit, and its last entry is still older than the best match
                    const bool b_hit = ((hitm >> (nbk * QS_NODE_CAP)) & 0x7full) != 0;
                    const bool b_full = ((limm >> last_lane) & 1ull) != 0;
                    if (node) node = (b_hit || !b_full || nxt == 0 || lastid >= gbest) ? 0u : nxt;
                }
#if defined(QS_CHAIN_PROF) || defined(QS_CHAIN_PROF2)
                const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
#endif
                double wx = 0, wy = 0;
                if (DENSE && dense) {
                    // self.landmarks in insertion order (:294): the first entry of the query's type within the radius among
                    // those old enough (:300); entries are in node order, so the scan ends at the first one that is too new
                    gbest = LL_MAX;
                    const QS_GLOBAL long long *const g_idx = (const QS_GLOBAL long long *)Gp->lm_idx;
                    const QS_GLOBAL unsigned char *const g_type = (const QS_GLOBAL unsigned char *)Gp->lm_type;
                    const QS_GLOBAL double *const g_x = (const QS_GLOBAL double *)Gp->lm_x, *const g_y = (const QS_GLOBAL double *)Gp->lm_y;
                    for (long long c0 = 0; c0 < nl; c0 += 2 * QS_WAVE) {
                        st_misc++;                                      // (counted with the side-list rounds: linear scans)
                        const long long k0 = c0 + lane, k1 = c0 + QS_WAVE + lane;
                        long long i0 = LL_MAX, i1 = LL_MAX; int t0 = 0, t1 = 0; double x0 = 0, y0 = 0, x1 = 0, y1 = 0;
                        if (k0 < nl) { i0 = g_idx[k0]; t0 = g_type[k0]; x0 = g_x[k0]; y0 = g_y[k0]; }
                        if (k1 < nl) { i1 = g_idx[k1]; t1 = g_type[k1]; x1 = g_x[k1]; y1 = g_y[k1]; }
                        const double ax = qx - x0, ay = qy - y0, cx2 = qx - x1, cy2 = qy - y1;
                        const bool c0ok = i0 <= limit && t0 == qtype && ax * ax + ay * ay < r2thr;
                        const bool c1ok = i1 <= limit && t1 == qtype && cx2 * cx2 + cy2 * cy2 < r2thr;
                        const unsigned long long m0 = __ballot(c0ok), m1 = __ballot(c1ok);
                        if (m0) { const int w = __ffsll((long long)m0) - 1; gbest = rl64(i0, w); wx = rlf64(x0, w); wy = rlf64(y0, w); break; }
                        if (m1) { const int w = __ffsll((long long)m1) - 1; gbest = rl64(i1, w); wx = rlf64(x1, w); wy = rlf64(y1, w); break; }
                        if (__ballot((k0 < nl && i0 > limit) || (k1 < nl && i1 > limit))) break;
                    }
                } else if (gbest != LL_MAX) {                           // uniform
                    const int w = __ffsll((long long)__ballot(best == gbest)) - 1;
                    wx = rlf64(bx, w); wy = rlf64(by, w);
                }
                // landmarks outside the directory: linear scan in insertion order (rare)
                if (nm > 0 && !(DENSE && dense)) {                     // (the log scan covers the side list's landmarks too)
                    const unsigned int *const misc = Gp->misc;
                    const long long *const lm_idx = Gp->lm_idx;
                    const unsigned char *const lm_type = Gp->lm_type;
                    const double *const lm_x = Gp->lm_x, *const lm_y = Gp->lm_y;
                    for (long long c0 = 0; c0 < nm; c0 += QS_WAVE) {
                        st_misc++;
                        const long long k2 = c0 + lane;
                        bool cand = false, beyond = false;
                        long long li = LL_MAX; double lx = 0, ly = 0;
                        if (k2 < nm) {
                            const unsigned int slot = misc[k2];
                            li = lm_idx[slot];
                            beyond = li > limit;
                            if (!beyond && lm_type[slot] == qtype) {
                                lx = lm_x[slot]; ly = lm_y[slot];
                                const double dx = qx - lx, dy = qy - ly;
                                cand = dx * dx + dy * dy < r2thr;
                            }
                        }
                        const unsigned long long cm = __ballot(cand);
                        if (cm) {
                            const int w = __ffsll((long long)cm) - 1;
                            const long long widx = rl64(li, w);
                            if (widx < gbest) { gbest = widx; wx = rlf64(lx, w); wy = rlf64(ly, w); }
                            break;
                        }
                        if (__ballot(beyond)) break;
                    }
                }
                if (l_idx < gbest) { gbest = l_idx; wx = l_x; wy = l_y; }
                if (gbest != LL_MAX) {
                    // the agent's later events cannot close (:304): off the wave's list (this one goes with the loop step)
                    qrem &= ~(__ballot(W.v_a == qa) & ~((2ull << src) - 1));
                    if (lane == 0) { w_ridx[par][src] = gbest; w_rx[par][src] = wx; w_ry[par][src] = wy; }
                    // the closure, applied to the agent's state by its owner (wave 0 writes the records)
                    const double ex = wx - qx, ey = wy - qy;                                   // :311-312
                    const double cdx = ex * corr, cdy = ey * corr;                             // :314-315
                    const double ndx = odx + cdx, ndy = ody + cdy;                             // :911-914
                    if (ONE) { c_dx = ndx; c_dy = ndy; c_last = qidx; }                         // :318
                    else {
                        if (lane == 0) { s_dx[par ^ 1][qa] = ndx; s_dy[par ^ 1][qa] = ndy; s_lastc[par ^ 1][qa] = qidx; }   // (after the carry-over above)
                        if (ownlane && W.v_a == qa) { nw_dx = ndx; nw_dy = ndy; nw_last = qidx; }
                    }
                } else st_nomatch++;
#ifdef QS_CHAIN_PROF
                { const unsigned long long tq3 = __builtin_amdgcn_s_memtime(); pq_a += tq1 - tq0; pq_b += tq2 - tq1; pq_c += tq3 - tq2; }
#endif
#ifdef QS_CHAIN_PROF2
                { const unsigned long long tq3 = __builtin_amdgcn_s_memtime(); p2_setup += tq1 - tq0; p2_scan += tq2 - tq1; p2_post += tq3 - tq2; }
#endif
            }
#ifdef QS_CHAIN_PROF2
            const unsigned long long p2_t2 = __builtin_amdgcn_s_memtime();
#endif
            if (ONE && active && lane == 0) { s_dx[par ^ 1][wave - 1] = c_dx; s_dy[par ^ 1][wave - 1] = c_dy; s_lastc[par ^ 1][wave - 1] = c_last; }
            // the window's landmarks get their final pose from their agent's owner: the drift at window start,
            // or -- later events of an agent that closed in this window -- the drift after the closure (:855-857)
            if (ONE) {                                                    // one agent per owner: its state is in lane 0
                const bool after = W.v_idx > c_last;
                const double ddx = after ? c_dx : o_dx, ddy = after ? c_dy : o_dy;
                if (ownlane) { i_x[ring][lane] = raw_pose ? W.px : W.px + ddx; i_y[ring][lane] = raw_pose ? W.py : W.py + ddy; }
            } else if (ownlane) {
                const bool after = W.v_idx > nw_last;
                const double ddx = after ? nw_dx : st_dx, ddy = after ? nw_dy : st_dy;
                i_x[ring][lane] = raw_pose ? W.px : W.px + ddx;
                i_y[ring][lane] = raw_pose ? W.py : W.py + ddy;
            }
#ifdef QS_CHAIN_PROF2
            p2_t4 = __builtin_amdgcn_s_memtime();
            p2_pub += p2_t4 - p2_t2;
#endif
            CH_PHASE_END(active, W.k);
        }
#ifdef QS_CHAIN_PROF2
        if (lane == 0 && wave == 1) {
            atomicAdd(&counters[QS_CNT_SLAM_CYC_A], p2_head); atomicAdd(&counters[QS_CNT_SLAM_CYC_B], p2_setup);
            atomicAdd(&counters[QS_CNT_SLAM_CYC_C], p2_scan); atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], p2_post);
            atomicAdd(&counters[QS_CNT_EKF_WRAP_CLAMP], p2_pub); atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], p2_bar);
        }
#endif
#ifdef QS_CHAIN_PROF3
        if (lane == 0) atomicAdd(&counters[wave == 1 ? QS_CNT_SLAM_MISC_ITERS : QS_CNT_EKF_WRAP_CLAMP], p3_busy);
#endif
        if (ONE) {
            if (lane == 0) { drift[2 * (bot0 + wave - 1)] = c_dx; drift[2 * (bot0 + wave - 1) + 1] = c_dy; last_closure[bot0 + wave - 1] = c_last; }
        } else if (own < nb) {                                // the states after the last window are in the buffer `par` names now
            drift[2 * (bot0 + own)] = s_dx[par][own];
            drift[2 * (bot0 + own) + 1] = s_dy[par][own];
            last_closure[bot0 + own] = s_lastc[par][own];
        }
        if (lane == 0) {
            if (st_nomatch && pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAINW_MISS - QS_FLAG_PILE, st_nomatch);
#if defined(QS_CHAIN_STATS) && !defined(QS_CHAIN_PROF2)
            atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], st_rounds);
            atomicAdd(&counters[QS_CNT_SLAM_NODE_ITERS], st_iters);
#endif
#if !defined(QS_CHAIN_PROF2) && !defined(QS_CHAIN_PROF3)
            if (st_misc) atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], st_misc);
#endif
#ifdef QS_CHAIN_PROF
            atomicAdd(&counters[QS_CNT_SLAM_CYC_A], pq_a); atomicAdd(&counters[QS_CNT_SLAM_CYC_B], pq_b); atomicAdd(&counters[QS_CNT_SLAM_CYC_C], pq_c);
#endif
        }
    }
#undef CH_PHASE_END
}

// ---- the chain, free-running form (graphs with at most CH_AGW agents) ---------------------------------------------------------
// The windowed kernel above synchronises every role once per window.  It does not have to: a closure decision needs its agent's
// own drift (the owner's registers) and the landmarks at least MIN_POSES_BETWEEN nodes older than the query, and
//   * landmarks enter the index in node order, so whatever the index holds is a PREFIX of self.landmarks;
//   * the first match in list order (:294-318) is the lowest node index among the matches.
// Hence a match found among the landmarks the index already holds is FINAL -- every landmark still on its way is newer than
// all of those -- and only a query that finds nothing has to wait until every landmark it may see has arrived.  On the
// reference's workloads nearly every query finds a match laps old, so:
//   owner waves (one per agent)  run their agent's recurrence at their own pace: the agent's next event that may close (:304),
//                  its pose, a query of the index (entries up to the committer's frontier), the closure, next.  An owner hands
//                  over its DECISIONS only (closing node, matched landmark, correction: an LDS ring) and how far it has decided.
//   the committer (one wave)     takes events in NODE order as far as every agent has decided, up to 64 at a time: it keeps every
//                  agent's drift as the decisions it has passed leave it (the owner's additions in the owner's order), poses
//                  the events, writes closure records (:317), the landmark log (:288), the bucket index; then -- its stores
//                  complete -- moves the frontier: the node index up to which the index is complete and visible.
// An owner that finds no match below the frontier while the frontier is still short of its limit waits for the committer and asks
// again; the owner with the oldest pending event never waits for anyone (everything older is final), so the scheme cannot
// lock up; a full decision ring holds its owner back until the committer -- which then has work -- drains it.
#define FR_RING 64
#define FR_PEND 256            // pending-pose ring of the free-running kernel (events)
// Every wait of this kernel ends by the argument above.  A kernel that never ends would take the GPU with it, so the waits are
// bounded all the same (~1 s): a wave that runs out of patience leaves a mark in QS_CNT_SLAM_ROUNDS (bit 40) and goes on --
// the results are then wrong and every parity check says so -- instead of hanging.
// (Once one wait has run out, every other one ends within 64 polls: a broken hand-over costs a second, not a second per event.)
#define FR_SPIN_MAX (1u << 24)
#define FR_SPIN(cond) do { unsigned int sp_ = 0; while (cond) {                                                                                   \
        if (++sp_ > FR_SPIN_MAX) { if (lane == 0) atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], 1ull << 40); break; }                                   \
        if ((sp_ & 63u) == 0 && (__hip_atomic_load(&counters[QS_CNT_SLAM_ROUNDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 40)) break;         \
        __builtin_amdgcn_s_sleep(1); } } while (0)

// one query: the reference's first match among the landmarks of type qtype with node index <= eff, within the radius of
// (qx, qy); lane = (bucket of the 3 x 3 neighbourhood, entry of that bucket's current node).  LL_MAX: none.
// (the pieces of a query's first round, so that an owner can have the NEXT decision's rows on their way while it works on this one)
struct FqRows { long long id, lastid; double nx, ny; unsigned int nxt; };
// the lane's node: the first node of its bucket of the 3 x 3 around (qx, qy); 0: none.  (qcx, qcy): the centre cell
__device__ inline unsigned int fq_node0(const QsBucketGeom &bg, double qx, double qy, int qtype, int lane, int &qcx, int &qcy)
{
    const int nbk = lane / QS_NODE_CAP;
    const int nb_dx = (nbk % 3) - 1, nb_dy = (nbk / 3) - 1;
    const bool indexed = bucket_cell(qx, qy, qtype, bg, qcx, qcy);
    return (indexed && lane < 9 * QS_NODE_CAP) ? 1u + (unsigned int)bucket_key(qtype, qcx + nb_dx, qcy + nb_dy, bg) : 0u;
}
__device__ inline void fq_load(QsNodeG g_nodes, QsU32G g_next, unsigned int node, int lane, FqRows &r)
{
    const int se = lane % QS_NODE_CAP;
    r.id = LL_MAX; r.lastid = LL_MAX; r.nx = 0; r.ny = 0; r.nxt = 0;
    if (node) {
        const QsNodeG nd = g_nodes + node;
        r.id = nd->idx[se]; r.lastid = nd->idx[QS_NODE_CAP - 1]; r.nx = nd->x[se]; r.ny = nd->y[se]; r.nxt = g_next[node];
    }
}
// pre: the rows of the first nodes (node0), loaded by the caller -- complete for every entry up to the frontier it read before
// it asked for them, which is what eff must not exceed
template <bool DENSE>
__device__ inline long long free_query(const QsGraphDev *Gp, QsNodeG g_nodes, QsU32G g_next, const QsBucketGeom &bg, double qx, double qy,
                                       int qtype, long long eff, double r2thr, long long nm, long long nl, int lane, double &wx, double &wy,
                                       unsigned long long &st_misc, unsigned int node0, bool use_pre, const FqRows &pre)
{
    const int nbk = lane / QS_NODE_CAP;
    const int last_lane = min(nbk * QS_NODE_CAP + QS_NODE_CAP - 1, 63);
    unsigned int node = node0;
    long long best = LL_MAX, gbest = LL_MAX;
    double bx = 0, by = 0;
    const int dense_after = (int)((nl >> 9) > 8 ? ((nl >> 9) < 100000 ? (nl >> 9) : 100000) : 8);
    int rounds = 0;
    bool dense = false;
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 5
    const unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
    while (__ballot(node != 0)) {
        if (DENSE && rounds++ == dense_after) { dense = true; break; }
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 2
        st_misc++;                                                       // (profile build: node rounds)
#endif
        FqRows r;
        if (use_pre) { r = pre; use_pre = false; }                        // (by value and a flag: a pointer would put the rows in scratch)
        else fq_load(g_nodes, g_next, node, lane, r);
        const long long id = r.id, lastid = r.lastid;
        const double nx = r.nx, ny = r.ny;
        const unsigned int nxt = r.nxt;
        const bool inlim = node != 0 && id <= eff;                       // empty slots read as a huge index
        bool newhit = false;
        if (inlim && best == LL_MAX) {
            const double dx = qx - nx, dy = qy - ny;
            if (dx * dx + dy * dy < r2thr) { best = id; bx = nx; by = ny; newhit = true; }   // :308-309
        }
        if (__ballot(newhit)) {
            const long long v = (unsigned long long)eff >> 32 ? wave_min_nonneg_i64(newhit ? best : LL_MAX)
                                                               : (long long)wave_min_u32(newhit ? (unsigned int)best : 0xffffffffu);
            gbest = v < gbest ? v : gbest;
        }
        const unsigned long long hitm = __ballot(best != LL_MAX);
        const unsigned long long limm = __ballot(inlim);
        const bool b_hit = ((hitm >> (nbk * QS_NODE_CAP)) & 0x7full) != 0;
        const bool b_full = ((limm >> last_lane) & 1ull) != 0;
        if (node) node = (b_hit || !b_full || nxt == 0 || lastid >= gbest) ? 0u : nxt;
    }
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 5
    st_misc += __builtin_amdgcn_s_memtime() - tl_;                       // (profile build 5: the node rounds)
#endif
    wx = 0; wy = 0;
    if (DENSE && dense) {
        // self.landmarks in insertion order (:294), a wave wide; the scan ends at the first entry that is too new
        gbest = LL_MAX;
        const QS_GLOBAL long long *const g_idx = (const QS_GLOBAL long long *)Gp->lm_idx;
        const QS_GLOBAL unsigned char *const g_type = (const QS_GLOBAL unsigned char *)Gp->lm_type;
        const QS_GLOBAL double *const g_x = (const QS_GLOBAL double *)Gp->lm_x, *const g_y = (const QS_GLOBAL double *)Gp->lm_y;
        for (long long c0 = 0; c0 < nl; c0 += QS_WAVE) {
            st_misc++;
            const long long k0 = c0 + lane;
            long long i0 = LL_MAX; int t0 = 0; double x0 = 0, y0 = 0;
            if (k0 < nl) { i0 = g_idx[k0]; t0 = g_type[k0]; x0 = g_x[k0]; y0 = g_y[k0]; }
            const double ax = qx - x0, ay = qy - y0;
            const bool ok = i0 <= eff && t0 == qtype && ax * ax + ay * ay < r2thr;
            const unsigned long long m0 = __ballot(ok);
            if (m0) { const int w = __ffsll((long long)m0) - 1; gbest = rl64(i0, w); wx = rlf64(x0, w); wy = rlf64(y0, w); break; }
            if (__ballot(k0 < nl && i0 > eff)) break;
        }
        return gbest;
    }
    if (gbest != LL_MAX) {
        const int w = __ffsll((long long)__ballot(best == gbest)) - 1;
        wx = rlf64(bx, w); wy = rlf64(by, w);
    }
    if (nm > 0) {                                                       // landmarks outside the directory (types > 5): linear, rare
        const unsigned int *const misc = Gp->misc;
        const long long *const lm_idx = Gp->lm_idx;
        const unsigned char *const lm_type = Gp->lm_type;
        const double *const lm_x = Gp->lm_x, *const lm_y = Gp->lm_y;
        for (long long c0 = 0; c0 < nm; c0 += QS_WAVE) {
            st_misc++;
            const long long k2 = c0 + lane;
            bool cand = false, beyond = false;
            long long li = LL_MAX; double lx = 0, ly = 0;
            if (k2 < nm) {
                const unsigned int slot = misc[k2];
                li = lm_idx[slot];
                beyond = li > eff;
                if (!beyond && lm_type[slot] == qtype) {
                    lx = lm_x[slot]; ly = lm_y[slot];
                    const double dx = qx - lx, dy = qy - ly;
                    cand = dx * dx + dy * dy < r2thr;
                }
            }
            const unsigned long long cm = __ballot(cand);
            if (cm) {
                const int w = __ffsll((long long)cm) - 1;
                const long long widx = rl64(li, w);
                if (widx < gbest) { gbest = widx; wx = rlf64(lx, w); wy = rlf64(ly, w); }
                break;
            }
            if (__ballot(beyond)) break;
        }
    }
    return gbest;
}

__device__ inline long long lds_ld64(const long long *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st64(long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline unsigned int lds_ld32(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st32(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// WAVES: 16, or 8 for graphs of up to 5 agents -- half the waves per SIMD is twice the registers per wave (256), and the owner keeps
// two chunks of events and two decisions' bucket rows in registers.
// POST: the owners post their events' landmark poses for the others' queries (below).  It costs every decision ~240 cycles, and a
// stream whose queries nearly always find their match in the index (the reference's sessions: 99.8 %) never looks at them: the
// library runs the instantiation without it until a batch had more than one decision in eight wait for the committer
// (qs_api.hip, chain_stats_poll).
template <bool DENSE, int WAVES, bool POST>
__global__ void __launch_bounds__(WAVES * QS_WAVE)
qs_slam_chain_free_kernel(QsGraphDev *__restrict__ graphs, QsSlamBatch sb, QsBucketGeom bg, int bots_per_graph,
                          int max_agent, int min_between, double r2thr, double corr,
                          double *__restrict__ drift, long long *__restrict__ last_closure,
                          unsigned long long *__restrict__ counters, int raw_pose, unsigned int *__restrict__ pile_flag)
{
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QsGraphDev *const Gp = graphs + g;
    const int bot0 = g * bots_per_graph + 1;
    const int nb = min(bots_per_graph, max_agent - bot0 + 1);
    constexpr int AGW = WAVES - 3, INS = WAVES - 1, THREADS = WAVES * QS_WAVE;   // owner waves 1 .. AGW; the committer
    const int n_ow = min(AGW, nb);                  // owner waves in use
    constexpr int NA = AGW;                         // (graphs with more agents: qs_slam_chain_dyn_kernel)
    constexpr int RD = FR_RING;                     // decisions an agent can be ahead of the committer

    // An owner hands the committer its DECISIONS only -- (closing node, matched landmark, correction), in the agent's order, in a
    // ring of RD slots -- and how far it has decided (s_prog).  Everything else about an event follows from those: the
    // committer walks the events in node order, keeps every agent's drift as the decisions it has passed leave it (the same
    // additions in the same order as the owner's: the same doubles), and poses, logs, indexes and records from that.
    __shared__ long long q_idx[NA][RD], q_midx[NA][RD];                         // closing node; matched landmark's node
    __shared__ double q_cdx[NA][RD], q_cdy[NA][RD];                             // the closure's correction (:314-315)
    __shared__ unsigned int s_push[NA], s_cons[NA];                             // decisions pushed by the owner / taken by the committer
    __shared__ long long s_prog[AGW];          // every event of the owner's agents with a node index below this is decided (LL_MAX: all)
    __shared__ long long s_frontier;              // every landmark with node index <= this is in the index, complete and visible
    __shared__ long long s_nmisc, s_nlms;         // side-list / log entries that go with that frontier
    // committer's own: every agent's drift as of the events it has passed, and where the agent's next closure record goes
    __shared__ double c_ddx[NA], c_ddy[NA];
    __shared__ unsigned int c_apos[NA];
    // Landmarks that are decided about but not in the index yet: every owner posts the pose each of its events' landmarks is appended
    // at (:288) as it passes the event, in a ring by position in the event list; a query that finds nothing in the index looks
    // there for what lies between the frontier and its limit instead of waiting for the committer (FR_PEND - 64: how far an owner
    // may be ahead of the committer).
    __shared__ long long p_node[POST ? FR_PEND : 1];   // the event that has the slot (its node index), written after ...
    __shared__ double p_x[POST ? FR_PEND : 1], p_y[POST ? FR_PEND : 1];   // ... its pose
    __shared__ unsigned int s_comm;               // events committed (count from e0): where the frontier stands in the list

    const unsigned int e0 = sb.ev_base[g], e1 = sb.ev_base[g + 1];
    if (POST) for (int t = tid; t < FR_PEND; t += THREADS) p_node[t] = -LL_MAX - 1;
    for (int t = tid; t < NA; t += THREADS) {
        s_push[t] = 0; s_cons[t] = 0;
        if (t < nb) { c_ddx[t] = drift[2 * (bot0 + t)]; c_ddy[t] = drift[2 * (bot0 + t) + 1]; c_apos[t] = sb.agent_ev[bot0 + t]; }
    }
    if (tid < AGW) s_prog[tid] = tid < n_ow ? (e0 < e1 ? sb.ev_node[e0] : LL_MAX) : LL_MAX;
    if (tid == 0) { s_frontier = e0 < e1 ? sb.ev_node[e0] - 1 : LL_MAX; s_nmisc = Gp->n_misc; s_nlms = Gp->n_lms; s_comm = 0; }
    __syncthreads();

    if (wave >= 1 && wave <= n_ow) {
        // =================================== owner of agent wave - 1 ===================================
        __builtin_amdgcn_s_setprio(3);             // the decisions are the critical path
        const int a = wave - 1;
        // its agent's state, in every lane (the authoritative copy; nothing else writes it)
        const int my_agent = a;
        const bool my_valid = my_agent < nb;
        double c_dx = my_valid ? drift[2 * (bot0 + my_agent)] : 0, c_dy = my_valid ? drift[2 * (bot0 + my_agent) + 1] : 0;
        long long c_last = my_valid ? last_closure[bot0 + my_agent] : 0;
        unsigned int pushed = 0, cons_c = 0;
        const QsNodeG g_nodes = (QsNodeG)Gp->nodes;
        const QsU32G g_next = (QsU32G)Gp->nd_next;
        unsigned long long st_misc = 0, st_wait = 0;
#ifdef QS_FREE_PROF
        unsigned long long pf_wait = 0, pf_query = 0, pf_post = 0, pf_total0 = __builtin_amdgcn_s_memtime();
#endif
        auto publish_prog = [&](long long nxt) {
            // what is handed over is the decision ring (LDS), written by this wave just before: the LDS unit takes a wave's
            // operations in the order they were issued, so all that is needed is that the compiler keeps that order
            __asm__ volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&s_prog[a], nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // the next chunk's events are requested while this chunk's are dealt with
        auto load_chunk = [&](unsigned int q0, int &ag, long long &idx, int &type, double &px, double &py, long long &next_first) {
            const unsigned int q = q0 + lane;
            const bool have = q < e1;
            ag = have ? (int)sb.ev_agent[q] : -1;
            idx = have ? sb.ev_node[q] : LL_MAX;
            type = have ? (int)sb.ev_type[q] : 0;
            px = have ? sb.ev_px[q] : 0; py = have ? sb.ev_py[q] : 0;
            next_first = q0 + QS_WAVE < e1 ? sb.ev_node[q0 + QS_WAVE] : LL_MAX;   // a lower bound on the owner's next event after this chunk
        };
        int ag_n = -1, type_n = 0; long long idx_n = LL_MAX, nf_n = LL_MAX; double px_n = 0, py_n = 0;
        if (e0 < e1) load_chunk(e0, ag_n, idx_n, type_n, px_n, py_n, nf_n);
        for (unsigned int q0 = e0; q0 < e1; q0 += QS_WAVE) {
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 4
            const unsigned long long tc_ = __builtin_amdgcn_s_memtime();
#endif
            const int ag = ag_n, type = type_n;
            const long long idx = idx_n, next_first = nf_n;
            const double px = px_n, py = py_n;
            if (q0 + QS_WAVE < e1) load_chunk(q0 + QS_WAVE, ag_n, idx_n, type_n, px_n, py_n, nf_n);
            const unsigned long long own = __ballot(ag == a);
            unsigned long long done = 0;                                            // own lanes decided so far
            unsigned long long posted = 0;                                          // own lanes whose landmark pose is in the pending ring
            // own events up to lane hi (with the agent's drift as it is now: the events since its last closure)
            auto post_upto = [&](int hi) {
                if (!POST) return;
                const unsigned long long m = own & ~posted & ((2ull << hi) - 1);
                if ((m >> lane) & 1) {
                    const unsigned int ps = (q0 - e0 + (unsigned int)lane) % FR_PEND;
                    p_x[ps] = raw_pose ? px : px + c_dx; p_y[ps] = raw_pose ? py : py + c_dy;   // rx += cdx  :856-857
                }
                __asm__ volatile("" ::: "memory");
                if ((m >> lane) & 1) __hip_atomic_store(&p_node[(q0 - e0 + (unsigned int)lane) % FR_PEND], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                posted |= m;
            };
            // (a slot's last holder is long committed)
            if (POST && own) FR_SPIN(q0 - e0 + QS_WAVE > __hip_atomic_load(&s_comm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + (unsigned int)(FR_PEND - QS_WAVE));
            // the events that may close a loop (:304: their agent is past its cool-down); the ones before the first need no decision
            auto eligible = [&]() -> unsigned long long {
                return own & ~done & __ballot(idx - c_last >= min_between);
            };
            unsigned long long elig = eligible();
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 4
            pf_wait += __builtin_amdgcn_s_memtime() - tc_;                         // (profile build 4: top of a chunk)
#endif
            // A decision: (event f of the chunk, its pose with the agent's drift as it is NOW, the first nodes of its nine buckets,
            // their rows on the way).  The rows of the NEXT decision are asked for as soon as this one's closure is known -- before
            // the hand-over to the committer, whose LDS traffic then runs in the shadow of the loads.
            int f = 0, qtype = 0;
            long long qidx = 0, fr_rows = 0;
            double qx = 0, qy = 0;
            unsigned int node0 = 0;
            FqRows rows = {LL_MAX, LL_MAX, 0, 0, 0u};
            auto start_decision = [&](unsigned long long e_) {
                // (the frontier first: the rows are asked for after it is read -- an older value is only more cautious)
                fr_rows = __hip_atomic_load(&s_frontier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                f = __ffsll((long long)e_) - 1;
                post_upto(f);
                qidx = rl64(idx, f);
                const double spx = rlf64(px, f), spy = rlf64(py, f);
                qx = raw_pose ? spx : spx + c_dx; qy = raw_pose ? spy : spy + c_dy;               // rx += cdx  :856-857
                qtype = __builtin_amdgcn_readlane(type, f);
                int qcx, qcy;
                node0 = fq_node0(bg, qx, qy, qtype, lane, qcx, qcy);
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fq_load(g_nodes, g_next, node0, lane, rows);
            };
            if (elig) start_decision(elig);
            while (elig) {
                const int qa = a;
                const long long qidx_c = qidx;                                      // (this decision's: start_decision below moves on)
                const double qx_c = qx, qy_c = qy;
                const int f_c = f, qtype_c = qtype;
                // :300 -- and a node never sees its own landmark (appended after the check, :288): with MIN_POSES_BETWEEN < 1
                // the newest landmark a query can see is still the one before it
                const long long limit = qidx_c - (min_between > 1 ? min_between : 1);
                long long gbest; double wx, wy;
#ifdef QS_FREE_PROF
                const unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
                {
                    bool pre = true;
                    for (long long fr = fr_rows;;) {
                        const long long nm = s_nmisc, nl = DENSE ? s_nlms : 0;
                        gbest = free_query<DENSE>(Gp, g_nodes, g_next, bg, qx_c, qy_c, qtype_c, fr < limit ? fr : limit, r2thr, nm, nl, lane, wx, wy, st_misc,
                                                  node0, pre, rows);
                        if (gbest != LL_MAX || fr >= limit) break;                  // a match below the frontier is final; so is "none" once all are in
                        pre = false;
                        // Nothing in the index up to fr.  The landmarks with fr < node <= limit are events the committer has not got
                        // to: their poses are in the pending ring once their owners have passed them.  64 events a round, youngest
                        // first; the OLDEST match is the reference's first match.
                        st_wait++;
                        publish_prog(qidx_c);                                       // (the committer has to get past this owner's older events)
                        if (!POST) {                                                // wait until it has everything up to the limit in the index
                            if (lds_ld64(&s_frontier) <= fr) FR_SPIN(lds_ld64(&s_frontier) < limit);
                            fr = lds_ld64(&s_frontier);
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");    // the index as of that frontier, not older
                            continue;
                        }
                        bool stale = false;
                        const unsigned int r_me = q0 - e0 + (unsigned int)f_c;      // this event's place in the list
                        for (unsigned int back = 0; back < r_me; back += QS_WAVE) {
                            const long long jr = (long long)r_me - 1 - (long long)back - lane;
                            const bool hv = jr >= 0;
                            const long long nd = hv ? sb.ev_node[e0 + (unsigned int)jr] : -LL_MAX - 1;
                            const int ty = hv ? (int)sb.ev_type[e0 + (unsigned int)jr] : 0;
                            const bool inr = hv && nd > fr && nd <= limit;
                            const unsigned int ps = hv ? (unsigned int)jr % FR_PEND : 0u;
                            // posted -- or, if the committer overtook it meanwhile (its slot may have a new holder), in the index by now
                            FR_SPIN(__ballot(inr && __hip_atomic_load(&p_node[ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != nd &&
                                             __hip_atomic_load(&s_frontier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nd) != 0);
                            __asm__ volatile("" ::: "memory");
                            if (__ballot(inr && __hip_atomic_load(&p_node[ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != nd)) { stale = true; break; }
                            bool hit = false;
                            double lx = 0, ly = 0;
                            if (inr && ty == qtype_c) {
                                lx = p_x[ps]; ly = p_y[ps];
                                const double dx = qx_c - lx, dy = qy_c - ly;
                                hit = dx * dx + dy * dy < r2thr;                    // :308-309
                            }
                            __asm__ volatile("" ::: "memory");
                            if (__ballot(inr && __hip_atomic_load(&p_node[ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != nd)) { stale = true; break; }
                            const unsigned long long hm = __ballot(hit);
                            if (hm) { const int w = 63 - __builtin_clzll(hm); gbest = rl64(nd, w); wx = rlf64(lx, w); wy = rlf64(ly, w); }
                            if (!__ballot(hv && lane == QS_WAVE - 1 && nd > fr)) break;  // the round's oldest event is in the index already
                        }
                        if (!stale) break;                                          // match or none: final (index up to fr, events up to limit)
                        gbest = LL_MAX;                                             // the index has grown under the scan: once more, from it
                        fr = lds_ld64(&s_frontier);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");        // the index as of that frontier, not older
                    }
                }
#ifdef QS_FREE_PROF
                pf_query += __builtin_amdgcn_s_memtime() - tq_;
#endif
                done |= (2ull << f_c) - 1;                                          // (lanes up to f: decided)
                double cdx = 0, cdy = 0;
                if (gbest != LL_MAX) {
                    const double ex = wx - qx_c, ey = wy - qy_c;                    // :311-312
                    cdx = ex * corr; cdy = ey * corr;                               // :314-315
                    c_dx += cdx; c_dy += cdy; c_last = qidx_c;                      // :911-914, :318
                }
                // the next decision, its rows on the way ...
                elig = eligible();
                if (elig) start_decision(elig);
                // ... and this one handed over
                if (gbest != LL_MAX) {
                    // the agent's decision ring: a slot must be free
                    unsigned int pq = pushed, cq = cons_c;
                    if (pq - cq >= (unsigned int)RD) {                              // looks full
                        cq = lds_ld32(&s_cons[qa]);
                        if (pq - cq >= (unsigned int)RD) {
#ifdef QS_FREE_PROF
                            const unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
                            publish_prog(qidx_c);
                            FR_SPIN(pq - (cq = lds_ld32(&s_cons[qa])) >= (unsigned int)RD);
#ifdef QS_FREE_PROF
                            pf_wait += __builtin_amdgcn_s_memtime() - t_;
#endif
                        }
                    }
                    if (lane == 0) {
                        const unsigned int sl = pq % RD;
                        q_idx[qa][sl] = qidx_c; q_midx[qa][sl] = gbest; q_cdx[qa][sl] = cdx; q_cdy[qa][sl] = cdy;
                    }
                    pushed = pq + 1; cons_c = cq;
                    __asm__ volatile("" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&s_push[qa], pq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // decided: everything of this owner below its next event that may close (or, failing one in this chunk, below the
                // next chunk's first event)
                publish_prog(elig ? qidx : next_first);
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 3
                pf_post += __builtin_amdgcn_s_memtime() - tq_;                      // (query + everything after it)
#endif
            }
            post_upto(QS_WAVE - 1);                                                 // (the agent's events after its last decision of the chunk)
            publish_prog(next_first);
        }
        publish_prog(LL_MAX);
        if (my_valid && lane == 0) {
            drift[2 * (bot0 + my_agent)] = c_dx; drift[2 * (bot0 + my_agent) + 1] = c_dy; last_closure[bot0 + my_agent] = c_last;
        }
        if (lane == 0) {
            if (st_misc) atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], st_misc);
            if (st_wait) { atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], st_wait); if (pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAIN_MISS - QS_FLAG_PILE, (unsigned int)st_wait); }
#ifdef QS_FREE_PROF
            if (a == 0) { atomicAdd(&counters[QS_CNT_SLAM_CYC_A], pf_wait); atomicAdd(&counters[QS_CNT_SLAM_CYC_B], pf_query);
                          if (QS_FREE_PROF >= 3) atomicAdd(&counters[QS_CNT_SLAM_CYC_C], pf_post);
                          atomicAdd(&counters[QS_CNT_SLAM_NODE_ITERS], __builtin_amdgcn_s_memtime() - pf_total0); }
#endif
        }
    } else if (wave == INS) {
        // =================================== the committer ===================================
        const QsGraphDev G = *Gp;
        long long n_lms = G.n_lms, n_misc = G.n_misc, n_cls = G.n_cls;
        unsigned int pool = G.nodes_used;
        bool pile = false;
        unsigned long long st_batches = 0;
        const unsigned long long t0_cyc = __builtin_amdgcn_s_memtime(), t0_real = __builtin_amdgcn_s_memrealtime();
        unsigned int e = e0, idle = 0;
#ifdef QS_FREE_PROF
        unsigned long long pf_idle = 0, pf_agents = 0, pf_insert = 0;
#endif
        // the events at the head of the list, a lane each: asked for again as soon as it is known how many of them go into the batch,
        // so that the next 64 arrive while this batch is dealt with
        long long node = LL_MAX, node_n = LL_MAX;
        int ag = 0, type_l = 0, ag_n = 0, type_n = 0;
        double px_l = 0, py_l = 0, px_n = 0, py_n = 0;
        auto load_events = [&](unsigned int from, long long &nd, int &a_, int &ty, double &x_, double &y_) {
            const unsigned int q = from + lane;
            const bool have = q < e1;
            nd = have ? sb.ev_node[q] : LL_MAX;
            a_ = have ? (int)sb.ev_agent[q] : 0;
            ty = have ? (int)sb.ev_type[q] : 0;
            x_ = have ? sb.ev_px[q] : 0; y_ = have ? sb.ev_py[q] : 0;
        };
        if (e < e1) load_events(e, node, ag, type_l, px_l, py_l);
        while (e < e1) {
            const bool have = e + lane < e1;
            // an event is ready when its agent has decided past it; the batch is the ready PREFIX (node order)
            // (relaxed, here and below: what the owners hand over is in LDS, which takes a wave's operations in the order they were
            // issued -- an acquire would make this wave wait for its own stores to HBM as well)
            const bool ready = have && __hip_atomic_load(&s_prog[ag], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > node;
            __asm__ volatile("" ::: "memory");
            const unsigned long long rm = __ballot(ready);
            const int k = rm == ~0ull ? 64 : (int)__builtin_ctzll(~rm);
            if (k == 0) {
                if (++idle > FR_SPIN_MAX) { if (lane == 0) atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], 1ull << 40); break; }   // (never: see FR_SPIN)
                if ((idle & 63u) == 0 && (__hip_atomic_load(&counters[QS_CNT_SLAM_ROUNDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 40)) break;
#ifdef QS_FREE_PROF
                pf_idle++;
#endif
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            idle = 0;
            if (e + k < e1) load_events(e + k, node_n, ag_n, type_n, px_n, py_n);
#ifdef QS_FREE_PROF
            const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
            const bool inw = lane < k;
            const int type = inw ? type_l : 0;
            const double px = inw ? px_l : 0, py = inw ? py_l : 0;
            // ---- every agent of the batch: its drift along its events, its decisions that fall into the batch ----
            double dx = 0, dy = 0, cdx = 0, cdy = 0;
            long long midx = LL_MAX;
            for (unsigned long long rem = __ballot(inw); rem;) {
                const int ld = __ffsll((long long)rem) - 1;
                const int aa = __builtin_amdgcn_readlane(ag, ld);
                const unsigned long long La = __ballot(inw && ag == aa);
                rem &= ~La;
                double cur_dx = c_ddx[aa], cur_dy = c_ddy[aa];
                unsigned int dc = s_cons[aa];
                const unsigned int dp = __hip_atomic_load(&s_push[aa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __asm__ volatile("" ::: "memory");
                unsigned int apos = c_apos[aa];
                unsigned long long left = La;                                        // lanes of aa not yet given their drift
                while (dc != dp) {
                    const unsigned int sl = dc % RD;
                    const long long qi = q_idx[aa][sl];
                    const unsigned long long lm = La & __ballot(node == qi);
                    if (!lm) break;                                                  // the agent's next decision is about a later event
                    const int lc = __ffsll((long long)lm) - 1;
                    const unsigned long long upto = left & ((2ull << lc) - 1);       // the closing event and the agent's events before it
                    if ((upto >> lane) & 1) { dx = cur_dx; dy = cur_dy; }            // matched / stored at the pose BEFORE the closure (:288, :308)
                    const double ecx = q_cdx[aa][sl], ecy = q_cdy[aa][sl];
                    if (lane == lc) { midx = q_midx[aa][sl]; cdx = ecx; cdy = ecy; }
                    cur_dx += ecx; cur_dy += ecy;                                    // drift_correction[agent] += ...  :911-914
                    if (lane == lc) { sb.acl_node[apos] = qi; sb.acl_dx[apos] = cur_dx; sb.acl_dy[apos] = cur_dy; }
                    apos++;
                    left &= ~upto;
                    dc++;
                }
                if ((left >> lane) & 1) { dx = cur_dx; dy = cur_dy; }
                __asm__ volatile("" ::: "memory");
                if (lane == ld) { c_ddx[aa] = cur_dx; c_ddy[aa] = cur_dy; c_apos[aa] = apos; __hip_atomic_store(&s_cons[aa], dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            }
#ifdef QS_FREE_PROF
            const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
            pf_agents += tp1 - tp0;
#endif
            const double x = raw_pose ? px : px + dx, y = raw_pose ? py : py + dy;   // rx += cdx, ry += cdy  :856-857
            // ---- closure records, in node order  (:317) ----
            const bool closes = inw && midx != LL_MAX;
            const unsigned long long cmask = __ballot(closes);
            if (closes) {
                const long long slot = n_cls + __popcll(cmask & ((1ull << lane) - 1));
                if (slot < G.cap_cls) {
                    G.cl_lm_idx[slot] = midx; G.cl_node_idx[slot] = node; G.cl_dx[slot] = cdx; G.cl_dy[slot] = cdy;
                    G.cl_agent[slot] = (unsigned char)(bot0 + ag);
                }
            }
            n_cls += __popcll(cmask);
            // ---- self.landmarks.append(...)  :288, and the bucket index ----
            int cx, cy;
            const long long kb = (inw && bucket_cell(x, y, type, bg, cx, cy)) ? bucket_key(type, cx, cy, bg) : -1;
            chain_insert_lanes(G, inw, lane, node, kb, x, y, type, k, lane, n_lms, n_misc, pool, pile);
            // ---- everything above complete, then the frontier moves ----
            // (workgroup scope: the readers are waves of this workgroup, on this CU; what is needed is that the stores are done)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const long long next_node = e + k < e1 ? rl64(node_n, 0) : LL_MAX;
            if (lane == 0) {
                s_nmisc = n_misc; s_nlms = n_lms;
                lds_st64(&s_frontier, next_node == LL_MAX ? LL_MAX : next_node - 1);
                __hip_atomic_store(&s_comm, e + k - e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            e += k;
            node = node_n; ag = ag_n; type_l = type_n; px_l = px_n; py_l = py_n;
            st_batches++;
#ifdef QS_FREE_PROF
            pf_insert += __builtin_amdgcn_s_memtime() - tp1;
#endif
        }
        if (lane == 0) {
            atomicAdd(&counters[QS_CNT_CLOSURES], (unsigned long long)(n_cls - G.n_cls));
            if (pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAIN_HIT - QS_FLAG_PILE, (unsigned int)(n_cls - G.n_cls));
            atomicAdd(&counters[QS_CNT_LANDMARKS], (unsigned long long)(n_lms - G.n_lms));
            atomicAdd(&counters[QS_CNT_SLAM_WINDOWS], st_batches);
#ifdef QS_FREE_PROF
            if (QS_FREE_PROF < 3) atomicAdd(&counters[QS_CNT_SLAM_CYC_C], pf_idle);
            atomicAdd(&counters[QS_CNT_EKF_WRAP_CLAMP], pf_agents); if (QS_FREE_PROF == 1) atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], pf_insert);
#endif
            atomicAdd(&counters[QS_CNT_SLAM_CYCLES], __builtin_amdgcn_s_memtime() - t0_cyc);
            atomicAdd(&counters[QS_CNT_SLAM_REALTIME], __builtin_amdgcn_s_memrealtime() - t0_real);
            Gp->n_nodes = G.n_nodes + sb.acc_total[g];
            Gp->n_cls = n_cls; Gp->n_lms = n_lms; Gp->n_misc = n_misc; Gp->nodes_used = pool;
            if (pile && pile_flag) *pile_flag = 1u;
        }
        for (int t = lane; t < nb; t += QS_WAVE) sb.acl_cnt[bot0 + t] = c_apos[t] - sb.agent_ev[bot0 + t];
    }
}

// ---- the free-running form for graphs with more agents than owner waves: events DEALT to the owners as they come free ----
// (up to 255 bots in ONE PoseGraphSLAM, dual_bot_mapper.py:267-275.)  With agents tied to waves the slowest wave sets the pace
// -- bucket chains differ in length from agent to agent --; here the 14 owner waves take the events off ONE counter, in node
// order.  An agent's decisions still happen one after the other: its state (drift, last closure: :911-914, :318) lives in LDS
// with a word saying up to which of its events it is decided, and whoever holds the agent's next event waits for that word.
// Which event that is -- the previous event of the same agent -- is worked out by wave 0 (the dispatcher), ahead of the owners.
// The committer is the one of qs_slam_chain_free_kernel; "ready" = below the oldest event any owner is working on.
// Waits point at OLDER events only (the agent's previous event, the frontier at q - MIN_POSES_BETWEEN, a ring slot held by an
// older decision of the agent; the dispatcher at events committed), so the oldest undecided event can always go on.
#define DY_RING 512             // dispatcher -> owners: per event, the node index of the agent's previous event
#define DY_PEND 256             // owners -> owners: per event under way or decided, the pose its landmark is stored at; an owner is
                                // never more than DY_PEND - 64 events ahead of the committer
#define DY_RING_SLOTS 1024      // decision slots in all: an agent can be 16 (up to 64 agents), 8 (up to 128) or 4 decisions ahead of the
                                // committer -- the further, the larger the committer's batches get when it is what everybody waits for
#define DY_OWNERS (CH_WAVES - 2)
template <bool DENSE>
__global__ void __launch_bounds__(CH_THREADS)
qs_slam_chain_dyn_kernel(QsGraphDev *__restrict__ graphs, QsSlamBatch sb, QsBucketGeom bg, int bots_per_graph,
                         int max_agent, int min_between, double r2thr, double corr,
                         double *__restrict__ drift, long long *__restrict__ last_closure,
                         unsigned long long *__restrict__ counters, int raw_pose, unsigned int *__restrict__ pile_flag)
{
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    QsGraphDev *const Gp = graphs + g;
    const int bot0 = g * bots_per_graph + 1;
    const int nb = min(bots_per_graph, max_agent - bot0 + 1);
    constexpr int NA = QS_MAX_AGENT + 1;
    constexpr long long LL_MIN_ = -LL_MAX - 1;

    __shared__ long long q_idx[DY_RING_SLOTS], q_midx[DY_RING_SLOTS];           // decisions [agent][slot]: closing node; matched landmark's node
    __shared__ double q_cdx[DY_RING_SLOTS], q_cdy[DY_RING_SLOTS];               // the closure's correction (:314-315)
    const unsigned int DY_RD = nb <= 64 ? 16u : nb <= 128 ? 8u : 4u;            // (slots per agent; a power of two)
    __shared__ unsigned int s_push[NA], s_cons[NA];                             // decisions pushed by the owners / taken by the committer
    __shared__ double a_dx[NA], a_dy[NA];                                       // the agent's drift and last closure as of ...
    __shared__ long long a_last[NA], a_node[NA];                                // ... its event a_node (decided up to and including it)
    __shared__ long long s_prog[CH_WAVES];        // the event an owner is working on (LL_MAX: none left): everything older than all of them is decided
    __shared__ long long s_frontier, s_nmisc, s_nlms;
    __shared__ unsigned int s_head, s_disp, s_comm;   // events handed out / prepared by the dispatcher / committed (counts from e0)
    __shared__ long long d_ring[DY_RING], d_last[NA];
    __shared__ long long p_node[DY_PEND];         // the event that has the slot (its node index), written after ...
    __shared__ double p_x[DY_PEND], p_y[DY_PEND]; // ... the pose its landmark is appended at (:288: before its own closure)
    __shared__ unsigned int d_tag[NA];
    __shared__ double c_ddx[NA], c_ddy[NA];       // committer: every agent's drift as of the events it has passed
    __shared__ unsigned int c_apos[NA], c_tag[NA];

    const unsigned int e0 = sb.ev_base[g], e1 = sb.ev_base[g + 1];
    for (int t = tid; t < DY_PEND; t += CH_THREADS) p_node[t] = LL_MIN_;
    for (int t = tid; t < NA; t += CH_THREADS) {
        s_push[t] = 0; s_cons[t] = 0; d_tag[t] = 0xffffffffu; c_tag[t] = 0xffffffffu; d_last[t] = LL_MIN_; a_node[t] = LL_MIN_;
        if (t < nb) {
            const double dx = drift[2 * (bot0 + t)], dy = drift[2 * (bot0 + t) + 1];
            a_dx[t] = dx; a_dy[t] = dy; c_ddx[t] = dx; c_ddy[t] = dy;
            a_last[t] = last_closure[bot0 + t]; c_apos[t] = sb.agent_ev[bot0 + t];
        }
    }
    if (tid < CH_WAVES) s_prog[tid] = (tid >= 1 && tid <= DY_OWNERS && e0 < e1) ? sb.ev_node[e0] : LL_MAX;
    if (tid == 0) { s_frontier = e0 < e1 ? sb.ev_node[e0] - 1 : LL_MAX; s_nmisc = Gp->n_misc; s_nlms = Gp->n_lms; s_head = 0; s_disp = 0; s_comm = 0; }
    __syncthreads();
#define LD_RLX(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define ST_RLX(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CBAR() __asm__ volatile("" ::: "memory")      // LDS takes a wave's operations in the order they were issued: keeping the compiler to it is all a hand-over needs

    if (wave == 0) {
        // =================================== the dispatcher ===================================
        for (unsigned int d = e0; d < e1; d += QS_WAVE) {
            const unsigned int r = d - e0;
            // a ring slot is free once the event that had it is committed
            FR_SPIN(r + QS_WAVE > DY_RING && LD_RLX(&s_comm) + DY_RING < r + QS_WAVE);
            const unsigned int q = d + lane;
            const bool have = q < e1;
            const int ag = have ? (int)sb.ev_agent[q] : 0;
            const long long node = have ? sb.ev_node[q] : LL_MAX;
            long long prev = LL_MIN_;
            bool todo = have;
            while (__ballot(todo)) {          // an agent's events of this chunk one per round, in node order
                if (todo) atomicMin(&d_tag[ag], (unsigned int)lane);
                CBAR();
                if (todo && LD_RLX(&d_tag[ag]) == (unsigned int)lane) { prev = d_last[ag]; d_last[ag] = node; d_tag[ag] = 0xffffffffu; todo = false; }
                CBAR();
            }
            if (have) d_ring[(r + lane) % DY_RING] = prev;
            CBAR();
            if (lane == 0) ST_RLX(&s_disp, min(r + QS_WAVE, e1 - e0));
        }
    } else if (wave <= DY_OWNERS) {
        // =================================== an owner ===================================
        __builtin_amdgcn_s_setprio(3);             // the decisions are the critical path
        const QsNodeG g_nodes = (QsNodeG)Gp->nodes;
        const QsU32G g_next = (QsU32G)Gp->nd_next;
        unsigned long long st_misc = 0, st_wait = 0;
        // two chunks of the event list in registers, a lane per event: the one the owner's event is in and the one after it
        unsigned int cb = 0xffffffffu, nbase = 0xffffffffu;
        int ag_c = 0, type_c = 0, ag_n = 0, type_n = 0;
        long long idx_c = LL_MAX, idx_n = LL_MAX;
        double px_c = 0, py_c = 0, px_n = 0, py_n = 0;
        auto load_chunk = [&](unsigned int q0, int &ag, long long &idx, int &type, double &px, double &py) {
            const unsigned int q = q0 + lane;
            const bool have = q < e1;
            ag = have ? (int)sb.ev_agent[q] : 0;
            idx = have ? sb.ev_node[q] : LL_MAX;
            type = have ? (int)sb.ev_type[q] : 0;
            px = have ? sb.ev_px[q] : 0; py = have ? sb.ev_py[q] : 0;
        };
#ifdef QS_FREE_PROF
        unsigned long long pf_wait = 0, pf_query = 0, pf_prev = 0, pf_grab = 0, pf_total0 = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
#ifdef QS_FREE_PROF
            const unsigned long long tg_ = __builtin_amdgcn_s_memtime();
#endif
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(&s_head, 1u);
            const unsigned int r = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            if (r >= e1 - e0) break;
            const unsigned int i = e0 + r;
            const unsigned int cbase = e0 + (r & ~(unsigned int)(QS_WAVE - 1));
            if (cbase != cb) {
                if (cbase == nbase) { ag_c = ag_n; idx_c = idx_n; type_c = type_n; px_c = px_n; py_c = py_n; }
                else load_chunk(cbase, ag_c, idx_c, type_c, px_c, py_c);
                cb = cbase; nbase = cbase + QS_WAVE;
                if (nbase < e1) load_chunk(nbase, ag_n, idx_n, type_n, px_n, py_n);
            }
            const int l = (int)(i - cb);
            const int qa = __builtin_amdgcn_readlane(ag_c, l);
            const long long qidx = rl64(idx_c, l);
            if (lane == 0) ST_RLX(&s_prog[wave], qidx);
            // the agent's previous event has to be decided
#ifdef QS_FREE_PROF
            const unsigned long long tv_ = __builtin_amdgcn_s_memtime();
            pf_grab += tv_ - tg_;
#endif
            FR_SPIN(LD_RLX(&s_disp) <= r);
            CBAR();
            const long long prev = d_ring[r % DY_RING];
            FR_SPIN(LD_RLX(&a_node[qa]) < prev);
            CBAR();
#ifdef QS_FREE_PROF
            pf_prev += __builtin_amdgcn_s_memtime() - tv_;
#endif
            const long long last = a_last[qa];
            const double odx = a_dx[qa], ody = a_dy[qa];
            const double spx = rlf64(px_c, l), spy = rlf64(py_c, l);
            const double qx = raw_pose ? spx : spx + odx, qy = raw_pose ? spy : spy + ody;       // rx += cdx  :856-857
            // the pose this event's landmark is appended at (:288) is known from here on: posted for the queries of younger events,
            // which then need not wait for the committer to put it into the index
            {
                FR_SPIN(r >= LD_RLX(&s_comm) + (unsigned int)(DY_PEND - QS_WAVE));  // (the slot's last holder is long committed)
                const unsigned int ps = r % DY_PEND;
                if (lane == 0) { p_x[ps] = qx; p_y[ps] = qy; }
                CBAR();
                if (lane == 0) ST_RLX(&p_node[ps], qidx);
            }
            if (qidx - last >= min_between) {                                       // :304
                const long long fr0 = LD_RLX(&s_frontier);
                const int qtype = __builtin_amdgcn_readlane(type_c, l);
                const long long limit = qidx - (min_between > 1 ? min_between : 1);  // :300; a node never sees its own landmark (:288)
                long long gbest; double wx, wy;
#ifdef QS_FREE_PROF
                const unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
                for (long long fr = fr0;; fr = lds_ld64(&s_frontier)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");            // the index as of that frontier, not older
                    const long long nm = s_nmisc, nl = DENSE ? s_nlms : 0;
                    int qcx_, qcy_;
                    gbest = free_query<DENSE>(Gp, g_nodes, g_next, bg, qx, qy, qtype, fr < limit ? fr : limit, r2thr, nm, nl, lane, wx, wy, st_misc,
                                              fq_node0(bg, qx, qy, qtype, lane, qcx_, qcy_), false, FqRows{LL_MAX, LL_MAX, 0, 0, 0u});
                    if (gbest != LL_MAX || fr >= limit) break;                      // a match below the frontier is final; so is "none" once all are in
                    // Nothing in the index up to fr.  The landmarks with fr < node <= limit are events the committer has not got to:
                    // their poses are in the pending ring as soon as their owners have started on them -- which waits for older events
                    // only.  64 events a round, youngest first; the OLDEST match is the reference's first match.
                    st_wait++;
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 6
                    const unsigned long long tw_ = __builtin_amdgcn_s_memtime();
#endif
                    bool stale = false;
                    for (unsigned int back = 0; back < r; back += QS_WAVE) {
                        const long long jr = (long long)r - 1 - (long long)back - lane;
                        const bool hv = jr >= 0;
                        const long long nd = hv ? sb.ev_node[e0 + (unsigned int)jr] : LL_MIN_;
                        const int ty = hv ? (int)sb.ev_type[e0 + (unsigned int)jr] : 0;
                        const bool inr = hv && nd > fr && nd <= limit;
                        const unsigned int ps = hv ? (unsigned int)jr % DY_PEND : 0u;
                        // posted -- or, if the committer overtook it meanwhile (its slot may have a new holder), in the index by now
                        FR_SPIN(__ballot(inr && LD_RLX(&p_node[ps]) != nd && LD_RLX(&s_frontier) < nd) != 0);
                        CBAR();
                        if (__ballot(inr && LD_RLX(&p_node[ps]) != nd)) { stale = true; break; }
                        bool hit = false;
                        double lx = 0, ly = 0;
                        if (inr && ty == qtype) {
                            lx = p_x[ps]; ly = p_y[ps];
                            const double dx = qx - lx, dy = qy - ly;
                            hit = dx * dx + dy * dy < r2thr;                        // :308-309
                        }
                        CBAR();
                        if (__ballot(inr && LD_RLX(&p_node[ps]) != nd)) { stale = true; break; }   // (the pose read belongs to that event)
                        const unsigned long long hm = __ballot(hit);
                        if (hm) { const int w = 63 - __builtin_clzll(hm); gbest = rl64(nd, w); wx = rlf64(lx, w); wy = rlf64(ly, w); }
                        if (!__ballot(hv && lane == QS_WAVE - 1 && nd > fr)) break;  // the round's oldest event is in the index already
                    }
#if defined(QS_FREE_PROF) && QS_FREE_PROF == 6
                    pf_wait += __builtin_amdgcn_s_memtime() - tw_;                   // (profile build 6: the pending scan)
#endif
                    if (!stale) break;                                              // match or none: final (index up to fr, events up to limit)
                    gbest = LL_MAX;                                                 // the index has grown under the scan: once more, from it
                }
#ifdef QS_FREE_PROF
                pf_query += __builtin_amdgcn_s_memtime() - tq_;
#endif
                if (gbest != LL_MAX) {
                    const double ex = wx - qx, ey = wy - qy;                        // :311-312
                    const double cdx = ex * corr, cdy = ey * corr;                  // :314-315
                    const unsigned int pq = s_push[qa];
                    unsigned int cq = LD_RLX(&s_cons[qa]);
                    if (pq - cq >= DY_RD) {
#ifdef QS_FREE_PROF
                        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
                        FR_SPIN(pq - (cq = LD_RLX(&s_cons[qa])) >= DY_RD);
#ifdef QS_FREE_PROF
                        pf_wait += __builtin_amdgcn_s_memtime() - t_;
#endif
                    }
                    CBAR();
                    if (lane == 0) {
                        const unsigned int sl = (unsigned int)qa * DY_RD + (pq & (DY_RD - 1));
                        q_idx[sl] = qidx; q_midx[sl] = gbest; q_cdx[sl] = cdx; q_cdy[sl] = cdy;
                        a_dx[qa] = odx + cdx; a_dy[qa] = ody + cdy; a_last[qa] = qidx;      // :911-914, :318
                    }
                    CBAR();
                    if (lane == 0) ST_RLX(&s_push[qa], pq + 1);
                }
            }
            CBAR();
            if (lane == 0) ST_RLX(&a_node[qa], qidx);
        }
        if (lane == 0) ST_RLX(&s_prog[wave], LL_MAX);
        if (lane == 0) {
            if (st_misc) atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], st_misc);
            if (st_wait) { atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], st_wait); if (pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAIN_MISS - QS_FLAG_PILE, (unsigned int)st_wait); }
#ifdef QS_FREE_PROF
            if (wave == 1) { atomicAdd(&counters[QS_CNT_SLAM_CYC_A], QS_FREE_PROF == 2 ? pf_prev : pf_wait);
                             atomicAdd(&counters[QS_CNT_SLAM_CYC_B], QS_FREE_PROF == 2 ? pf_grab : pf_query);
                             atomicAdd(&counters[QS_CNT_SLAM_NODE_ITERS], __builtin_amdgcn_s_memtime() - pf_total0); }
#endif
        }
    } else if (wave == CH_WAVES - 1) {
        // =================================== the committer ===================================
        const QsGraphDev G = *Gp;
        long long n_lms = G.n_lms, n_misc = G.n_misc, n_cls = G.n_cls;
        unsigned int pool = G.nodes_used;
        bool pile = false;
        unsigned long long st_batches = 0;
        const unsigned long long t0_cyc = __builtin_amdgcn_s_memtime(), t0_real = __builtin_amdgcn_s_memrealtime();
        unsigned int e = e0, idle = 0;
#ifdef QS_FREE_PROF
        unsigned long long pf_idle = 0, pf_agents = 0, pf_insert = 0;
#endif
        long long node = LL_MAX, node_n = LL_MAX;
        int ag = 0, type_l = 0, ag_n = 0, type_n = 0;
        double px_l = 0, py_l = 0, px_n = 0, py_n = 0;
        auto load_events = [&](unsigned int from, long long &nd, int &a_, int &ty, double &x_, double &y_) {
            const unsigned int q = from + lane;
            const bool have = q < e1;
            nd = have ? sb.ev_node[q] : LL_MAX;
            a_ = have ? (int)sb.ev_agent[q] : 0;
            ty = have ? (int)sb.ev_type[q] : 0;
            x_ = have ? sb.ev_px[q] : 0; y_ = have ? sb.ev_py[q] : 0;
        };
        if (e < e1) load_events(e, node, ag, type_l, px_l, py_l);
        while (e < e1) {
            const bool have = e + lane < e1;
            // the batch: the events older than everything an owner is still working on (a PREFIX of the list)
            const long long mp = wave_min_nonneg_i64((lane >= 1 && lane <= DY_OWNERS) ? LD_RLX(&s_prog[lane]) : LL_MAX);
            CBAR();
            const unsigned long long rm = __ballot(have && node < mp);
            const int k = rm == ~0ull ? 64 : (int)__builtin_ctzll(~rm);
            if (k == 0) {
                if (++idle > FR_SPIN_MAX) { if (lane == 0) atomicAdd(&counters[QS_CNT_SLAM_ROUNDS], 1ull << 40); break; }   // (never: see FR_SPIN)
                if ((idle & 63u) == 0 && (__hip_atomic_load(&counters[QS_CNT_SLAM_ROUNDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 40)) break;
#ifdef QS_FREE_PROF
                pf_idle++;
#endif
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            idle = 0;
            if (e + k < e1) load_events(e + k, node_n, ag_n, type_n, px_n, py_n);
#ifdef QS_FREE_PROF
            const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
            const bool inw = lane < k;
            const int type = inw ? type_l : 0;
            const double px = inw ? px_l : 0, py = inw ? py_l : 0;
            // ---- the agents' drifts along the batch, from their decisions: a lane per event; an agent with several events in the
            // batch has them handled one per round, in node order ----
            double dx = 0, dy = 0, cdx = 0, cdy = 0;
            long long midx = LL_MAX;
            bool todo = inw;
            while (__ballot(todo)) {
                if (todo) atomicMin(&c_tag[ag], (unsigned int)lane);
                CBAR();
                if (todo && LD_RLX(&c_tag[ag]) == (unsigned int)lane) {
                    const unsigned int dc = s_cons[ag];
                    const unsigned int dp = LD_RLX(&s_push[ag]);
                    dx = c_ddx[ag]; dy = c_ddy[ag];                                  // matched / stored at the pose BEFORE the closure (:288, :308)
                    CBAR();
                    const unsigned int sl = (unsigned int)ag * DY_RD + (dc & (DY_RD - 1));
                    if (dc != dp && q_idx[sl] == node) {                             // (else the agent's next decision is about a later event)
                        midx = q_midx[sl]; cdx = q_cdx[sl]; cdy = q_cdy[sl];
                        const double ndx = dx + cdx, ndy = dy + cdy;                 // drift_correction[agent] += ...  :911-914
                        const unsigned int apos = c_apos[ag];
                        sb.acl_node[apos] = node; sb.acl_dx[apos] = ndx; sb.acl_dy[apos] = ndy;
                        c_ddx[ag] = ndx; c_ddy[ag] = ndy; c_apos[ag] = apos + 1;
                        CBAR();
                        ST_RLX(&s_cons[ag], dc + 1);
                    }
                    c_tag[ag] = 0xffffffffu;
                    todo = false;
                }
                CBAR();
            }
#ifdef QS_FREE_PROF
            const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
            pf_agents += tp1 - tp0;
#endif
            const double x = raw_pose ? px : px + dx, y = raw_pose ? py : py + dy;   // rx += cdx, ry += cdy  :856-857
            // ---- closure records, in node order  (:317) ----
            const bool closes = inw && midx != LL_MAX;
            const unsigned long long cmask = __ballot(closes);
            if (closes) {
                const long long slot = n_cls + __popcll(cmask & ((1ull << lane) - 1));
                if (slot < G.cap_cls) {
                    G.cl_lm_idx[slot] = midx; G.cl_node_idx[slot] = node; G.cl_dx[slot] = cdx; G.cl_dy[slot] = cdy;
                    G.cl_agent[slot] = (unsigned char)(bot0 + ag);
                }
            }
            n_cls += __popcll(cmask);
            // ---- self.landmarks.append(...)  :288, and the bucket index ----
            int cx, cy;
            const long long kb = (inw && bucket_cell(x, y, type, bg, cx, cy)) ? bucket_key(type, cx, cy, bg) : -1;
            chain_insert_lanes(G, inw, lane, node, kb, x, y, type, k, lane, n_lms, n_misc, pool, pile);
            // ---- everything above complete, then the frontier moves ----
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const long long next_node = e + k < e1 ? rl64(node_n, 0) : LL_MAX;
            e += k;
            if (lane == 0) {
                s_nmisc = n_misc; s_nlms = n_lms;
                lds_st64(&s_frontier, next_node == LL_MAX ? LL_MAX : next_node - 1);
                ST_RLX(&s_comm, e - e0);
            }
            node = node_n; ag = ag_n; type_l = type_n; px_l = px_n; py_l = py_n;
            st_batches++;
#ifdef QS_FREE_PROF
            pf_insert += __builtin_amdgcn_s_memtime() - tp1;
#endif
        }
        if (lane == 0) {
            atomicAdd(&counters[QS_CNT_CLOSURES], (unsigned long long)(n_cls - G.n_cls));
            if (pile_flag) atomicAdd(pile_flag + QS_FLAG_CHAIN_HIT - QS_FLAG_PILE, (unsigned int)(n_cls - G.n_cls));
            atomicAdd(&counters[QS_CNT_LANDMARKS], (unsigned long long)(n_lms - G.n_lms));
            atomicAdd(&counters[QS_CNT_SLAM_WINDOWS], st_batches);
#ifdef QS_FREE_PROF
            atomicAdd(&counters[QS_CNT_SLAM_CYC_C], pf_idle);
            atomicAdd(&counters[QS_CNT_EKF_WRAP_CLAMP], pf_agents); if (QS_FREE_PROF == 1) atomicAdd(&counters[QS_CNT_SLAM_MISC_ITERS], pf_insert);
#endif
            atomicAdd(&counters[QS_CNT_SLAM_CYCLES], __builtin_amdgcn_s_memtime() - t0_cyc);
            atomicAdd(&counters[QS_CNT_SLAM_REALTIME], __builtin_amdgcn_s_memrealtime() - t0_real);
            Gp->n_nodes = G.n_nodes + sb.acc_total[g];
            Gp->n_cls = n_cls; Gp->n_lms = n_lms; Gp->n_misc = n_misc; Gp->nodes_used = pool;
            if (pile && pile_flag) *pile_flag = 1u;
        }
        for (int t = lane; t < nb; t += QS_WAVE) sb.acl_cnt[bot0 + t] = c_apos[t] - sb.agent_ev[bot0 + t];
    }
    // the agents' states go back to where the next ingest finds them
    __syncthreads();
    for (int t = tid; t < nb; t += CH_THREADS) {
        drift[2 * (bot0 + t)] = a_dx[t]; drift[2 * (bot0 + t) + 1] = a_dy[t]; last_closure[bot0 + t] = a_last[t];
    }
#undef LD_RLX
#undef ST_RLX
#undef CBAR
}

// ---- pose: rx, ry of every accepted record (dual_bot_mapper.py:855-857) ---------------------------
// drift of the record's bot = drift after that bot's last closure at a node index < the record's
// (a closure at node j is applied to packets after j, :910-914), else the drift at batch start.
__global__ void __launch_bounds__(256)
qs_slam_pose_kernel(size_t n, QsBatch b, QsSlamBatch sb)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !b.accept[i]) return;
    const int agent = b.agent[i];
    const long long node = sb.node[i];
    const unsigned int base = sb.agent_ev[agent];
    int lo = 0, hi = (int)sb.acl_cnt[agent];       // first closure with acl_node >= node
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sb.acl_node[base + mid] < node) lo = mid + 1; else hi = mid; }
    double dx, dy;
    if (lo == 0) { dx = sb.drift_start[2 * agent]; dy = sb.drift_start[2 * agent + 1]; }
    else { dx = sb.acl_dx[base + lo - 1]; dy = sb.acl_dy[base + lo - 1]; }
    b.rx[i] = b.px[i] + dx;
    b.ry[i] = b.py[i] + dy;
}

// ---- reset: empties what a session used of the bucket index ------------------------------------------
// The first node of every directory entry makes the node array as large as the table (>100 MB per graph
// for a 4096^2 world): a reset clears only the entries the landmark log names -- the log keeps the pose
// each landmark was indexed at -- and the pool nodes that were handed out.
__global__ void __launch_bounds__(256)
qs_slam_reset_index_kernel(const QsGraphDev *__restrict__ graphs, QsBucketGeom bg, unsigned int first_pool)
{
    const QsGraphDev G = graphs[blockIdx.y];
    if (!G.nodes) return;
    const long long n_lms = G.n_lms < G.cap_lms ? G.n_lms : G.cap_lms;
    const long long pool_n = (long long)G.nodes_used - first_pool;
    const long long total = (n_lms + (pool_n > 0 ? pool_n : 0)) * 32;          // 32 lanes per node: 24 words + next + directory entry
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long i = t >> 5;
        const int r = (int)(t & 31);
        long long node, key = -1;
        if (i < n_lms) {
            int cx, cy;
            const int type = G.lm_type[i];
            if (!bucket_cell(G.lm_x[i], G.lm_y[i], type, bg, cx, cy)) continue;     // side list: not in the index
            key = bucket_key(type, cx, cy, bg);
            node = 1 + key;
        } else {
            node = first_pool + (i - n_lms);
        }
        if (r < 24) ((unsigned long long *)(G.nodes + node))[r] = 0x7f7f7f7f7f7f7f7full;
        else if (r == 24) G.nd_next[node] = 0;
        else if (r == 25 && key >= 0) G.dir[key] = QsDirEntry{0, 0, 0, 0};
    }
}

// ... and then the graphs' counters and the bots' last closure (:271), on the device: a reset is all
// enqueued work, no host staging to wait for
__global__ void __launch_bounds__(256)
qs_slam_reset_counters_kernel(QsGraphDev *__restrict__ graphs, int n_graphs, unsigned int first_pool,
                              long long *__restrict__ last_closure, int nb, long long lc_value)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_graphs) {
        QsGraphDev *G = graphs + t;
        G->n_nodes = 0; G->n_lms = 0; G->n_cls = 0; G->n_misc = 0; G->nodes_used = first_pool;
    }
    if (t < nb) last_closure[t] = lc_value;
}

hipError_t qs_launch_slam_reset_index(qs_ctx *c)
{
    if (!c->d_graphs || c->n_graphs <= 0) return hipSuccess;
    const unsigned int first_pool = (unsigned int)(1 + c->dir_entries);
    const int nb = c->cfg.max_agent + 1, m = nb > c->n_graphs ? nb : c->n_graphs;
    // (one graph can hold the whole session's landmarks -- 2 x 10^5 entries of 32 words after one configs[1] step --: enough
    // workgroups to fill the chip whatever the number of graphs; 64 per graph took 0.28 ms there, 2048 take 0.03)
    const int per_graph = c->n_graphs >= 32 ? 64 : 2048 / c->n_graphs;
    hipLaunchKernelGGL(qs_slam_reset_index_kernel, dim3(per_graph, c->n_graphs), dim3(256), 0, c->stream, c->d_graphs, c->bg, first_pool);
    hipLaunchKernelGGL(qs_slam_reset_counters_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, c->d_graphs, c->n_graphs,
                       first_pool, c->d_last_closure, nb, -(long long)c->cfg.min_poses_between);
    return hipGetLastError();
}

hipError_t qs_launch_slam(qs_ctx *c, size_t n, bool raw_pose)
{
    if (n == 0) return hipSuccess;
    QsSlamBatch sb = c->sb;
    sb.n_blocks = qs_slam_blocks(n);
    const int G = c->n_graphs;
    hipLaunchKernelGGL(qs_slam_count_kernel, dim3(sb.n_blocks), dim3(IDX_BLOCK), 0, c->stream, n, c->b, sb,
                       c->bots_per_graph, G);
    hipLaunchKernelGGL(qs_slam_blockscan_kernel, dim3(G), dim3(256), 0, c->stream, sb);
    hipLaunchKernelGGL(qs_slam_prefix_kernel, dim3(1), dim3(256), 0, c->stream, sb, G, c->cfg.max_agent, c->d_drift);
    hipLaunchKernelGGL(qs_slam_index_kernel, dim3(sb.n_blocks), dim3(IDX_BLOCK), (size_t)IDX_WAVES * G * 2 * sizeof(unsigned int),
                       c->stream, n, c->b, sb, c->d_graphs, c->bots_per_graph, G);
    StageTimer t_chain(c, QS_STAGE_SLAM_CHAIN);
#define CH_LAUNCH(ONE_, DENSE_) hipLaunchKernelGGL((qs_slam_chain_kernel<ONE_, DENSE_>), dim3(G), dim3(CH_THREADS), 0, c->stream, c->d_graphs, sb, c->bg, \
                           c->bots_per_graph, c->cfg.max_agent, c->win, c->cfg.min_poses_between, c->r2_threshold,                          \
                           c->cfg.closure_correction, c->d_drift, c->d_last_closure, c->d_counters, raw_pose ? 1 : 0, c->d_flags + QS_FLAG_PILE)
    const bool one = c->bots_per_graph <= CH_AGW;
    // which form (qs_set_chain_form; QS_CHAIN_MODE at qs_create).  Left to itself the library runs the free-running form; for
    // graphs of up to CH_AGW agents without the posting of poses until a batch had more than 1 decision in 8 wait for the
    // committer -- a stream whose queries mostly find nothing in the index --, with it until fewer than 1 in 16 need it, and the
    // per-window kernel while even so there are more scans than closures
    // (chain_stats_poll in qs_api.hip reads the counts, without waiting for anything).  Same results either way.
    const bool free_mode = c->chain_form != QS_CHAIN_WINDOW && !(c->chain_form == QS_CHAIN_AUTO && one && c->chain_windowed);
    c->chain_last_posting = false;
    c->chain_last_free = free_mode;
#define FR_LAUNCH(DENSE_, WAVES_, POST_) hipLaunchKernelGGL((qs_slam_chain_free_kernel<DENSE_, WAVES_, POST_>), dim3(G), dim3(WAVES_ * QS_WAVE), 0, c->stream, c->d_graphs, sb, c->bg, \
                           c->bots_per_graph, c->cfg.max_agent, c->cfg.min_poses_between, c->r2_threshold,                                  \
                           c->cfg.closure_correction, c->d_drift, c->d_last_closure, c->d_counters, raw_pose ? 1 : 0, c->d_flags + QS_FLAG_PILE)
#define DY_LAUNCH(DENSE_) hipLaunchKernelGGL((qs_slam_chain_dyn_kernel<DENSE_>), dim3(G), dim3(CH_THREADS), 0, c->stream, c->d_graphs, sb, c->bg, \
                           c->bots_per_graph, c->cfg.max_agent, c->cfg.min_poses_between, c->r2_threshold,                                  \
                           c->cfg.closure_correction, c->d_drift, c->d_last_closure, c->d_counters, raw_pose ? 1 : 0, c->d_flags + QS_FLAG_PILE)
    if (free_mode) {
        if (one) {
            const bool post = c->chain_form == QS_CHAIN_FREE_POSTING || (c->chain_form == QS_CHAIN_AUTO && c->chain_posting);
            const int sel = (c->pile_mode ? 4 : 0) | (c->bots_per_graph <= 5 ? 2 : 0) | (post ? 1 : 0);
            switch (sel) {
            case 0: FR_LAUNCH(false, 16, false); break;  case 1: FR_LAUNCH(false, 16, true); break;
            case 2: FR_LAUNCH(false, 8, false); break;   case 3: FR_LAUNCH(false, 8, true); break;
            case 4: FR_LAUNCH(true, 16, false); break;   case 5: FR_LAUNCH(true, 16, true); break;
            case 6: FR_LAUNCH(true, 8, false); break;    default: FR_LAUNCH(true, 8, true); break;
            }
            c->chain_last_posting = post;
        }
        else { if (c->pile_mode) DY_LAUNCH(true); else DY_LAUNCH(false); }
    }
#undef DY_LAUNCH
    else if (c->pile_mode) { if (one) CH_LAUNCH(true, true); else CH_LAUNCH(false, true); }
    else { if (one) CH_LAUNCH(true, false); else CH_LAUNCH(false, false); }
#undef FR_LAUNCH
#undef CH_LAUNCH
    t_chain.stop();
    if (raw_pose) return hipGetLastError();
    hipLaunchKernelGGL(qs_slam_pose_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, c->stream, n, c->b, sb);
    return hipGetLastError();
}
