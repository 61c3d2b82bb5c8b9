// qs_internal.h -- shared declarations of the HIP implementation behind include/quasar_slam.h
// gfx950 only.  All device arithmetic that decides a cell index or a loop closure is fp64
// with -ffp-contract=off (the reference is CPython double arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/quasar_slam.h"

#define QS_WAVE 64
#define QS_MAX_AGENT 255          // agent_id is one byte on the wire (dual_bot_mapper.py:41)
#define QS_WIN_MAX 32             // SLAM window: min(min_poses_between, 32) consecutive nodes

// ---- monotone double <-> uint64 map for atomic min/max of zone boxes --------------------
__host__ __device__ inline unsigned long long qs_ord_from_double(double d)
{
    unsigned long long b;
    __builtin_memcpy(&b, &d, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double qs_double_from_ord(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
}
#define QS_ORD_MIN_IDENT 0xffffffffffffffffull   // identity for atomicMin
#define QS_ORD_MAX_IDENT 0ull                    // identity for atomicMax

struct QsNcclId { char internal[QS_RCCL_ID_BYTES]; };     // ncclUniqueId (rccl.h), passed by value to ncclCommInitRank

// ---- per pose-graph device state (PoseGraphSLAM, dual_bot_mapper.py:261-271) -------------
// The landmark list is kept twice: as the reference's insertion-ordered log (read-back, and
// the fallback scan), and as a spatial index: a directory (hash table over the bucket cells) of buckets
// of edge >= CLOSURE_RADIUS per landmark type, each bucket a chain of 7-entry nodes in insertion order.  A query looks at
// the 3x3 buckets around it; the first match in list order is the lowest node index among them.
#define QS_NTYPES 5               // landmark types 1..5 are indexed (LM_CORNER_L..LM_OPEN, :68-74)
#define QS_NODE_CAP 7
struct alignas(64) QsLmNode { long long idx[8]; double x[8]; double y[8]; };   // idx 0x7f7f.. = empty slot
struct QsDirEntry { unsigned int head, tail, tail_cnt, pad; };                  // head 0 = empty bucket (insert-side bookkeeping)
struct QsBucketGeom { double bx0, by0, cell, inv_cell; unsigned int hmask, pad; };   // hmask + 1 = table entries per type

struct QsGraphDev {
    long long n_nodes;     // len(self.nodes)
    long long n_lms;       // len(self.landmarks)
    long long n_cls;       // len(self.closures)
    long long cap_lms, cap_cls;
    double *lm_x, *lm_y;   // landmark position (pose of the storing packet, pre-closure)
    long long *lm_idx;     // node index of the storing packet (ascending)
    unsigned char *lm_type;
    long long *cl_lm_idx, *cl_node_idx;
    double *cl_dx, *cl_dy;
    unsigned char *cl_agent;   // agent_id of the closing node (nodes[node_idx].agent_id, :335)
    QsDirEntry *dir;       // [QS_NTYPES][hmask + 1]
    QsLmNode *nodes;       // [node_cap]: node 0 is the null node, node 1 + t the FIRST node of directory entry t (a query
                           // goes straight to it: no directory round trip), the pool of overflow nodes after those
    unsigned int *nd_next; // [node_cap]
    unsigned int *misc;    // [cap_lms] log slots of landmarks the directory does not cover
    long long n_misc;
    unsigned int nodes_used, pad0;   // next free pool node
    long long node_cap;    // 1 + directory entries + cap_lms
};

// ---- per-batch scratch of the SLAM stage ---------------------------------------------------
struct QsSlamBatch {
    long long *node;                       // [n] node index of each record (-1: rejected)
    long long *ev_node;                    // [n] landmark events, grouped by graph, in node order
    unsigned char *ev_agent, *ev_type;     //     agent index local to the graph, landmark type
    double *ev_px, *ev_py;                 //     pose before drift
    unsigned int *ev_base;                 // [n_graphs + 1] event range of each graph
    unsigned int *acc_total;               // [n_graphs] accepted records of each graph
    unsigned int *blk_acc, *blk_ev;        // [n_graphs][n_blocks] per-block counts -> exclusive offsets
    unsigned int *agent_ev;                // [max_agent + 2] events per bot -> exclusive prefix
    long long *acl_node;                   // [n] per-bot closure regions: node index of the closure,
    double *acl_dx, *acl_dy;               //     drift of the bot AFTER it
    unsigned int *acl_cnt;                 // [max_agent + 1] closures per bot in this batch
    double *drift_start;                   // [(max_agent + 1) * 2] drift at batch start
    int n_blocks;
};

// ---- geometry / constants passed by value to kernels ------------------------------------
struct QsGeom {
    int size;
    double res, ox, oy;
    double min_dist, max_dist;
    double inv_res;          // 1.0 / res: screens world_to_grid quotients, never decides one (raycast_common.h)
    unsigned int *dirty;     // sparse fuse (sparse_fuse.hip): one bit per QS_DIRTY_BLOCK_H x QS_DIRTY_BLOCK_W block of cells written
    int dirty_pitch;         //   since the last fuse, rows of dirty_pitch 32-bit words; nullptr = no tracking
};

// bit of the block that holds cell (x, y): word index and mask
__host__ __device__ inline size_t qs_dirty_word(int x, int y, int pitch) { return (size_t)(y / QS_DIRTY_BLOCK_H) * pitch + (x / QS_DIRTY_BLOCK_W) / 32; }
__host__ __device__ inline unsigned int qs_dirty_mask(int x) { return 1u << ((x / QS_DIRTY_BLOCK_W) & 31); }

// a ray left to the host (exact-trig mode): everything needed to cast it later, whatever has happened to its batch since
// device words read by the host at synchronisation points (qs_ctx::d_flags; qs_reset clears the first four)
enum { QS_FLAG_EDGE_N = 0,        // exact-trig mode: edge rays waiting for the host
       QS_FLAG_PILE = 1,          // a landmark pile has formed (slam.hip, DENSE)
       QS_FLAG_EDGE_OVF = 2,      // edge rays that found the waiting list full
       // loop-closure chain, running totals (never cleared; the host looks at differences -- qs_api.hip, chain_stats_poll):
       QS_FLAG_CHAIN_MISS = 4,    // free-running form: decisions that waited for the frontier ...
       QS_FLAG_CHAIN_HIT = 5,     // ... and its closures
       QS_FLAG_CHAINW_MISS = 6,   // per-window form: eligible queries that found nothing ...
       QS_FLAG_CHAINW_HIT = 7,    // ... and its closures
       QS_N_FLAGS = 8 };
struct QsEdgeRec { double rx, ry, yaw; float d; unsigned int key_free; };     // key_free: stamp of its free cells ((ordinal << 1) | 0)
#define QS_EDGE_CAP (1u << 18)

// ---- decoded batch (SoA, one slot per datagram of the batch) -----------------------------
struct QsBatch {
    size_t n;
    unsigned char *accept;   // 1 = passes dual_bot_mapper.py:826-843
    unsigned char *map_ok;   // 1 = accepted AND this context casts its rays / runs its filter: the same array as accept
                             // unless the context is one shard of a replicated-pose-graph deployment (qs_config.shard_bots)
    int own_lo, own_hi;      // agents whose rays this context casts (1..max_agent when not sharded)
    QsEdgeRec *edge;         // exact-trig mode: rays whose end point lies within 1e-9 cells of a cell boundary are not cast by the
    unsigned int *edge_n;    //   device but appended here, self-contained (pose, distance, stamp): the host resolves them at the next
    unsigned int edge_cap;   //   point the map is observed (qs_api.hip: flush_edge_rays); edge_n[0] = records so far, [2] = rays that
                             //   found the list full and were cast with the device's trig after all
    unsigned char *agent;    // agent_id
    unsigned char *lm;       // landmark_type (0 for v1 packets)
    double *px, *py, *yaw;   // f32 fields widened; px already has the bot offset (:851-852)
    float4 *dist;            // front, left, back, right (metres)
    int *enc;                // encoder ticks
    double *rx, *ry;         // pose after drift correction (:855-857), written by the SLAM stage
    double2 *hit;            // 4 per datagram: ray end points, filled on request (qs_launch_hits)
    unsigned char *hit_valid; // MIN < d <= MAX per ray (:888); the tiled raycast writes it on the ingest path
};

struct qs_ctx {
    qs_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool ekf_stream_shared = false;              // the device's one CU-masked stream (qs_api.hip): not this context's to destroy
    hipStream_t ekf_stream = nullptr;            // the EKF is independent of the map: own stream,
    hipEvent_t ev_decoded = nullptr, ev_ekf_done = nullptr;   // forked after decode, joined at the end
    std::string err;
    size_t cells = 0;
    int n_graphs = 0, bots_per_graph = 0, win = 30;
    double r2_threshold = 0.0;   // s < r2_threshold  <=>  sqrt(s) < closure_radius
    QsGeom geom;

    unsigned int *d_stamps = nullptr;            // [size][size]; 0 = UNKNOWN, else (ordinal<<1)|occ
    unsigned long long *d_counts = nullptr;      // [size][size]; hi32 = hits, lo32 = misses (this context's own writes)
    unsigned long long *d_counts_fused = nullptr; // [size][size]; snapshot of d_counts that a collective sums over the ranks
    bool counts_view_fused = false;              // qs_grid_counts / qs_grid_logodds read the fused snapshot
    bool dirty_since_fuse = false;               // cells written since the last qs_mark_fused (sharded streams: rebase guard)
    // sparse fuse (sparse_fuse.hip)
    unsigned int *d_dirty = nullptr;             // live bitmap [blocks_y][dirty_pitch]
    size_t dirty_words = 0;
    int blocks_x = 0, blocks_y = 0;
    unsigned long long *d_counts_sent = nullptr; // [size][size]: this context's counters as of its last sparse fuse (deltas travel)
    int sf_world = 0, sf_rank = 0;
    unsigned int *d_sf_bitmaps = nullptr;        // [sf_world][dirty_words]: every rank's bitmap of the fuse in flight
    unsigned int *d_sf_lists = nullptr;          // [sf_world][dirty_words * 32] block ids, ascending
    unsigned int *d_sf_counts = nullptr;         // [sf_world] blocks per rank
    unsigned char *d_sf_payload = nullptr; size_t sf_payload_bytes = 0;
    std::vector<unsigned int> sf_n;              // host copy of d_sf_counts
    std::vector<size_t> sf_off;                  // [sf_world + 1] byte offsets of the ranks' segments in the payload
    int sf_state = 0;                            // 0 idle, 1 begun, 2 planned
    double *d_offset = nullptr;                  // [max_agent+1]
    double *d_drift = nullptr;                   // [max_agent+1][2]
    long long *d_last_closure = nullptr;         // [max_agent+1]
    unsigned long long *d_zone = nullptr;        // [max_agent+1][4] ordered-u64 minx,miny,maxx,maxy
    unsigned long long *d_counters = nullptr;    // [QS_CNT_N]
    unsigned long long *d_graph_batch = nullptr; // [n_graphs][2]: accepted, landmark events of the batch
    double *d_ekf = nullptr;                     // [max_agent+1][44]
    double *d_ekf_prev = nullptr;                // [max_agent+1][4]

    QsGraphDev *d_graphs = nullptr;
    std::vector<QsGraphDev> h_graphs;            // host mirror of pointers/capacities
    std::vector<long long> lms_upper, cls_upper; // host upper bounds on n_lms / n_cls

    // batch staging
    size_t cap_batch = 0;
    unsigned char *d_pkts = nullptr; size_t cap_pkts_bytes = 0;
    unsigned short *d_lens = nullptr;
    double *d_time = nullptr;
    QsBatch b{};
    QsSlamBatch sb{};
    QsBucketGeom bg{};
    size_t dir_entries = 0;      // QS_NTYPES * (hmask + 1)
    size_t last_n = 0;
    bool last_has_poses = false;

    // tile-binned raycast workspace
    void *d_bin_ws = nullptr; size_t bin_ws_bytes = 0;
    void *d_frontier_ws = nullptr;               // frontier labelling workspace (allocated on first use)
    void *d_io_ws = nullptr; size_t io_ws_bytes = 0;     // staging of the object-API calls (qs_update_rays, views), grown on demand
    void *d_ekf_ws = nullptr; size_t ekf_ws_bytes = 0;   // parallel-in-time EKF workspace (ekf_scan.hip)

    uint64_t next_seq = 0, epoch_base = 0, n_rebases = 0;
    unsigned int *d_flags = nullptr;             // [QS_N_FLAGS] device words the host reads at synchronisation points (QS_FLAG_*)
    QsEdgeRec *d_edge = nullptr;                 // [QS_EDGE_CAP] the waiting rays
    bool edge_maybe = false;                     // an exact-trig ingest has run since the last flush: the list may hold rays
    bool pile_mode = false;                      // launch the chain kernel's DENSE variant
    bool flags_maybe = false;                    // a loop-closure chain has run since the flags were read last
    int chain_form = 0;                          // QS_CHAIN_AUTO / _FREE / _WINDOW (qs_set_chain_form)
    bool chain_posting = false;                  // QS_CHAIN_AUTO's present choice for graphs of few agents: the free-running kernel WITH the
                                                 // owners posting their landmarks' poses (slam.hip, qs_launch_slam)
    bool chain_windowed = false;                 // ... and beyond that: the per-window kernel (even with posted poses most queries find nothing)
    bool chain_last_free = true, chain_last_posting = false;   // the form the last launch used
    unsigned int *h_chain_stat = nullptr;        // pinned: [0..3] the four running totals as copied last, [4..7] as consumed last
    hipEvent_t ev_chain_stat = nullptr;          // ... complete when the copy has landed
    bool chain_stat_pending = false;
    unsigned int chain_stat_tick = 0;            // ingests since qs_create (the copy is asked for by one in four)
    uint64_t edge_rays_total = 0;                // exact-trig mode: rays resolved on the host since the last reset
    uint64_t edge_overflow_total = 0;            //   ... and rays that found the list full (cast with the device's trig)

    // timing
    bool timing = false;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    double stage_ms[QS_STAGE_N]{};
    uint64_t stage_launches[QS_STAGE_N]{};
};

// ---- HIP-event timing of stages and of single kernels, on the stream they are launched on -------
static inline hipEvent_t qs_ev_get(qs_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
struct StageTimer {
    qs_ctx *c; int stage; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    StageTimer(qs_ctx *c_, int s, hipStream_t st_ = nullptr) : c(c_), stage(s), st(st_ ? st_ : c_->stream)
    { if (c->timing) { a = qs_ev_get(c); b = qs_ev_get(c); hipEventRecord(a, st); } }
    void stop() { if (c->timing && a) { hipEventRecord(b, st); c->pending.push_back({stage, a, b}); a = nullptr; } }
};

// ---- kernel launchers (each defined next to its kernel) ----------------------------------
// decode.hip
hipError_t qs_launch_decode(qs_ctx *c, const unsigned char *d_pkts, size_t n, size_t stride,
                            const unsigned short *d_lens);
#define QS_SLAM_IDX_BLOCK 1024   // records per block of the SLAM index tables
// slam.hip
hipError_t qs_launch_slam(qs_ctx *c, size_t n, bool raw_pose = false);
hipError_t qs_launch_slam_reset_index(qs_ctx *c);              // empties the bucket index of every graph (what was used of it)
int qs_slam_blocks(size_t n);
// raycast.hip
#define QS_DIRECT_MAX_BATCH 256   // raycast_mode auto: batches up to this size take the direct kernel
hipError_t qs_launch_raycast_direct(qs_ctx *c, size_t n, uint64_t seq0);
hipError_t qs_launch_edge_cast(qs_ctx *c, unsigned int n_edge, const double *d_in);
hipError_t qs_launch_hits(qs_ctx *c, size_t n);                 // ray end points of the resident batch (qs_last_hits)
hipError_t qs_launch_update_rays(qs_ctx *c, const double *rx, const double *ry, const double *hx,
                                 const double *hy, const unsigned char *valid, size_t n,
                                 uint64_t seq0);
hipError_t qs_launch_world_to_grid(qs_ctx *c, const double *w, size_t n, int axis, long long *out);
// raycast_tiled.hip
hipError_t qs_launch_raycast_tiled(qs_ctx *c, size_t n, uint64_t seq0);
size_t qs_tiled_workspace_bytes(const qs_ctx *c, size_t n);
bool qs_tiled_supported(const qs_ctx *c);
// grid_ops.hip
hipError_t qs_launch_view_i8(qs_ctx *c, signed char *out_dev);
hipError_t qs_launch_logodds(qs_ctx *c, float l_occ, float l_free, float lmin, float lmax, float *out_dev);
hipError_t qs_launch_split_counts(qs_ctx *c, int *hits_dev, int *misses_dev);
hipError_t qs_launch_rebase(qs_ctx *c);
hipError_t qs_launch_fuse(qs_ctx *c, const unsigned int *const *d_src_stamps,
                          const unsigned long long *const *d_src_counts, size_t n_src, size_t cell_off, size_t n_cells,
                          unsigned long long *dst_counts);
hipError_t qs_launch_fill_zone_identity(qs_ctx *c);
hipError_t qs_launch_reset_small(qs_ctx *c);
hipError_t qs_launch_grid_to_pcd(qs_ctx *c, const signed char *d_grid, int h, int w, double res,
                                 double ox, double oy, double *d_xy, size_t cap,
                                 unsigned long long *d_count, unsigned int *d_rowcount);
hipError_t qs_launch_rasterise(qs_ctx *c, const double *d_xy, size_t n, double res, double minx,
                               double miny, int h, int w, signed char *d_grid);
hipError_t qs_launch_bbox(qs_ctx *c, const double *d_xy, size_t n, unsigned long long *d_box4);
// sparse_fuse.hip
size_t qs_sf_block_bytes(const qs_ctx *c);
hipError_t qs_launch_sf_mark_range(qs_ctx *c, size_t cell_off, size_t n_cells);
hipError_t qs_launch_sf_lists(qs_ctx *c);
hipError_t qs_launch_sf_pack(qs_ctx *c, unsigned int n_own, unsigned char *dst);
hipError_t qs_launch_sf_apply(qs_ctx *c);
hipError_t qs_launch_sf_popcount(qs_ctx *c, unsigned long long *d_out);
// diag.hip
hipError_t qs_launch_diag_latencies(qs_ctx *c, const unsigned int *d_chase_l2, const unsigned int *d_chase_l1, double *d_out);
// frontier.hip
size_t qs_frontier_workspace_bytes(const qs_ctx *c);
hipError_t qs_launch_frontier_label(qs_ctx *c, void *ws, bool with_clusters);
hipError_t qs_launch_frontier_compact(qs_ctx *c, void *ws, int mode, int phase, int *d_xy, long long *d_stats, size_t cap);
unsigned long long *qs_frontier_total_ptr(const qs_ctx *c, void *ws);
// icp.hip
hipError_t qs_launch_icp_nn(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, size_t n_dst,
                            double max_d2, int *corr, double *d2);
hipError_t qs_launch_mfma_f64_rate(qs_ctx *c, int blocks, int iters, double *sink);
hipError_t qs_launch_icp_prep(qs_ctx *c, const double2 *dst, size_t n_dst, size_t n_pad, double cx, double cy, double *planes);
void qs_icp_nn_plan(size_t n_src, size_t n_pad, unsigned int *n_groups, unsigned int *n_parts, unsigned int *chunks_per_part);
hipError_t qs_launch_icp_nn_mfma(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, size_t n_dst,
                                 const double *planes, size_t n_pad, double cx, double cy, double t2max, double max_d2,
                                 int *corr, double *d2, int *part_j, double *part_d2, double *thr_seed);
hipError_t qs_launch_icp_sums(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, const int *corr,
                              const double *d2, int pass, const double means[4], double *partial, double *out6);
hipError_t qs_launch_icp_transform(qs_ctx *c, double2 *pts, size_t n, double cs, double sn, double tx, double ty);
hipError_t qs_launch_voxel_keys(qs_ctx *c, const double2 *pts, size_t n, double minx, double miny, double voxel,
                                unsigned long long *keys);
// ekf.hip
hipError_t qs_launch_ekf_ingest(qs_ctx *c, size_t n, const double *d_time, hipStream_t st);
// ekf_scan.hip: the same filter over a large batch, parallel in time
#define QS_EKF_SCAN_MIN_BATCH 4096
hipError_t qs_launch_ekf_scan(qs_ctx *c, size_t n, const double *d_time, hipStream_t st);
hipError_t qs_launch_ekf_step(qs_ctx *c, const int *d_bots, const double *d_omega, const double *d_t,
                              const double *d_zv, const double *d_zo, size_t n, int do_update);
