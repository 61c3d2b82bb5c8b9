/*
 * quasar_slam.h -- C ABI of the MI355X-native central-mapper hot path.
 *
 * The reference (deevinandu/Distributed-Multi-Agent-SLAM-Swarm-Robotics-System) has no
 * FFI: its boundary is the Python object API used by server_nodes/dual_bot_mapper.py's
 * main() and MapRenderer.  Every entry point below names the reference interface it
 * replaces (file:line, relative to the reference tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every call returns 0 on success, <0 on error (QS_E_*); qs_last_error() gives text;
 *   - one context = one GPU = one mapper instance; calls on one context are serialised by
 *     the caller (the reference is single-threaded); contexts are independent;
 *   - the caller owns every buffer it passes; the library owns all device state;
 *   - "host" pointers are ordinary memory, "device" pointers are HIP device memory on the
 *     context's GPU; work is enqueued on the context's stream and host-visible results are
 *     complete when the call returns (device variants: after qs_sync()).
 */
#ifndef QUASAR_SLAM_H
#define QUASAR_SLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_OK 0
#define QS_E_INVAL (-1)    /* bad argument */
#define QS_E_HIP (-2)      /* HIP runtime error (text in qs_last_error) */
#define QS_E_NOMEM (-3)
#define QS_E_RANGE (-4)    /* index (bot, graph, capacity) out of range */
#define QS_E_NODEV (-5)    /* no usable GPU */
#define QS_E_STATE (-6)    /* call not valid in the context's current state (text in qs_last_error) */

/* wire formats: dual_bot_mapper.py:41-54 */
#define QS_PACKET_SIZE 42     /* QuasarPacket v2 '<4sBfffiIffffB' */
#define QS_PACKET_SIZE_V1 41  /* QuasarPacket v1 '<4sBfffiIffff'  */
#define QS_ZONE_SIZE 20       /* ZONE '<4sffff' */

typedef struct qs_ctx qs_ctx;

typedef struct qs_config {
    /* OccupancyGrid(size, resolution, origin_x, origin_y)  dual_bot_mapper.py:113-119 */
    int32_t size;
    double res, ox, oy;
    double separation;          /* --separation, added to bot 2's x   :715, :851-852 */
    double min_dist, max_dist;  /* MIN_DIST_M, MAX_DIST_M            :57-58  */
    double closure_radius;      /* CLOSURE_RADIUS                     :97  */
    int32_t min_poses_between;  /* MIN_POSES_BETWEEN                  :98  */
    double closure_correction;  /* CLOSURE_CORRECTION                 :99  */
    int32_t max_agent;          /* accept agent_id 1..max_agent; reference: 2 (:842) */
    int32_t bots_per_graph;     /* bots sharing one PoseGraphSLAM; 0 = all (reference) */
    int32_t enable_counts;      /* keep per-cell hit/miss counters (build extension) */
    int32_t enable_ekf;         /* run the firmware EKF per bot on ingest (ekf.cpp) */
    double ekf_metres_per_tick; /* encoder scale for the EKF wiring (generator: 0.0107) */
    int32_t device;             /* HIP device ordinal */
    int32_t raycast_mode;       /* 0 = auto, 1 = direct global atomics, 2 = LDS tile-binned */
    int32_t seq_stride;         /* arrival index of record i = seq0 + i*seq_stride (0 = 1); a rank of an
                                   N-way round-robin sharded stream uses seq0 = base + rank, stride N */
    int32_t shard_bots;         /* > 0: this context is shard `shard_rank` of a deployment that keeps ONE pose graph over all
                                   bots (PoseGraphSLAM is global across bots, :275, :294-309): every shard ingests the whole
                                   interleaved stream (decode + loop closure replicated, identical on all shards) but casts
                                   rays, keeps zones and runs the EKF only for its own agents
                                   shard_rank*shard_bots+1 .. (shard_rank+1)*shard_bots.  0 = the context owns every agent */
    int32_t shard_rank;
    int32_t exact_trig;         /* 1 (default): rays whose end point falls within 1e-9 cells of a cell boundary -- where a last-bit
                                   difference between the device's sin / cos and glibc's (CPython's math.cos / math.sin) could
                                   change a cell index -- are not cast by the device but wait, self-contained, for the host to
                                   recompute their end points with libm: at the next call that observes the map (grid / counter /
                                   frontier reads, qs_device_buffers, a fuse, qs_counters, qs_sync).  qs_ingest_device itself never
                                   waits for the GPU.  0: device trig only */
    int32_t reserved[3];
} qs_config;

/* reference constants (dual_bot_mapper.py:57-99) */
int qs_config_default(qs_config *cfg);

int qs_create(const qs_config *cfg, qs_ctx **out);
int qs_destroy(qs_ctx *ctx);
const char *qs_last_error(const qs_ctx *ctx);   /* ctx may be NULL: last create error */
/* run on the caller's hipStream_t (e.g. torch's current stream); NULL = own stream */
int qs_set_stream(qs_ctx *ctx, void *hip_stream);
int qs_sync(qs_ctx *ctx);
/* new session: grid -> UNKNOWN, pose graphs, drift, zones, EKF cleared (main() start, :755-785).
 * Enqueued on the context's stream like an ingest; does not wait for the GPU. */
int qs_reset(qs_ctx *ctx);
/* per-bot x offset; bot 2 defaults to cfg.separation (:851-852) */
int qs_set_bot_offset(qs_ctx *ctx, int32_t bot, double off_x);

/* ---- ingest: the batched body of the recv loop, dual_bot_mapper.py:826-919 ------------
 * n datagrams, record i at pkts + i*stride, its length lens[i] (NULL: every length ==
 * stride).  42-byte v2 and 41-byte v1 records are decoded, anything else, a bad magic or
 * an agent outside 1..max_agent is dropped (:826-843).  recv_time (seconds, may be NULL)
 * feeds only the EKF wiring.  seq0 = arrival index of record 0 in the global stream
 * (UINT64_MAX: continue this context's own counter); later records win grid cells. */
int qs_ingest(qs_ctx *ctx, const uint8_t *pkts, size_t n, size_t stride,
              const uint16_t *lens, const double *recv_time, uint64_t seq0);
/* same with device-resident inputs; asynchronous on the context's stream */
int qs_ingest_device(qs_ctx *ctx, const uint8_t *d_pkts, size_t n, size_t stride,
                     const uint16_t *d_lens, const double *d_recv_time, uint64_t seq0);
/* per record of the LAST ingest: accepted flag and pose after offset+drift (:850-857) */
int qs_last_batch(qs_ctx *ctx, uint8_t *accepted, double *pose_xyyaw /* n x 3 */, size_t n);
/* valid hit points of the LAST ingest (point_clouds[agent][sensor].append, :892):
 * 4 slots per record in sensor order front,left,back,right; valid[i*4+s] marks used slots.
 * Computed on request from the batch that is still resident (one extra kernel): the map update itself
 * only needs the rays' grid cells. */
int qs_last_hits(qs_ctx *ctx, double *xy /* n x 4 x 2 */, uint8_t *valid /* n x 4 */, size_t n);

/* ---- OccupancyGrid object API ----------------------------------------------------------
 * batched OccupancyGrid.update_ray(robot_x, robot_y, hit_x, hit_y, hit_valid)  :136-156 */
int qs_update_rays(qs_ctx *ctx, const double *rx, const double *ry, const double *hx,
                   const double *hy, const uint8_t *valid, size_t n, uint64_t seq0);
/* OccupancyGrid.world_to_grid  :121-125 (device evaluation of the same fp64 expression) */
int qs_world_to_grid(qs_ctx *ctx, const double *w, size_t n, int32_t axis, int64_t *out);
/* OccupancyGrid.grid: int8 [size][size], row = gy, values -1/0/100  :92-94, :119 */
int qs_grid_i8(qs_ctx *ctx, int8_t *out_host);
int qs_grid_i8_device(qs_ctx *ctx, int8_t *out_dev);
/* build extension: per-cell counts of OCCUPIED / FREE writes, and a log-odds view
 * clamp(hits*l_occ - misses*l_free, lmin, lmax) */
int qs_grid_counts(qs_ctx *ctx, int32_t *hits_host, int32_t *misses_host);
int qs_grid_logodds(qs_ctx *ctx, float l_occ, float l_free, float lmin, float lmax,
                    float *out_host);
/* raw device state, for collectives (RCCL all-reduce MAX on int32 stamps, SUM on counts) */
int qs_device_buffers(qs_ctx *ctx, void **stamps_dev, size_t *stamps_bytes,
                      void **counts_dev, size_t *counts_bytes);

/* ---- PoseGraphSLAM state  dual_bot_mapper.py:261-338 ---------------------------------- */
int qs_slam_sizes(qs_ctx *ctx, int32_t graph, int64_t *n_nodes, int64_t *n_landmarks,
                  int64_t *n_closures);
/* slam.closures: (lm_idx, node_idx), (corr_dx, corr_dy)  :270, :317 */
int qs_slam_closures(qs_ctx *ctx, int32_t graph, int64_t *idx2, double *corr2, size_t cap);
/* nodes[node_idx].agent_id of every closure's closing node (get_correction_for_agent, :328-338) */
int qs_slam_closure_agents(qs_ctx *ctx, int32_t graph, uint8_t *agents, size_t cap);
/* slam.landmarks: (x, y), (type, node_idx) in insertion order  :269, :288 */
int qs_slam_landmarks(qs_ctx *ctx, int32_t graph, double *xy, int64_t *type_idx, size_t cap);
/* PoseGraphSLAM.add_pose(x, y, yaw, agent_id, landmark_type, timestamp) -> (closure_detected,
 * correction_dx, correction_dy), batched  :273-290.  Object API: the poses are used AS GIVEN -- the
 * caller has already applied its drift correction, as main() does (:855-857) before it calls
 * add_pose (:908).  No rays are cast.  closed / corr2 (n x 2) may be NULL. */
int qs_slam_add_poses(qs_ctx *ctx, const double *x, const double *y, const uint8_t *agent,
                      const uint8_t *landmark, size_t n, uint8_t *closed, double *corr2);
/* drift_correction[bot]  :782, :910-914 */
int qs_drift(qs_ctx *ctx, int32_t bot, double out[2]);

/* ---- ZONE output  dual_bot_mapper.py:675-688, :702-706, :922-945 ----------------------- */
/* bbox over bot's valid hit points U path; *valid = 0 when the bot has no points yet */
int qs_zone(qs_ctx *ctx, int32_t bot, double out[4], int32_t *valid);
/* the 20-byte datagram sent to the OTHER bot; online == 0 lifts the zone (999,999,-999,-999) */
int qs_zone_packet(qs_ctx *ctx, int32_t bot, int32_t online, uint8_t out[QS_ZONE_SIZE]);

/* ---- grid merge ------------------------------------------------------------------------
 * shared-grid semantics of dual_bot_mapper.py:785: dst <- fuse(dst, srcs...) cell-wise
 * (latest stamp wins, counts add).  All contexts on dst's GPU, same geometry. */
int qs_fuse(qs_ctx *dst, qs_ctx *const *srcs, size_t n);
/* same over raw device buffers (e.g. peers' grids gathered by the caller) */
int qs_fuse_buffers(qs_ctx *dst, const void *const *stamps_dev, const void *const *counts_dev,
                    size_t n);
/* fuse of a RANGE of cells: dst cells [cell_offset, cell_offset + n_cells) <- fuse(dst, sources), every source pointer
 * naming the source's first cell of that range (the receive buffers of a reduce-scatter: each rank folds its peers'
 * copies of ITS slice, then the slices are all-gathered).  Either pointer array may be NULL (stamps only / counters
 * only).  counts_into_fused != 0 adds into the snapshot of qs_fused_counts instead of the local counters. */
int qs_fuse_buffers_range(qs_ctx *dst, const void *const *stamps_dev, const void *const *counts_dev,
                          size_t n, size_t cell_offset, size_t n_cells, int32_t counts_into_fused);
/* ---- sharded streams (one context per GPU, the shared grid of dual_bot_mapper.py:785 kept in N pieces) -------------
 * The local counters hold this context's own writes only and are never the target of a collective: qs_fused_counts
 * copies them into a second buffer (on the context's stream) and returns it; the caller sums THAT over the ranks, as
 * often as it likes.  qs_counts_source(ctx, 1) makes qs_grid_counts / qs_grid_logodds read the fused snapshot. */
int qs_fused_counts(qs_ctx *ctx, void **fused_dev, size_t *bytes);
int qs_counts_source(qs_ctx *ctx, int32_t fused);
/* Stamp epochs: ordinals are 30 bits, so a batch that would pass 2^28 arrival indices first rebases the grid (every
 * written cell -> ordinal 1).  With seq_stride > 1 the shards must have exchanged their stamps (MAX all-reduce) since
 * their last write before that happens, or two ranks' writes to one cell would tie: qs_epoch_query tells whether the
 * next ingest (same seq0 / n) would rebase -- the answer is the same on every rank --, qs_mark_fused records that the
 * exchange has happened; an ingest that needs a rebase with unfused writes fails with QS_E_STATE. */
int qs_epoch_query(qs_ctx *ctx, uint64_t seq0, size_t n, int32_t *would_rebase);
int qs_mark_fused(qs_ctx *ctx);
/* ---- sparse fuse: only the blocks a shard has written since its last fuse travel -------------------------------------
 * The shared grid of dual_bot_mapper.py:785 kept in N pieces, fused without moving the whole map: a shard by agent writes
 * a few rooms, not the world.  With tracking on, every writer of the grid (tiled raster merge, direct rays, edge rays,
 * qs_update_rays, qs_fuse*) sets one bit per QS_DIRTY_BLOCK_H x QS_DIRTY_BLOCK_W block of cells it touches.  A fuse is
 *   qs_sparse_fuse_begin   own bitmap -> slot `rank` of a [world][bitmap_bytes] device array (live bitmap cleared);
 *                          the caller all-gathers that array over the ranks (RCCL; the slots are equal-sized);
 *   qs_sparse_fuse_plan    block lists of every rank (ascending block index), this rank's blocks packed -- 64 stamps and,
 *                          with counters, 64 counter DELTAS since this rank's previous sparse fuse -- at offsets[rank] of
 *                          one payload buffer; n_blocks / offsets (bytes, world + 1 entries) tell the caller what to send
 *                          (its own segment, to every peer) and where to receive (peer p's segment at offsets[p]);
 *   qs_sparse_fuse_apply   every received block folded in: stamps MAX into the grid, deltas ADDED to the fused counters
 *                          (which qs_grid_counts / qs_grid_logodds then read); qs_mark_fused implied.
 * Result = the dense fuse (MAX all-reduce of the stamps, SUM of the counters) bit for bit, as long as every rank's grid
 * was equal after the previous fuse (true from qs_reset on).  In this mode the fused counters accumulate deltas: do not
 * mix with qs_fused_counts (refused while tracking is on).  world <= QS_SPARSE_MAX_WORLD. */
#define QS_DIRTY_BLOCK_W 16
#define QS_DIRTY_BLOCK_H 4
#define QS_SPARSE_MAX_WORLD 64
int qs_dirty_tracking(qs_ctx *ctx, int32_t enable);
/* diagnostic: number of blocks marked since the last sparse fuse (waits for the stream) */
int qs_dirty_blocks(qs_ctx *ctx, size_t *n_blocks, size_t *block_cells);
int qs_sparse_fuse_begin(qs_ctx *ctx, int32_t world, int32_t rank, void **bitmaps_dev, size_t *bitmap_bytes);
int qs_sparse_fuse_plan(qs_ctx *ctx, uint32_t *n_blocks /* [world] */, size_t *offsets /* [world + 1] */,
                        void **payload_dev, size_t *block_bytes);
int qs_sparse_fuse_apply(qs_ctx *ctx);
/* The same fuse with its two exchanges on RCCL, for hosts that are not torch (dist.py drives the three steps above through
 * torch.distributed): one process per GPU, one communicator over the node's xGMI links.  qs_rccl_unique_id on rank 0, the 128
 * bytes to every rank by whatever channel the host has, qs_rccl_comm_init on every rank, then qs_sparse_fuse_rccl as often as the
 * map is to be fused: ncclAllGather of the bitmaps, one group of ncclSend / ncclRecv for the packed blocks (point to point, all
 * links at once), the fold.  stats (may be NULL): {blocks packed, bytes packed, bytes sent, bytes received} of this rank.
 * RCCL is loaded on demand (dlopen): QS_E_NODEV if it is absent.  Untested beyond one rank: the build has one GPU. */
#define QS_RCCL_ID_BYTES 128
int qs_rccl_unique_id(uint8_t out[QS_RCCL_ID_BYTES]);
int qs_rccl_comm_init(qs_ctx *ctx, const uint8_t id[QS_RCCL_ID_BYTES], int32_t world, int32_t rank, void **comm);
int qs_rccl_comm_destroy(void *comm);
int qs_sparse_fuse_rccl(qs_ctx *ctx, void *comm, int32_t world, int32_t rank, uint64_t stats[4]);
/* the fused counters as they stand (no snapshot is taken): the sum over the ranks after a fuse; NULL before the first one */
int qs_fused_counts_buffer(qs_ctx *ctx, void **fused_dev, size_t *bytes);
/* MapMerger.grid_to_pcd  server_nodes/map_merger.py:64-85: cells > 50 -> (col*res+ox,
 * row*res+oy) in row-major order.  grid: host int8 [h][w]; returns the count in *n_out. */
int qs_grid_to_pcd(qs_ctx *ctx, const int8_t *grid, int32_t h, int32_t w, double res,
                   double ox, double oy, double *xy, size_t cap, size_t *n_out);
/* MapMerger.publish_global_map  map_merger.py:87-127: points -> int8 canvas; call with
 * grid == NULL to obtain dims {h,w} and origin {min_x,min_y} first */
int qs_rasterise(qs_ctx *ctx, const double *xy, size_t n, double res, int32_t dims[2],
                 double origin[2], int8_t *grid);

/* ---- ICP registration and voxel down-sampling: MapMerger.map_callback, map_merger.py:45-60 ------
 * registration_icp(source, target, max_dist, identity, PointToPoint, max_iteration) on planar
 * clouds (Open3D semantics; parity unpinned: Open3D is not available).  T = 3x3 row-major planar
 * rigid transform source -> target; fitness = #correspondences / n_src; rmse over correspondences;
 * stops early when |d fitness| < rel_fitness and |d rmse| < rel_rmse (Open3D defaults: 1e-6). */
int qs_icp(qs_ctx *ctx, const double *src_xy, size_t n_src, const double *dst_xy, size_t n_dst,
           double max_dist, int32_t max_iter, double rel_fitness, double rel_rmse, double T[9],
           double *fitness, double *rmse, int32_t *iters);
/* The correspondence search of registration_icp on its own (build extension; map_merger.py:48-52 reaches it through
 * Open3D): corr[i] = nearest target of source i if closer than max_dist else -1 (ties: lowest index), d2[i] = its squared
 * distance.  mode 0 = auto, 1 = scalar fp64 brute force, 2 = distance matrix on the matrix cores
 * (v_mfma_f64_16x16x4_f64 as a screen with margin, fp64 re-evaluation of what passes): same results bit for bit.
 * ms (may be NULL): HIP-event milliseconds of {the search kernel, the operand preparation}. */
int qs_nn_search(qs_ctx *ctx, const double *src_xy, size_t n_src, const double *dst_xy, size_t n_dst,
                 double max_dist, int32_t mode, int32_t *corr, double *d2, float ms[2]);
/* Loop-closure chain (PoseGraphSLAM.check_loop_closure, dual_bot_mapper.py:292-326): which device form runs -- same closures,
 * landmarks and drifts whichever.  QS_CHAIN_AUTO (default; the environment's QS_CHAIN_MODE=free|free_posting|window at qs_create
 * overrides): the free-running form; for graphs of up to 13 bots without the owner waves posting their landmarks' poses for each
 * other (QS_CHAIN_FREE) as long as its decisions rarely find nothing in the index, with it (QS_CHAIN_FREE_POSTING) for streams
 * whose queries mostly find nothing -- decided from counts the kernels leave: every ingest has them copied to pinned memory behind
 * itself and the next ingest looks at whatever has arrived, nobody waits; kept over qs_reset: it describes the stream, not the
 * session.  QS_CHAIN_WINDOW: the per-window kernel (one barrier per window of MIN_POSES_BETWEEN nodes): the second opinion of the
 * tests, and QS_CHAIN_AUTO's last resort for a stream that hardly ever matches (more scans of posted poses than closures).
 * qs_chain_form: the form the last ingest used. */
#define QS_CHAIN_AUTO 0
#define QS_CHAIN_FREE 1
#define QS_CHAIN_WINDOW 2
#define QS_CHAIN_FREE_POSTING 3
int qs_set_chain_form(qs_ctx *ctx, int form);
int qs_chain_form(qs_ctx *ctx);

/* diagnostic: measured dense fp64 MFMA rate of this GPU (TFLOP/s), the ceiling qs_nn_search mode 2 is priced against */
int qs_diag_mfma_f64_rate(qs_ctx *ctx, double *tflops);
/* diagnostic: latencies of the primitives one loop-closure decision chains together, measured on this GPU by ONE workgroup
 * (as the loop-closure chain kernels run), in shader-clock cycles:
 *   out[0] dependent global load, L2 hit     out[1] dependent global load, L1 hit     out[2] dependent LDS read
 *   out[3] dependent v_fma_f64               out[4] dependent DPP / VALU step          out[5] v_readlane -> VALU step
 *   out[6] workgroup barrier + LDS fences, 16 waves      out[7] same, 5 waves          out[8] shader clock in MHz */
#define QS_DIAG_LAT_N 9
int qs_diag_latencies(qs_ctx *ctx, double out[QS_DIAG_LAT_N]);
/* PointCloud.voxel_down_sample(voxel): mean of the points of each voxel, ascending voxel order
 * (Open3D's order is unspecified).  out_xy == NULL queries the count. */
int qs_voxel_downsample(qs_ctx *ctx, const double *xy, size_t n, double voxel, double *out_xy,
                        size_t cap, size_t *n_out);

/* ---- frontiers  dual_bot_mapper.py:181-237, :948-956 -------------------------------------------
 * OccupancyGrid.get_frontiers: interior FREE cells with a 4-neighbour UNKNOWN, row-major order
 * (gx, gy pairs).  xy == NULL queries the count. */
int qs_frontier_cells(qs_ctx *ctx, int32_t *xy, size_t cap, size_t *n_out);
/* cluster_frontiers + the sums cluster_centroid_world divides: 4-connected clusters of at least
 * min_cluster cells (reference: FRONTIER_MIN_CLUSTER = 3, :102) in the reference's order (by
 * first cell, row-major); 5 values per cluster: size, first_gx, first_gy, sum_gx, sum_gy.
 * stats5 == NULL queries the count. */
int qs_frontier_clusters(qs_ctx *ctx, int32_t min_cluster, int64_t *stats5, size_t cap, size_t *n_out);

/* cluster membership: every frontier cell (row-major order) with the linear index (gy*size + gx) of the first cell of
 * its 4-connected cluster; 3 values per cell: gx, gy, root.  xy_root == NULL queries the count. */
int qs_frontier_members(qs_ctx *ctx, int32_t *xy_root, size_t cap, size_t *n_out);

/* ---- EKF  AgentFirmware_Bot1/ekf.cpp:5-92 ---------------------------------------------- */
/* On ingest (qs_config.enable_ekf) the filter of every bot runs over the batch: batches of >= 4096
 * packets in a parallel-in-time form that agrees with the step-by-step filter to rounding (~1e-12
 * relative), smaller ones step by step.  QS_CNT_EKF_WRAP_CLAMP counts chunks whose heading-wrap count
 * had to be clamped (a sign of absurd inputs; 0 on every stream seen). */
/* batched over bots: for k in 0..n: predict(omega_m[k], t[k]) then update(z_v[k], z_omega[k])
 * on bot_ids[k] (each bot at most once per call); do_update == 0: predict only */
int qs_ekf_init(qs_ctx *ctx, int32_t bot, double t, const double x0[6]);
int qs_ekf_step(qs_ctx *ctx, const int32_t *bot_ids, const double *omega_m, const double *t,
                const double *z_v, const double *z_omega, size_t n, int32_t do_update);
int qs_ekf_state(qs_ctx *ctx, int32_t bot, double x[6], double P[36]);

/* ---- counters / timing ------------------------------------------------------------------ */
enum { QS_CNT_DATAGRAMS = 0, QS_CNT_ACCEPTED, QS_CNT_RAYS, QS_CNT_CELLS, QS_CNT_HITS,
       QS_CNT_CLOSURES, QS_CNT_LANDMARKS, QS_CNT_REBASES, QS_CNT_SLAM_WINDOWS, QS_CNT_SLAM_ROUNDS,
       QS_CNT_SLAM_NODE_ITERS, QS_CNT_SLAM_MISC_ITERS, QS_CNT_SLAM_CYCLES, QS_CNT_SLAM_REALTIME, QS_CNT_SLAM_CYC_A, QS_CNT_SLAM_CYC_B,
       QS_CNT_SLAM_CYC_C, QS_CNT_EKF_WRAP_CLAMP, QS_CNT_EDGE_RAYS /* exact_trig: rays resolved on the host */,
       QS_CNT_EDGE_OVERFLOW /* exact_trig: rays that found the waiting list full and were cast with the device's trig */, QS_CNT_N };
int qs_counters(qs_ctx *ctx, uint64_t out[QS_CNT_N]);
/* HIP-event timing of the pipeline stages on the context's stream.  enable != 0 brackets
 * each stage of every ingest with events; qs_stage_times returns accumulated ms and the
 * launch count per stage since the last call with reset != 0. */
enum { QS_STAGE_DECODE = 0, QS_STAGE_SLAM, QS_STAGE_RAYCAST, QS_STAGE_EKF,
       /* single kernels inside the stages above (their time is part of the stage's too) */
       QS_STAGE_SLAM_CHAIN,                     /* the loop-closure chain kernel alone (inside SLAM) */
       QS_STAGE_RC_RAYS, QS_STAGE_RC_SORT, QS_STAGE_RC_RASTER,   /* tiled raycast: pass A; passes B + C; pass D */
       QS_STAGE_N };
int qs_timing_enable(qs_ctx *ctx, int32_t enable);
int qs_stage_times(qs_ctx *ctx, double ms[QS_STAGE_N], uint64_t launches[QS_STAGE_N],
                   int32_t reset);

const char *qs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QUASAR_SLAM_H */
