// qs_api.hip -- the C ABI of include/quasar_slam.h: context, device memory, and the per-batch
// pipeline  decode (K0) -> SLAM drift (K4) -> raycast (K1) [-> EKF (K5)]  on one HIP stream.
#include <math.h>
#include <algorithm>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qs_internal.h"

static thread_local std::string g_create_err;
static hipStream_t g_masked[64] = {};       // per device: the CU-masked stream its contexts' filters run on (ingest_device)
static int g_masked_users[64] = {};
static std::mutex g_masked_mutex;           // (contexts are independent: two threads may create / destroy theirs at the same time)
static void chain_stats_poll(qs_ctx *c, bool synced, const unsigned int *fresh);
static int flush_edge_rays(qs_ctx *c);      // exact-trig mode: rays waiting for libm end points (defined with the ingest path)
#define FLUSHCHK(c) do { int rcf__ = flush_edge_rays(c); if (rcf__ != QS_OK) return rcf__; } while (0)

static int qs_fail(qs_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(c, x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return qs_fail((c), QS_E_HIP, #x, e__); } while (0)
#define ARGCHK(c, cond) do { if (!(cond)) return qs_fail((c), QS_E_INVAL, "invalid argument: " #cond); } while (0)

template <typename T>
static hipError_t dev_realloc(T **p, size_t count)
{
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
    if (count == 0) return hipSuccess;
    return hipMalloc((void **)p, count * sizeof(T));
}

extern "C" const char *qs_version(void) { return "quasar-slam-amd 0.1 (gfx950)"; }

extern "C" int qs_config_default(qs_config *cfg)
{
    if (!cfg) return QS_E_INVAL;
    memset(cfg, 0, sizeof *cfg);
    cfg->size = 200; cfg->res = 0.05; cfg->ox = -5.0; cfg->oy = -5.0;      // dual_bot_mapper.py:87-90
    cfg->separation = 0.0;                                                  // :715
    cfg->min_dist = 0.05; cfg->max_dist = 1.20;                             // :57-58
    cfg->closure_radius = 0.60; cfg->min_poses_between = 30; cfg->closure_correction = 0.5;  // :97-99
    cfg->max_agent = 2;                                                     // :842
    cfg->bots_per_graph = 0;
    cfg->enable_counts = 1;
    cfg->enable_ekf = 0;
    cfg->ekf_metres_per_tick = 0.0107;        // simulation_tools/generate_fake_dual_session.py:462
    cfg->device = 0;
    cfg->raycast_mode = 0;
    cfg->exact_trig = 1;
    return QS_OK;
}

// smallest double T with sqrt(T) >= radius: (s < T) <=> (sqrt(s) < radius) for correctly rounded sqrt
static double r2_threshold_for(double radius)
{
    if (!(radius > 0)) return 0.0;
    double t = radius * radius;
    while (sqrt(t) >= radius) t = nextafter(t, 0.0);
    while (sqrt(t) < radius) t = nextafter(t, INFINITY);
    return t;
}

static void graph_free(QsGraphDev &g)
{
    hipFree(g.lm_x); hipFree(g.lm_y); hipFree(g.lm_idx); hipFree(g.lm_type);
    hipFree(g.cl_lm_idx); hipFree(g.cl_node_idx); hipFree(g.cl_dx); hipFree(g.cl_dy); hipFree(g.cl_agent);
    hipFree(g.dir); hipFree(g.nodes); hipFree(g.nd_next); hipFree(g.misc);
    memset(&g, 0, sizeof g);
}

template <typename T>
static hipError_t grow_array(T **p, long long old_n, long long new_cap, hipStream_t st)
{
    T *q = nullptr;
    hipError_t e = hipMalloc((void **)&q, (size_t)new_cap * sizeof(T));
    if (e != hipSuccess) return e;
    if (*p && old_n > 0) {
        e = hipMemcpyAsync(q, *p, (size_t)old_n * sizeof(T), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
    }
    if (*p) hipFree(*p);
    *p = q;
    return hipSuccess;
}

// nodes (first node of every directory entry + the overflow pool): "empty" (idx bytes 0x7f -> a huge node
// index) and unlinked (next 0)
static hipError_t grow_pool(qs_ctx *c, QsGraphDev &G, long long old_cap, long long new_cap)
{
    const size_t fixed = 1 + c->dir_entries, n_new = fixed + (size_t)new_cap, n_old = fixed + (size_t)old_cap;
    QsLmNode *nodes = nullptr; unsigned int *next = nullptr, *misc = nullptr;
    hipError_t e = hipMalloc((void **)&nodes, n_new * sizeof(QsLmNode));
    if (e == hipSuccess) e = hipMalloc((void **)&next, n_new * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void **)&misc, (size_t)new_cap * sizeof(unsigned int));
    if (e == hipSuccess && G.nodes) {
        e = hipMemsetAsync(nodes + n_old, 0x7f, (n_new - n_old) * sizeof(QsLmNode), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(next + n_old, 0, (n_new - n_old) * sizeof(unsigned int), c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nodes, G.nodes, n_old * sizeof(QsLmNode), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(next, G.nd_next, n_old * sizeof(unsigned int), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess && old_cap > 0) e = hipMemcpyAsync(misc, G.misc, (size_t)old_cap * sizeof(unsigned int), hipMemcpyDeviceToDevice, c->stream);
    } else if (e == hipSuccess) {
        e = hipMemsetAsync(nodes, 0x7f, n_new * sizeof(QsLmNode), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(next, 0, n_new * sizeof(unsigned int), c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { hipFree(nodes); hipFree(next); hipFree(misc); return e; }
    hipFree(G.nodes); hipFree(G.nd_next); hipFree(G.misc);
    G.nodes = nodes; G.nd_next = next; G.misc = misc;
    G.node_cap = (long long)n_new;
    return hipSuccess;
}

static int graph_reserve(qs_ctx *c, int g, long long need_lms, long long need_cls, long long have_lms,
                         long long have_cls)
{
    QsGraphDev &G = c->h_graphs[g];
    bool changed = false;
    if (!G.dir) {
        HIPCHK(c, hipMalloc((void **)&G.dir, c->dir_entries * sizeof(QsDirEntry)));
        HIPCHK(c, hipMemsetAsync(G.dir, 0, c->dir_entries * sizeof(QsDirEntry), c->stream));
        changed = true;
    }
    if (need_lms > G.cap_lms) {
        long long cap = G.cap_lms ? G.cap_lms : 1024;
        while (cap < need_lms) cap *= 2;
        HIPCHK(c, grow_array(&G.lm_x, have_lms, cap, c->stream));
        HIPCHK(c, grow_array(&G.lm_y, have_lms, cap, c->stream));
        HIPCHK(c, grow_array(&G.lm_idx, have_lms, cap, c->stream));
        HIPCHK(c, grow_array(&G.lm_type, have_lms, cap, c->stream));
        HIPCHK(c, grow_pool(c, G, G.cap_lms, cap));
        G.cap_lms = cap; changed = true;
    }
    if (need_cls > G.cap_cls) {
        long long cap = G.cap_cls ? G.cap_cls : 256;
        while (cap < need_cls) cap *= 2;
        HIPCHK(c, grow_array(&G.cl_lm_idx, have_cls, cap, c->stream));
        HIPCHK(c, grow_array(&G.cl_node_idx, have_cls, cap, c->stream));
        HIPCHK(c, grow_array(&G.cl_dx, have_cls, cap, c->stream));
        HIPCHK(c, grow_array(&G.cl_dy, have_cls, cap, c->stream));
        HIPCHK(c, grow_array(&G.cl_agent, have_cls, cap, c->stream));
        G.cap_cls = cap; changed = true;
    }
    if (changed) {
        // pointers and capacities change; the counters live on the device and are preserved
        QsGraphDev cur;
        HIPCHK(c, hipMemcpyAsync(&cur, c->d_graphs + g, sizeof cur, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        QsGraphDev upd = G;
        upd.n_nodes = cur.n_nodes; upd.n_lms = cur.n_lms; upd.n_cls = cur.n_cls;
        upd.n_misc = cur.n_misc; upd.nodes_used = cur.nodes_used ? cur.nodes_used : (unsigned int)(1 + c->dir_entries);
        HIPCHK(c, hipMemcpyAsync(c->d_graphs + g, &upd, sizeof upd, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return QS_OK;
}

static int reset_state(qs_ctx *c)
{
    HIPCHK(c, hipMemsetAsync(c->d_stamps, 0, c->cells * sizeof(unsigned int), c->stream));
    if (c->d_counts) HIPCHK(c, hipMemsetAsync(c->d_counts, 0, c->cells * sizeof(unsigned long long), c->stream));
    if (c->d_counts_fused) HIPCHK(c, hipMemsetAsync(c->d_counts_fused, 0, c->cells * sizeof(unsigned long long), c->stream));
    c->dirty_since_fuse = false;
    if (c->d_dirty) HIPCHK(c, hipMemsetAsync(c->d_dirty, 0, c->dirty_words * sizeof(unsigned int), c->stream));
    if (c->d_counts_sent) HIPCHK(c, hipMemsetAsync(c->d_counts_sent, 0, c->cells * sizeof(unsigned long long), c->stream));
    c->sf_state = 0;
    HIPCHK(c, qs_launch_reset_small(c));      // drift, zone boxes, counters, per-graph batch counts, EKF state, flags: one launch
    // the bucket index of every graph: only what the session used of it (directory entries, first nodes,
    // pool nodes), found from the landmark log on the device; then the graphs' counters and the bots' last
    // closure (:271).  All enqueued: a reset does not wait for the GPU.
    HIPCHK(c, qs_launch_slam_reset_index(c));
    for (int g = 0; g < c->n_graphs; g++) { c->lms_upper[g] = 0; c->cls_upper[g] = 0; }
    c->next_seq = 0; c->epoch_base = 0; c->last_n = 0; c->last_has_poses = false; c->n_rebases = 0; c->edge_rays_total = 0;
    c->pile_mode = false;
    c->edge_maybe = false; c->edge_overflow_total = 0;      // (rays still waiting belonged to the old session: the flags are cleared above)
    return QS_OK;
}

extern "C" int qs_create(const qs_config *cfg, qs_ctx **out)
{
    if (!cfg || !out) return qs_fail(nullptr, QS_E_INVAL, "qs_create: null argument");
    *out = nullptr;
    if (cfg->size < 4 || cfg->size % 4 != 0 || cfg->size > 32768)
        return qs_fail(nullptr, QS_E_INVAL, "qs_create: size must be a multiple of 4 in [4, 32768]");
    if (!(cfg->res > 0) || !isfinite(cfg->ox) || !isfinite(cfg->oy))
        return qs_fail(nullptr, QS_E_INVAL, "qs_create: bad resolution/origin");
    if (cfg->max_agent < 1 || cfg->max_agent > QS_MAX_AGENT)
        return qs_fail(nullptr, QS_E_INVAL, "qs_create: max_agent must be in [1, 255]");
    if (cfg->shard_bots < 0 || cfg->shard_rank < 0 || (cfg->shard_bots > 0 && (int64_t)cfg->shard_rank * cfg->shard_bots >= cfg->max_agent))
        return qs_fail(nullptr, QS_E_INVAL, "qs_create: shard_rank * shard_bots must lie below max_agent");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return qs_fail(nullptr, QS_E_NODEV, "qs_create: no HIP device (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return qs_fail(nullptr, QS_E_NODEV, "qs_create: bad device ordinal");

    qs_ctx *c = new qs_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
    c->bots_per_graph = cfg->bots_per_graph > 0 ? cfg->bots_per_graph : cfg->max_agent;
    c->n_graphs = (cfg->max_agent + c->bots_per_graph - 1) / c->bots_per_graph;
    c->win = cfg->min_poses_between < 1 ? 1 : (cfg->min_poses_between > QS_WIN_MAX ? QS_WIN_MAX : cfg->min_poses_between);
    c->r2_threshold = r2_threshold_for(cfg->closure_radius);
    c->cells = (size_t)cfg->size * cfg->size;
    c->geom = QsGeom{cfg->size, cfg->res, cfg->ox, cfg->oy, cfg->min_dist, cfg->max_dist, 1.0 / cfg->res};
    {   // landmark buckets: edge a hair above the closure radius, so that two points closer than the
        // radius are never two buckets apart whatever the rounding of (v - b0) / cell.  The directory is a
        // hash table over the cells, sized for one entry per cell of the configured world (2^20 at most).
        const double cell = cfg->closure_radius > 0 ? cfg->closure_radius * (1.0 + 1e-9) : 1.0;
        double nbd = ceil(cfg->size * cfg->res / cell) + 1.0;
        if (!(nbd >= 1)) nbd = 1;
        if (nbd > 1024) nbd = 1024;
        unsigned int slab = 256;
        while ((double)slab < nbd * nbd) slab <<= 1;
        c->bg = QsBucketGeom{cfg->ox, cfg->oy, cell, 1.0 / cell, slab - 1, 0};
        c->dir_entries = (size_t)QS_NTYPES * slab;
    }
    const int nb = cfg->max_agent + 1;
#define CREATE_CHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { int rc__ = qs_fail(nullptr, QS_E_HIP, #x, e__); qs_destroy(c); return rc__; } } while (0)
    CREATE_CHK(hipSetDevice(c->device));
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    CREATE_CHK(hipMalloc((void **)&c->d_stamps, c->cells * sizeof(unsigned int)));
    if (cfg->enable_counts) CREATE_CHK(hipMalloc((void **)&c->d_counts, c->cells * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void **)&c->d_offset, nb * sizeof(double)));
    CREATE_CHK(hipMalloc((void **)&c->d_drift, nb * 2 * sizeof(double)));
    CREATE_CHK(hipMalloc((void **)&c->d_last_closure, nb * sizeof(long long)));
    CREATE_CHK(hipMalloc((void **)&c->d_zone, nb * 4 * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void **)&c->d_counters, QS_CNT_N * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void **)&c->d_graph_batch, (size_t)c->n_graphs * 2 * sizeof(unsigned long long)));
    CREATE_CHK(hipMalloc((void **)&c->d_ekf, (size_t)nb * 44 * sizeof(double)));
    CREATE_CHK(hipMalloc((void **)&c->d_ekf_prev, (size_t)nb * 4 * sizeof(double)));
    CREATE_CHK(hipMalloc((void **)&c->d_flags, QS_N_FLAGS * sizeof(unsigned int)));
    CREATE_CHK(hipMemset(c->d_flags, 0, QS_N_FLAGS * sizeof(unsigned int)));
    CREATE_CHK(hipHostMalloc((void **)&c->h_chain_stat, 8 * sizeof(unsigned int), hipHostMallocDefault));
    memset(c->h_chain_stat, 0, 8 * sizeof(unsigned int));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_chain_stat, hipEventDisableTiming));
    if (const char *e = getenv("QS_CHAIN_MODE")) c->chain_form = strcmp(e, "window") == 0 ? QS_CHAIN_WINDOW : strcmp(e, "free") == 0 ? QS_CHAIN_FREE : strcmp(e, "free_posting") == 0 ? QS_CHAIN_FREE_POSTING : QS_CHAIN_AUTO;
    CREATE_CHK(hipMalloc((void **)&c->d_graphs, (size_t)c->n_graphs * sizeof(QsGraphDev)));
    CREATE_CHK(hipMemset(c->d_graphs, 0, (size_t)c->n_graphs * sizeof(QsGraphDev)));
    c->h_graphs.assign(c->n_graphs, QsGraphDev{});
    c->lms_upper.assign(c->n_graphs, 0);
    c->cls_upper.assign(c->n_graphs, 0);
    std::vector<double> off(nb, 0.0);
    if (cfg->max_agent >= 2) off[2] = cfg->separation;                       // :851-852
    CREATE_CHK(hipMemcpy(c->d_offset, off.data(), nb * sizeof(double), hipMemcpyHostToDevice));
#undef CREATE_CHK
    for (int g = 0; g < c->n_graphs; g++) {
        int rc = graph_reserve(c, g, 1024, 256, 0, 0);
        if (rc != QS_OK) { g_create_err = c->err; qs_destroy(c); return rc; }
    }
    int rc = reset_state(c);
    if (rc != QS_OK) { g_create_err = c->err; qs_destroy(c); return rc; }
    *out = c;
    return QS_OK;
}

static void free_batch(qs_ctx *c)
{
    QsBatch &b = c->b;
    hipFree(b.accept); hipFree(b.agent); hipFree(b.lm); hipFree(b.px); hipFree(b.py); hipFree(b.yaw);
    hipFree(b.dist); hipFree(b.enc); hipFree(b.rx); hipFree(b.ry); hipFree(b.hit); hipFree(b.hit_valid);
    if (b.map_ok != b.accept) hipFree(b.map_ok);
    memset(&b, 0, sizeof b);
    QsSlamBatch &sb = c->sb;
    hipFree(sb.node); hipFree(sb.ev_node); hipFree(sb.ev_agent); hipFree(sb.ev_type); hipFree(sb.ev_px); hipFree(sb.ev_py);
    hipFree(sb.ev_base); hipFree(sb.acc_total); hipFree(sb.blk_acc); hipFree(sb.blk_ev); hipFree(sb.agent_ev);
    hipFree(sb.acl_node); hipFree(sb.acl_dx); hipFree(sb.acl_dy); hipFree(sb.acl_cnt); hipFree(sb.drift_start);
    memset(&sb, 0, sizeof sb);
    c->cap_batch = 0;
}

extern "C" int qs_destroy(qs_ctx *c)
{
    if (!c) return QS_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &g : c->h_graphs) graph_free(g);
    free_batch(c);
    hipFree(c->d_stamps); hipFree(c->d_counts); hipFree(c->d_counts_fused); hipFree(c->d_io_ws); hipFree(c->d_offset); hipFree(c->d_drift);
    hipFree(c->d_last_closure); hipFree(c->d_zone); hipFree(c->d_counters); hipFree(c->d_graph_batch);
    hipFree(c->d_ekf); hipFree(c->d_ekf_prev); hipFree(c->d_ekf_ws); hipFree(c->d_graphs); hipFree(c->d_flags); if (c->h_chain_stat) hipHostFree(c->h_chain_stat); if (c->ev_chain_stat) hipEventDestroy(c->ev_chain_stat); hipFree(c->d_pkts); hipFree(c->d_lens);
    hipFree(c->d_time); hipFree(c->d_bin_ws); hipFree(c->d_frontier_ws);
    hipFree(c->d_edge);
    hipFree(c->d_dirty); hipFree(c->d_counts_sent); hipFree(c->d_sf_bitmaps); hipFree(c->d_sf_lists); hipFree(c->d_sf_counts); hipFree(c->d_sf_payload);
    for (auto &p : c->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->ekf_stream) {
        hipStreamSynchronize(c->ekf_stream);
        if (!c->ekf_stream_shared) hipStreamDestroy(c->ekf_stream);
        else if (c->device < 64) {
            std::lock_guard<std::mutex> lk(g_masked_mutex);
            if (--g_masked_users[c->device] == 0) {                          // the last context of the device takes the shared stream with it
                hipStreamDestroy(g_masked[c->device]);                       // (a profiler's exit handler trips over a CU-masked queue left behind)
                g_masked[c->device] = nullptr;
            }
        }
    }
    if (c->ev_decoded) hipEventDestroy(c->ev_decoded);
    if (c->ev_ekf_done) hipEventDestroy(c->ev_ekf_done);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return QS_OK;
}

extern "C" const char *qs_last_error(const qs_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" int qs_set_stream(qs_ctx *c, void *hip_stream)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hip_stream) {
        if (c->own_stream) { hipStreamDestroy(c->stream); c->own_stream = false; }
        c->stream = (hipStream_t)hip_stream;
    } else if (!c->own_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return QS_OK;
}

extern "C" int qs_set_chain_form(qs_ctx *c, int form)
{
    ARGCHK(c, c != nullptr);
    ARGCHK(c, form == QS_CHAIN_AUTO || form == QS_CHAIN_FREE || form == QS_CHAIN_WINDOW || form == QS_CHAIN_FREE_POSTING);
    c->chain_form = form;
    return QS_OK;
}

extern "C" int qs_chain_form(qs_ctx *c)
{
    if (!c) return QS_E_INVAL;
    return !c->chain_last_free ? QS_CHAIN_WINDOW : c->chain_last_posting ? QS_CHAIN_FREE_POSTING : QS_CHAIN_FREE;
}

extern "C" int qs_sync(qs_ctx *c)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

extern "C" int qs_reset(qs_ctx *c)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    return reset_state(c);
}

extern "C" int qs_set_bot_offset(qs_ctx *c, int32_t bot, double off_x)
{
    ARGCHK(c, c != nullptr);
    if (bot < 1 || bot > c->cfg.max_agent) return qs_fail(c, QS_E_RANGE, "qs_set_bot_offset: bot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_offset + bot, &off_x, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

// ---- timing (StageTimer: qs_internal.h) -----------------------------------------------------------
extern "C" int qs_timing_enable(qs_ctx *c, int32_t enable)
{
    ARGCHK(c, c != nullptr);
    c->timing = enable != 0;
    return QS_OK;
}

extern "C" int qs_stage_times(qs_ctx *c, double ms[QS_STAGE_N], uint64_t launches[QS_STAGE_N], int32_t reset)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { c->stage_ms[p.stage] += t; c->stage_launches[p.stage]++; }
        c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
    for (int s = 0; s < QS_STAGE_N; s++) { if (ms) ms[s] = c->stage_ms[s]; if (launches) launches[s] = c->stage_launches[s]; }
    if (reset) for (int s = 0; s < QS_STAGE_N; s++) { c->stage_ms[s] = 0; c->stage_launches[s] = 0; }
    return QS_OK;
}

// ---- batch buffers --------------------------------------------------------------------------
static int ensure_batch(qs_ctx *c, size_t n)
{
    if (n <= c->cap_batch) return QS_OK;
    size_t cap = c->cap_batch ? c->cap_batch : 1024;
    while (cap < n) cap *= 2;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    QsBatch &b = c->b;
    HIPCHK(c, dev_realloc(&b.accept, cap)); HIPCHK(c, dev_realloc(&b.agent, cap)); HIPCHK(c, dev_realloc(&b.lm, cap));
    HIPCHK(c, dev_realloc(&b.px, cap)); HIPCHK(c, dev_realloc(&b.py, cap)); HIPCHK(c, dev_realloc(&b.yaw, cap));
    HIPCHK(c, dev_realloc(&b.dist, cap)); HIPCHK(c, dev_realloc(&b.enc, cap));
    HIPCHK(c, dev_realloc(&b.rx, cap)); HIPCHK(c, dev_realloc(&b.ry, cap));
    HIPCHK(c, dev_realloc(&b.hit, 4 * cap)); HIPCHK(c, dev_realloc(&b.hit_valid, 4 * cap));
    if (c->cfg.shard_bots > 0) {
        if (b.map_ok == b.accept) b.map_ok = nullptr;        // (accept was re-allocated above)
        HIPCHK(c, dev_realloc(&b.map_ok, cap));
        b.own_lo = c->cfg.shard_rank * c->cfg.shard_bots + 1;
        b.own_hi = std::min(c->cfg.max_agent, (c->cfg.shard_rank + 1) * c->cfg.shard_bots);
    } else { b.map_ok = b.accept; b.own_lo = 1; b.own_hi = c->cfg.max_agent; }
    if (c->cfg.exact_trig) {
        if (!c->d_edge) HIPCHK(c, hipMalloc((void **)&c->d_edge, (size_t)QS_EDGE_CAP * sizeof(QsEdgeRec)));
        b.edge = c->d_edge; b.edge_n = c->d_flags; b.edge_cap = QS_EDGE_CAP;
    }
    QsSlamBatch &sb = c->sb;
    const size_t nblk = (size_t)qs_slam_blocks(cap), G = (size_t)c->n_graphs, nb = (size_t)c->cfg.max_agent + 2;
    HIPCHK(c, dev_realloc(&sb.node, cap)); HIPCHK(c, dev_realloc(&sb.ev_node, cap));
    HIPCHK(c, dev_realloc(&sb.ev_agent, cap)); HIPCHK(c, dev_realloc(&sb.ev_type, cap));
    HIPCHK(c, dev_realloc(&sb.ev_px, cap)); HIPCHK(c, dev_realloc(&sb.ev_py, cap));
    HIPCHK(c, dev_realloc(&sb.ev_base, G + 1)); HIPCHK(c, dev_realloc(&sb.acc_total, G));
    HIPCHK(c, dev_realloc(&sb.blk_acc, G * nblk)); HIPCHK(c, dev_realloc(&sb.blk_ev, G * nblk));
    HIPCHK(c, dev_realloc(&sb.agent_ev, nb)); HIPCHK(c, dev_realloc(&sb.acl_cnt, nb));
    HIPCHK(c, dev_realloc(&sb.acl_node, cap)); HIPCHK(c, dev_realloc(&sb.acl_dx, cap)); HIPCHK(c, dev_realloc(&sb.acl_dy, cap));
    HIPCHK(c, dev_realloc(&sb.drift_start, 2 * nb));
    c->cap_batch = cap;
    return QS_OK;
}

// Stamp ordinals are 30 bits (so stamps stay below 2^31 and an int32 MAX all-reduce is valid).
static const uint64_t QS_EPOCH_LIMIT = (1ull << 28) - 2;
static bool epoch_would_rebase(const qs_ctx *c, uint64_t seq0, size_t n_seq)
{
    return seq0 + n_seq - c->epoch_base > QS_EPOCH_LIMIT;
}
static int ensure_epoch(qs_ctx *c, uint64_t seq0, size_t n_seq)
{
    if (seq0 < c->epoch_base) return qs_fail(c, QS_E_INVAL, "seq0 precedes the current stamp epoch (sequence numbers must not decrease)");
    if (n_seq > QS_EPOCH_LIMIT) return qs_fail(c, QS_E_RANGE, "batch too large for one stamp epoch (2^28 records)");
    if (epoch_would_rebase(c, seq0, n_seq)) {
        // A rebase collapses every written cell to ordinal 1.  In one mapper that keeps the order against all later
        // writes; in a shard -- of a round-robin stream (seq_stride > 1) or of a replicated pose graph (shard_bots > 0:
        // every rank sees every packet but casts only its own agents' rays) -- two ranks' unfused writes to one cell
        // would tie afterwards, so the shards must have exchanged their stamps first (dist.ShardedMapper does: it asks
        // qs_epoch_query before every ingest).
        if ((c->cfg.seq_stride > 1 || c->cfg.shard_bots > 0) && c->dirty_since_fuse)
            return qs_fail(c, QS_E_STATE, "this batch crosses a stamp epoch: fuse the shards' grids (all-reduce + qs_mark_fused) first");
        { int rcf = flush_edge_rays(c); if (rcf != QS_OK) return rcf; }     // waiting rays carry stamps of the epoch that ends here
        HIPCHK(c, qs_launch_rebase(c));
        c->epoch_base = seq0 ? seq0 - 1 : 0;
        c->n_rebases++;
    }
    return QS_OK;
}

extern "C" int qs_epoch_query(qs_ctx *c, uint64_t seq0, size_t n, int32_t *would_rebase)
{
    ARGCHK(c, c != nullptr && would_rebase != nullptr);
    if (seq0 == UINT64_MAX) seq0 = c->next_seq;
    const uint64_t sstride = c->cfg.seq_stride > 0 ? (uint64_t)c->cfg.seq_stride : 1;
    *would_rebase = (n > 0 && epoch_would_rebase(c, seq0 - seq0 % sstride, n * sstride)) ? 1 : 0;
    return QS_OK;
}

extern "C" int qs_mark_fused(qs_ctx *c)
{
    ARGCHK(c, c != nullptr);
    c->dirty_since_fuse = false;
    return QS_OK;
}

static int reserve_graphs_for_batch(qs_ctx *c, size_t n)
{
    bool need_sync = false;
    for (int g = 0; g < c->n_graphs; g++)
        if (c->lms_upper[g] + (long long)n > c->h_graphs[g].cap_lms || c->cls_upper[g] + (long long)n > c->h_graphs[g].cap_cls)
            need_sync = true;
    if (!need_sync) {
        for (int g = 0; g < c->n_graphs; g++) { c->lms_upper[g] += (long long)n; c->cls_upper[g] += (long long)n; }
        return QS_OK;
    }
    // The batch's landmark events per graph are known on the device only.  Asking costs a host sync in the
    // middle of the pipeline (every launch after the decode waits for it), so when memory allows, the graphs
    // that are short are simply grown to the safe bound -- every record a landmark of that graph -- and the
    // next batches of this size go through without a question: ~260 B per unit of capacity (log, closures,
    // side list, worst-case node pool), against 288 GB.
    {
        const double unit = 2 * 8 + 8 + 1 + 2 * 8 + 2 * 8 + 4 + sizeof(QsLmNode) + 4;
        double extra = 0;
        for (int g = 0; g < c->n_graphs; g++) {
            const long long nl = c->lms_upper[g] + (long long)n, nc = c->cls_upper[g] + (long long)n;
            if (nl > c->h_graphs[g].cap_lms) extra += (double)(2 * nl - c->h_graphs[g].cap_lms) * unit;     // (capacities double)
            if (!c->h_graphs[g].nodes) extra += (double)(1 + c->dir_entries) * (sizeof(QsLmNode) + 4 + sizeof(QsDirEntry));   // first nodes
            if (nc > c->h_graphs[g].cap_cls) extra += (double)(2 * nc - c->h_graphs[g].cap_cls) * 32.0;
        }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && extra <= 0.25 * (double)free_b) {
            for (int g = 0; g < c->n_graphs; g++) {
                const long long nl = c->lms_upper[g] + (long long)n, nc = c->cls_upper[g] + (long long)n;
                int rc = graph_reserve(c, g, nl, nc, c->lms_upper[g], c->cls_upper[g]);
                if (rc != QS_OK) return rc;
                c->lms_upper[g] = nl; c->cls_upper[g] = nc;
            }
            return QS_OK;
        }
    }
    // tighten the bounds with the exact device-side numbers, then grow what is really short
    std::vector<QsGraphDev> cur(c->n_graphs);
    std::vector<unsigned long long> gb((size_t)c->n_graphs * 2);
    HIPCHK(c, hipMemcpyAsync(cur.data(), c->d_graphs, cur.size() * sizeof(QsGraphDev), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(gb.data(), c->d_graph_batch, gb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int g = 0; g < c->n_graphs; g++) {
        const long long ev = (long long)gb[2 * g + 1];
        const long long need_l = cur[g].n_lms + ev, need_c = cur[g].n_cls + ev;
        int rc = graph_reserve(c, g, need_l, need_c, cur[g].n_lms, cur[g].n_cls);
        if (rc != QS_OK) return rc;
        c->lms_upper[g] = need_l; c->cls_upper[g] = need_c;
    }
    return QS_OK;
}

static int io_reserve(qs_ctx *c, size_t bytes);
// Exact-trig mode (qs_config.exact_trig, default on): rays the device did not decide (raycast_common.h, qs_edge_ray) wait in
// a list of self-contained records (pose, distance, stamp) and get their end points from libm here -- math.cos / math.sin of
// the reference are glibc's -- before they are cast with the stamps their ingest gave them (stamps make the order
// irrelevant).  An ingest does not wait for this: the list is flushed at the next point where the map can be OBSERVED (any
// call that reads or hands out the grid, the counters or the pose graphs; qs_sync; before a stamp rebase) and dropped by
// qs_reset.  The same synchronisation brings the graphs' real landmark / closure counts (capacity planning starts from them,
// not from "every record so far was a landmark") and the pile flag of the loop-closure chain.
// Which instantiation of the free-running loop-closure chain suits the stream (slam.hip, qs_launch_slam): with or without the
// owners posting their landmarks' poses.  The kernels keep running totals in device words (decisions that had to wait for the
// committer / scans of the posted poses; closures); every ingest asks for a copy of them into pinned memory behind itself and
// looks, before it launches its own chain, at whatever copy has landed by then -- nobody waits.  More than 1 in 8: posting on;
// fewer than 1 in 16: off again (both instantiations count the same events); more scans than closures even so (a stream
// that hardly ever matches: the adversarial one spread over an 8192^2 world): the per-window kernel, until its queries
// that find nothing are fewer than half its closures.
static void chain_stats_poll(qs_ctx *c, bool synced, const unsigned int *fresh)
{
    // fresh: the four totals as a synchronising call has just read them (newer than any copy in flight, which has landed too)
    if (!fresh && !c->chain_stat_pending) return;
    if (!fresh && !synced && hipEventQuery(c->ev_chain_stat) != hipSuccess) { (void)hipGetLastError(); return; }
    c->chain_stat_pending = false;
    unsigned int *now = c->h_chain_stat, *seen = c->h_chain_stat + 4;
    if (fresh) for (int i = 0; i < 4; i++) now[i] = fresh[i];
    const uint64_t f_miss = now[0] - seen[0], f_hit = now[1] - seen[1], w_miss = now[2] - seen[2], w_hit = now[3] - seen[3];
    if (c->chain_windowed) { if (w_miss + w_hit >= 256 && w_miss * 2 < w_hit) c->chain_windowed = false; }   // (back to posting)
    else if (f_miss + f_hit >= 256) {
        if (!c->chain_posting) { if (f_miss * 8 > f_hit) c->chain_posting = true; }
        else if (f_miss > f_hit) c->chain_windowed = true;   // more scans than closures: the per-window kernel's LDS windows are cheaper
        else if (f_miss * 16 < f_hit) c->chain_posting = false;
    }
    for (int i = 0; i < 4; i++) seen[i] = now[i];
}
static int chain_stats_request(qs_ctx *c)
{
    if (c->chain_stat_pending) return QS_OK;                 // (the copy in flight will do)
    // one ingest in four: the copy is a blit kernel with a barrier either side (~20 us of a 1.4 ms step when the stream is
    // 64 bots), and what it carries only ever changes the choice of an instantiation
    if ((c->chain_stat_tick++ & 3u) != 0) return QS_OK;
    HIPCHK(c, hipMemcpyAsync(c->h_chain_stat, c->d_flags + QS_FLAG_CHAIN_MISS, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_chain_stat, c->stream));
    c->chain_stat_pending = true;
    return QS_OK;
}

static int flush_edge_rays(qs_ctx *c)
{
    if (!c->edge_maybe && !c->flags_maybe) return QS_OK;
    unsigned int fl[QS_N_FLAGS] = {0};
    HIPCHK(c, hipMemcpyAsync(fl, c->d_flags, sizeof fl, hipMemcpyDeviceToHost, c->stream));
    std::vector<QsGraphDev> cur((size_t)c->n_graphs);
    HIPCHK(c, hipMemcpyAsync(cur.data(), c->d_graphs, cur.size() * sizeof(QsGraphDev), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->edge_maybe = false; c->flags_maybe = false;
    for (int g = 0; g < c->n_graphs; g++) { c->lms_upper[g] = cur[g].n_lms; c->cls_upper[g] = cur[g].n_cls; }
    if (fl[QS_FLAG_PILE]) c->pile_mode = true;           // a landmark pile has formed: the chain kernel's DENSE variant from now on
    chain_stats_poll(c, true, fl + QS_FLAG_CHAIN_MISS);
    const unsigned int n_edge = fl[QS_FLAG_EDGE_N] < QS_EDGE_CAP ? fl[QS_FLAG_EDGE_N] : QS_EDGE_CAP;
    c->edge_overflow_total += fl[QS_FLAG_EDGE_OVF];
    if (n_edge == 0) return QS_OK;
    c->edge_rays_total += n_edge;
    std::vector<QsEdgeRec> recs(n_edge);
    HIPCHK(c, hipMemcpy(recs.data(), c->d_edge, (size_t)n_edge * sizeof(QsEdgeRec), hipMemcpyDeviceToHost));
    const size_t bytes = (size_t)n_edge * 4 * sizeof(double);
    int rc = io_reserve(c, bytes);
    if (rc != QS_OK) return rc;
    double *d = (double *)c->d_io_ws;
    std::vector<double> h((size_t)n_edge * 4);
    static const double kPi = 3.141592653589793;                                              // math.pi
    static const double off[4] = {0.0, kPi / 2, kPi, -kPi / 2};                               // :61-66
    for (unsigned int e = 0; e < n_edge; e++) {
        const QsEdgeRec &r = recs[e];
        const double dd = (double)r.d;
        const int sensor = (int)(((r.key_free >> 1) - 1) & 3);                                 // ordinal = 4 * arrival index + sensor + 1
        const double a = r.yaw + off[sensor];                                                  // :887
        const bool valid = (c->cfg.min_dist < dd) && (dd <= c->cfg.max_dist);                  // :888
        const double range = valid ? dd : ((dd > c->cfg.min_dist) ? ((c->cfg.max_dist < dd) ? c->cfg.max_dist : dd) : c->cfg.max_dist);   // :900
        h[4 * e] = r.rx + range * cos(a);                                                      // :890 / :901
        h[4 * e + 1] = r.ry + range * sin(a);                                                  // :891 / :902
        h[4 * e + 2] = valid ? 1.0 : 0.0;
        h[4 * e + 3] = 0.0;
    }
    HIPCHK(c, hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, qs_launch_edge_cast(c, n_edge, d));
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, sizeof(unsigned int), c->stream));                 // the list is empty again
    HIPCHK(c, hipMemsetAsync(c->d_flags + 2, 0, sizeof(unsigned int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                                                // h goes out of scope
    return QS_OK;
}

// at a point where the host waits for the stream anyway: the chain's flags (pile, form statistics)
static int read_pile_flag(qs_ctx *c)
{
    c->flags_maybe = true;
    return flush_edge_rays(c);
}

static int ingest_device(qs_ctx *c, const uint8_t *d_pkts, size_t n, size_t stride, const uint16_t *d_lens,
                         const double *d_time, uint64_t seq0)
{
    if (seq0 == UINT64_MAX) seq0 = c->next_seq;
    c->last_n = n; c->last_has_poses = true;
    if (n == 0) return QS_OK;
    int rc = ensure_batch(c, n);
    if (rc != QS_OK) return rc;
    c->b.n = n;
    const uint64_t sstride = c->cfg.seq_stride > 0 ? (uint64_t)c->cfg.seq_stride : 1;
    // epoch decisions use the stride-aligned range so that all ranks of a sharded stream agree
    rc = ensure_epoch(c, seq0 - seq0 % sstride, n * sstride);
    if (rc != QS_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_graph_batch, 0, (size_t)c->n_graphs * 2 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->sb.agent_ev, 0, ((size_t)c->cfg.max_agent + 2) * sizeof(unsigned int), c->stream));
    { StageTimer t(c, QS_STAGE_DECODE); HIPCHK(c, qs_launch_decode(c, d_pkts, n, stride, d_lens)); t.stop(); }
    rc = reserve_graphs_for_batch(c, n);
    if (rc != QS_OK) return rc;
    if (c->cfg.enable_ekf) {
        // fork: the filter only needs the decoded fields, never the map (and the map never the filter)
        if (!c->ekf_stream) {
            // The filter's stream keeps off the lowest 32 CUs.  Its kernels run beside the loop-closure chain, whose workgroups
            // (one per pose graph, each a whole CU's worth of latency-bound waves) lose ~10 % when scan kernels share their
            // SIMDs; with 32 CUs left alone the dispatcher puts the chain there (64 bots / 32 graphs: chain 1.26 -> 1.15 ms,
            // step 1.89 -> 1.80 ms; tools/ekf_cu_mask_probe.sh).  QS_EKF_CU_MASK = hex words (lowest CUs first) overrides,
            // "none" switches the mask off; a device too small for it, or a refusal, falls back to an ordinary stream.
            const char *mk = getenv("QS_EKF_CU_MASK");
            std::vector<uint32_t> words;
            if (mk && *mk && strcmp(mk, "none") != 0) {
                char *end = nullptr;
                for (const char *q = mk; *q;) { words.push_back((uint32_t)strtoul(q, &end, 16)); if (end == q) break; q = (*end == ',') ? end + 1 : end; }
            } else if (!mk || !*mk) {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount >= 128) {
                    words.assign((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
                    words[0] = 0u;
                }
            }
            // ONE masked stream per device, shared by its contexts (the last one destroys it): a second CU-masked queue on the same GPU
            // slows every kernel of the process by 30-50 % (measured: two contexts, each with its own masked stream, 1.81 ->
            // 2.79 ms per 64-bot step; tools/secondary_probe.py).  Contexts of one process then run their filters one after
            // the other, which is how they are driven anyway (a caller serialises the calls on a context).
            if (!words.empty() && c->device < 64) {
                std::lock_guard<std::mutex> lk(g_masked_mutex);
                if (!g_masked[c->device] && hipExtStreamCreateWithCUMask(&g_masked[c->device], (uint32_t)words.size(), words.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    g_masked[c->device] = nullptr;
                }
                c->ekf_stream = g_masked[c->device];
                c->ekf_stream_shared = c->ekf_stream != nullptr;
                if (c->ekf_stream_shared) g_masked_users[c->device]++;
            }
            if (!c->ekf_stream)
                HIPCHK(c, hipStreamCreateWithFlags(&c->ekf_stream, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_decoded, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_ekf_done, hipEventDisableTiming));
        }
        HIPCHK(c, hipEventRecord(c->ev_decoded, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->ekf_stream, c->ev_decoded, 0));
        { StageTimer t(c, QS_STAGE_EKF, c->ekf_stream); HIPCHK(c, n >= QS_EKF_SCAN_MIN_BATCH ? qs_launch_ekf_scan(c, n, d_time, c->ekf_stream)
                                                   : qs_launch_ekf_ingest(c, n, d_time, c->ekf_stream)); t.stop(); }
        HIPCHK(c, hipEventRecord(c->ev_ekf_done, c->ekf_stream));
    }
    chain_stats_poll(c, false, nullptr);
    { StageTimer t(c, QS_STAGE_SLAM); HIPCHK(c, qs_launch_slam(c, n)); t.stop(); }
    { int rcs = chain_stats_request(c); if (rcs != QS_OK) return rcs; }
    {
        StageTimer t(c, QS_STAGE_RAYCAST);
        // auto (0): a handful of packets (the live UDP path: <= 20 per frame) is one direct kernel instead
        // of the four tiled passes -- same cells either way; 1 = always direct, 2 = always tiled
        if (c->cfg.raycast_mode == 1 || (c->cfg.raycast_mode == 0 && n <= QS_DIRECT_MAX_BATCH))
            HIPCHK(c, qs_launch_raycast_direct(c, n, seq0));
        else HIPCHK(c, qs_launch_raycast_tiled(c, n, seq0));
        t.stop();
    }
    if (c->cfg.enable_ekf) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_ekf_done, 0));   // join
    if (c->b.edge) c->edge_maybe = true;                 // resolved at the next point the map is observed (flush_edge_rays)
    c->flags_maybe = true;
    c->next_seq = seq0 + n * sstride;
    c->dirty_since_fuse = true;
    return QS_OK;
}

extern "C" int qs_ingest_device(qs_ctx *c, const uint8_t *d_pkts, size_t n, size_t stride, const uint16_t *d_lens,
                                const double *d_time, uint64_t seq0)
{
    ARGCHK(c, c != nullptr);
    ARGCHK(c, n == 0 || (d_pkts != nullptr && stride >= QS_PACKET_SIZE_V1));
    HIPCHK(c, hipSetDevice(c->device));
    return ingest_device(c, d_pkts, n, stride, d_lens, d_time, seq0);
}

extern "C" int qs_ingest(qs_ctx *c, const uint8_t *pkts, size_t n, size_t stride, const uint16_t *lens,
                         const double *recv_time, uint64_t seq0)
{
    ARGCHK(c, c != nullptr);
    ARGCHK(c, n == 0 || (pkts != nullptr && stride >= QS_PACKET_SIZE_V1));
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) { c->last_n = 0; return QS_OK; }
    const size_t bytes = n * stride;
    if (bytes > c->cap_pkts_bytes) {
        size_t cap = c->cap_pkts_bytes ? c->cap_pkts_bytes : (1u << 16);
        while (cap < bytes) cap *= 2;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_realloc(&c->d_pkts, cap));
        HIPCHK(c, dev_realloc(&c->d_lens, cap / QS_PACKET_SIZE_V1 + 1));
        HIPCHK(c, dev_realloc(&c->d_time, cap / QS_PACKET_SIZE_V1 + 1));
        c->cap_pkts_bytes = cap;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_pkts, pkts, bytes, hipMemcpyHostToDevice, c->stream));
    if (lens) HIPCHK(c, hipMemcpyAsync(c->d_lens, lens, n * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    if (recv_time) HIPCHK(c, hipMemcpyAsync(c->d_time, recv_time, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = ingest_device(c, c->d_pkts, n, stride, lens ? c->d_lens : nullptr, recv_time ? c->d_time : nullptr, seq0);
    if (rc != QS_OK) return rc;
    // this call waits for the GPU anyway (the caller's buffers are free when it returns): the waiting edge rays are resolved
    // now, and the graphs' real landmark / closure counts and the pile flag come along
    if (c->cfg.exact_trig) c->edge_maybe = true;
    return read_pile_flag(c);
}

extern "C" int qs_last_batch(qs_ctx *c, uint8_t *accepted, double *pose, size_t n)
{
    ARGCHK(c, c != nullptr);
    if (!c->last_has_poses || n != c->last_n) return qs_fail(c, QS_E_INVAL, "qs_last_batch: n does not match the last ingest");
    if (n == 0) return QS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint8_t> acc(n);
    HIPCHK(c, hipMemcpyAsync(acc.data(), c->b.accept, n, hipMemcpyDeviceToHost, c->stream));
    std::vector<double> rx, ry, yaw;
    if (pose) {
        rx.resize(n); ry.resize(n); yaw.resize(n);
        HIPCHK(c, hipMemcpyAsync(rx.data(), c->b.rx, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(ry.data(), c->b.ry, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(yaw.data(), c->b.yaw, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        if (accepted) accepted[i] = acc[i];
        if (pose) {
            pose[3 * i] = acc[i] ? rx[i] : NAN; pose[3 * i + 1] = acc[i] ? ry[i] : NAN; pose[3 * i + 2] = acc[i] ? yaw[i] : NAN;
        }
    }
    return QS_OK;
}

extern "C" int qs_last_hits(qs_ctx *c, double *xy, uint8_t *valid, size_t n)
{
    ARGCHK(c, c != nullptr && xy != nullptr && valid != nullptr);
    if (!c->last_has_poses || n != c->last_n) return qs_fail(c, QS_E_INVAL, "qs_last_hits: n does not match the last ingest");
    if (n == 0) return QS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint8_t> acc(n);
    HIPCHK(c, qs_launch_hits(c, n));
    HIPCHK(c, hipMemcpyAsync(acc.data(), c->b.accept, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(xy, c->b.hit, 4 * n * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(valid, c->b.hit_valid, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++)
        if (!acc[i]) for (int s = 0; s < 4; s++) valid[4 * i + s] = 0;
    return QS_OK;
}

// ---- OccupancyGrid object API ---------------------------------------------------------------
static int io_reserve(qs_ctx *c, size_t bytes)
{
    if (bytes <= c->io_ws_bytes) return QS_OK;
    size_t cap = c->io_ws_bytes ? c->io_ws_bytes : (1u << 16);
    while (cap < bytes) cap *= 2;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_io_ws) { HIPCHK(c, hipFree(c->d_io_ws)); c->d_io_ws = nullptr; c->io_ws_bytes = 0; }
    HIPCHK(c, hipMalloc(&c->d_io_ws, cap));
    c->io_ws_bytes = cap;
    return QS_OK;
}

extern "C" int qs_update_rays(qs_ctx *c, const double *rx, const double *ry, const double *hx, const double *hy,
                              const uint8_t *valid, size_t n, uint64_t seq0)
{
    ARGCHK(c, c != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(c, rx && ry && hx && hy && valid);
    HIPCHK(c, hipSetDevice(c->device));
    if (seq0 == UINT64_MAX) seq0 = c->next_seq;
    const size_t n_seq = (n + 3) / 4;
    int rc = ensure_epoch(c, seq0, n_seq);
    if (rc != QS_OK) return rc;
    // staging lives with the context (grown on demand): the object API's update_ray is one ray per call
    int rc2 = io_reserve(c, 4 * n * sizeof(double) + n);
    if (rc2 != QS_OK) return rc2;
    double *d = (double *)c->d_io_ws; unsigned char *dv = (unsigned char *)(d + 4 * n);
    hipError_t e = hipSuccess;
    const double *src[4] = {rx, ry, hx, hy};
    for (int q = 0; q < 4 && e == hipSuccess; q++)
        e = hipMemcpyAsync(d + q * n, src[q], n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dv, valid, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_update_rays(c, d, d + n, d + 2 * n, d + 3 * n, dv, n, seq0);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_update_rays", e);
    c->dirty_since_fuse = true;
    c->next_seq = seq0 + n_seq;
    c->last_has_poses = false;
    return QS_OK;
}

extern "C" int qs_world_to_grid(qs_ctx *c, const double *w, size_t n, int32_t axis, int64_t *out)
{
    ARGCHK(c, c != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(c, w && out);
    HIPCHK(c, hipSetDevice(c->device));
    double *d = nullptr; long long *o = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, n * sizeof(double)));
    HIPCHK(c, hipMalloc((void **)&o, n * sizeof(long long)));
    hipError_t e = hipMemcpyAsync(d, w, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_world_to_grid(c, d, n, axis, o);
    if (e == hipSuccess) e = hipMemcpyAsync(out, o, n * sizeof(long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d); hipFree(o);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_world_to_grid", e);
    return QS_OK;
}

extern "C" int qs_grid_i8_device(qs_ctx *c, int8_t *out_dev)
{
    ARGCHK(c, c != nullptr && out_dev != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    HIPCHK(c, qs_launch_view_i8(c, (signed char *)out_dev));
    return QS_OK;
}

extern "C" int qs_grid_i8(qs_ctx *c, int8_t *out_host)
{
    ARGCHK(c, c != nullptr && out_host != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    signed char *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, c->cells));
    hipError_t e = qs_launch_view_i8(c, d);
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, c->cells, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_grid_i8", e);
    return QS_OK;
}

extern "C" int qs_grid_counts(qs_ctx *c, int32_t *hits_host, int32_t *misses_host)
{
    ARGCHK(c, c != nullptr && hits_host && misses_host);
    if (!c->d_counts) return qs_fail(c, QS_E_INVAL, "qs_grid_counts: context created with enable_counts = 0");
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    int *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, 2 * c->cells * sizeof(int)));
    hipError_t e = qs_launch_split_counts(c, d, d + c->cells);
    if (e == hipSuccess) e = hipMemcpyAsync(hits_host, d, c->cells * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(misses_host, d + c->cells, c->cells * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_grid_counts", e);
    return QS_OK;
}

extern "C" int qs_grid_logodds(qs_ctx *c, float l_occ, float l_free, float lmin, float lmax, float *out_host)
{
    ARGCHK(c, c != nullptr && out_host);
    if (!c->d_counts) return qs_fail(c, QS_E_INVAL, "qs_grid_logodds: context created with enable_counts = 0");
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    float *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, c->cells * sizeof(float)));
    hipError_t e = qs_launch_logodds(c, l_occ, l_free, lmin, lmax, d);
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, c->cells * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_grid_logodds", e);
    return QS_OK;
}

extern "C" int qs_device_buffers(qs_ctx *c, void **stamps_dev, size_t *stamps_bytes, void **counts_dev, size_t *counts_bytes)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);                                   // whoever gets the buffers may read them (a collective)
    if (stamps_dev) *stamps_dev = c->d_stamps;
    if (stamps_bytes) *stamps_bytes = c->cells * sizeof(unsigned int);
    if (counts_dev) *counts_dev = c->d_counts;
    if (counts_bytes) *counts_bytes = c->d_counts ? c->cells * sizeof(unsigned long long) : 0;
    return QS_OK;
}

// ---- SLAM state -------------------------------------------------------------------------------
static int read_graph(qs_ctx *c, int32_t graph, QsGraphDev &g)
{
    if (graph < 0 || graph >= c->n_graphs) return qs_fail(c, QS_E_RANGE, "graph index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(&g, c->d_graphs + graph, sizeof g, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->lms_upper[graph] = g.n_lms; c->cls_upper[graph] = g.n_cls;      // exact now: nothing is in flight
    return QS_OK;
}

extern "C" int qs_slam_sizes(qs_ctx *c, int32_t graph, int64_t *n_nodes, int64_t *n_landmarks, int64_t *n_closures)
{
    ARGCHK(c, c != nullptr);
    QsGraphDev g;
    int rc = read_graph(c, graph, g);
    if (rc != QS_OK) return rc;
    if (n_nodes) *n_nodes = g.n_nodes;
    if (n_landmarks) *n_landmarks = g.n_lms;
    if (n_closures) *n_closures = g.n_cls;
    return QS_OK;
}

extern "C" int qs_slam_closures(qs_ctx *c, int32_t graph, int64_t *idx2, double *corr2, size_t cap)
{
    ARGCHK(c, c != nullptr && idx2 && corr2);
    QsGraphDev g;
    int rc = read_graph(c, graph, g);
    if (rc != QS_OK) return rc;
    const size_t n = (size_t)g.n_cls;
    if (n > cap) return qs_fail(c, QS_E_RANGE, "qs_slam_closures: capacity too small");
    if (n == 0) return QS_OK;
    std::vector<long long> a(n), b(n); std::vector<double> dx(n), dy(n);
    HIPCHK(c, hipMemcpy(a.data(), g.cl_lm_idx, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(b.data(), g.cl_node_idx, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(dx.data(), g.cl_dx, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(dy.data(), g.cl_dy, n * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) { idx2[2 * i] = a[i]; idx2[2 * i + 1] = b[i]; corr2[2 * i] = dx[i]; corr2[2 * i + 1] = dy[i]; }
    return QS_OK;
}

// agent_id of every closure's closing node (what get_correction_for_agent reads through self.nodes[node_idx], :335)
extern "C" int qs_slam_closure_agents(qs_ctx *c, int32_t graph, uint8_t *agents, size_t cap)
{
    ARGCHK(c, c != nullptr && agents);
    QsGraphDev g;
    int rc = read_graph(c, graph, g);
    if (rc != QS_OK) return rc;
    const size_t n = (size_t)g.n_cls;
    if (n > cap) return qs_fail(c, QS_E_RANGE, "qs_slam_closure_agents: capacity too small");
    if (n) HIPCHK(c, hipMemcpy(agents, g.cl_agent, n, hipMemcpyDeviceToHost));
    return QS_OK;
}

extern "C" int qs_slam_landmarks(qs_ctx *c, int32_t graph, double *xy, int64_t *type_idx, size_t cap)
{
    ARGCHK(c, c != nullptr && xy && type_idx);
    QsGraphDev g;
    int rc = read_graph(c, graph, g);
    if (rc != QS_OK) return rc;
    const size_t n = (size_t)g.n_lms;
    if (n > cap) return qs_fail(c, QS_E_RANGE, "qs_slam_landmarks: capacity too small");
    if (n == 0) return QS_OK;
    std::vector<double> x(n), y(n); std::vector<long long> idx(n); std::vector<unsigned char> t(n);
    HIPCHK(c, hipMemcpy(x.data(), g.lm_x, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(y.data(), g.lm_y, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(idx.data(), g.lm_idx, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(t.data(), g.lm_type, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) { xy[2 * i] = x[i]; xy[2 * i + 1] = y[i]; type_idx[2 * i] = t[i]; type_idx[2 * i + 1] = idx[i]; }
    return QS_OK;
}

// PoseGraphSLAM.add_pose, batched (object API): poses as given, no rays, no EKF.
extern "C" int qs_slam_add_poses(qs_ctx *c, const double *x, const double *y, const uint8_t *agent, const uint8_t *landmark,
                                 size_t n, uint8_t *closed, double *corr2)
{
    ARGCHK(c, c != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(c, x && y && agent && landmark);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_batch(c, n);
    if (rc != QS_OK) return rc;
    const int G = c->n_graphs, nb = c->cfg.max_agent + 2;
    std::vector<unsigned char> acc(n);
    std::vector<unsigned long long> gb((size_t)G * 2, 0);
    std::vector<unsigned int> aev(nb, 0);
    for (size_t i = 0; i < n; i++) {
        const bool ok = agent[i] >= 1 && agent[i] <= c->cfg.max_agent && isfinite(x[i]) && isfinite(y[i]);
        acc[i] = ok ? 1 : 0;
        if (!ok) continue;
        const int g = (agent[i] - 1) / c->bots_per_graph;
        gb[2 * g]++;
        if (landmark[i]) { gb[2 * g + 1]++; aev[agent[i]]++; }
    }
    HIPCHK(c, hipMemcpyAsync(c->b.accept, acc.data(), n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b.agent, agent, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b.lm, landmark, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b.px, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->b.py, y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_graph_batch, gb.data(), gb.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sb.agent_ev, aev.data(), aev.size() * sizeof(unsigned int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // host vectors above go out of use
    c->b.n = n;
    rc = reserve_graphs_for_batch(c, n);
    if (rc != QS_OK) return rc;
    std::vector<QsGraphDev> before(G), after(G);
    HIPCHK(c, hipMemcpyAsync(before.data(), c->d_graphs, (size_t)G * sizeof(QsGraphDev), hipMemcpyDeviceToHost, c->stream));
    chain_stats_poll(c, false, nullptr);
    HIPCHK(c, qs_launch_slam(c, n, true));
    { int rcs = chain_stats_request(c); if (rcs != QS_OK) return rcs; }
    HIPCHK(c, hipMemcpyAsync(after.data(), c->d_graphs, (size_t)G * sizeof(QsGraphDev), hipMemcpyDeviceToHost, c->stream));
    std::vector<long long> node(n);
    HIPCHK(c, hipMemcpyAsync(node.data(), c->sb.node, n * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = read_pile_flag(c);
    if (rc != QS_OK) return rc;
    c->last_has_poses = false;
    if (closed) memset(closed, 0, n);
    if (corr2) for (size_t i = 0; i < 2 * n; i++) corr2[i] = 0.0;
    if (!closed && !corr2) return QS_OK;
    for (int g = 0; g < G; g++) {
        const long long k0 = before[g].n_cls, k1 = after[g].n_cls;
        if (k1 <= k0) continue;
        const size_t m = (size_t)(k1 - k0);
        std::vector<long long> cn(m); std::vector<double> dx(m), dy(m);
        HIPCHK(c, hipMemcpy(cn.data(), after[g].cl_node_idx + k0, m * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(dx.data(), after[g].cl_dx + k0, m * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(dy.data(), after[g].cl_dy + k0, m * 8, hipMemcpyDeviceToHost));
        size_t q = 0;     // closures and poses of one graph are both in node order
        for (size_t i = 0; i < n && q < m; i++) {
            if (!acc[i] || (agent[i] - 1) / c->bots_per_graph != g) continue;
            if (node[i] == cn[q]) {
                if (closed) closed[i] = 1;
                if (corr2) { corr2[2 * i] = dx[q]; corr2[2 * i + 1] = dy[q]; }
                q++;
            }
        }
    }
    return QS_OK;
}

extern "C" int qs_drift(qs_ctx *c, int32_t bot, double out[2])
{
    ARGCHK(c, c != nullptr && out);
    if (bot < 1 || bot > c->cfg.max_agent) return qs_fail(c, QS_E_RANGE, "qs_drift: bot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_drift + 2 * bot, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

// ---- ZONE ---------------------------------------------------------------------------------------
extern "C" int qs_zone(qs_ctx *c, int32_t bot, double out[4], int32_t *valid)
{
    ARGCHK(c, c != nullptr && out && valid);
    if (bot < 1 || bot > c->cfg.max_agent) return qs_fail(c, QS_E_RANGE, "qs_zone: bot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long z[4];
    HIPCHK(c, hipMemcpyAsync(z, c->d_zone + 4 * bot, sizeof z, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *valid = z[0] != QS_ORD_MIN_IDENT;                      // compute_bounding_box: None if no points  :704
    for (int i = 0; i < 4; i++) out[i] = *valid ? qs_double_from_ord(z[i]) : NAN;
    return QS_OK;
}

extern "C" int qs_zone_packet(qs_ctx *c, int32_t bot, int32_t online, uint8_t out[QS_ZONE_SIZE])
{
    ARGCHK(c, c != nullptr && out);
    float f[4] = {999.0f, 999.0f, -999.0f, -999.0f};       // send_zone_to_bot(None)  :679-681
    if (online) {
        double z[4]; int32_t valid = 0;
        int rc = qs_zone(c, bot, z, &valid);
        if (rc != QS_OK) return rc;
        if (valid) for (int i = 0; i < 4; i++) f[i] = (float)z[i];   // struct.pack('<4sffff')  :683-684
    }
    memcpy(out, "ZONE", 4);
    memcpy(out + 4, f, 16);
    return QS_OK;
}

// ---- fuse / merge ---------------------------------------------------------------------------------
extern "C" int qs_fuse_buffers_range(qs_ctx *c, const void *const *stamps_dev, const void *const *counts_dev, size_t n,
                                     size_t cell_offset, size_t n_cells, int32_t counts_into_fused)
{
    ARGCHK(c, c != nullptr);
    if (n == 0 || n_cells == 0) return QS_OK;
    ARGCHK(c, stamps_dev != nullptr || counts_dev != nullptr);
    ARGCHK(c, cell_offset % 4 == 0 && n_cells % 4 == 0 && cell_offset + n_cells <= c->cells);
    if (counts_dev && counts_into_fused && !c->d_counts_fused)
        return qs_fail(c, QS_E_INVAL, "qs_fuse_buffers_range: no fused counter snapshot (call qs_fused_counts first)");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long *dc = c->cfg.enable_counts ? (counts_into_fused ? c->d_counts_fused : c->d_counts) : nullptr;
    HIPCHK(c, qs_launch_fuse(c, (const unsigned int *const *)stamps_dev, (const unsigned long long *const *)counts_dev, n,
                             cell_offset, n_cells, dc));
    if (!counts_into_fused) HIPCHK(c, qs_launch_sf_mark_range(c, cell_offset, n_cells));   // a local fold writes the grid too
    return QS_OK;
}

extern "C" int qs_fuse_buffers(qs_ctx *c, const void *const *stamps_dev, const void *const *counts_dev, size_t n)
{
    ARGCHK(c, c != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(c, stamps_dev != nullptr);
    return qs_fuse_buffers_range(c, stamps_dev, counts_dev, n, 0, c->cells, 0);
}

// Counters are per-context sums of this context's own writes.  A collective must not add into them (a second
// all-reduce would add the peers' totals again): it sums a SNAPSHOT.  This call copies the local counters into the
// context's second buffer (allocated on first use) on the context's stream and returns it; the caller sums it over the
// ranks in place.  qs_counts_source(ctx, 1) points the counter / log-odds views at it.
extern "C" int qs_fused_counts(qs_ctx *c, void **fused_dev, size_t *bytes)
{
    ARGCHK(c, c != nullptr && fused_dev != nullptr);
    if (!c->d_counts) return qs_fail(c, QS_E_INVAL, "qs_fused_counts: context created with enable_counts = 0");
    if (c->d_dirty) return qs_fail(c, QS_E_STATE, "qs_fused_counts: dirty tracking is on -- the fused counters accumulate the sparse fuse's deltas");
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    const size_t nb = c->cells * sizeof(unsigned long long);
    if (!c->d_counts_fused) HIPCHK(c, hipMalloc((void **)&c->d_counts_fused, nb));
    HIPCHK(c, hipMemcpyAsync(c->d_counts_fused, c->d_counts, nb, hipMemcpyDeviceToDevice, c->stream));
    *fused_dev = c->d_counts_fused;
    if (bytes) *bytes = nb;
    return QS_OK;
}

extern "C" int qs_fused_counts_buffer(qs_ctx *c, void **fused_dev, size_t *bytes)
{
    ARGCHK(c, c != nullptr && fused_dev != nullptr);
    *fused_dev = c->d_counts_fused;
    if (bytes) *bytes = c->d_counts_fused ? c->cells * sizeof(unsigned long long) : 0;
    return QS_OK;
}

extern "C" int qs_counts_source(qs_ctx *c, int32_t fused)
{
    ARGCHK(c, c != nullptr);
    if (fused && !c->d_counts_fused) return qs_fail(c, QS_E_INVAL, "qs_counts_source: no fused snapshot yet (qs_fused_counts)");
    c->counts_view_fused = fused != 0;
    return QS_OK;
}

// ---- sparse fuse (sparse_fuse.hip; protocol in include/quasar_slam.h) --------------------------------------------------
extern "C" int qs_dirty_tracking(qs_ctx *c, int32_t enable)
{
    ARGCHK(c, c != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!enable) {
        hipFree(c->d_dirty); c->d_dirty = nullptr; c->geom.dirty = nullptr; c->geom.dirty_pitch = 0;
        c->sf_state = 0;
        return QS_OK;
    }
    if (c->d_dirty) return QS_OK;
    if (c->dirty_since_fuse) return qs_fail(c, QS_E_STATE, "qs_dirty_tracking: the grid has unfused writes (enable it after qs_create / qs_reset / a fuse)");
    c->blocks_x = (c->cfg.size + QS_DIRTY_BLOCK_W - 1) / QS_DIRTY_BLOCK_W;
    c->blocks_y = (c->cfg.size + QS_DIRTY_BLOCK_H - 1) / QS_DIRTY_BLOCK_H;
    const int pitch = (c->blocks_x + 31) / 32;
    c->dirty_words = (size_t)c->blocks_y * pitch;
    HIPCHK(c, hipMalloc((void **)&c->d_dirty, c->dirty_words * sizeof(unsigned int)));
    HIPCHK(c, hipMemsetAsync(c->d_dirty, 0, c->dirty_words * sizeof(unsigned int), c->stream));
    if (c->d_counts) {
        const size_t nb = c->cells * sizeof(unsigned long long);
        if (!c->d_counts_sent) HIPCHK(c, hipMalloc((void **)&c->d_counts_sent, nb));
        // the fused counters accumulate deltas from here on: they start as "nothing sent", the local counters as all delta
        HIPCHK(c, hipMemsetAsync(c->d_counts_sent, 0, nb, c->stream));
        if (!c->d_counts_fused) HIPCHK(c, hipMalloc((void **)&c->d_counts_fused, nb));
        HIPCHK(c, hipMemsetAsync(c->d_counts_fused, 0, nb, c->stream));
        // counters written before tracking was switched on have no dirty bit: everything is marked once
    }
    c->geom.dirty = c->d_dirty; c->geom.dirty_pitch = pitch;
    if (c->next_seq != 0) HIPCHK(c, qs_launch_sf_mark_range(c, 0, c->cells));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

extern "C" int qs_dirty_blocks(qs_ctx *c, size_t *n_blocks, size_t *block_cells)
{
    ARGCHK(c, c != nullptr && n_blocks != nullptr);
    if (!c->d_dirty) return qs_fail(c, QS_E_STATE, "qs_dirty_blocks: dirty tracking is off (qs_dirty_tracking)");
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    int rc = io_reserve(c, sizeof(unsigned long long));
    if (rc != QS_OK) return rc;
    unsigned long long v = 0;
    HIPCHK(c, qs_launch_sf_popcount(c, (unsigned long long *)c->d_io_ws));
    HIPCHK(c, hipMemcpyAsync(&v, c->d_io_ws, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_blocks = (size_t)v;
    if (block_cells) *block_cells = (size_t)QS_DIRTY_BLOCK_W * QS_DIRTY_BLOCK_H;
    return QS_OK;
}

extern "C" int qs_sparse_fuse_begin(qs_ctx *c, int32_t world, int32_t rank, void **bitmaps_dev, size_t *bitmap_bytes)
{
    ARGCHK(c, c != nullptr && bitmaps_dev != nullptr && bitmap_bytes != nullptr);
    ARGCHK(c, world >= 1 && world <= QS_SPARSE_MAX_WORLD && rank >= 0 && rank < world);
    if (!c->d_dirty) return qs_fail(c, QS_E_STATE, "qs_sparse_fuse_begin: dirty tracking is off (qs_dirty_tracking)");
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    if (world != c->sf_world) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_realloc(&c->d_sf_bitmaps, (size_t)world * c->dirty_words));
        HIPCHK(c, dev_realloc(&c->d_sf_lists, (size_t)world * c->dirty_words * 32));
        HIPCHK(c, dev_realloc(&c->d_sf_counts, (size_t)world));
        c->sf_world = world;
        c->sf_n.assign(world, 0); c->sf_off.assign((size_t)world + 1, 0);
    }
    c->sf_rank = rank;
    const size_t nb = c->dirty_words * sizeof(unsigned int);
    HIPCHK(c, hipMemcpyAsync(c->d_sf_bitmaps + (size_t)rank * c->dirty_words, c->d_dirty, nb, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dirty, 0, nb, c->stream));
    *bitmaps_dev = c->d_sf_bitmaps; *bitmap_bytes = nb;
    c->sf_state = 1;
    return QS_OK;
}

extern "C" int qs_sparse_fuse_plan(qs_ctx *c, uint32_t *n_blocks, size_t *offsets, void **payload_dev, size_t *block_bytes)
{
    ARGCHK(c, c != nullptr && n_blocks != nullptr && offsets != nullptr && payload_dev != nullptr);
    if (c->sf_state != 1) return qs_fail(c, QS_E_STATE, "qs_sparse_fuse_plan: call qs_sparse_fuse_begin (and all-gather the bitmaps) first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, qs_launch_sf_lists(c));
    HIPCHK(c, hipMemcpyAsync(c->sf_n.data(), c->d_sf_counts, (size_t)c->sf_world * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t bb = qs_sf_block_bytes(c);
    size_t run = 0;
    for (int s = 0; s < c->sf_world; s++) { c->sf_off[s] = run; run += (size_t)c->sf_n[s] * bb; n_blocks[s] = c->sf_n[s]; offsets[s] = c->sf_off[s]; }
    c->sf_off[c->sf_world] = run; offsets[c->sf_world] = run;
    if (run > c->sf_payload_bytes) {
        size_t cap = c->sf_payload_bytes ? c->sf_payload_bytes : ((size_t)1 << 20);
        while (cap < run) cap *= 2;
        HIPCHK(c, dev_realloc(&c->d_sf_payload, cap));
        c->sf_payload_bytes = cap;
    }
    HIPCHK(c, qs_launch_sf_pack(c, c->sf_n[c->sf_rank], c->d_sf_payload + c->sf_off[c->sf_rank]));
    *payload_dev = c->d_sf_payload;
    if (block_bytes) *block_bytes = bb;
    c->sf_state = 2;
    return QS_OK;
}

extern "C" int qs_sparse_fuse_apply(qs_ctx *c)
{
    ARGCHK(c, c != nullptr);
    if (c->sf_state != 2) return qs_fail(c, QS_E_STATE, "qs_sparse_fuse_apply: call qs_sparse_fuse_plan (and exchange the segments) first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, qs_launch_sf_apply(c));
    c->sf_state = 0;
    c->dirty_since_fuse = false;
    if (c->d_counts) c->counts_view_fused = true;
    return QS_OK;
}

extern "C" int qs_fuse(qs_ctx *dst, qs_ctx *const *srcs, size_t n)
{
    ARGCHK(dst, dst != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(dst, srcs != nullptr);
    std::vector<const void *> st(n), ct(n);
    bool counts = dst->d_counts != nullptr;
    for (size_t i = 0; i < n; i++) {
        qs_ctx *s = srcs[i];
        if (!s || s->device != dst->device || s->cfg.size != dst->cfg.size || s->cfg.res != dst->cfg.res ||
            s->cfg.ox != dst->cfg.ox || s->cfg.oy != dst->cfg.oy)
            return qs_fail(dst, QS_E_INVAL, "qs_fuse: source grids must share device and geometry with dst");
        if (s->epoch_base != dst->epoch_base || s->n_rebases != dst->n_rebases)
            return qs_fail(dst, QS_E_INVAL, "qs_fuse: source and destination are in different stamp epochs");
        { HIPCHK(dst, hipSetDevice(s->device)); int rcs = flush_edge_rays(s); if (rcs != QS_OK) return qs_fail(dst, rcs, s->err.c_str()); }
        HIPCHK(dst, hipStreamSynchronize(s->stream));
        st[i] = s->d_stamps; ct[i] = s->d_counts;
        if (!s->d_counts) counts = false;
    }
    int rc = qs_fuse_buffers(dst, st.data(), counts ? ct.data() : nullptr, n);
    if (rc != QS_OK) return rc;
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    return QS_OK;
}

extern "C" int qs_grid_to_pcd(qs_ctx *c, const int8_t *grid, int32_t h, int32_t w, double res, double ox, double oy,
                              double *xy, size_t cap, size_t *n_out)
{
    ARGCHK(c, c != nullptr && grid != nullptr && n_out != nullptr && h > 0 && w > 0);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t cells = (size_t)h * w, n_chunks = (cells + 1023) / 1024;
    signed char *dg = nullptr; unsigned int *dchunk = nullptr; unsigned long long *dcount = nullptr; double *dxy = nullptr;
    hipError_t e = hipMalloc((void **)&dg, cells);
    if (e == hipSuccess) e = hipMalloc((void **)&dchunk, n_chunks * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void **)&dcount, sizeof(unsigned long long));
    unsigned long long total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(dg, grid, cells, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_grid_to_pcd(c, dg, h, w, res, ox, oy, nullptr, 0, dcount, dchunk);
    if (e == hipSuccess) e = hipMemcpyAsync(&total, dcount, sizeof total, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    *n_out = (size_t)total;
    if (e == hipSuccess && xy && total > 0) {
        const size_t m = total < cap ? (size_t)total : cap;
        e = hipMalloc((void **)&dxy, 2 * (size_t)total * sizeof(double));
        if (e == hipSuccess) e = qs_launch_grid_to_pcd(c, dg, h, w, res, ox, oy, dxy, (size_t)total, dcount, dchunk);
        if (e == hipSuccess) e = hipMemcpyAsync(xy, dxy, 2 * m * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    hipFree(dg); hipFree(dchunk); hipFree(dcount); hipFree(dxy);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_grid_to_pcd", e);
    return QS_OK;
}

extern "C" int qs_rasterise(qs_ctx *c, const double *xy, size_t n, double res, int32_t dims[2], double origin[2], int8_t *grid)
{
    ARGCHK(c, c != nullptr && dims && origin && res > 0);
    if (n == 0) { dims[0] = dims[1] = 0; return QS_OK; }      // publish_global_map returns early  :88-93
    ARGCHK(c, xy != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    double *dxy = nullptr; unsigned long long *dbox = nullptr; signed char *dg = nullptr;
    unsigned long long box[4] = {QS_ORD_MIN_IDENT, QS_ORD_MIN_IDENT, QS_ORD_MAX_IDENT, QS_ORD_MAX_IDENT};
    hipError_t e = hipMalloc((void **)&dxy, 2 * n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&dbox, sizeof box);
    if (e == hipSuccess) e = hipMemcpyAsync(dxy, xy, 2 * n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dbox, box, sizeof box, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_bbox(c, dxy, n, dbox);
    if (e == hipSuccess) e = hipMemcpyAsync(box, dbox, sizeof box, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    int rc = QS_OK;
    if (e == hipSuccess) {
        const double mnx = qs_double_from_ord(box[0]), mny = qs_double_from_ord(box[1]);
        const double mxx = qs_double_from_ord(box[2]), mxy = qs_double_from_ord(box[3]);
        const double wd = ceil((mxx - mnx) / res), hd = ceil((mxy - mny) / res);     // :103-104
        if (!(wd >= 0 && wd < 65536 && hd >= 0 && hd < 65536)) rc = qs_fail(c, QS_E_RANGE, "qs_rasterise: canvas too large");
        else {
            const int w = (int)wd + 1, h = (int)hd + 1;
            dims[0] = h; dims[1] = w; origin[0] = mnx; origin[1] = mny;
            if (grid) {
                e = hipMalloc((void **)&dg, (size_t)h * w);
                if (e == hipSuccess) e = qs_launch_rasterise(c, dxy, n, res, mnx, mny, h, w, dg);
                if (e == hipSuccess) e = hipMemcpyAsync(grid, dg, (size_t)h * w, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            }
        }
    }
    hipFree(dxy); hipFree(dbox); hipFree(dg);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_rasterise", e);
    return rc;
}

// ---- ICP / voxel down-sample (map_merger.py:45-60; Open3D semantics, parity unpinned) ------------------
// The correspondence search (nearest target of every source point) has two implementations with identical results:
// the scalar fp64 brute force and the MFMA-screened one (icp.hip).  mode 0 = auto (MFMA from 64 targets up).
struct NnPlan { double cx, cy, t2max; size_t n_pad; double *planes; bool mfma; int *part_j; double *part_d2, *thr_seed; };
static void nn_free(NnPlan &pl) { hipFree(pl.planes); hipFree(pl.part_j); hipFree(pl.part_d2); hipFree(pl.thr_seed); pl.planes = nullptr; }

static hipError_t nn_prepare(qs_ctx *c, const double *dst_xy, size_t n_dst, const double2 *d_dst, int mode, NnPlan &pl, size_t n_src)
{
    pl = NnPlan{0, 0, 0, 0, nullptr, false, nullptr, nullptr, nullptr};
    pl.mfma = mode == 2 || (mode == 0 && n_dst >= 64);
    if (!pl.mfma) return hipSuccess;
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (size_t j = 0; j < n_dst; j++) {
        const double x = dst_xy[2 * j], y = dst_xy[2 * j + 1];
        if (isfinite(x)) { mnx = x < mnx ? x : mnx; mxx = x > mxx ? x : mxx; }
        if (isfinite(y)) { mny = y < mny ? y : mny; mxy = y > mxy ? y : mxy; }
    }
    pl.cx = isfinite(mnx) ? 0.5 * (mnx + mxx) : 0.0; pl.cy = isfinite(mny) ? 0.5 * (mny + mxy) : 0.0;
    const double hx = isfinite(mnx) ? mxx - pl.cx : 0.0, hy = isfinite(mny) ? mxy - pl.cy : 0.0;
    pl.t2max = 1.0001 * (hx * hx + hy * hy) + 1e-300;          // >= every finite target's centred squared norm
    pl.n_pad = (n_dst + 15) / 16 * 16;
    hipError_t e = hipMalloc((void **)&pl.planes, 3 * pl.n_pad * sizeof(double));
    if (e == hipSuccess) e = qs_launch_icp_prep(c, d_dst, n_dst, pl.n_pad, pl.cx, pl.cy, pl.planes);
    // per-part results and the sources' threshold seeds (the targets are cut into parts: icp.hip)
    unsigned int groups, parts, cpp;
    qs_icp_nn_plan(n_src, pl.n_pad, &groups, &parts, &cpp);
    if (e == hipSuccess) e = hipMalloc((void **)&pl.part_j, (size_t)parts * n_src * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&pl.part_d2, (size_t)parts * n_src * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&pl.thr_seed, n_src * sizeof(double));
    return e;
}

static hipError_t nn_run(qs_ctx *c, const NnPlan &pl, const double2 *d_src, size_t n_src, const double2 *d_dst, size_t n_dst,
                         double max_d2, int *d_corr, double *d_d2)
{
    if (pl.mfma) return qs_launch_icp_nn_mfma(c, d_src, n_src, d_dst, n_dst, pl.planes, pl.n_pad, pl.cx, pl.cy, pl.t2max, max_d2, d_corr, d_d2,
                                              pl.part_j, pl.part_d2, pl.thr_seed);
    return qs_launch_icp_nn(c, d_src, n_src, d_dst, n_dst, max_d2, d_corr, d_d2);
}

// Build extension (the correspondence step of registration_icp on its own; used by the tests and tools/bench_icp_nn.py):
// corr[i] = index of the target nearest to source i if closer than max_dist, else -1 (ties: lowest index); d2[i] its squared
// distance (0 without a correspondence).  ms (may be NULL): HIP-event time of {the search kernel, the operand preparation}.
extern "C" int qs_nn_search(qs_ctx *c, const double *src_xy, size_t n_src, const double *dst_xy, size_t n_dst, double max_dist,
                            int32_t mode, int32_t *corr, double *d2, float ms[2])
{
    ARGCHK(c, c != nullptr && corr != nullptr && d2 != nullptr);
    ARGCHK(c, n_src > 0 && n_dst > 0 && src_xy && dst_xy && max_dist > 0 && mode >= 0 && mode <= 2);
    ARGCHK(c, n_dst < (size_t)1 << 31);
    HIPCHK(c, hipSetDevice(c->device));
    double2 *d_src = nullptr, *d_dst = nullptr; int *d_corr = nullptr; double *d_d2 = nullptr;
    NnPlan pl{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipMalloc((void **)&d_src, n_src * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&d_dst, n_dst * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&d_corr, n_src * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_d2, n_src * sizeof(double));
    for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreate(&ev[k]);
    if (e == hipSuccess) e = hipMemcpyAsync(d_src, src_xy, n_src * sizeof(double2), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dst, dst_xy, n_dst * sizeof(double2), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipEventRecord(ev[0], c->stream);
    if (e == hipSuccess) e = nn_prepare(c, dst_xy, n_dst, d_dst, mode, pl, n_src);
    if (e == hipSuccess) e = hipEventRecord(ev[1], c->stream);
    if (e == hipSuccess) e = nn_run(c, pl, d_src, n_src, d_dst, n_dst, max_dist * max_dist, d_corr, d_d2);      // warm (code load, caches)
    if (e == hipSuccess) e = hipEventRecord(ev[2], c->stream);
    if (e == hipSuccess) e = nn_run(c, pl, d_src, n_src, d_dst, n_dst, max_dist * max_dist, d_corr, d_d2);
    if (e == hipSuccess) e = hipEventRecord(ev[3], c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(corr, d_corr, n_src * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, d_d2, n_src * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && ms) { hipEventElapsedTime(&ms[0], ev[2], ev[3]); hipEventElapsedTime(&ms[1], ev[0], ev[1]); }
    for (int k = 0; k < 4; k++) if (ev[k]) hipEventDestroy(ev[k]);
    hipFree(d_src); hipFree(d_dst); hipFree(d_corr); hipFree(d_d2); nn_free(pl);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_nn_search", e);
    return QS_OK;
}

extern "C" int qs_icp(qs_ctx *c, const double *src_xy, size_t n_src, const double *dst_xy, size_t n_dst, double max_dist,
                      int32_t max_iter, double rel_fitness, double rel_rmse, double T[9], double *fitness, double *rmse,
                      int32_t *iters)
{
    ARGCHK(c, c != nullptr && T != nullptr && fitness != nullptr && rmse != nullptr);
    ARGCHK(c, n_src > 0 && n_dst > 0 && src_xy && dst_xy && max_dist > 0 && max_iter >= 0);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = (n_src + 255) / 256;
    double2 *d_src = nullptr, *d_dst = nullptr; int *d_corr = nullptr; double *d_d2 = nullptr, *d_part = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_src, n_src * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&d_dst, n_dst * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&d_corr, n_src * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_d2, n_src * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_part, nb * 6 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, 6 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_src, src_xy, n_src * sizeof(double2), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dst, dst_xy, n_dst * sizeof(double2), hipMemcpyHostToDevice, c->stream);
    NnPlan pl{};
    if (e == hipSuccess) e = nn_prepare(c, dst_xy, n_dst, d_dst, 0, pl, n_src);     // the targets do not move: operands once per registration
    double tc = 1.0, ts = 0.0, tx = 0.0, ty = 0.0;          // accumulated transform
    double out[6] = {0};
    const double zero4[4] = {0, 0, 0, 0};
    auto evaluate = [&](double &fit, double &rm) -> hipError_t {
        hipError_t ee = nn_run(c, pl, d_src, n_src, d_dst, n_dst, max_dist * max_dist, d_corr, d_d2);
        if (ee == hipSuccess) ee = qs_launch_icp_sums(c, d_src, n_src, d_dst, d_corr, d_d2, 0, zero4, d_part, d_out);
        if (ee == hipSuccess) ee = hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, c->stream);
        if (ee == hipSuccess) ee = hipStreamSynchronize(c->stream);
        fit = out[0] / (double)n_src;
        rm = out[0] > 0 ? sqrt(out[1] / out[0]) : 0.0;
        return ee;
    };
    double fit = 0, rm = 0;
    int it = 0;
    if (e == hipSuccess) e = evaluate(fit, rm);
    for (; e == hipSuccess && it < max_iter; it++) {
        double uc = 1.0, us = 0.0, ux = 0.0, uy = 0.0;       // ComputeTransformation: identity without correspondences
        if (out[0] > 0) {
            const double nn = out[0];
            const double means[4] = {out[2] / nn, out[3] / nn, out[4] / nn, out[5] / nn};
            double o2[6];
            e = qs_launch_icp_sums(c, d_src, n_src, d_dst, d_corr, d_d2, 1, means, d_part, d_out);
            if (e == hipSuccess) e = hipMemcpyAsync(o2, d_out, sizeof o2, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) break;
            const double theta = atan2(o2[1], o2[0]);
            uc = cos(theta); us = sin(theta);
            ux = means[2] - (uc * means[0] - us * means[1]);
            uy = means[3] - (us * means[0] + uc * means[1]);
        }
        // transformation = update * transformation
        const double nc = uc * tc - us * ts, ns = us * tc + uc * ts;
        const double nx = uc * tx - us * ty + ux, ny = us * tx + uc * ty + uy;
        tc = nc; ts = ns; tx = nx; ty = ny;
        e = qs_launch_icp_transform(c, d_src, n_src, uc, us, ux, uy);
        const double bfit = fit, brm = rm;
        if (e == hipSuccess) e = evaluate(fit, rm);
        if (e == hipSuccess && fabs(bfit - fit) < rel_fitness && fabs(brm - rm) < rel_rmse) { it++; break; }
    }
    hipFree(d_src); hipFree(d_dst); hipFree(d_corr); hipFree(d_d2); hipFree(d_part); hipFree(d_out); nn_free(pl);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_icp", e);
    T[0] = tc; T[1] = -ts; T[2] = tx; T[3] = ts; T[4] = tc; T[5] = ty; T[6] = 0; T[7] = 0; T[8] = 1;
    *fitness = fit; *rmse = rm;
    if (iters) *iters = it;
    return QS_OK;
}

// Diagnostic: measured fp64 MFMA rate of this GPU (dense v_mfma_f64_16x16x4_f64, every CU, 2 waves per SIMD), TFLOP/s.
extern "C" int qs_diag_mfma_f64_rate(qs_ctx *c, double *tflops)
{
    ARGCHK(c, c != nullptr && tflops != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    double *sink = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    const int blocks = 256 * 2, iters = 20000;             // 2 workgroups of 4 waves per CU
    hipError_t e = hipMalloc((void **)&sink, 8);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = qs_launch_mfma_f64_rate(c, blocks, 1000, sink);
    if (e == hipSuccess) e = hipEventRecord(a, c->stream);
    if (e == hipSuccess) e = qs_launch_mfma_f64_rate(c, blocks, iters, sink);
    if (e == hipSuccess) e = hipEventRecord(b, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (a) hipEventDestroy(a);
    if (b) hipEventDestroy(b);
    hipFree(sink);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_diag_mfma_f64_rate", e);
    const double flops = (double)blocks * 4 /* waves */ * iters * 4 /* MFMAs */ * (2.0 * 16 * 16 * 4);
    *tflops = flops / (ms * 1e-3) / 1e12;
    return QS_OK;
}

// Diagnostic: measured latencies of the primitives of one loop-closure decision (diag.hip), shader-clock cycles.
extern "C" int qs_diag_latencies(qs_ctx *c, double out[QS_DIAG_LAT_N])
{
    ARGCHK(c, c != nullptr && out != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned int n2 = 1u << 18, n1 = 1u << 11;            // 1 MiB: past the 32 KiB L1, inside the 4 MiB L2; 8 KiB: inside L1
    std::vector<unsigned int> h2(n2), h1(n1);
    for (unsigned int k = 0; k < n2; k++) h2[k] = (k * 1664525u + 1013904223u) & (n2 - 1);     // full-period LCG: one cycle through all entries
    for (unsigned int k = 0; k < n1; k++) h1[k] = (k * 1664525u + 1013904223u) & (n1 - 1);
    unsigned int *d2 = nullptr, *d1 = nullptr; double *d_out = nullptr;
    double h_out[16] = {0};
    hipError_t e = hipMalloc((void **)&d2, n2 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d1, n1 * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof h_out);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, h2.data(), n2 * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d1, h1.data(), n1 * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof h_out, c->stream);
    if (e == hipSuccess) e = qs_launch_diag_latencies(c, d2, d1, d_out);          // (warm: code load)
    if (e == hipSuccess) e = qs_launch_diag_latencies(c, d2, d1, d_out);
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d2); hipFree(d1); hipFree(d_out);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_diag_latencies", e);
    for (int i = 0; i < QS_DIAG_LAT_N; i++) out[i] = h_out[i];
    return QS_OK;
}

extern "C" int qs_voxel_downsample(qs_ctx *c, const double *xy, size_t n, double voxel, double *out_xy, size_t cap, size_t *n_out)
{
    ARGCHK(c, c != nullptr && n_out != nullptr && voxel > 0);
    *n_out = 0;
    if (n == 0) return QS_OK;
    ARGCHK(c, xy != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    double mnx = xy[0], mny = xy[1];
    for (size_t i = 1; i < n; i++) { if (xy[2 * i] < mnx) mnx = xy[2 * i]; if (xy[2 * i + 1] < mny) mny = xy[2 * i + 1]; }
    mnx -= voxel * 0.5; mny -= voxel * 0.5;                 // voxel_min_bound = min_bound - voxel_size / 2
    double2 *d = nullptr; unsigned long long *dk = nullptr;
    std::vector<unsigned long long> keys(n);
    hipError_t e = hipMalloc((void **)&d, n * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&dk, n * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(d, xy, n * sizeof(double2), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_voxel_keys(c, d, n, mnx, mny, voxel, dk);
    if (e == hipSuccess) e = hipMemcpyAsync(keys.data(), dk, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d); hipFree(dk);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_voxel_downsample", e);
    // group by voxel (ascending key), average in input order: a handful of points per ROS callback
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return keys[a] < keys[b]; });
    size_t k = 0;
    for (size_t p = 0; p < n;) {
        size_t q = p; double sx = 0, sy = 0;
        while (q < n && keys[order[q]] == keys[order[p]]) { sx += xy[2 * order[q]]; sy += xy[2 * order[q] + 1]; q++; }
        if (out_xy && k < cap) { out_xy[2 * k] = sx / (double)(q - p); out_xy[2 * k + 1] = sy / (double)(q - p); }
        k++; p = q;
    }
    *n_out = k;
    return QS_OK;
}

// ---- frontiers ------------------------------------------------------------------------------------
static int frontier_run(qs_ctx *c, int mode, int32_t min_cluster, int32_t *xy, int64_t *stats5, size_t cap, size_t *n_out)
{
    ARGCHK(c, c != nullptr && n_out != nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    if (!c->d_frontier_ws) HIPCHK(c, hipMalloc(&c->d_frontier_ws, qs_frontier_workspace_bytes(c)));
    void *ws = c->d_frontier_ws;
    HIPCHK(c, qs_launch_frontier_label(c, ws, mode != 0));
    HIPCHK(c, qs_launch_frontier_compact(c, ws, mode == 2 ? 0 : mode, 0, nullptr, nullptr, 0));
    unsigned long long total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, qs_frontier_total_ptr(c, ws), sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mode == 0) {
        *n_out = (size_t)total;
        if (!xy || total == 0) return QS_OK;
        int *d = nullptr;
        HIPCHK(c, hipMalloc((void **)&d, 2 * (size_t)total * sizeof(int)));
        hipError_t e = qs_launch_frontier_compact(c, ws, 0, 1, d, nullptr, (size_t)total);
        const size_t m = total < cap ? (size_t)total : cap;
        if (e == hipSuccess) e = hipMemcpyAsync(xy, d, 2 * m * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(d);
        if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_frontier_cells", e);
        return QS_OK;
    }
    if (mode == 2) {
        // every frontier cell with the first cell (row-major) of its 4-connected cluster: gx, gy, root linear index
        *n_out = (size_t)total;
        if (!xy || total == 0) return QS_OK;
        int *d = nullptr;
        HIPCHK(c, hipMalloc((void **)&d, 3 * (size_t)total * sizeof(int)));
        hipError_t e = qs_launch_frontier_compact(c, ws, 2, 1, d, nullptr, (size_t)total);
        const size_t m = total < cap ? (size_t)total : cap;
        if (e == hipSuccess) e = hipMemcpyAsync(xy, d, 3 * m * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(d);
        if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_frontier_members", e);
        return QS_OK;
    }
    // clusters: all components come back in first-cell order; the size filter keeps that order (:228-229)
    std::vector<long long> all(5 * (size_t)total);
    if (total) {
        long long *d = nullptr;
        HIPCHK(c, hipMalloc((void **)&d, 5 * (size_t)total * sizeof(long long)));
        hipError_t e = qs_launch_frontier_compact(c, ws, 1, 1, nullptr, d, (size_t)total);
        if (e == hipSuccess) e = hipMemcpyAsync(all.data(), d, all.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(d);
        if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_frontier_clusters", e);
    }
    size_t k = 0;
    for (size_t i = 0; i < (size_t)total; i++) {
        if (all[5 * i] < min_cluster) continue;
        if (stats5 && k < cap) memcpy(stats5 + 5 * k, &all[5 * i], 5 * sizeof(long long));
        k++;
    }
    *n_out = k;
    return QS_OK;
}

extern "C" int qs_frontier_cells(qs_ctx *c, int32_t *xy, size_t cap, size_t *n_out)
{ return frontier_run(c, 0, 0, xy, nullptr, cap, n_out); }

extern "C" int qs_frontier_members(qs_ctx *c, int32_t *xy_root, size_t cap, size_t *n_out)
{ return frontier_run(c, 2, 0, xy_root, nullptr, cap, n_out); }

extern "C" int qs_frontier_clusters(qs_ctx *c, int32_t min_cluster, int64_t *stats5, size_t cap, size_t *n_out)
{ return frontier_run(c, 1, min_cluster, nullptr, stats5, cap, n_out); }

// ---- EKF --------------------------------------------------------------------------------------------
extern "C" int qs_ekf_init(qs_ctx *c, int32_t bot, double t, const double x0[6])
{
    ARGCHK(c, c != nullptr);
    if (bot < 1 || bot > c->cfg.max_agent) return qs_fail(c, QS_E_RANGE, "qs_ekf_init: bot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    double f[44] = {0};
    for (int i = 0; i < 6; i++) { f[i] = x0 ? x0[i] : 0.0; f[6 + 7 * i] = 1.0; }     // x0, P = I  ekf.cpp:5-19
    f[42] = t; f[43] = 1.0;
    HIPCHK(c, hipMemcpyAsync(c->d_ekf + (size_t)bot * 44, f, sizeof f, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

extern "C" int qs_ekf_step(qs_ctx *c, const int32_t *bot_ids, const double *omega_m, const double *t, const double *z_v,
                           const double *z_omega, size_t n, int32_t do_update)
{
    ARGCHK(c, c != nullptr);
    if (n == 0) return QS_OK;
    ARGCHK(c, bot_ids && omega_m && t && (!do_update || (z_v && z_omega)));
    HIPCHK(c, hipSetDevice(c->device));
    int *db = nullptr; double *dd = nullptr;
    hipError_t e = hipMalloc((void **)&db, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&dd, 4 * n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(db, bot_ids, n * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, omega_m, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dd + n, t, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && do_update) e = hipMemcpyAsync(dd + 2 * n, z_v, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && do_update) e = hipMemcpyAsync(dd + 3 * n, z_omega, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = qs_launch_ekf_step(c, db, dd, dd + n, dd + 2 * n, dd + 3 * n, n, do_update);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(db); hipFree(dd);
    if (e != hipSuccess) return qs_fail(c, QS_E_HIP, "qs_ekf_step", e);
    return QS_OK;
}

extern "C" int qs_ekf_state(qs_ctx *c, int32_t bot, double x[6], double P[36])
{
    ARGCHK(c, c != nullptr);
    if (bot < 1 || bot > c->cfg.max_agent) return qs_fail(c, QS_E_RANGE, "qs_ekf_state: bot out of range");
    HIPCHK(c, hipSetDevice(c->device));
    double f[44];
    HIPCHK(c, hipMemcpyAsync(f, c->d_ekf + (size_t)bot * 44, sizeof f, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (x) memcpy(x, f, 6 * sizeof(double));
    if (P) memcpy(P, f + 6, 36 * sizeof(double));
    return QS_OK;
}

extern "C" int qs_counters(qs_ctx *c, uint64_t out[QS_CNT_N])
{
    ARGCHK(c, c != nullptr && out);
    HIPCHK(c, hipSetDevice(c->device));
    FLUSHCHK(c);
    unsigned long long v[QS_CNT_N];
    HIPCHK(c, hipMemcpyAsync(v, c->d_counters, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < QS_CNT_N; i++) out[i] = v[i];
    out[QS_CNT_REBASES] = c->n_rebases;
    out[QS_CNT_EDGE_RAYS] = c->edge_rays_total;
    out[QS_CNT_EDGE_OVERFLOW] = c->edge_overflow_total;
    return QS_OK;
}
