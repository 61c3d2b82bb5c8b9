/*
 * oracle.c -- CPU restatement of the reference mapper's per-packet hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED for the mapper path (rows A1-A6, A8-A10 of SURVEY.md section 8)
 * against fixtures produced by importing the reference's own Python in the build
 * container (tests/golden/make_golden.py -> the .npz files and kat.json under tests/golden).
 * PARITY UNPINNED for the EKF (A7: reference is Arduino/Eigen C++, not buildable here,
 * no vectors exist) and for map_merger's rasterise (A11: needs rclpy/open3d, absent);
 * those two are restated from the source text and checked with hand-derived cases.
 *
 * All citations are into /root/reference/.  Plain C, fp64, libm cos/sin/sqrt/pow --
 * the same libm CPython's math module calls.  Compile with -O2 -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants: server_nodes/dual_bot_mapper.py:57-66, :92-99 --------------------- */
#define MAX_DIST_M 1.20
#define MIN_DIST_M 0.05
#define CELL_UNKNOWN (-1)
#define CELL_FREE 0
#define CELL_OCCUPIED 100
#define CLOSURE_RADIUS 0.60
#define MIN_POSES_BETWEEN 30
#define CLOSURE_CORRECTION 0.5
#define PACKET_SIZE 42    /* '<4sBfffiIffffB'  dual_bot_mapper.py:41 */
#define PACKET_SIZE_V1 41 /* '<4sBfffiIffff'   dual_bot_mapper.py:45 */

static const double SENSOR_ANGLES_RAD[4] = {
    /* front, left, back, right -- dict order of dual_bot_mapper.py:61-66 */
    0.0, 3.14159265358979323846 / 2, 3.14159265358979323846, -3.14159265358979323846 / 2};

typedef struct { double x, y; int type; long idx; } landmark_t;
typedef struct { long lm_idx, node_idx; double dx, dy; } closure_t;

/* Optional spatial index over self.landmarks (qso_use_index): NOT the reference's algorithm -- the reference scans the whole
 * list (:294) -- but the same answer: cells of edge >= CLOSURE_RADIUS per landmark type, each cell's landmarks in insertion
 * order; the first match in list order is the lowest list index among the first matches of the 3 x 3 cells around the query.
 * It exists so that the CPU baseline can also be quoted with the data structure the GPU path uses (bench.py: cpu_baseline.indexed):
 * what the hardware buys, separated from what the index buys. */
typedef struct { long long key; int *it; int n, cap; int used; } lmcell_t;
typedef struct { lmcell_t *tab; size_t cap, used; } lmindex_t;

typedef struct {
    long n_nodes;                 /* len(self.nodes)            :268 */
    landmark_t *lms; long n_lms, cap_lms;   /* self.landmarks   :269 */
    closure_t *cls; long n_cls, cap_cls;    /* self.closures    :270 */
    lmindex_t ix;
} graph_t;

typedef struct {
    int size; double res, ox, oy;          /* OccupancyGrid ctor :113-119 */
    int8_t *grid;                           /* np.full((size,size), -1, int8), [gy][gx] */
    int32_t *hits, *misses;                 /* build extension: per-cell write counts */
    uint32_t *stamps;                       /* build extension: (ordinal << 1) | occ of the latest write,
                                               ordinal = 4 * arrival index + sensor + 1 (0 = never written) */
    uint64_t cur_seq; int cur_sensor; uint64_t seq_next, seq_stride;
    int max_agent;                          /* accept agent_id 1..max_agent (reference: 2, :842) */
    int bots_per_graph;                     /* bots sharing one PoseGraphSLAM (reference: all) */
    int n_graphs;
    graph_t *graphs;
    double *offset_x;                       /* per bot; bot 2 = --separation  :851-852 */
    double *drift;                          /* per bot (cdx, cdy)  :782 */
    long *last_closure;                     /* per bot, init -MIN_POSES_BETWEEN  :271 */
    double closure_radius, closure_correction; long min_poses_between;   /* the reference's module constants (:99-101);
                                               settable here because the product's qs_config exposes them */
    double *zone;                           /* per bot (minx,miny,maxx,maxy) over hits U path */
    long *zone_n;                           /* per bot number of points folded in */
    long *pkt_count;                        /* per bot  :848 */
    /* per accepted packet log */
    double *pose; long n_pose, cap_pose;    /* (rx, ry, ryaw) after offset+drift */
    int32_t *pose_agent; long *pose_src;
    /* per valid hit log, arrival order */
    double *hit; int32_t *hit_agent_sensor; long n_hit, cap_hit;
    long n_rays, n_cells_written, n_datagrams;
    /* optional per-bot EKF on the build-defined telemetry wiring (see qso_ekf_packet) */
    int ekf_on; double ekf_mpt, cur_time; double *ekf, *ekf_prev;
    /* build extension (qs_config.shard_bots): this mapper is one shard of a deployment that keeps ONE pose graph over
     * all bots: it runs add_pose for every packet but casts rays / keeps zones / runs the EKF for agents own_lo..own_hi */
    int own_lo, own_hi;
    int use_index;                          /* closure search through the spatial index instead of the list scan (same result) */
} mapper_t;

/* ---- OccupancyGrid ---------------------------------------------------------------- */

/* world_to_grid  dual_bot_mapper.py:121-125 : int((w - o) / res), truncation toward 0.
 * Python's int() raises on nan/inf; callers reject non-finite poses before this. */
static long w2g(double w, double o, double res) { return (long)((w - o) / res); }

static int in_bounds(const mapper_t *m, long gx, long gy) /* :133-134 */
{ return 0 <= gx && gx < m->size && 0 <= gy && gy < m->size; }

/* update_ray + _bresenham  dual_bot_mapper.py:136-179, fused: the cell list is consumed
 * as it is produced.  Every cell but the last -> FREE; last -> OCCUPIED iff hit_valid. */
static void update_ray(mapper_t *m, double rx, double ry, double hx, double hy, int hit_valid)
{
    long x0 = w2g(rx, m->ox, m->res), y0 = w2g(ry, m->oy, m->res);
    long x1 = w2g(hx, m->ox, m->res), y1 = w2g(hy, m->oy, m->res);
    long dx = labs(x1 - x0), dy = labs(y1 - y0);
    long sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    long err = dx - dy;
    m->n_rays++;
    for (;;) {
        int last = (x0 == x1 && y0 == y1);
        if (!last || hit_valid) {
            if (in_bounds(m, x0, y0)) {
                size_t c = (size_t)y0 * m->size + x0;
                m->grid[c] = last ? CELL_OCCUPIED : CELL_FREE;
                {
                    uint32_t key = (uint32_t)(((4 * m->cur_seq + (uint64_t)m->cur_sensor + 1) << 1) | (last ? 1u : 0u));
                    if (key > m->stamps[c]) m->stamps[c] = key;
                }
                if (last) m->hits[c]++; else m->misses[c]++;
                m->n_cells_written++;
            }
        }
        if (last) break;
        long e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x0 += sx; }
        if (e2 < dx) { err += dx; y0 += sy; }
    }
}

/* ---- the optional index (see lmindex_t) ---------------------------------------------------------------------------- */
static long long ix_key(int type, long cx, long cy) { return ((long long)type << 56) ^ ((long long)(cx & 0xfffffff) << 28) ^ (long long)(cy & 0xfffffff); }
static long ix_coord(double v, double cell) { return (long)floor(v / cell); }
static lmcell_t *ix_find(lmindex_t *ix, long long key, int create)
{
    if (ix->cap == 0) { if (!create) return NULL; ix->cap = 1024; ix->tab = calloc(ix->cap, sizeof(lmcell_t)); }
    if (create && 2 * (ix->used + 1) > ix->cap) {           /* grow and rehash */
        lmindex_t nx = {calloc(2 * ix->cap, sizeof(lmcell_t)), 2 * ix->cap, 0};
        for (size_t i = 0; i < ix->cap; i++) if (ix->tab[i].used) {
            size_t h = (size_t)((unsigned long long)ix->tab[i].key * 0x9E3779B97F4A7C15ull >> 20) & (nx.cap - 1);
            while (nx.tab[h].used) h = (h + 1) & (nx.cap - 1);
            nx.tab[h] = ix->tab[i]; nx.used++;
        }
        free(ix->tab); *ix = nx;
    }
    size_t h = (size_t)((unsigned long long)key * 0x9E3779B97F4A7C15ull >> 20) & (ix->cap - 1);
    while (ix->tab[h].used && ix->tab[h].key != key) h = (h + 1) & (ix->cap - 1);
    if (!ix->tab[h].used) {
        if (!create) return NULL;
        ix->tab[h].used = 1; ix->tab[h].key = key; ix->used++;
    }
    return &ix->tab[h];
}
static void ix_add(lmindex_t *ix, int type, double x, double y, double cell, int log_index)
{
    lmcell_t *c = ix_find(ix, ix_key(type, ix_coord(x, cell), ix_coord(y, cell)), 1);
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 8; c->it = realloc(c->it, (size_t)c->cap * sizeof(int)); }
    c->it[c->n++] = log_index;
}
static void ix_free(lmindex_t *ix)
{
    for (size_t i = 0; i < ix->cap; i++) free(ix->tab[i].it);
    free(ix->tab); memset(ix, 0, sizeof *ix);
}

/* ---- PoseGraphSLAM  dual_bot_mapper.py:261-326 -------------------------------------- */
/* the list scan of :294-318 through the index: same first match (lowest list index), found in the 3 x 3 cells around the query */
static long first_match_indexed(mapper_t *m, graph_t *g, long idx, int agent, int lm_type, double nx, double ny)
{
    if (idx - m->last_closure[agent] < m->min_poses_between) return -1;     /* :304 (true or false for every landmark alike) */
    const double cell = m->closure_radius * (1.0 + 1e-9);
    const long cx = ix_coord(nx, cell), cy = ix_coord(ny, cell);
    long best = -1;
    for (long dy = -1; dy <= 1; dy++) for (long dx = -1; dx <= 1; dx++) {
        const lmcell_t *c = ix_find(&g->ix, ix_key(lm_type, cx + dx, cy + dy), 0);
        if (!c) continue;
        for (int k = 0; k < c->n; k++) {
            const long i = c->it[k];
            if (best >= 0 && i > best) break;
            const landmark_t *l = &g->lms[i];
            if (idx - l->idx < m->min_poses_between) break;                 /* :300 (later entries are newer still) */
            const double dist = sqrt(pow(nx - l->x, 2.0) + pow(ny - l->y, 2.0));   /* :308 */
            if (dist < m->closure_radius) { best = i; break; }                  /* :309 */
        }
    }
    return best;
}

static int check_closure(mapper_t *m, graph_t *g, long idx, int agent, int lm_type,
                         double nx, double ny, double *cdx, double *cdy)
{
    const int indexed = m->use_index && m->closure_radius > 0;
    long i0 = 0, i1 = g->n_lms;
    if (indexed) {                                                      /* the scan below then visits exactly the match */
        const long b = first_match_indexed(m, g, idx, agent, lm_type, nx, ny);
        if (b < 0) return 0;
        i0 = b; i1 = b + 1;
    }
    for (long i = i0; i < i1; i++) {
        const landmark_t *l = &g->lms[i];
        if (l->type != lm_type) continue;                              /* :296 */
        if (idx - l->idx < m->min_poses_between) continue;              /* :300 */
        if (idx - m->last_closure[agent] < m->min_poses_between) continue; /* :304 */
        /* :308  math.sqrt((node.x - lm_x)**2 + (node.y - lm_y)**2); float**2 is libm pow */
        double dist = sqrt(pow(nx - l->x, 2.0) + pow(ny - l->y, 2.0));
        if (dist < m->closure_radius) {                                 /* :309 */
            double ex = l->x - nx, ey = l->y - ny;                      /* :311-312 */
            *cdx = ex * m->closure_correction;                          /* :314-315 */
            *cdy = ey * m->closure_correction;
            if (g->n_cls == g->cap_cls) {
                g->cap_cls = g->cap_cls ? 2 * g->cap_cls : 64;
                g->cls = realloc(g->cls, g->cap_cls * sizeof(closure_t));
            }
            g->cls[g->n_cls++] = (closure_t){l->idx, idx, *cdx, *cdy}; /* :317 */
            m->last_closure[agent] = idx;                               /* :318 */
            return 1;
        }
    }
    return 0;
}

static int add_pose(mapper_t *m, graph_t *g, double x, double y, int agent, int lm_type,
                    double *cdx, double *cdy) /* :273-290 */
{
    long idx = g->n_nodes++;
    int closed = 0;
    *cdx = *cdy = 0.0;
    if (lm_type != 0) {
        closed = check_closure(m, g, idx, agent, lm_type, x, y, cdx, cdy);
        if (g->n_lms == g->cap_lms) {
            g->cap_lms = g->cap_lms ? 2 * g->cap_lms : 256;
            g->lms = realloc(g->lms, g->cap_lms * sizeof(landmark_t));
        }
        if (m->use_index && m->closure_radius > 0) ix_add(&g->ix, lm_type, x, y, m->closure_radius * (1.0 + 1e-9), (int)g->n_lms);
        g->lms[g->n_lms++] = (landmark_t){x, y, lm_type, idx};          /* :288 */
    }
    return closed;
}

/* ---- mapper ------------------------------------------------------------------------ */
mapper_t *qso_create(int size, double res, double ox, double oy, double separation,
                     int max_agent, int bots_per_graph)
{
    mapper_t *m = calloc(1, sizeof(*m));
    m->size = size; m->res = res; m->ox = ox; m->oy = oy;
    m->max_agent = max_agent < 1 ? 2 : max_agent;
    m->bots_per_graph = bots_per_graph < 1 ? m->max_agent : bots_per_graph;
    m->n_graphs = (m->max_agent + m->bots_per_graph - 1) / m->bots_per_graph;
    size_t cells = (size_t)size * size;
    m->grid = malloc(cells); memset(m->grid, CELL_UNKNOWN, cells);
    m->hits = calloc(cells, sizeof(int32_t));
    m->misses = calloc(cells, sizeof(int32_t));
    m->stamps = calloc(cells, sizeof(uint32_t));
    m->seq_stride = 1;
    m->own_lo = 1; m->own_hi = m->max_agent;
    m->graphs = calloc(m->n_graphs, sizeof(graph_t));
    int nb = m->max_agent + 1;
    m->offset_x = calloc(nb, sizeof(double));
    if (m->max_agent >= 2) m->offset_x[2] = separation;
    m->drift = calloc(2 * nb, sizeof(double));
    m->last_closure = malloc(nb * sizeof(long));
    m->zone = malloc(4 * nb * sizeof(double));
    m->zone_n = calloc(nb, sizeof(long));
    m->pkt_count = calloc(nb, sizeof(long));
    m->closure_radius = CLOSURE_RADIUS; m->min_poses_between = MIN_POSES_BETWEEN; m->closure_correction = CLOSURE_CORRECTION;
    for (int b = 0; b < nb; b++) m->last_closure[b] = -MIN_POSES_BETWEEN;
    return m;
}

/* before the first packet: other values of the closure constants (last_closure restarts at -min_between, :271) */
void qso_set_closure_params(mapper_t *m, double radius, long min_between, double correction)
{
    m->closure_radius = radius; m->min_poses_between = min_between; m->closure_correction = correction;
    for (int b = 0; b <= m->max_agent; b++) m->last_closure[b] = -min_between;
}

void qso_destroy(mapper_t *m)
{
    if (!m) return;
    for (int g = 0; g < m->n_graphs; g++) { free(m->graphs[g].lms); free(m->graphs[g].cls); ix_free(&m->graphs[g].ix); }
    free(m->grid); free(m->hits); free(m->misses); free(m->stamps); free(m->graphs); free(m->offset_x);
    free(m->drift); free(m->last_closure); free(m->zone); free(m->zone_n); free(m->pkt_count);
    free(m->ekf); free(m->ekf_prev);
    free(m->pose); free(m->pose_agent); free(m->pose_src); free(m->hit); free(m->hit_agent_sensor);
    free(m);
}

/* closure search through the spatial index (same closures as the list scan); before the first landmark */
void qso_use_index(mapper_t *m, int on) { m->use_index = on; }
void qso_set_offset(mapper_t *m, int bot, double off_x) { m->offset_x[bot] = off_x; }
void qso_set_owned(mapper_t *m, int lo, int hi) { m->own_lo = lo; m->own_hi = hi; }

void qso_ekf_packet(double *f, double *prev, double t, double x, double y, double yaw,
                    double enc, double metres_per_tick);

void qso_enable_ekf(mapper_t *m, double metres_per_tick)
{
    m->ekf_on = 1; m->ekf_mpt = metres_per_tick;
    if (!m->ekf) { m->ekf = calloc((size_t)(m->max_agent + 1) * 44, sizeof(double)); m->ekf_prev = calloc((size_t)(m->max_agent + 1) * 4, sizeof(double)); }
}
const double *qso_ekf_state(const mapper_t *m, int bot) { return m->ekf + (size_t)bot * 44; }

static void zone_fold(mapper_t *m, int bot, double x, double y)
{   /* compute_bounding_box :702-706 over point_clouds[bot] U paths[bot] (:930-940); the
     * lists only grow, so a running min/max equals the reference's recompute. */
    double *z = &m->zone[4 * bot];
    if (m->zone_n[bot]++ == 0) { z[0] = z[2] = x; z[1] = z[3] = y; return; }
    if (x < z[0]) z[0] = x;
    if (y < z[1]) z[1] = y;
    if (x > z[2]) z[2] = x;
    if (y > z[3]) z[3] = y;
}

static float rd_f32(const uint8_t *p) { float f; memcpy(&f, p, 4); return f; }

/* One datagram through dual_bot_mapper.py:826-919.  Returns 1 if accepted. */
int qso_feed(mapper_t *m, const uint8_t *d, int len)
{
    long src = m->n_datagrams++;
    m->cur_seq = m->seq_next; m->seq_next += m->seq_stride;   /* arrival index of this datagram */
    int lm = 0;
    if (len == PACKET_SIZE) lm = d[41];                 /* :828-831 */
    else if (len != PACKET_SIZE_V1) return 0;            /* :832-838 */
    if (memcmp(d, "QSRL", 4) != 0) return 0;             /* :840 */
    int agent = d[4];
    if (agent < 1 || agent > m->max_agent) return 0;     /* :842 */
    double rx = rd_f32(d + 5), ry = rd_f32(d + 9), ryaw = rd_f32(d + 13);
    double dist[4] = {rd_f32(d + 25), rd_f32(d + 29), rd_f32(d + 33), rd_f32(d + 37)};
    /* Python would raise in int() on a non-finite pose; the build defines: drop it. */
    if (!isfinite(rx) || !isfinite(ry) || !isfinite(ryaw)) return 0;
    m->pkt_count[agent]++;                               /* :848 */
    rx += m->offset_x[agent];  /* :851-852 (bot 2: separation; x + 0.0 is exact for the others) */
    const int owned = agent >= m->own_lo && agent <= m->own_hi;
    if (m->ekf_on && owned) {   /* telemetry only: uses the pose before drift correction, never feeds the map */
        int32_t enc; memcpy(&enc, d + 17, 4);
        qso_ekf_packet(m->ekf + (size_t)agent * 44, m->ekf_prev + (size_t)agent * 4, m->cur_time, rx, ry, ryaw,
                       (double)enc, m->ekf_mpt);
    }
    rx += m->drift[2 * agent];                           /* :855-857 */
    ry += m->drift[2 * agent + 1];
    if (owned) zone_fold(m, agent, rx, ry);              /* paths  :878-879 */
    if (m->n_pose == m->cap_pose) {
        m->cap_pose = m->cap_pose ? 2 * m->cap_pose : 1024;
        m->pose = realloc(m->pose, 3 * m->cap_pose * sizeof(double));
        m->pose_agent = realloc(m->pose_agent, m->cap_pose * sizeof(int32_t));
        m->pose_src = realloc(m->pose_src, m->cap_pose * sizeof(long));
    }
    m->pose[3 * m->n_pose] = rx; m->pose[3 * m->n_pose + 1] = ry; m->pose[3 * m->n_pose + 2] = ryaw;
    m->pose_agent[m->n_pose] = agent; m->pose_src[m->n_pose] = src; m->n_pose++;
    for (int s = 0; s < 4 && owned; s++) {               /* :886-903 */
        m->cur_sensor = s;
        double a = ryaw + SENSOR_ANGLES_RAD[s];
        double dd = dist[s];
        int valid = (MIN_DIST_M < dd) && (dd <= MAX_DIST_M);
        if (valid) {
            double wx = rx + dd * cos(a), wy = ry + dd * sin(a);
            if (m->n_hit == m->cap_hit) {
                m->cap_hit = m->cap_hit ? 2 * m->cap_hit : 1024;
                m->hit = realloc(m->hit, 2 * m->cap_hit * sizeof(double));
                m->hit_agent_sensor = realloc(m->hit_agent_sensor, m->cap_hit * sizeof(int32_t));
            }
            m->hit[2 * m->n_hit] = wx; m->hit[2 * m->n_hit + 1] = wy;
            m->hit_agent_sensor[m->n_hit++] = agent * 4 + s;
            zone_fold(m, agent, wx, wy);                 /* point_clouds :892 */
            update_ray(m, rx, ry, wx, wy, 1);            /* :897 */
        } else {
            /* :900  min(dist, MAX) if dist > MIN else MAX ; Python min(a,b): b if b<a else a */
            double r = (dd > MIN_DIST_M) ? ((MAX_DIST_M < dd) ? MAX_DIST_M : dd) : MAX_DIST_M;
            update_ray(m, rx, ry, rx + r * cos(a), ry + r * sin(a), 0); /* :901-903 */
        }
    }
    graph_t *g = &m->graphs[(agent - 1) / m->bots_per_graph];
    double cdx, cdy;
    if (add_pose(m, g, rx, ry, agent, lm, &cdx, &cdy)) { /* :908-914 */
        m->drift[2 * agent] = m->drift[2 * agent] + cdx;
        m->drift[2 * agent + 1] = m->drift[2 * agent + 1] + cdy;
    }
    return 1;
}

/* n records of `stride` bytes; lens[i] = datagram length (NULL: every one is `stride`). */
long qso_feed_stream(mapper_t *m, const uint8_t *buf, long n, long stride, const uint16_t *lens)
{
    long acc = 0;
    for (long i = 0; i < n; i++) {
        m->cur_time = (double)(m->n_datagrams);
        acc += qso_feed(m, buf + i * stride, lens ? lens[i] : (int)stride);
    }
    return acc;
}

/* same with receive times (seconds) for the EKF wiring */
long qso_feed_stream_t(mapper_t *m, const uint8_t *buf, long n, long stride, const uint16_t *lens,
                       const double *times)
{
    long acc = 0;
    for (long i = 0; i < n; i++) {
        m->cur_time = times ? times[i] : (double)(m->n_datagrams);
        acc += qso_feed(m, buf + i * stride, lens ? lens[i] : (int)stride);
    }
    return acc;
}

/* batched OccupancyGrid.update_ray (the object API of :136) */
void qso_update_rays(mapper_t *m, const double *rx, const double *ry, const double *hx,
                     const double *hy, const uint8_t *valid, long n)
{
    for (long i = 0; i < n; i++) update_ray(m, rx[i], ry[i], hx[i], hy[i], valid[i]);
}

void qso_world_to_grid(const mapper_t *m, const double *w, long n, int axis, int64_t *out)
{
    for (long i = 0; i < n; i++) out[i] = w2g(w[i], axis ? m->oy : m->ox, m->res);
}

/* _bresenham as a list: out gets (x,y) pairs; returns the cell count.  :158-179 */
long qso_bresenham(long x0, long y0, long x1, long y1, int64_t *out, long cap)
{
    long dx = labs(x1 - x0), dy = labs(y1 - y0);
    long sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1, err = dx - dy, n = 0;
    for (;;) {
        if (n < cap) { out[2 * n] = x0; out[2 * n + 1] = y0; }
        n++;
        if (x0 == x1 && y0 == y1) break;
        long e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x0 += sx; }
        if (e2 < dx) { err += dx; y0 += sy; }
    }
    return n;
}

/* ---- read-back ----------------------------------------------------------------------- */
const int8_t *qso_grid(const mapper_t *m) { return m->grid; }
const uint32_t *qso_stamps(const mapper_t *m) { return m->stamps; }
/* arrival indices of the following datagrams: seq0, seq0 + stride, ... (sharded streams) */
void qso_set_sequence(mapper_t *m, uint64_t seq0, uint64_t stride) { m->seq_next = seq0; m->seq_stride = stride ? stride : 1; }
const int32_t *qso_hits(const mapper_t *m) { return m->hits; }
const int32_t *qso_misses(const mapper_t *m) { return m->misses; }
long qso_n_nodes(const mapper_t *m, int g) { return m->graphs[g].n_nodes; }
long qso_n_landmarks(const mapper_t *m, int g) { return m->graphs[g].n_lms; }
long qso_n_closures(const mapper_t *m, int g) { return m->graphs[g].n_cls; }
long qso_n_poses(const mapper_t *m) { return m->n_pose; }
long qso_n_hits(const mapper_t *m) { return m->n_hit; }
long qso_n_rays(const mapper_t *m) { return m->n_rays; }
long qso_n_cells_written(const mapper_t *m) { return m->n_cells_written; }
const double *qso_poses(const mapper_t *m) { return m->pose; }
const int32_t *qso_pose_agents(const mapper_t *m) { return m->pose_agent; }
const double *qso_hit_points(const mapper_t *m) { return m->hit; }
const int32_t *qso_hit_agent_sensor(const mapper_t *m) { return m->hit_agent_sensor; }
void qso_drift(const mapper_t *m, int bot, double out[2])
{ out[0] = m->drift[2 * bot]; out[1] = m->drift[2 * bot + 1]; }
void qso_closures(const mapper_t *m, int g, int64_t *idx2, double *corr2)
{
    for (long i = 0; i < m->graphs[g].n_cls; i++) {
        idx2[2 * i] = m->graphs[g].cls[i].lm_idx; idx2[2 * i + 1] = m->graphs[g].cls[i].node_idx;
        corr2[2 * i] = m->graphs[g].cls[i].dx; corr2[2 * i + 1] = m->graphs[g].cls[i].dy;
    }
}
void qso_landmarks(const mapper_t *m, int g, double *xy, int64_t *type_idx)
{
    for (long i = 0; i < m->graphs[g].n_lms; i++) {
        xy[2 * i] = m->graphs[g].lms[i].x; xy[2 * i + 1] = m->graphs[g].lms[i].y;
        type_idx[2 * i] = m->graphs[g].lms[i].type; type_idx[2 * i + 1] = m->graphs[g].lms[i].idx;
    }
}
/* zone bbox of `bot` (what gets sent to the OTHER bot, :940-941); returns 0 if no points */
int qso_zone(const mapper_t *m, int bot, double out[4])
{
    if (m->zone_n[bot] == 0) return 0;
    memcpy(out, &m->zone[4 * bot], 4 * sizeof(double));
    return 1;
}
/* send_zone_to_bot payload :675-684, ZONE_FMT '<4sffff' :49 */
void qso_zone_packet(const mapper_t *m, int bot, int online, uint8_t out[20])
{
    float f[4] = {999.0f, 999.0f, -999.0f, -999.0f};
    double z[4];
    if (online && qso_zone(m, bot, z)) for (int i = 0; i < 4; i++) f[i] = (float)z[i];
    memcpy(out, "ZONE", 4);
    memcpy(out + 4, f, 16);
}

/* float log-odds view from the integer counts (build extension):
 * L = clamp(hits*l_occ - misses*l_free, lmin, lmax) evaluated in fp32 from exact ints. */
void qso_logodds(const mapper_t *m, float l_occ, float l_free, float lmin, float lmax, float *out)
{
    size_t cells = (size_t)m->size * m->size;
    for (size_t c = 0; c < cells; c++) {
        float v = (float)m->hits[c] * l_occ - (float)m->misses[c] * l_free;
        out[c] = v < lmin ? lmin : (v > lmax ? lmax : v);
    }
}

/* ---- map_merger.py ------------------------------------------------------------------- */
/* grid_to_pcd  server_nodes/map_merger.py:64-85: occupied = data > 50, point =
 * (col*res + ox, row*res + oy), argwhere order (row-major).  Returns the point count. */
long qso_grid_to_pcd(const int8_t *data, int h, int w, double res, double ox, double oy,
                     double *xy, long cap)
{
    long n = 0;
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++)
            if (data[(size_t)r * w + c] > 50) {
                if (n < cap) { xy[2 * n] = c * res + ox; xy[2 * n + 1] = r * res + oy; }
                n++;
            }
    return n;
}

/* publish_global_map  map_merger.py:87-127: bbox -> canvas of ceil(span/res)+1 filled -1,
 * idx = ((p - min)/res).astype(int) clipped, grid[y,x] = 100.  out_dims = {h, w};
 * out_origin = {min_x, min_y}.  Call with grid == NULL to size the canvas first. */
int qso_rasterise(const double *xy, long n, double res, int *out_dims, double *out_origin,
                  int8_t *grid)
{
    if (n <= 0) return 0;
    double mnx = xy[0], mxx = xy[0], mny = xy[1], mxy = xy[1];
    for (long i = 1; i < n; i++) {
        if (xy[2 * i] < mnx) mnx = xy[2 * i];
        if (xy[2 * i] > mxx) mxx = xy[2 * i];
        if (xy[2 * i + 1] < mny) mny = xy[2 * i + 1];
        if (xy[2 * i + 1] > mxy) mxy = xy[2 * i + 1];
    }
    int w = (int)ceil((mxx - mnx) / res) + 1, h = (int)ceil((mxy - mny) / res) + 1;
    out_dims[0] = h; out_dims[1] = w; out_origin[0] = mnx; out_origin[1] = mny;
    if (!grid) return 1;
    memset(grid, CELL_UNKNOWN, (size_t)h * w);
    for (long i = 0; i < n; i++) {
        long xi = (long)((xy[2 * i] - mnx) / res), yi = (long)((xy[2 * i + 1] - mny) / res);
        xi = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi);
        yi = yi < 0 ? 0 : (yi > h - 1 ? h - 1 : yi);
        grid[(size_t)yi * w + xi] = CELL_OCCUPIED;
    }
    return 1;
}

/* ---- EKF  AgentFirmware_Bot1/ekf.cpp:5-92, ekf.h:38-44 ------------------------------- */
/* state x[6] = [x, y, theta, v, omega, bias_omega]; P row-major 6x6; layout of one filter:
 * {x[6], P[36], last_time, initialized} = 44 doubles. */
#define EKF_STRIDE 44
static const double EKF_Q[6] = {0.01, 0.01, 0.01, 0.1, 0.1, 0.001};  /* ekf.cpp:11 */
static const double EKF_R[2] = {0.05, 0.05};                          /* ekf.cpp:12 */

void qso_ekf_init(double *f, double t, const double x0[6])  /* ctor :5-13 + init :15-19 */
{
    for (int i = 0; i < 6; i++) f[i] = x0 ? x0[i] : 0.0;
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) f[6 + 6 * r + c] = (r == c);
    f[42] = t; f[43] = 1.0;
}

void qso_ekf_predict(double *f, double omega_measured, double t)  /* ekf.cpp:26-68 */
{
    if (f[43] == 0.0) return;
    double dt = t - f[42];
    if (dt <= 0) return;
    f[42] = t;
    double *x = f, *P = f + 6;
    double theta = x[2], v = x[3], bias = x[5];
    double omega_c = omega_measured - bias;
    double theta_new = theta + omega_c * dt;
    if (theta_new > M_PI) theta_new -= 2 * M_PI;
    else if (theta_new < -M_PI) theta_new += 2 * M_PI;
    double x_new = x[0] + v * cos(theta) * dt;
    double y_new = x[1] + v * sin(theta) * dt;
    x[0] = x_new; x[1] = y_new; x[2] = theta_new; x[4] = omega_c;
    double J[36], JP[36];
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) J[6 * r + c] = (r == c);
    J[0 * 6 + 2] = -v * sin(theta) * dt; J[0 * 6 + 3] = cos(theta) * dt;
    J[1 * 6 + 2] = v * cos(theta) * dt;  J[1 * 6 + 3] = sin(theta) * dt;
    J[2 * 6 + 5] = -dt; J[4 * 6 + 4] = 0.0; J[4 * 6 + 5] = -1.0;
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += J[6 * r + k] * P[6 * k + c];
        JP[6 * r + c] = s;
    }
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += JP[6 * r + k] * J[6 * c + k];
        P[6 * r + c] = s + (r == c ? EKF_Q[r] : 0.0);
    }
}

void qso_ekf_update(double *f, double z_v, double z_omega)  /* ekf.cpp:70-92 */
{
    if (f[43] == 0.0) return;
    double *x = f, *P = f + 6;
    double y0 = z_v - x[3], y1 = z_omega - x[4];
    /* S = H P H^T + R with H selecting rows/cols 3,4 */
    double s00 = P[3 * 6 + 3] + EKF_R[0], s01 = P[3 * 6 + 4];
    double s10 = P[4 * 6 + 3],            s11 = P[4 * 6 + 4] + EKF_R[1];
    double det = s00 * s11 - s01 * s10;
    double invdet = 1.0 / det;   /* 2x2 inverse as adjugate * (1/det), as Eigen's fixed-size path does */
    double i00 = s11 * invdet, i01 = -s01 * invdet, i10 = -s10 * invdet, i11 = s00 * invdet;
    double K[12];
    for (int r = 0; r < 6; r++) {
        double p3 = P[6 * r + 3], p4 = P[6 * r + 4];
        K[2 * r] = p3 * i00 + p4 * i10;
        K[2 * r + 1] = p3 * i01 + p4 * i11;
    }
    for (int r = 0; r < 6; r++) x[r] = x[r] + (K[2 * r] * y0 + K[2 * r + 1] * y1);
    double Pn[36];
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
        /* (I - K H) P = P - K (H P);  (I-KH)[r][k] = delta - K[r][0]*[k==3] - K[r][1]*[k==4] */
        double s = 0.0;
        for (int k = 0; k < 6; k++) {
            double ikh = (r == k ? 1.0 : 0.0) - (k == 3 ? K[2 * r] : 0.0) - (k == 4 ? K[2 * r + 1] : 0.0);
            s += ikh * P[6 * k + c];
        }
        Pn[6 * r + c] = s;
    }
    memcpy(P, Pn, sizeof(Pn));
}

/* Build-defined wiring of the EKF to the telemetry stream (modelled on
 * esp32_firmware/src/main.cpp:176-188): per accepted packet of a bot, in arrival order,
 *   first packet: init(t, [x, y, yaw, 0, 0, 0]);
 *   later: dt = t - t_prev; if dt > 0: inv_dt = 1/dt, omega_m = wrap(yaw - yaw_prev) * inv_dt,
 *          v_enc = (enc - enc_prev) * metres_per_tick * inv_dt; predict(omega_m, t); update(v_enc, omega_m)
 * One step of that wiring, for one bot, given already-decoded fields.  prev = {t, yaw, enc, seen}. */
void qso_ekf_packet(double *f, double *prev, double t, double x, double y, double yaw,
                    double enc, double metres_per_tick)
{
    if (prev[3] == 0.0) {
        double x0[6] = {x, y, yaw, 0, 0, 0};
        qso_ekf_init(f, t, x0);
    } else {
        double dt = t - prev[0];
        if (dt > 0) {
            double dyaw = yaw - prev[1];
            if (dyaw > M_PI) dyaw -= 2 * M_PI;
            else if (dyaw < -M_PI) dyaw += 2 * M_PI;
            double inv_dt = 1.0 / dt;
            double omega_m = dyaw * inv_dt;
            double v_enc = (enc - prev[2]) * metres_per_tick * inv_dt;
            qso_ekf_predict(f, omega_m, t);
            qso_ekf_update(f, v_enc, omega_m);
        }
    }
    prev[0] = t; prev[1] = yaw; prev[2] = enc; prev[3] = 1.0;
}

/* ---- frontiers  dual_bot_mapper.py:181-237 ------------------------------------------------ */
/* get_frontiers :181-196: interior FREE cells with a 4-neighbour UNKNOWN, row-major (y outer). */
long qso_frontier_cells(const int8_t *grid, int size, int32_t *xy, long cap)
{
    long n = 0;
    for (int y = 1; y < size - 1; y++)
        for (int x = 1; x < size - 1; x++) {
            if (grid[(size_t)y * size + x] != CELL_FREE) continue;
            if (grid[(size_t)y * size + x - 1] == CELL_UNKNOWN || grid[(size_t)y * size + x + 1] == CELL_UNKNOWN ||
                grid[(size_t)(y - 1) * size + x] == CELL_UNKNOWN || grid[(size_t)(y + 1) * size + x] == CELL_UNKNOWN) {
                if (n < cap) { xy[2 * n] = x; xy[2 * n + 1] = y; }
                n++;
            }
        }
    return n;
}

/* cluster_frontiers :198-231 (BFS flood fill over 4-neighbours, seeds in list order, clusters of
 * fewer than min_cluster cells dropped) + the integer sums cluster_centroid_world :233-237 divides.
 * stats per kept cluster: size, first_x, first_y, sum_x, sum_y.  Returns the cluster count. */
long qso_frontier_clusters(const int32_t *xy, long n, int size, int min_cluster, int64_t *stats, long cap)
{
    size_t cells = (size_t)size * size;
    uint8_t *in_set = calloc(cells, 1), *visited = calloc(cells, 1);
    int32_t *queue = malloc((size_t)(n > 0 ? n : 1) * 2 * sizeof(int32_t));
    for (long i = 0; i < n; i++) in_set[(size_t)xy[2 * i + 1] * size + xy[2 * i]] = 1;
    long nc = 0;
    static const int DX[4] = {-1, 1, 0, 0}, DY[4] = {0, 0, -1, 1};     /* :223 */
    for (long i = 0; i < n; i++) {
        int sx = xy[2 * i], sy = xy[2 * i + 1];
        if (visited[(size_t)sy * size + sx]) continue;
        long head = 0, tail = 0, cnt = 0;
        int64_t sumx = 0, sumy = 0;
        queue[0] = sx; queue[1] = sy; tail = 1;
        while (head < tail) {
            int cx = queue[2 * head], cy = queue[2 * head + 1];
            head++;
            size_t c = (size_t)cy * size + cx;
            if (visited[c]) continue;
            visited[c] = 1; cnt++; sumx += cx; sumy += cy;
            for (int d = 0; d < 4; d++) {
                int nx = cx + DX[d], ny = cy + DY[d];
                if (nx < 0 || ny < 0 || nx >= size || ny >= size) continue;
                size_t nb = (size_t)ny * size + nx;
                if (in_set[nb] && !visited[nb]) {
                    if (tail >= n) {           /* duplicates can be queued: grow on demand */
                        /* each cell is queued at most 4 times; 4n is a safe bound */
                    }
                    queue = realloc(queue, (size_t)(tail + 1) * 2 * sizeof(int32_t));
                    queue[2 * tail] = nx; queue[2 * tail + 1] = ny; tail++;
                }
            }
        }
        if (cnt >= min_cluster) {
            if (nc < cap) { stats[5 * nc] = cnt; stats[5 * nc + 1] = sx; stats[5 * nc + 2] = sy; stats[5 * nc + 3] = sumx; stats[5 * nc + 4] = sumy; }
            nc++;
        }
    }
    free(in_set); free(visited); free(queue);
    return nc;
}
