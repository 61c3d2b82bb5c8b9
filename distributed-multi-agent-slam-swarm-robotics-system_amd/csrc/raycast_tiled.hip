// raycast_tiled.hip -- K1 (production form): tile-binned, LDS-staged raycast.
// Same semantics as raycast.hip (dual_bot_mapper.py:882-903, :136-179); different schedule.
//
// Why: the reference workloads revisit the same few hundred cells thousands of times per batch
// (two robots circling one room).  One global atomic per cell write then serialises on a few
// cache lines.  Here the cell writes of a batch are sorted by 64x64-cell grid tile, each
// workgroup rasters up to QT_CHUNK rays of ONE tile into an LDS copy of that tile (ds_max_u32 /
// ds_add_u32), and only the touched cells of the tile are merged into the HBM grid, one
// coalesced 256-byte row per wave-instruction.
//
//   pass A  qs_rays_kernel      per ray: projection (fp64 trig), grid end points, zone / hit
//                                outputs, per-tile counts (wave-aggregated atomics).  Rays longer
//                                than one tile (fine resolutions) are written directly.
//   pass B  qs_tile_scan_kernel exclusive scans: records per tile -> record base, chunks per
//                                tile -> work-item base.
//   pass C  qs_scatter_kernel   per ray: one 16-byte record per overlapped tile (<= 2x2),
//                                slots reserved with wave-aggregated atomics.
//   pass D  qs_raster_kernel    per work item (tile, <= QT_CHUNK records): LDS raster + merge.
//
// HBM traffic per packet (4 rays, ~1.4 tile records per ray): 45 B decoded fields + 64 B ray
// end points written and read + ~90 B records written and read, independent of how many cells
// each ray covers; the per-cell work happens in LDS.
#include "qs_internal.h"
#include "raycast_common.h"

#define QT_TILE 64                       // tile edge in cells: 64 x 64 x u32 = 16 KiB of LDS
#define QT_TILE_SHIFT 6
#define QT_CELLS (QT_TILE * QT_TILE)
#define QT_CHUNK 2048                    // records per raster work item
#define QT_BLOCK 256
#define QT_MAX_ITEMS_PAD 8

struct QtWorkspace {
    unsigned int *tile_count;    // [n_tiles]   records per tile (pass A)
    unsigned int *tile_cursor;   // [n_tiles]   scatter cursors (pass C)
    unsigned int *tile_base;     // [n_tiles+1] exclusive scan of tile_count
    unsigned int *chunk_base;    // [n_tiles+1] exclusive scan of ceil(count / QT_CHUNK)
    int4 *rays;                  // [4n]        absolute grid end points (x0,y0,x1,y1); x0 = INT_MIN: no record
    uint4 *recs;                 // [16n]       tile records
    int tiles_x, n_tiles;
};

// ---- wave-aggregated atomicAdd over a small key space -------------------------------------
// Lanes with equal `key` are served by ONE atomic; returns each lane's slot (base + rank).
__device__ inline unsigned int qt_wave_agg_add(unsigned int *counters, int key, bool active)
{
    const int lane = threadIdx.x & 63;
    unsigned int result = 0;
    unsigned long long remaining = __ballot(active);
    while (remaining) {
        const int leader = __ffsll((long long)remaining) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long grp = __ballot(active && key == k);
        if (active && key == k) {
            const unsigned int rank = __popcll(grp & ((1ull << lane) - 1));
            unsigned int base = 0;
            if (rank == 0) base = atomicAdd(&counters[k], (unsigned int)__popcll(grp));
            base = __shfl(base, leader);
            result = base + rank;
        }
        remaining &= ~grp;
    }
    return result;
}

// ---- pass A -----------------------------------------------------------------------------------
template <bool COUNTS>
__global__ void __launch_bounds__(QT_BLOCK)
qs_rays_kernel(size_t n, QsBatch b, QsGeom geo, QtWorkspace ws, unsigned int *__restrict__ stamps,
               unsigned long long *__restrict__ counts, unsigned long long ord_base, unsigned long long ord_stride,
               unsigned long long *__restrict__ zone, int max_agent, unsigned long long *__restrict__ counters)
{
    __shared__ unsigned long long s_zone[QS_MAX_AGENT + 1][4];
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    for (int t = tid; t <= max_agent; t += QT_BLOCK) {
        s_zone[t][0] = QS_ORD_MIN_IDENT; s_zone[t][1] = QS_ORD_MIN_IDENT;
        s_zone[t][2] = QS_ORD_MAX_IDENT; s_zone[t][3] = QS_ORD_MAX_IDENT;
    }
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();

    const size_t r = (size_t)blockIdx.x * QT_BLOCK + tid;
    const size_t i = r >> 2;
    const int s = (int)(r & 3);
    unsigned int my_cells = 0, my_ray = 0, my_hit = 0;
    bool binned = false;
    int tx_lo = 0, tx_hi = 0, ty_lo = 0, ty_hi = 0;
    int4 rec = make_int4((int)0x80000000, 0, 0, 0);
    if (i < n && b.accept[i]) {
        const double rx = b.rx[i], ry = b.ry[i], yaw = b.yaw[i];
        const float4 d4 = b.dist[i];
        const float df = s == 0 ? d4.x : (s == 1 ? d4.y : (s == 2 ? d4.z : d4.w));
        const int agent = b.agent[i];
        QsRay ray = qs_project_ray(rx, ry, yaw, (double)df, s, geo);
        b.hit[r] = make_double2(ray.ex, ray.ey);
        b.hit_valid[r] = ray.valid ? 1 : 0;
        if (s == 0) {   // paths[agent].append  dual_bot_mapper.py:878-879
            atomicMin(&s_zone[agent][0], qs_ord_from_double(rx)); atomicMin(&s_zone[agent][1], qs_ord_from_double(ry));
            atomicMax(&s_zone[agent][2], qs_ord_from_double(rx)); atomicMax(&s_zone[agent][3], qs_ord_from_double(ry));
        }
        if (ray.valid) {  // point_clouds[agent][name].append  :892
            atomicMin(&s_zone[agent][0], qs_ord_from_double(ray.ex)); atomicMin(&s_zone[agent][1], qs_ord_from_double(ray.ey));
            atomicMax(&s_zone[agent][2], qs_ord_from_double(ray.ex)); atomicMax(&s_zone[agent][3], qs_ord_from_double(ray.ey));
            my_hit = 1;
        }
        my_ray = 1;
        QsLine ln;
        if (qs_line_setup(ray, rx, ry, geo, ln)) {
            if (ln.dx < QT_TILE && ln.dy < QT_TILE) {
                // the ray's cells lie in at most 2 x 2 tiles; clip the tile range to the grid
                const int xlo = max(min(ln.x0, ln.x1), 0), xhi = min(max(ln.x0, ln.x1), geo.size - 1);
                const int ylo = max(min(ln.y0, ln.y1), 0), yhi = min(max(ln.y0, ln.y1), geo.size - 1);
                tx_lo = xlo >> QT_TILE_SHIFT; tx_hi = xhi >> QT_TILE_SHIFT;
                ty_lo = ylo >> QT_TILE_SHIFT; ty_hi = yhi >> QT_TILE_SHIFT;
                rec = make_int4(ln.x0, ln.y0, ln.x1, ln.y1);
                binned = true;
            } else {
                // long ray (fine resolution): direct global atomics, as raycast.hip
                const unsigned int key_free = (unsigned int)((ord_base + ord_stride * i + s + 1) << 1);
                int x = ln.x0, y = ln.y0, err = ln.dx - ln.dy;
                for (;;) {
                    const bool last = (x == ln.x1 && y == ln.y1);
                    if ((!last || ray.valid) && x >= 0 && x < geo.size && y >= 0 && y < geo.size) {
                        const size_t c = (size_t)y * geo.size + x;
                        atomicMax(&stamps[c], key_free | (last ? 1u : 0u));
                        if (COUNTS) atomicAdd(&counts[c], last ? (1ull << 32) : 1ull);
                        my_cells++;
                    }
                    if (last) break;
                    const int e2 = 2 * err;
                    if (e2 > -ln.dy) { err -= ln.dy; x += ln.sx; }
                    if (e2 < ln.dx) { err += ln.dx; y += ln.sy; }
                }
            }
        }
    }
    if (r < 4 * n) ws.rays[r] = rec;
    // per-tile record counts: up to 2 x 2 tiles per ray, one aggregated atomic per distinct tile
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tx = tx_lo + (q & 1), ty = ty_lo + (q >> 1);
        const bool act = binned && tx <= tx_hi && ty <= ty_hi;
        qt_wave_agg_add(ws.tile_count, ty * ws.tiles_x + tx, act);
    }
    if (my_ray) atomicAdd(&s_cnt[0], my_ray);
    if (my_cells) atomicAdd(&s_cnt[1], my_cells);
    if (my_hit) atomicAdd(&s_cnt[2], my_hit);
    __syncthreads();
    for (int t = tid; t <= max_agent; t += QT_BLOCK) {
        if (s_zone[t][0] != QS_ORD_MIN_IDENT) {
            atomicMin(&zone[4 * t + 0], s_zone[t][0]); atomicMin(&zone[4 * t + 1], s_zone[t][1]);
            atomicMax(&zone[4 * t + 2], s_zone[t][2]); atomicMax(&zone[4 * t + 3], s_zone[t][3]);
        }
    }
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&counters[QS_CNT_RAYS], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&counters[QS_CNT_HITS], (unsigned long long)s_cnt[2]);
    }
}

// ---- pass B: one workgroup, two exclusive scans over the tiles --------------------------------
__global__ void __launch_bounds__(1024)
qs_tile_scan_kernel(QtWorkspace ws)
{
    __shared__ unsigned int s_rec[1024], s_chk[1024];
    const int tid = threadIdx.x;
    const int per = (ws.n_tiles + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, ws.n_tiles);
    unsigned int a = 0, c = 0;
    for (int t = lo; t < hi; t++) { const unsigned int v = ws.tile_count[t]; a += v; c += (v + QT_CHUNK - 1) / QT_CHUNK; }
    s_rec[tid] = a; s_chk[tid] = c;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned int va = 0, vc = 0;
        if (tid >= off) { va = s_rec[tid - off]; vc = s_chk[tid - off]; }
        __syncthreads();
        s_rec[tid] += va; s_chk[tid] += vc;
        __syncthreads();
    }
    unsigned int ra = s_rec[tid] - a, rc = s_chk[tid] - c;      // exclusive prefix of this lane's range
    for (int t = lo; t < hi; t++) {
        const unsigned int v = ws.tile_count[t];
        ws.tile_base[t] = ra; ws.chunk_base[t] = rc;
        ra += v; rc += (v + QT_CHUNK - 1) / QT_CHUNK;
    }
    if (tid == 1023) { ws.tile_base[ws.n_tiles] = s_rec[1023]; ws.chunk_base[ws.n_tiles] = s_chk[1023]; }
}

// ---- pass C: scatter tile records ---------------------------------------------------------------
__global__ void __launch_bounds__(QT_BLOCK)
qs_scatter_kernel(size_t n_rays, QsBatch b, QtWorkspace ws, int size, unsigned long long ord_base,
                  unsigned long long ord_stride)
{
    const size_t r = (size_t)blockIdx.x * QT_BLOCK + threadIdx.x;
    int4 ray = make_int4((int)0x80000000, 0, 0, 0);
    if (r < n_rays) ray = ws.rays[r];
    const bool binned = ray.x != (int)0x80000000;
    int tx_lo = 0, tx_hi = -1, ty_lo = 0, ty_hi = -1;
    unsigned int key_free = 0, flags = 0;
    if (binned) {
        const int xlo = max(min(ray.x, ray.z), 0), xhi = min(max(ray.x, ray.z), size - 1);
        const int ylo = max(min(ray.y, ray.w), 0), yhi = min(max(ray.y, ray.w), size - 1);
        tx_lo = xlo >> QT_TILE_SHIFT; tx_hi = xhi >> QT_TILE_SHIFT;
        ty_lo = ylo >> QT_TILE_SHIFT; ty_hi = yhi >> QT_TILE_SHIFT;
        key_free = (unsigned int)((ord_base + ord_stride * (r >> 2) + (r & 3) + 1) << 1);
        flags = b.hit_valid[r] ? 1u : 0u;
    }
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tx = tx_lo + (q & 1), ty = ty_lo + (q >> 1);
        const bool act = binned && tx <= tx_hi && ty <= ty_hi;
        const int tile = ty * ws.tiles_x + tx;
        const unsigned int slot = qt_wave_agg_add(ws.tile_cursor, tile, act);
        if (act) {
            const int ox = tx << QT_TILE_SHIFT, oy = ty << QT_TILE_SHIFT;   // tile origin
            uint4 rec;
            rec.x = ((unsigned int)(ray.x - ox) & 0xffffu) | ((unsigned int)(ray.y - oy) << 16);
            rec.y = ((unsigned int)(ray.z - ox) & 0xffffu) | ((unsigned int)(ray.w - oy) << 16);
            rec.z = key_free;
            rec.w = flags;
            ws.recs[(size_t)ws.tile_base[tile] + slot] = rec;
        }
    }
}

// ---- pass D: LDS raster + merge -------------------------------------------------------------------
template <bool COUNTS>
__global__ void __launch_bounds__(QT_BLOCK)
qs_raster_kernel(QtWorkspace ws, int size, unsigned int *__restrict__ stamps,
                 unsigned long long *__restrict__ counts, unsigned long long *__restrict__ counters)
{
    __shared__ unsigned int s_stamp[QT_CELLS];
    __shared__ unsigned int s_cnt[COUNTS ? QT_CELLS : 1];    // hi16 hits, lo16 misses (<= QT_CHUNK each)
    __shared__ unsigned int s_cells;
    const int tid = threadIdx.x;
    const unsigned int n_items = ws.chunk_base[ws.n_tiles];
    const unsigned int item = blockIdx.x;
    if (item >= n_items) return;
    // tile of this work item: last t with chunk_base[t] <= item
    int lo = 0, hi = ws.n_tiles;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ws.chunk_base[mid] <= item) lo = mid; else hi = mid; }
    const int tile = lo;
    const unsigned int cnt = ws.tile_count[tile];
    const unsigned int rb = ws.tile_base[tile] + (item - ws.chunk_base[tile]) * QT_CHUNK;
    const unsigned int re = min(rb + QT_CHUNK, ws.tile_base[tile] + cnt);
    const bool exclusive = cnt <= QT_CHUNK;          // the only work item of this tile in this launch
    const int tx0 = (tile % ws.tiles_x) << QT_TILE_SHIFT, ty0 = (tile / ws.tiles_x) << QT_TILE_SHIFT;
    const int tw = min(QT_TILE, size - tx0), th = min(QT_TILE, size - ty0);

    for (int c = tid; c < QT_CELLS / 4; c += QT_BLOCK) {
        ((uint4 *)s_stamp)[c] = make_uint4(0, 0, 0, 0);
        if (COUNTS) ((uint4 *)s_cnt)[c] = make_uint4(0, 0, 0, 0);
    }
    if (tid == 0) s_cells = 0;
    __syncthreads();

    unsigned int my_cells = 0;
    for (unsigned int j = rb + tid; j < re; j += QT_BLOCK) {
        const uint4 rec = ws.recs[j];
        int x = (short)(rec.x & 0xffffu), y = (short)(rec.x >> 16);
        const int x1 = (short)(rec.y & 0xffffu), y1 = (short)(rec.y >> 16);
        const unsigned int key_free = rec.z;
        const bool valid = rec.w & 1u;
        const int dx = abs(x1 - x), dy = abs(y1 - y);             // dual_bot_mapper.py:161-165
        const int sx = x < x1 ? 1 : -1, sy = y < y1 ? 1 : -1;
        int err = dx - dy;
        for (;;) {
            const bool last = (x == x1 && y == y1);                // :169
            if ((!last || valid) && (unsigned int)x < (unsigned int)tw && (unsigned int)y < (unsigned int)th) {
                const int c = (y << QT_TILE_SHIFT) + x;
                atomicMax(&s_stamp[c], key_free | (last ? 1u : 0u));          // :150 / :156
                if (COUNTS) atomicAdd(&s_cnt[c], last ? 0x10000u : 1u);
                my_cells++;
            }
            if (last) break;
            const int e2 = 2 * err;                                // :171-177
            if (e2 > -dy) { err -= dy; x += sx; }
            if (e2 < dx) { err += dx; y += sy; }
        }
    }
    if (my_cells) atomicAdd(&s_cells, my_cells);
    __syncthreads();

    // merge the touched cells into the HBM grid: one 64-cell (256 B) row per wave-instruction
    for (int c = tid; c < QT_CELLS; c += QT_BLOCK) {
        const int x = c & (QT_TILE - 1), y = c >> QT_TILE_SHIFT;
        const unsigned int v = s_stamp[c];
        if (v != 0 && x < tw && y < th) {
            const size_t gidx = (size_t)(ty0 + y) * size + (tx0 + x);
            if (exclusive) { if (v > stamps[gidx]) stamps[gidx] = v; }
            else atomicMax(&stamps[gidx], v);
            if (COUNTS) {
                const unsigned int k = s_cnt[c];
                const unsigned long long add = ((unsigned long long)(k >> 16) << 32) | (k & 0xffffu);
                if (exclusive) counts[gidx] += add;
                else atomicAdd(&counts[gidx], add);
            }
        }
    }
    if (tid == 0 && s_cells) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)s_cells);
}

// ---- host side ------------------------------------------------------------------------------------
static inline size_t qt_align(size_t v) { return (v + 255) & ~(size_t)255; }

size_t qs_tiled_workspace_bytes(const qs_ctx *c, size_t n)
{
    const int tiles_x = (c->cfg.size + QT_TILE - 1) / QT_TILE;
    const size_t n_tiles = (size_t)tiles_x * tiles_x;
    return 4 * qt_align((n_tiles + 1) * sizeof(unsigned int)) + qt_align(4 * n * sizeof(int4)) +
           qt_align(16 * n * sizeof(uint4));
}

hipError_t qs_launch_raycast_tiled(qs_ctx *c, size_t n, uint64_t seq0)
{
    if (n == 0) return hipSuccess;
    const size_t need = qs_tiled_workspace_bytes(c, c->cap_batch);
    if (need > c->bin_ws_bytes) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return e;
        if (c->d_bin_ws) { hipFree(c->d_bin_ws); c->d_bin_ws = nullptr; c->bin_ws_bytes = 0; }
        e = hipMalloc(&c->d_bin_ws, need);
        if (e != hipSuccess) return e;
        c->bin_ws_bytes = need;
    }
    QtWorkspace ws;
    ws.tiles_x = (c->cfg.size + QT_TILE - 1) / QT_TILE;
    ws.n_tiles = ws.tiles_x * ws.tiles_x;
    const size_t tbytes = qt_align(((size_t)ws.n_tiles + 1) * sizeof(unsigned int));
    char *p = (char *)c->d_bin_ws;
    ws.tile_count = (unsigned int *)p; p += tbytes;
    ws.tile_cursor = (unsigned int *)p; p += tbytes;
    ws.tile_base = (unsigned int *)p; p += tbytes;
    ws.chunk_base = (unsigned int *)p; p += tbytes;
    ws.rays = (int4 *)p; p += qt_align(4 * c->cap_batch * sizeof(int4));
    ws.recs = (uint4 *)p;
    hipError_t e = hipMemsetAsync(ws.tile_count, 0, 2 * tbytes, c->stream);   // counts + cursors
    if (e != hipSuccess) return e;

    const unsigned long long ord_base = 4ull * (seq0 - c->epoch_base);
    const unsigned long long ord_stride = 4ull * (unsigned long long)(c->cfg.seq_stride > 0 ? c->cfg.seq_stride : 1);
    const size_t n_rays = 4 * n;
    const unsigned int ray_blocks = (unsigned int)((n_rays + QT_BLOCK - 1) / QT_BLOCK);
    // upper bound on raster work items: every record in a full chunk, plus one partial chunk per tile
    size_t max_items = (4 * n_rays) / QT_CHUNK + (size_t)ws.n_tiles + QT_MAX_ITEMS_PAD;
    if (max_items > 4 * n_rays) max_items = 4 * n_rays;
    if (c->cfg.enable_counts) {
        hipLaunchKernelGGL(qs_rays_kernel<true>, dim3(ray_blocks), dim3(QT_BLOCK), 0, c->stream, n, c->b, c->geom, ws,
                           c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone, c->cfg.max_agent, c->d_counters);
    } else {
        hipLaunchKernelGGL(qs_rays_kernel<false>, dim3(ray_blocks), dim3(QT_BLOCK), 0, c->stream, n, c->b, c->geom, ws,
                           c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone, c->cfg.max_agent, c->d_counters);
    }
    hipLaunchKernelGGL(qs_tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, ws);
    hipLaunchKernelGGL(qs_scatter_kernel, dim3(ray_blocks), dim3(QT_BLOCK), 0, c->stream, n_rays, c->b, ws,
                       c->cfg.size, ord_base, ord_stride);
    if (c->cfg.enable_counts)
        hipLaunchKernelGGL(qs_raster_kernel<true>, dim3((unsigned int)max_items), dim3(QT_BLOCK), 0, c->stream, ws,
                           c->cfg.size, c->d_stamps, c->d_counts, c->d_counters);
    else
        hipLaunchKernelGGL(qs_raster_kernel<false>, dim3((unsigned int)max_items), dim3(QT_BLOCK), 0, c->stream, ws,
                           c->cfg.size, c->d_stamps, c->d_counts, c->d_counters);
    return hipGetLastError();
}
