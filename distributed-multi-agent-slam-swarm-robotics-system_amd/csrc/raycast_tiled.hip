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
// The sort is a counting sort with NO global atomics (measured: single-lane global atomics on a
// handful of hot lines cost ~6.5 ns each and dominated the first version, profiles/r01):
//   pass A  qs_rays_kernel      NWG persistent workgroups, one THREAD per packet: projection of
//                                the 4 rays (fp64 sincos), grid end points, zone / hit outputs,
//                                per-tile record counts in an LDS histogram, written as one row
//                                of the table T[wg][tile].  Rays longer than one tile (fine
//                                resolutions) are written to the grid directly.
//   pass B1 qs_table_scan_kernel per tile: exclusive scan of its column of T over the workgroups.
//   pass B2 (inside pass C)      exclusive scans over the tiles: record base and work-item base.
//   pass C  qs_scatter_kernel    same workgroup decomposition as A: LDS cursors (= table row +
//                                tile base) hand out record slots; one 16-byte record per
//                                overlapped tile (<= 2x2 per ray).
//   pass D  qs_raster_kernel     per work item (tile, <= QT_CHUNK records): LDS raster + merge.
//
// HBM traffic per packet (4 rays, ~1.4 tile records per ray): 45 B decoded fields + 64 B ray
// end points written and read + ~90 B records written and read, independent of how many cells
// each ray covers; the per-cell work happens in LDS.
#include "qs_internal.h"
#include "raycast_common.h"
#include <cstdlib>

#define QT_TILE 64                       // tile edge in cells: 64 x 64 x u32 = 16 KiB of LDS
#define QT_TILE_SHIFT 6
#define QT_CELLS (QT_TILE * QT_TILE)
#define QT_CHUNK 2048                    // records per raster work item
#ifndef QT_BLOCK
#define QT_BLOCK 512                     // raster workgroup: 8 waves share one 32 KiB LDS tile (A/B: 256 -> 512 threads = -5..-18 % stage time)
#endif
#ifndef QT_BIN_BLOCK
#define QT_BIN_BLOCK 1024                // pass A / C workgroup
#endif
#ifndef QT_MAX_WG
#define QT_MAX_WG 512                    // persistent workgroups of pass A / C (2 per CU)
#endif
#define QT_MAX_TILES 16384               // LDS histogram limit: 64 KiB (8192^2 cells)
#define QT_NO_RAY (-32768)              // x0 of "no ray": grids are <= 16384 cells wide, rays < 64 cells past an edge
#ifndef QT_RASTER_WGS
#define QT_RASTER_WGS 1024               // persistent raster workgroups (4 per CU)
#endif
#ifndef QT_PITCH
#define QT_PITCH 67                      // LDS row pitch in cells: bank = (x + 3 y) mod 32, see qs_raster_kernel
#endif
#define QT_LDS_CELLS (QT_TILE * QT_PITCH)

struct QtWorkspace {
    unsigned int *table;         // [nwg][n_tiles] per-workgroup record counts -> exclusive offsets
    unsigned int *tile_count;    // [n_tiles]   records per tile
    unsigned int *tile_base;     // [n_tiles+1] exclusive scan of tile_count
    unsigned int *chunk_base;    // [n_tiles+1] exclusive scan of ceil(count / QT_CHUNK)
    uint2 *rays;                 // [4n]        absolute grid end points, i16 x 4: (x0 | y0 << 16, x1 | y1 << 16);
                                 //             x0 = QT_NO_RAY: none
    uint2 *recs;                 // [16n]       tile records: tile-relative end points, i8 x 4; stamp | observed-hit bit
    int tiles_x, n_tiles, nwg;
    size_t pk_per_wg;            // packets per workgroup (multiple of 64)
    unsigned int *dirty;         // sparse fuse: bitmap of written 4 x 16-cell blocks (QsGeom::dirty), or nullptr
    int dirty_pitch;
};

__device__ inline void qt_tile_range(int x0, int y0, int x1, int y1, int size, int &tx_lo, int &tx_hi,
                                     int &ty_lo, int &ty_hi)
{
    const int xlo = max(min(x0, x1), 0), xhi = min(max(x0, x1), size - 1);
    const int ylo = max(min(y0, y1), 0), yhi = min(max(y0, y1), size - 1);
    tx_lo = xlo >> QT_TILE_SHIFT; tx_hi = xhi >> QT_TILE_SHIFT;
    ty_lo = ylo >> QT_TILE_SHIFT; ty_hi = yhi >> QT_TILE_SHIFT;
}

// ---- pass A -----------------------------------------------------------------------------------
template <bool COUNTS>
__global__ void __launch_bounds__(QT_BIN_BLOCK)
qs_rays_kernel(size_t n, QsBatch b, QsGeom geo, QtWorkspace ws, unsigned int *__restrict__ stamps,
               unsigned long long *__restrict__ counts, unsigned long long ord_base, unsigned long long ord_stride,
               unsigned long long *__restrict__ zone, int max_agent, unsigned long long *__restrict__ counters)
{
    extern __shared__ unsigned int s_hist[];                       // [n_tiles]
    __shared__ double s_zone[QS_MAX_AGENT + 1][4];
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    for (int t = tid; t < ws.n_tiles; t += QT_BIN_BLOCK) s_hist[t] = 0;
    for (int t = tid; t <= max_agent; t += QT_BIN_BLOCK) QS_ZONE_LDS_INIT(s_zone, t);
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();

    // one THREAD per ray (4 per packet): small per-thread state, so twice the waves of a
    // thread-per-packet form fit a SIMD, and every store of the pass is a full coalesced row
    const size_t r0 = 4 * (size_t)blockIdx.x * ws.pk_per_wg;
    const size_t r1 = (r0 + 4 * ws.pk_per_wg < 4 * n) ? r0 + 4 * ws.pk_per_wg : 4 * n;
    unsigned int my_cells = 0, my_rays = 0, my_hits = 0;
    // the inputs of the NEXT ray are requested before the current one is processed: a thread's rays
    // are 256 packets apart, so every iteration would otherwise start with a full HBM round trip
    unsigned char acc_n = 0, agent_n = 0;
    double rx_n = 0, ry_n = 0, yaw_n = 0;
    float df_n = 0;
    if (r0 + tid < r1) {
        const size_t r = r0 + tid, i = r >> 2;
        acc_n = b.map_ok[i]; agent_n = b.agent[i]; rx_n = b.rx[i]; ry_n = b.ry[i]; yaw_n = b.yaw[i];
        df_n = ((const float *)b.dist)[r];
    }
    for (size_t r = r0 + tid; r < r1; r += QT_BIN_BLOCK) {
        const size_t i = r >> 2;
        const int s = (int)(r & 3);
        const unsigned char acc = acc_n;
        const int agent = agent_n;
        const double rx = rx_n, ry = ry_n, yaw = yaw_n;
        const float df = df_n;
        if (r + QT_BIN_BLOCK < r1) {
            const size_t rn = r + QT_BIN_BLOCK, in = rn >> 2;
            acc_n = b.map_ok[in]; agent_n = b.agent[in]; rx_n = b.rx[in]; ry_n = b.ry[in]; yaw_n = b.yaw[in];
            df_n = ((const float *)b.dist)[rn];
        }
        uint2 rec = make_uint2((unsigned int)QT_NO_RAY & 0xffffu, 0u);
        bool valid = false;
        if (acc) {
            const QsRay ray = qs_project_ray(rx, ry, yaw, (double)df, s, geo);
            valid = ray.valid;
            // compute_bounding_box over hits U path (:702-706, :930-940): exact min/max, any order
            if (s == 0) qs_zone_point(s_zone, agent, rx, ry);                // paths[agent].append  :878-879
            if (valid) qs_zone_point(s_zone, agent, ray.ex, ray.ey);         // point_clouds[..].append  :892
            my_hits += valid ? 1u : 0u;
            my_rays++;
            QsLine ln;
            if (b.edge && qs_edge_ray(ray, geo) &&
                qs_edge_defer(b, rx, ry, yaw, df, (unsigned int)((ord_base + ord_stride * i + s + 1) << 1))) {
                // the host decides this ray's cells (qs_api.hip: flush_edge_rays)
            } else if (qs_line_setup(ray, rx, ry, geo, ln)) {
                if (ln.dx < QT_TILE && ln.dy < QT_TILE) {
                    // the ray's cells lie in at most 2 x 2 tiles
                    int tx_lo, tx_hi, ty_lo, ty_hi;
                    qt_tile_range(ln.x0, ln.y0, ln.x1, ln.y1, geo.size, tx_lo, tx_hi, ty_lo, ty_hi);
                    const int t00 = ty_lo * ws.tiles_x + tx_lo;
                    const bool wx = tx_hi > tx_lo, wy = ty_hi > ty_lo;      // dx, dy < 64: at most one boundary each
                    atomicAdd(&s_hist[t00], 1u);
                    if (wx) atomicAdd(&s_hist[t00 + 1], 1u);
                    if (wy) atomicAdd(&s_hist[t00 + ws.tiles_x], 1u);
                    if (wx && wy) atomicAdd(&s_hist[t00 + ws.tiles_x + 1], 1u);
                    rec = make_uint2(((unsigned int)ln.x0 & 0xffffu) | ((unsigned int)ln.y0 << 16),
                                     ((unsigned int)ln.x1 & 0xffffu) | ((unsigned int)ln.y1 << 16));
                } else {
                    // long ray (fine resolution): direct global atomics, as raycast.hip
                    const unsigned int key_free = (unsigned int)((ord_base + ord_stride * i + s + 1) << 1);
                    int x = ln.x0, y = ln.y0, err = ln.dx - ln.dy;
                    for (;;) {
                        const bool last = (x == ln.x1 && y == ln.y1);
                        if ((!last || valid) && x >= 0 && x < geo.size && y >= 0 && y < geo.size) {
                            const size_t c = (size_t)y * geo.size + x;
                            atomicMax(&stamps[c], key_free | (last ? 1u : 0u));
                            qs_mark_dirty(geo, x, y);
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
        ws.rays[r] = rec;
        b.hit_valid[r] = valid ? 1 : 0;
    }
    if (my_rays) atomicAdd(&s_cnt[0], my_rays);
    if (my_cells) atomicAdd(&s_cnt[1], my_cells);
    if (my_hits) atomicAdd(&s_cnt[2], my_hits);
    __syncthreads();
    unsigned int *row = ws.table + (size_t)blockIdx.x * ws.n_tiles;
    for (int t = tid; t < ws.n_tiles; t += QT_BIN_BLOCK) row[t] = s_hist[t];
    for (int t = tid; t <= max_agent; t += QT_BIN_BLOCK) qs_zone_commit(s_zone, t, zone);
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&counters[QS_CNT_RAYS], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&counters[QS_CNT_HITS], (unsigned long long)s_cnt[2]);
    }
}

// ---- pass B1: per tile, exclusive scan of its table column over the workgroups -------------------
// One 1024-thread workgroup per 64 tiles: thread (seg, t) owns rows [seg * R, (seg + 1) * R) of
// tile column t (R = ceil(nwg / 16) <= 32); a wave reads one 256-byte row segment per instruction
// and all R loads of a thread are in flight together, so the column costs one HBM round trip.  The 16
// segment sums of a column are combined through LDS and the running offsets written back.
#define QT_SCAN_TILES 64
#define QT_SCAN_SEGS 16
#define QT_SCAN_ROWS (QT_MAX_WG / QT_SCAN_SEGS)

__global__ void __launch_bounds__(QT_SCAN_TILES * QT_SCAN_SEGS)
qs_table_scan_kernel(QtWorkspace ws)
{
    __shared__ unsigned int s_part[QT_SCAN_SEGS][QT_SCAN_TILES];
    const int tid = threadIdx.x;
    const int lt = tid & (QT_SCAN_TILES - 1), seg = tid >> 6;
    const int t = blockIdx.x * QT_SCAN_TILES + lt;
    const int R = (ws.nwg + QT_SCAN_SEGS - 1) / QT_SCAN_SEGS;
    const int w0 = seg * R;
    const bool live = t < ws.n_tiles;
    unsigned int v[QT_SCAN_ROWS];
    unsigned int sum = 0;
    #pragma unroll
    for (int q = 0; q < QT_SCAN_ROWS; q++) {
        v[q] = (live && q < R && w0 + q < ws.nwg) ? ws.table[(size_t)(w0 + q) * ws.n_tiles + t] : 0u;
    }
    #pragma unroll
    for (int q = 0; q < QT_SCAN_ROWS; q++) sum += v[q];
    s_part[seg][lt] = sum;
    __syncthreads();
    unsigned int run = 0, total = 0;
    #pragma unroll
    for (int q = 0; q < QT_SCAN_SEGS; q++) { const unsigned int p = s_part[q][lt]; if (q < seg) run += p; total += p; }
    #pragma unroll
    for (int q = 0; q < QT_SCAN_ROWS; q++) {
        if (live && q < R && w0 + q < ws.nwg) ws.table[(size_t)(w0 + q) * ws.n_tiles + t] = run;
        run += v[q];
    }
    if (live && seg == 0) ws.tile_count[t] = total;
}

// ---- pass C: scatter tile records ---------------------------------------------------------------
__global__ void __launch_bounds__(QT_BIN_BLOCK)
qs_scatter_kernel(size_t n, QsBatch b, QtWorkspace ws, int size, unsigned long long ord_base,
                  unsigned long long ord_stride)
{
    extern __shared__ unsigned int s_cur[];                        // [n_tiles] next record slot per tile
    __shared__ unsigned int s_wrec[QT_BIN_BLOCK / QS_WAVE], s_wchk[QT_BIN_BLOCK / QS_WAVE];
    const int tid = threadIdx.x, lane = tid & (QS_WAVE - 1), wave = tid >> 6;
    const unsigned int *row = ws.table + (size_t)blockIdx.x * ws.n_tiles;
    // pass B2, folded in: every workgroup scans the tile counts itself (16 KB of L2-resident reads, two
    // barriers) instead of waiting for a one-workgroup kernel in between: record base of every tile into
    // the LDS cursors; workgroup 0 also publishes record base and work-item base for the raster pass
    {
        const int per = (ws.n_tiles + QT_BIN_BLOCK - 1) / QT_BIN_BLOCK;
        const int lo = min(tid * per, ws.n_tiles), hi = min(lo + per, ws.n_tiles);
        unsigned int a = 0, c = 0;
        for (int t = lo; t < hi; t++) { const unsigned int v = ws.tile_count[t]; a += v; c += (v + QT_CHUNK - 1) / QT_CHUNK; }
        unsigned int ia = a, ic = c;                               // inclusive scan inside the wave
        #pragma unroll
        for (int off = 1; off < QS_WAVE; off <<= 1) {
            const unsigned int va = __shfl_up(ia, off), vc = __shfl_up(ic, off);
            if (lane >= off) { ia += va; ic += vc; }
        }
        if (lane == QS_WAVE - 1) { s_wrec[wave] = ia; s_wchk[wave] = ic; }
        __syncthreads();
        unsigned int ra = ia - a, rc = ic - c;
        for (int w = 0; w < wave; w++) { ra += s_wrec[w]; rc += s_wchk[w]; }
        const bool pub = blockIdx.x == 0;
        for (int t = lo; t < hi; t++) {
            const unsigned int v = ws.tile_count[t];
            s_cur[t] = ra + row[t];
            if (pub) { ws.tile_base[t] = ra; ws.chunk_base[t] = rc; }
            ra += v; rc += (v + QT_CHUNK - 1) / QT_CHUNK;
        }
        if (pub && tid == QT_BIN_BLOCK - 1) { ws.tile_base[ws.n_tiles] = ra; ws.chunk_base[ws.n_tiles] = rc; }
    }
    __syncthreads();
    const size_t r0 = 4 * (size_t)blockIdx.x * ws.pk_per_wg;
    const size_t r1 = (r0 + 4 * ws.pk_per_wg < 4 * n) ? r0 + 4 * ws.pk_per_wg : 4 * n;
    for (size_t r = r0 + tid; r < r1; r += QT_BIN_BLOCK) {
        const uint2 ray = ws.rays[r];
        const int x0 = (short)(ray.x & 0xffffu), y0 = (short)(ray.x >> 16);
        if (x0 == QT_NO_RAY) continue;
        const int x1 = (short)(ray.y & 0xffffu), y1 = (short)(ray.y >> 16);
        int tx_lo, tx_hi, ty_lo, ty_hi;
        qt_tile_range(x0, y0, x1, y1, size, tx_lo, tx_hi, ty_lo, ty_hi);
        // stamp of the ray's free cells; bit 0 (the occupied bit of a cell stamp) carries "the end
        // cell was observed" into the raster pass
        const unsigned int key = (unsigned int)((ord_base + ord_stride * (r >> 2) + (r & 3) + 1) << 1) | (b.hit_valid[r] ? 1u : 0u);
        // dx, dy < 64: the tile box is 1 x 1, 2 x 1, 1 x 2 or 2 x 2 (same enumeration as pass A)
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const int tx = tx_lo + (q & 1), ty = ty_lo + (q >> 1);
            if (tx > tx_hi || ty > ty_hi) continue;
            const unsigned int slot = atomicAdd(&s_cur[ty * ws.tiles_x + tx], 1u);
            const int ox = tx << QT_TILE_SHIFT, oy = ty << QT_TILE_SHIFT;   // tile origin: |coord - origin| < 128
            ws.recs[slot] = make_uint2(((unsigned int)(x0 - ox) & 0xffu) | (((unsigned int)(y0 - oy) & 0xffu) << 8) |
                                       (((unsigned int)(x1 - ox) & 0xffu) << 16) | ((unsigned int)(y1 - oy) << 24), key);
        }
    }
}

// ---- experiment (off in production: -DQT_CLIP=1 builds it, tools/ab_raster_clip.sh measures it) ----------
// Clip a record's walk to its tile in closed form: the walk of dual_bot_mapper.py:166-178 in major / minor form has, after t
// steps, minor offset m(t) = (2 t dmin + dmaj - 1) div (2 dmaj) and error term E(t) = dmaj - dmin + m(t) dmaj - t dmin
// (checked on the CPU against the step-by-step walk for every |d| <= 40; the in-tile steps are one interval [ta, tb] because both coordinates are monotone).
#ifndef QT_CLIP
#define QT_CLIP 0
#endif
__device__ inline int qt_idiv(int n, int d) { return __float2int_rz(__fdividef((float)n + 0.5f, (float)d)); }   // exact: 0 <= n < 2^14, 0 < d < 2^8
__device__ inline int qt_cdiv(int a, int b) { return qt_idiv(a + b - 1, b); }
// first in-tile step ta and the number of in-tile FREE cells of the walk (x, y) -> (x1, y1) in a tile of tw x th cells
__device__ inline int qt_clip_span(int x, int y, int x1, int y1, int tw, int th, int &ta)
{
    const int dx = abs(x1 - x), dy = abs(y1 - y), sx = x < x1 ? 1 : -1, sy = y < y1 ? 1 : -1;
    const bool xmaj = dx >= dy;
    const int dmaj = xmaj ? dx : dy, dmin = xmaj ? dy : dx;
    const int cm0 = xmaj ? x : y, sm = xmaj ? sx : sy, Wm = xmaj ? tw : th;
    const int cn0 = xmaj ? y : x, sn = xmaj ? sy : sx, Wn = xmaj ? th : tw;
    int a = 0, b = dmaj - 1;
    if (sm > 0) { a = max(a, -cm0); b = min(b, Wm - 1 - cm0); } else { a = max(a, cm0 - (Wm - 1)); b = min(b, cm0); }
    const int lo = sn > 0 ? max(0, -cn0) : max(0, cn0 - (Wn - 1)), hi = sn > 0 ? Wn - 1 - cn0 : cn0;
    if (hi < 0) b = -1;
    else if (dmin == 0) { if (lo > 0) b = -1; }
    else {
        if (lo > 0) a = max(a, qt_cdiv(2 * dmaj * lo - dmaj + 1, 2 * dmin));
        b = min(b, qt_cdiv(2 * dmaj * (hi + 1) - dmaj + 1, 2 * dmin) - 1);
    }
    ta = a;
    return max(0, b - a + 1);
}
#if QT_CLIP
// experiment only: the records of every tile re-ordered by clipped span (ascending), so that the 64 records a wave walks
// together have walks of similar length -- what a regrouping scatter pass would deliver, without its cost
__global__ void __launch_bounds__(256)
qs_clip_sort_kernel(QtWorkspace ws, int size, uint2 *__restrict__ tmp)
{
    __shared__ unsigned int s_bin[66];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const unsigned int n = ws.tile_count[tile], base = ws.tile_base[tile];
    if (n == 0) return;
    const int tx0 = (tile % ws.tiles_x) << QT_TILE_SHIFT, ty0 = (tile / ws.tiles_x) << QT_TILE_SHIFT;
    const int tw = min(QT_TILE, size - tx0), th = min(QT_TILE, size - ty0);
    for (int t = tid; t < 66; t += 256) s_bin[t] = 0;
    __syncthreads();
    auto span_of = [&](uint2 rec) {
        const int x = (signed char)(rec.x & 0xffu), y = (signed char)((rec.x >> 8) & 0xffu);
        const int x1 = (signed char)((rec.x >> 16) & 0xffu), y1 = (signed char)(rec.x >> 24);
        int ta; return qt_clip_span(x, y, x1, y1, tw, th, ta);
    };
    for (unsigned int j = tid; j < n; j += 256) atomicAdd(&s_bin[span_of(ws.recs[base + j]) + 1], 1u);
    __syncthreads();
    if (tid == 0) for (int t = 1; t < 66; t++) s_bin[t] += s_bin[t - 1];
    __syncthreads();
    for (unsigned int j = tid; j < n; j += 256) { const uint2 r = ws.recs[base + j]; tmp[base + atomicAdd(&s_bin[span_of(r)], 1u)] = r; }
    __syncthreads();
    for (unsigned int j = tid; j < n; j += 256) ws.recs[base + j] = tmp[base + j];
}
#endif

// ---- pass D: LDS raster + merge -------------------------------------------------------------------
// Persistent workgroups: workgroup w takes the contiguous run of work items [w * per, (w + 1) * per).
// Items are in tile order, so consecutive items mostly belong to the same tile (two robots in one
// room: ~300 items per tile); the LDS tile keeps accumulating across them and is merged into the HBM
// grid only when the tile changes.  Every merge of a shared tile is ~3000 device-scope atomics on
// the same few hundred cache lines as every other workgroup of that tile, so merging once per run
// instead of once per item takes most of that traffic away (and the empty workgroups of a
// one-item-per-workgroup launch with it).
#define QT_FLUSH_ITEMS 31                // 16-bit LDS counters: <= 31 * QT_CHUNK writes per cell between merges

template <bool COUNTS>
__device__ inline void qt_merge_tile(unsigned int *s_stamp, unsigned int *s_cnt, unsigned int *s_cells, int tid, int tile,
                                     bool exclusive, const QtWorkspace &ws, int size, unsigned int *__restrict__ stamps,
                                     unsigned long long *__restrict__ counts)
{
    // one 64-cell (256 B) grid row per wave-instruction.  A thread owns QT_CELLS / QT_BLOCK cells of
    // one column; all its grid loads are issued before the first store so an exclusive merge costs
    // one HBM round trip, not one per row.
    constexpr int PER = QT_CELLS / QT_BLOCK;
    constexpr int ROWS = QT_BLOCK / QT_TILE;              // rows covered by the workgroup per step
    const int tx0 = (tile % ws.tiles_x) << QT_TILE_SHIFT, ty0 = (tile / ws.tiles_x) << QT_TILE_SHIFT;
    const int mx = tid & (QT_TILE - 1), my = tid >> QT_TILE_SHIFT;
    // cells beyond the grid edge (partial tiles) are never written in LDS: v == 0 / kk == 0 there
    const size_t g0 = (size_t)(ty0 + my) * size + (tx0 + mx);
    const size_t gstep = (size_t)ROWS * size;
    unsigned int writes = 0;
    {   // stamps: cell-wise max  (:150 / :156 last writer wins)
        unsigned int v[PER], g[PER];
        #pragma unroll
        for (int q = 0; q < PER; q++) v[q] = s_stamp[(my + q * ROWS) * QT_PITCH + mx];
        if (ws.dirty) {
            // sparse fuse: a wave holds one 64-cell tile row per q = four blocks side by side; every cell written in LDS
            // has a stamp there, so the ballot of v != 0 is the row's written cells
            #pragma unroll
            for (int q = 0; q < PER; q++) {
                const unsigned long long wm = __builtin_amdgcn_ballot_w64(v[q] != 0);
                if (wm != 0 && (tid & (QS_WAVE - 1)) == 0) {
                    const unsigned int nib = ((wm & 0xffffull) ? 1u : 0u) | ((wm & 0xffff0000ull) ? 2u : 0u) |
                                             ((wm & 0xffff00000000ull) ? 4u : 0u) | ((wm >> 48) ? 8u : 0u);
                    const int gy = ty0 + my + q * ROWS;
                    unsigned int *w = ws.dirty + qs_dirty_word(tx0, gy, ws.dirty_pitch);
                    const unsigned int m = nib << ((tx0 / QS_DIRTY_BLOCK_W) & 31);   // tx0 is a multiple of 64: the nibble stays in one word
                    if ((__atomic_load_n(w, __ATOMIC_RELAXED) & m) != m) atomicOr(w, m);
                }
            }
        }
        if (exclusive) {
            #pragma unroll
            for (int q = 0; q < PER; q++) g[q] = v[q] != 0 ? stamps[g0 + q * gstep] : 0xffffffffu;
            #pragma unroll
            for (int q = 0; q < PER; q++) if (v[q] > g[q]) stamps[g0 + q * gstep] = v[q];
        } else {
            #pragma unroll
            for (int q = 0; q < PER; q++) if (v[q] != 0) atomicMax(&stamps[g0 + q * gstep], v[q]);
        }
    }
    if (COUNTS) {   // hit / miss counters: cell-wise sum
        unsigned int kk[PER];
        unsigned long long gc[PER];
        #pragma unroll
        for (int q = 0; q < PER; q++) {
            kk[q] = s_cnt[(my + q * ROWS) * QT_PITCH + mx];
            writes += (kk[q] >> 16) + (kk[q] & 0xffffu);
        }
        if (exclusive) {
            #pragma unroll
            for (int q = 0; q < PER; q++) gc[q] = kk[q] != 0 ? counts[g0 + q * gstep] : 0;
            #pragma unroll
            for (int q = 0; q < PER; q++)
                if (kk[q] != 0) counts[g0 + q * gstep] = gc[q] + (((unsigned long long)(kk[q] >> 16) << 32) | (kk[q] & 0xffffu));
        } else {
            #pragma unroll
            for (int q = 0; q < PER; q++)
                if (kk[q] != 0) atomicAdd(&counts[g0 + q * gstep], ((unsigned long long)(kk[q] >> 16) << 32) | (kk[q] & 0xffffu));
        }
        // cell writes (statistic): the LDS counters hold them
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) writes += __shfl_xor(writes, off);
        if ((tid & (QS_WAVE - 1)) == 0 && writes) atomicAdd(s_cells, writes);
    }
}

template <bool COUNTS>
__global__ void __launch_bounds__(QT_BLOCK, 8)
qs_raster_kernel(QtWorkspace ws, int size, unsigned int *__restrict__ stamps,
                 unsigned long long *__restrict__ counts, unsigned long long *__restrict__ counters)
{
    // + 64 scratch cells: a lane whose current cell is outside the tile (or whose walk has ended)
    // aims its two LDS atomics at its own scratch word instead of branching around them
    __shared__ unsigned int s_stamp[QT_LDS_CELLS + QS_WAVE];
    __shared__ unsigned int s_cnt[COUNTS ? QT_LDS_CELLS + QS_WAVE : 1];   // hi16 hits, lo16 misses
    __shared__ unsigned int s_cells;
    const int tid = threadIdx.x;
    const unsigned int n_items = ws.chunk_base[ws.n_tiles];
    const unsigned int per = (n_items + gridDim.x - 1) / gridDim.x;
    const unsigned int first = blockIdx.x * per;
    const unsigned int last = min(first + per, n_items);
    if (first >= last) return;
    if (tid == 0) s_cells = 0;
    // tile of the first work item: last t with chunk_base[t] <= first
    int tile;
    {
        int lo = 0, hi = ws.n_tiles;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ws.chunk_base[mid] <= first) lo = mid; else hi = mid; }
        tile = lo;
    }
    // Lane -> record mapping.  Records of a tile are in arrival order, so neighbouring records are the
    // four rays of one packet and the packets of the same bot a few centimetres apart: 64 consecutive
    // records walk nearly the same cells in lockstep and their LDS atomics serialise on the same
    // addresses.  Each wave therefore takes 8 groups of 8 consecutive records (two packets: 8
    // different rays), the groups QT_BLOCK / 8 records apart.
    const int lane = tid & (QS_WAVE - 1), wave = tid >> 6;
#if QT_CLIP
    const unsigned int slot = (unsigned int)tid;       // (experiment: records come sorted by clipped span -- neighbours walk alike in LENGTH, not in place)
#else
    const unsigned int slot = (unsigned int)((lane >> 3) * (QT_BLOCK / 8) + wave * 8 + (lane & 7));
#endif
    const int scratch4 = 4 * (QT_LDS_CELLS + lane);     // byte offset of this lane's scratch cell
    unsigned int wave_cells = 0;                        // wave-uniform count of cell writes (!COUNTS)
    int cur_tile = -1, since_flush = 0;

    for (unsigned int item = first; item < last; item++) {
        while (ws.chunk_base[tile + 1] <= item) tile++;          // tiles without records own no items
        if (tile != cur_tile || since_flush == QT_FLUSH_ITEMS) {
            if (cur_tile >= 0) {
                __syncthreads();
                const bool excl = ws.chunk_base[cur_tile] >= first && ws.chunk_base[cur_tile + 1] <= last;
                qt_merge_tile<COUNTS>(s_stamp, s_cnt, &s_cells, tid, cur_tile, excl, ws, size, stamps, counts);
                __syncthreads();
            }
            for (int c = tid; c < QT_LDS_CELLS / 4; c += QT_BLOCK) {
                ((uint4 *)s_stamp)[c] = make_uint4(0, 0, 0, 0);
                if (COUNTS) ((uint4 *)s_cnt)[c] = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
            cur_tile = tile; since_flush = 0;
        }
        since_flush++;
        const unsigned int rb = ws.tile_base[tile] + (item - ws.chunk_base[tile]) * QT_CHUNK;
        const unsigned int re = min(rb + QT_CHUNK, ws.tile_base[tile] + ws.tile_count[tile]);
        const int tx0 = (tile % ws.tiles_x) << QT_TILE_SHIFT, ty0 = (tile / ws.tiles_x) << QT_TILE_SHIFT;
        const int tw = min(QT_TILE, size - tx0), th = min(QT_TILE, size - ty0);
        const unsigned int tw4 = 4u * (unsigned int)tw, tha = 4u * QT_PITCH * (unsigned int)th;
        for (unsigned int j0 = rb; j0 < re; j0 += QT_BLOCK) {
            const unsigned int j = j0 + slot;
            // Walk state in major/minor form.  The walk of dual_bot_mapper.py:166-178 advances its
            // major axis in EVERY iteration and reaches (x1, y1) after exactly max(dx, dy) of them
            // (property-tested on the CPU: tests/, test_bresenham_major_axis_property), so with
            //   E = err (x-major, dx >= dy) or -err (y-major)
            // the pair of tests `e2 > -dy`, `e2 < dx` (:172-177) is ONE test per cell: the minor axis
            // steps iff 2 E < dmaj, i.e. E < (dmaj + 1) >> 1, and E += minor ? dmaj - dmin : -dmin.
            // The last cell (x1, y1) is the only one that can be marked occupied (:150): it is written
            // before the loop, and `x == x1 and y == y1` (:169) becomes a countdown over the
            // k = max(dx, dy) free cells.
            // x4 = 4 x and ya = 4 QT_PITCH y: the cell's LDS byte offset is their sum
            int x4 = 0, ya = 0, k = 0, E = 0, H = 0, incA = 0, incB = 0, sx_c = 0, sx_n = 0, sy_c = 0, sy_n = 0;
            unsigned int key_free = 0;
            bool wl = false;
            if (j < re) {
                const uint2 rec = ws.recs[j];
                const int x = (signed char)(rec.x & 0xffu), x1 = (signed char)((rec.x >> 16) & 0xffu);
                const int y1 = (signed char)(rec.x >> 24);
                const int y = (signed char)((rec.x >> 8) & 0xffu);
                key_free = rec.y & ~1u;
                const int dx = abs(x1 - x), dy = abs(y1 - y);                  // :161-162
                const int sx4 = x < x1 ? 4 : -4, sy = y < y1 ? 4 * QT_PITCH : -4 * QT_PITCH;   // :163-164, in bytes
                const bool xmaj = dx >= dy;
                const int dmaj = xmaj ? dx : dy, dmin = xmaj ? dy : dx;
                k = dmaj; E = dmaj - dmin; H = (dmaj + 1) >> 1; incA = dmaj - dmin; incB = -dmin;
                sx_c = sx4; sx_n = xmaj ? sx4 : 0; sy_c = sy; sy_n = xmaj ? 0 : sy;
                x4 = x << 2; ya = y * (4 * QT_PITCH);
#if QT_CLIP
                {   // start at the first in-tile step, walk the in-tile steps only
                    int ta;
                    k = qt_clip_span(x, y, x1, y1, tw, th, ta);
                    const int mn = dmaj ? qt_idiv(2 * ta * dmin + dmaj - 1, 2 * dmaj) : 0;
                    E = dmaj - dmin + mn * dmaj - ta * dmin;
                    const int ax = xmaj ? ta : mn, ay = xmaj ? mn : ta;          // steps taken along x / y
                    x4 += ax * sx4; ya += ay * sy;
                }
#endif
                wl = (rec.y & 1u) && (unsigned int)x1 < (unsigned int)tw && (unsigned int)y1 < (unsigned int)th;
                if (wl) {                                                      // :148-150 occupied end cell
                    const int c = y1 * QT_PITCH + x1;
                    atomicMax(&s_stamp[c], key_free | 1u);
                    if (COUNTS) atomicAdd(&s_cnt[c], 0x10000u);
                }
            }
            if (!COUNTS) wave_cells += (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(wl));
            for (int it = 0; __any(it < k); it++) {                            // :152-156 free cells
#if QT_CLIP
                const bool w = it < k;                                          // (every remaining step is inside the tile)
#else
                const bool w = it < k && (unsigned int)x4 < tw4 && (unsigned int)ya < tha;
#endif
                const int a = w ? ya + x4 : scratch4;
                atomicMax((unsigned int *)((char *)s_stamp + a), key_free);
                if (COUNTS) atomicAdd((unsigned int *)((char *)s_cnt + a), 1u);
                if (!COUNTS) wave_cells += (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w));
                const bool minor = E < H;
                E += minor ? incA : incB;
                x4 += minor ? sx_c : sx_n;
                ya += minor ? sy_c : sy_n;
            }
        }
    }
    if (!COUNTS && lane == 0 && wave_cells) atomicAdd(&s_cells, wave_cells);
    __syncthreads();
    {
        const bool excl = ws.chunk_base[cur_tile] >= first && ws.chunk_base[cur_tile + 1] <= last;
        qt_merge_tile<COUNTS>(s_stamp, s_cnt, &s_cells, tid, cur_tile, excl, ws, size, stamps, counts);
    }
    __syncthreads();
    if (tid == 0 && s_cells) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)s_cells);
}

// ---- host side ------------------------------------------------------------------------------------
static inline size_t qt_align(size_t v) { return (v + 255) & ~(size_t)255; }

static inline int qt_n_tiles(const qs_ctx *c)
{
    const int tiles_x = (c->cfg.size + QT_TILE - 1) / QT_TILE;
    return tiles_x * tiles_x;
}

bool qs_tiled_supported(const qs_ctx *c) { return qt_n_tiles(c) <= QT_MAX_TILES; }

size_t qs_tiled_workspace_bytes(const qs_ctx *c, size_t n)
{
    const size_t n_tiles = (size_t)qt_n_tiles(c);
    return qt_align((size_t)QT_MAX_WG * n_tiles * sizeof(unsigned int)) + 3 * qt_align((n_tiles + 1) * sizeof(unsigned int)) +
           qt_align(4 * n * sizeof(uint2)) + qt_align(16 * n * sizeof(uint2));
}

hipError_t qs_launch_raycast_tiled(qs_ctx *c, size_t n, uint64_t seq0)
{
    if (n == 0) return hipSuccess;
    if (!qs_tiled_supported(c)) return qs_launch_raycast_direct(c, n, seq0);
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
    // packets per workgroup: a multiple of 64, at least 256, and at most QT_MAX_WG workgroups
    size_t per = (n + QT_MAX_WG - 1) / QT_MAX_WG;
    per = per < 256 ? 256 : ((per + 63) & ~(size_t)63);
    ws.pk_per_wg = per;
    ws.nwg = (int)((n + per - 1) / per);
    const size_t tbytes = qt_align(((size_t)ws.n_tiles + 1) * sizeof(unsigned int));
    char *p = (char *)c->d_bin_ws;
    ws.table = (unsigned int *)p; p += qt_align((size_t)QT_MAX_WG * ws.n_tiles * sizeof(unsigned int));
    ws.tile_count = (unsigned int *)p; p += tbytes;
    ws.tile_base = (unsigned int *)p; p += tbytes;
    ws.chunk_base = (unsigned int *)p; p += tbytes;
    ws.rays = (uint2 *)p; p += qt_align(4 * c->cap_batch * sizeof(uint2));
    ws.recs = (uint2 *)p;
    ws.dirty = c->geom.dirty; ws.dirty_pitch = c->geom.dirty_pitch;

    const unsigned long long ord_base = 4ull * (seq0 - c->epoch_base);
    const unsigned long long ord_stride = 4ull * (unsigned long long)(c->cfg.seq_stride > 0 ? c->cfg.seq_stride : 1);
    const size_t n_rays = 4 * n;
    const size_t lds = (size_t)ws.n_tiles * sizeof(unsigned int);
    // upper bound on raster work items: every record in a full chunk, plus one partial chunk per tile
    size_t max_items = (4 * n_rays) / QT_CHUNK + (size_t)ws.n_tiles + 8;
    if (max_items > 4 * n_rays) max_items = 4 * n_rays;
    if (lds > 32 * 1024) {
        // large grids (up to 8192^2: 16384 tiles) need more dynamic LDS than the 64 KiB default allows
        // next to the static arrays; the CU has 160 KiB
        hipError_t ea = hipFuncSetAttribute((const void *)qs_rays_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea == hipSuccess) ea = hipFuncSetAttribute((const void *)qs_rays_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea == hipSuccess) ea = hipFuncSetAttribute((const void *)qs_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    StageTimer t_rays(c, QS_STAGE_RC_RAYS);
    if (c->cfg.enable_counts) {
        hipLaunchKernelGGL(qs_rays_kernel<true>, dim3(ws.nwg), dim3(QT_BIN_BLOCK), lds, c->stream, n, c->b, c->geom, ws,
                           c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone, c->cfg.max_agent, c->d_counters);
    } else {
        hipLaunchKernelGGL(qs_rays_kernel<false>, dim3(ws.nwg), dim3(QT_BIN_BLOCK), lds, c->stream, n, c->b, c->geom, ws,
                           c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone, c->cfg.max_agent, c->d_counters);
    }
    t_rays.stop();
    StageTimer t_sort(c, QS_STAGE_RC_SORT);
    hipLaunchKernelGGL(qs_table_scan_kernel, dim3((ws.n_tiles + QT_SCAN_TILES - 1) / QT_SCAN_TILES),
                       dim3(QT_SCAN_TILES * QT_SCAN_SEGS), 0, c->stream, ws);
    hipLaunchKernelGGL(qs_scatter_kernel, dim3(ws.nwg), dim3(QT_BIN_BLOCK), lds, c->stream, n, c->b, ws,
                       c->cfg.size, ord_base, ord_stride);
    t_sort.stop();
    // QS_RASTER_WGS (environment, read once): fewer persistent raster workgroups than the default -- a tuning
    // knob, and how the tests reach the long-run paths (tile changes inside a run, the 31-item flush)
    static const int env_wgs = [] { const char *e = getenv("QS_RASTER_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : QT_RASTER_WGS; }();
    const unsigned int raster_wgs = (unsigned int)(max_items < (size_t)env_wgs ? max_items : (size_t)env_wgs);
#if QT_CLIP
    {   // experiment: regroup by clipped span, untimed (what a regrouping scatter would deliver)
        static const bool do_sort = [] { const char *e = getenv("QS_RASTER_SORT"); return !e || atoi(e) != 0; }();
        static uint2 *d_tmp = nullptr; static size_t tmp_cap = 0;
        if (do_sort) {
            if (tmp_cap < 16 * c->cap_batch) { hipStreamSynchronize(c->stream); hipFree(d_tmp); hipMalloc((void **)&d_tmp, 16 * c->cap_batch * sizeof(uint2)); tmp_cap = 16 * c->cap_batch; }
            hipLaunchKernelGGL(qs_clip_sort_kernel, dim3(ws.n_tiles), dim3(256), 0, c->stream, ws, c->cfg.size, d_tmp);
        }
    }
#endif
    StageTimer t_raster(c, QS_STAGE_RC_RASTER);
    if (c->cfg.enable_counts)
        hipLaunchKernelGGL(qs_raster_kernel<true>, dim3(raster_wgs), dim3(QT_BLOCK), 0, c->stream, ws,
                           c->cfg.size, c->d_stamps, c->d_counts, c->d_counters);
    else
        hipLaunchKernelGGL(qs_raster_kernel<false>, dim3(raster_wgs), dim3(QT_BLOCK), 0, c->stream, ws,
                           c->cfg.size, c->d_stamps, c->d_counts, c->d_counters);
    t_raster.stop();
    return hipGetLastError();
}
