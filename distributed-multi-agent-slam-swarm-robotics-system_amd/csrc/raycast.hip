// raycast.hip -- K1 (direct form): 4-ray inverse sensor model into the global grid with one
// global atomic per cell.  Semantics: server_nodes/dual_bot_mapper.py:882-903 (projection and
// trust filter), :136-179 (update_ray + the strict-inequality Bresenham variant), :121-125
// (world_to_grid: fp64 subtract, fp64 divide, truncate).
//
// Order dependence: the reference overwrites cells, so the last write in arrival order wins.
// Every cell write carries a stamp (ordinal << 1) | occ with ordinal = 4*seq + sensor + 1;
// atomicMax makes any schedule produce the sequential answer.  Cells of one ray are distinct,
// so ties never occur.  Hit/miss counters are a 64-bit atomicAdd (hi32 hits, lo32 misses).
//
// This direct kernel is the reference point for correctness and the fallback for streams
// with no spatial locality; raycast_tiled.hip is the LDS-staged production path.
#include "qs_internal.h"
#include "raycast_common.h"

#define RC_BLOCK 256

template <bool COUNTS>
__global__ void __launch_bounds__(RC_BLOCK)
qs_raycast_direct_kernel(size_t n, QsBatch b, QsGeom geo, unsigned int *__restrict__ stamps,
                         unsigned long long *__restrict__ counts, unsigned long long ord_base,
                         unsigned long long ord_stride, unsigned long long *__restrict__ zone, int max_agent,
                         unsigned long long *__restrict__ counters)
{
    __shared__ double s_zone[QS_MAX_AGENT + 1][4];
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    for (int t = tid; t <= max_agent; t += RC_BLOCK) QS_ZONE_LDS_INIT(s_zone, t);
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();

    const size_t r = (size_t)blockIdx.x * RC_BLOCK + tid;   // ray id: 4 per datagram
    const size_t i = r >> 2;
    const int s = (int)(r & 3);
    unsigned int my_cells = 0, my_ray = 0, my_hit = 0;
    if (i < n && b.map_ok[i]) {
        const double rx = b.rx[i], ry = b.ry[i], yaw = b.yaw[i];
        const float4 d4 = b.dist[i];
        const float df = s == 0 ? d4.x : (s == 1 ? d4.y : (s == 2 ? d4.z : d4.w));
        const int agent = b.agent[i];
        QsRay ray = qs_project_ray(rx, ry, yaw, (double)df, s, geo);
        if (s == 0) qs_zone_point(s_zone, agent, rx, ry);                          // paths[agent].append  :878-879
        if (ray.valid) qs_zone_point(s_zone, agent, ray.ex, ray.ey);               // point_clouds[..].append  :892
        my_hit = ray.valid ? 1u : 0u;
        my_ray = 1;
        const unsigned int key_free = (unsigned int)((ord_base + ord_stride * i + s + 1) << 1);
        QsLine ln;
        if (b.edge && qs_edge_ray(ray, geo) && qs_edge_defer(b, rx, ry, yaw, df, key_free)) {
            // the host decides this ray's cells (qs_api.hip: flush_edge_rays)
        } else if (qs_line_setup(ray, rx, ry, geo, ln)) {
            int x = ln.x0, y = ln.y0, err = ln.dx - ln.dy;
            for (;;) {
                const bool last = (x == ln.x1 && y == ln.y1);
                if ((!last || ray.valid) && x >= 0 && x < geo.size && y >= 0 && y < geo.size) {
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
    // block-level counters and zone flush
    if (my_ray) atomicAdd(&s_cnt[0], my_ray);
    if (my_cells) atomicAdd(&s_cnt[1], my_cells);
    if (my_hit) atomicAdd(&s_cnt[2], my_hit);
    __syncthreads();
    for (int t = tid; t <= max_agent; t += RC_BLOCK) qs_zone_commit(s_zone, t, zone);
    if (tid == 0) {
        if (s_cnt[0]) atomicAdd(&counters[QS_CNT_RAYS], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&counters[QS_CNT_HITS], (unsigned long long)s_cnt[2]);
    }
}

hipError_t qs_launch_raycast_direct(qs_ctx *c, size_t n, uint64_t seq0)
{
    if (n == 0) return hipSuccess;
    const unsigned int blocks = (unsigned int)((4 * n + RC_BLOCK - 1) / RC_BLOCK);
    const unsigned long long ord_base = 4ull * (seq0 - c->epoch_base);
    const unsigned long long ord_stride = 4ull * (unsigned long long)(c->cfg.seq_stride > 0 ? c->cfg.seq_stride : 1);
    if (c->cfg.enable_counts)
        hipLaunchKernelGGL(qs_raycast_direct_kernel<true>, dim3(blocks), dim3(RC_BLOCK), 0, c->stream, n,
                           c->b, c->geom, c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone,
                           c->cfg.max_agent, c->d_counters);
    else
        hipLaunchKernelGGL(qs_raycast_direct_kernel<false>, dim3(blocks), dim3(RC_BLOCK), 0, c->stream, n,
                           c->b, c->geom, c->d_stamps, c->d_counts, ord_base, ord_stride, c->d_zone,
                           c->cfg.max_agent, c->d_counters);
    return hipGetLastError();
}

// ---- OccupancyGrid.update_ray, batched (object API, :136-156) ----------------------------
// Ray k is written after ray k-1: ordinal = seq0*4 + k + 1 in the same stamp space.
template <bool COUNTS>
__global__ void __launch_bounds__(RC_BLOCK)
qs_update_rays_kernel(size_t n, const double *__restrict__ rx, const double *__restrict__ ry,
                      const double *__restrict__ hx, const double *__restrict__ hy,
                      const unsigned char *__restrict__ valid, QsGeom geo,
                      unsigned int *__restrict__ stamps, unsigned long long *__restrict__ counts,
                      unsigned long long ord_base, unsigned long long *__restrict__ counters)
{
    const size_t k = (size_t)blockIdx.x * RC_BLOCK + threadIdx.x;
    if (k >= n) return;
    QsRay ray; ray.ex = hx[k]; ray.ey = hy[k]; ray.valid = valid[k] != 0;
    const unsigned int key_free = (unsigned int)((ord_base + k + 1) << 1);
    QsLine ln;
    unsigned int cells = 0;
    if (isfinite(rx[k]) && isfinite(ry[k]) && isfinite(ray.ex) && isfinite(ray.ey) &&
        qs_line_setup(ray, rx[k], ry[k], geo, ln)) {
        int x = ln.x0, y = ln.y0, err = ln.dx - ln.dy;
        for (;;) {
            const bool last = (x == ln.x1 && y == ln.y1);
            if ((!last || ray.valid) && x >= 0 && x < geo.size && y >= 0 && y < geo.size) {
                const size_t c = (size_t)y * geo.size + x;
                atomicMax(&stamps[c], key_free | (last ? 1u : 0u));
                qs_mark_dirty(geo, x, y);
                if (COUNTS) atomicAdd(&counts[c], last ? (1ull << 32) : 1ull);
                cells++;
            }
            if (last) break;
            const int e2 = 2 * err;
            if (e2 > -ln.dy) { err -= ln.dy; x += ln.sx; }
            if (e2 < ln.dx) { err += ln.dx; y += ln.sy; }
        }
    }
    atomicAdd(&counters[QS_CNT_RAYS], 1ull);
    if (cells) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)cells);
}

hipError_t qs_launch_update_rays(qs_ctx *c, const double *rx, const double *ry, const double *hx,
                                 const double *hy, const unsigned char *valid, size_t n, uint64_t seq0)
{
    if (n == 0) return hipSuccess;
    const unsigned int blocks = (unsigned int)((n + RC_BLOCK - 1) / RC_BLOCK);
    const unsigned long long ord_base = 4ull * (seq0 - c->epoch_base);
    if (c->cfg.enable_counts)
        hipLaunchKernelGGL(qs_update_rays_kernel<true>, dim3(blocks), dim3(RC_BLOCK), 0, c->stream, n, rx, ry,
                           hx, hy, valid, c->geom, c->d_stamps, c->d_counts, ord_base, c->d_counters);
    else
        hipLaunchKernelGGL(qs_update_rays_kernel<false>, dim3(blocks), dim3(RC_BLOCK), 0, c->stream, n, rx, ry,
                           hx, hy, valid, c->geom, c->d_stamps, c->d_counts, ord_base, c->d_counters);
    return hipGetLastError();
}

// ---- exact-trig mode: the waiting rays cast with their host-computed end points --------------------------------------
// in: [n_edge][4] = ex, ey, hit_valid, -  (end point from libm on the host); pose and stamp come from the ray's record
template <bool COUNTS>
__global__ void qs_edge_cast_kernel(unsigned int n_edge, const QsEdgeRec *__restrict__ recs, const double *__restrict__ in,
                                    QsGeom geo, unsigned int *__restrict__ stamps, unsigned long long *__restrict__ counts,
                                    unsigned long long *__restrict__ counters)
{
    const unsigned int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edge) return;
    const QsEdgeRec rec = recs[e];
    QsRay ray; ray.ex = in[4 * e]; ray.ey = in[4 * e + 1]; ray.valid = in[4 * e + 2] != 0.0;
    const unsigned int key_free = rec.key_free;
    QsLine ln;
    unsigned int cells = 0;
    if (qs_line_setup(ray, rec.rx, rec.ry, geo, ln)) {
        int x = ln.x0, y = ln.y0, err = ln.dx - ln.dy;
        for (;;) {
            const bool last = (x == ln.x1 && y == ln.y1);
            if ((!last || ray.valid) && x >= 0 && x < geo.size && y >= 0 && y < geo.size) {
                const size_t c = (size_t)y * geo.size + x;
                atomicMax(&stamps[c], key_free | (last ? 1u : 0u));
                qs_mark_dirty(geo, x, y);
                if (COUNTS) atomicAdd(&counts[c], last ? (1ull << 32) : 1ull);
                cells++;
            }
            if (last) break;
            const int e2 = 2 * err;
            if (e2 > -ln.dy) { err -= ln.dy; x += ln.sx; }
            if (e2 < ln.dx) { err += ln.dx; y += ln.sy; }
        }
    }
    if (cells) atomicAdd(&counters[QS_CNT_CELLS], (unsigned long long)cells);
}
hipError_t qs_launch_edge_cast(qs_ctx *c, unsigned int n_edge, const double *d_in)
{
    if (c->cfg.enable_counts)
        hipLaunchKernelGGL(qs_edge_cast_kernel<true>, dim3((n_edge + 255) / 256), dim3(256), 0, c->stream, n_edge, c->d_edge, d_in, c->geom,
                           c->d_stamps, c->d_counts, c->d_counters);
    else
        hipLaunchKernelGGL(qs_edge_cast_kernel<false>, dim3((n_edge + 255) / 256), dim3(256), 0, c->stream, n_edge, c->d_edge, d_in, c->geom,
                           c->d_stamps, c->d_counts, c->d_counters);
    return hipGetLastError();
}

__global__ void qs_world_to_grid_kernel(const double *__restrict__ w, size_t n, double o, double res,
                                        long long *__restrict__ out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = qs_w2g_ll(w[k], o, res);
}

hipError_t qs_launch_world_to_grid(qs_ctx *c, const double *w, size_t n, int axis, long long *out)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_world_to_grid_kernel, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, c->stream,
                       w, n, axis ? c->geom.oy : c->geom.ox, c->geom.res, out);
    return hipGetLastError();
}

// ---- hit points of the last batch, on request ---------------------------------------------------
// point_clouds[agent][name].append((hx, hy))  dual_bot_mapper.py:889-892.  The map update needs only
// the grid cells of a ray, so the world-frame end points (64 B per packet) are not written on the
// ingest path; qs_last_hits recomputes them with the same projection from the batch that is still
// resident, which gives the same bits.
__global__ void __launch_bounds__(RC_BLOCK)
qs_hits_kernel(size_t n, QsBatch b, QsGeom geo)
{
    const size_t r = (size_t)blockIdx.x * RC_BLOCK + threadIdx.x;
    const size_t i = r >> 2;
    if (i >= n) return;
    double2 h = make_double2(0.0, 0.0);
    unsigned char v = 0;
    if (b.map_ok[i]) {
        const QsRay ray = qs_project_ray(b.rx[i], b.ry[i], b.yaw[i], (double)((const float *)b.dist)[r], (int)(r & 3), geo);
        h = make_double2(ray.ex, ray.ey);
        v = ray.valid ? 1 : 0;
    }
    b.hit[r] = h;
    b.hit_valid[r] = v;
}

hipError_t qs_launch_hits(qs_ctx *c, size_t n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_hits_kernel, dim3((unsigned int)((4 * n + RC_BLOCK - 1) / RC_BLOCK)), dim3(RC_BLOCK), 0,
                       c->stream, n, c->b, c->geom);
    return hipGetLastError();
}
