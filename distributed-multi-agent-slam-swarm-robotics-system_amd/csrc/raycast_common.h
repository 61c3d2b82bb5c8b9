// raycast_common.h -- device helpers shared by the direct and the tile-binned raycast.
#pragma once
#include "qs_internal.h"

#define QS_LL_BAD (-0x7fffffffffffffffll - 1)
#define QS_MAX_RAY_CELLS (1 << 24)   // ABI limit on one ray's length in cells (update_rays API)

// OccupancyGrid.world_to_grid  dual_bot_mapper.py:121-125: int((w - o) / res) -- fp64
// subtract, fp64 DIVIDE (not a multiply by 1/res: (0.05+5)/0.05 = 100.99999999999999 -> 100),
// truncation toward zero (not floor: (-5.049+5)/0.05 = -0.98 -> 0).
__device__ inline long long qs_w2g_ll(double w, double o, double res)
{
    const double q = (w - o) / res;
    if (!(fabs(q) < 9.0e15)) return QS_LL_BAD;   // CPython would raise / walk 1e15 cells
    return (long long)q;
}

// Same value as an int, without the divide in the common case.  qa = (w - o) * (1 / res) is within
// 2^-51 |q| of the correctly rounded quotient q; for |qa| < 1e9 that is < 5e-7, so when qa is more than
// 1e-4 away from every integer, no integer lies between qa and q and int(q) == int(qa).  Otherwise
// (one coordinate in ~5000, e.g. the 100.99999999999999 case above) the quotient itself decides.
// Cells beyond +-2^30 are clamped there: every grid is < 2^14 wide, so such an end point can only
// make a ray miss the grid or exceed the QS_MAX_RAY_CELLS limit, clamped or not.
#define QS_CELL_CLAMP (1 << 30)
__device__ inline bool qs_w2g_i32(double w, double o, const QsGeom &geo, int &out)
{
    const double qa = (w - o) * geo.inv_res;
    if (fabs(qa) < 1.0e9 && fabs(qa - rint(qa)) > 1.0e-4) { out = (int)qa; return true; }
    const long long q = qs_w2g_ll(w, o, geo.res);
    if (q == QS_LL_BAD) return false;
    out = q > QS_CELL_CLAMP ? QS_CELL_CLAMP : (q < -QS_CELL_CLAMP ? -QS_CELL_CLAMP : (int)q);
    return true;
}

// Sparse fuse (sparse_fuse.hip): a writer that goes straight to the grid marks the block of every cell it writes.  The bit
// is read first: a ray's cells share a block four to sixteen at a time and the rays of a batch share rooms, so almost
// every mark finds its bit set and costs one cached load (a stale 0 only repeats the atomic; bits are never cleared
// while writers run).
__device__ inline void qs_mark_dirty(const QsGeom &geo, int x, int y)
{
    if (geo.dirty) {
        unsigned int *w = geo.dirty + qs_dirty_word(x, y, geo.dirty_pitch);
        const unsigned int m = qs_dirty_mask(x);
        if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & m)) atomicOr(w, m);
    }
}

struct QsRay { double ex, ey; bool valid; };

// sin and cos of a ray heading: the device library's fp64 sincos (one argument reduction for both;
// same bits as its separate sin and cos).  A Cody-Waite + fdlibm-kernel version for |a| < 1e5 was
// measured at -2 us per 1 M packets on this pass and dropped: not worth a second implementation of a
// value that decides cells.
__device__ inline void qs_sincos(double x, double *sn, double *cs) { sincos(x, sn, cs); }

// dual_bot_mapper.py:886-903: sensor order front(0), left(+pi/2), back(pi), right(-pi/2);
// hit_valid = MIN < d <= MAX; invalid rays extend to min(d, MAX) if d > MIN else MAX and
// only clear cells.  NaN compares false everywhere, which gives the reference's behaviour
// (a 1.2 m free ray) without a special case.
__device__ inline QsRay qs_project_ray(double rx, double ry, double yaw, double d, int s, const QsGeom &geo)
{
    const double kPi = 3.141592653589793;   // math.pi
    const double ang = s == 0 ? 0.0 : (s == 1 ? kPi / 2 : (s == 2 ? kPi : -kPi / 2));   // :61-66
    const double a = yaw + ang;                                                       // :887
    QsRay r;
    r.valid = (geo.min_dist < d) && (d <= geo.max_dist);                               // :888
    const double range = r.valid ? d
                       : ((d > geo.min_dist) ? ((geo.max_dist < d) ? geo.max_dist : d) : geo.max_dist);  // :900
    double sa, ca;
    qs_sincos(a, &sa, &ca);
    r.ex = rx + range * ca;                                                           // :890 / :901
    r.ey = ry + range * sa;                                                           // :891 / :902
    return r;
}

// Exact-trig mode.  The device library's sin / cos may differ from glibc's (the reference's math.cos / math.sin) in
// the last bit; after `r * cos + rx` that is at most one ulp of the end point (< 6e-13 cells at 200 m and 5 cm), which
// can only change int((w - o) / res) if the quotient lies that close to an integer.  Rays whose end-point quotient is
// within 1e-9 (+ 1e-14 relative) of an integer on either axis are therefore not decided here: the host recomputes their
// end points with libm and casts them with their own stamps (any order gives the same grid).
__device__ inline bool qs_edge_coord(double w, double o, const QsGeom &geo)
{
    const double qa = (w - o) * geo.inv_res;
    return fabs(qa - rint(qa)) <= 1.0e-9 + fabs(qa) * 1.0e-14;
}
__device__ inline bool qs_edge_ray(const QsRay &ray, const QsGeom &geo)
{
    return qs_edge_coord(ray.ex, geo.ox, geo) || qs_edge_coord(ray.ey, geo.oy, geo);
}

// leave the ray to the host; false: the list is full (the caller casts the ray with the device's end point after all)
__device__ inline bool qs_edge_defer(const QsBatch &b, double rx, double ry, double yaw, float d, unsigned int key_free)
{
    const unsigned int slot = atomicAdd(b.edge_n, 1u);
    if (slot >= b.edge_cap) { atomicAdd(b.edge_n + 2, 1u); return false; }
    QsEdgeRec rec; rec.rx = rx; rec.ry = ry; rec.yaw = yaw; rec.d = d; rec.key_free = key_free;
    b.edge[slot] = rec;
    return true;
}

struct QsLine { int x0, y0, x1, y1, dx, dy, sx, sy; };

// Grid end points of a ray and the Bresenham set-up of :158-165.  Returns false when no cell
// of the ray can be inside the grid (or the ray exceeds the ABI length limit): the reference
// would walk the same cells and skip every one of them (:149, :155).
__device__ inline bool qs_line_setup(const QsRay &ray, double rx, double ry, const QsGeom &geo, QsLine &ln)
{
    int x0, y0, x1, y1;
    bool ok = qs_w2g_i32(rx, geo.ox, geo, x0);                                                      // :142
    ok = qs_w2g_i32(ry, geo.oy, geo, y0) && ok;
    ok = qs_w2g_i32(ray.ex, geo.ox, geo, x1) && ok;                                                 // :143
    ok = qs_w2g_i32(ray.ey, geo.oy, geo, y1) && ok;
    if (!ok) return false;
    const int xlo = min(x0, x1), xhi = max(x0, x1), ylo = min(y0, y1), yhi = max(y0, y1);
    // |coordinates| <= 2^30: the differences below are exact in 32-bit unsigned arithmetic
    const unsigned int dx = (unsigned int)xhi - (unsigned int)xlo, dy = (unsigned int)yhi - (unsigned int)ylo;
    if (xhi < 0 || yhi < 0 || xlo >= geo.size || ylo >= geo.size) return false;
    if (dx > QS_MAX_RAY_CELLS || dy > QS_MAX_RAY_CELLS) return false;
    ln.x0 = x0; ln.y0 = y0; ln.x1 = x1; ln.y1 = y1;
    ln.dx = (int)dx; ln.dy = (int)dy;                                                                // :161-162
    ln.sx = x0 < x1 ? 1 : -1; ln.sy = y0 < y1 ? 1 : -1;                                             // :163-164
    return true;
}

// Bounding box of a bot's hit points and path (compute_bounding_box, dual_bot_mapper.py:702-706),
// exact min / max in any order.  The workgroup's copy lives in LDS as doubles and takes native
// ds_min_f64 / ds_max_f64 atomics without return: eight instructions per ray, no key conversion and
// nothing to wait for.  (Measured alternatives, 1 M packets: read-compare-then-atomic on ordered u64
// keys +25 us on the pass, per-thread register boxes flushed on a bot change +15 us -- the bot of a
// thread's next ray differs 60 % of the time on the 2-bot stream; this form +6 us.)
#define QS_ZONE_LDS_INIT(z, t) do { (z)[t][0] = __builtin_inf(); (z)[t][1] = __builtin_inf(); \
                                    (z)[t][2] = -__builtin_inf(); (z)[t][3] = -__builtin_inf(); } while (0)

__device__ inline void qs_zone_point(double (*s_zone)[4], int agent, double x, double y)
{
    double *z = s_zone[agent];
    __hip_atomic_fetch_min(&z[0], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(&z[1], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(&z[2], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(&z[3], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// fold the workgroup's box of bot t into the session's (order-preserving u64 keys in HBM).  The
// session's box rarely moves: read first, serialise on its 4 words only when it does.
__device__ inline void qs_zone_commit(const double (*s_zone)[4], int t, unsigned long long *__restrict__ zone)
{
    if (!(s_zone[t][0] <= s_zone[t][2])) return;        // no point of this bot in this workgroup
    const volatile unsigned long long *zg = zone + 4 * t;
    const unsigned long long k0 = qs_ord_from_double(s_zone[t][0]), k1 = qs_ord_from_double(s_zone[t][1]);
    const unsigned long long k2 = qs_ord_from_double(s_zone[t][2]), k3 = qs_ord_from_double(s_zone[t][3]);
    if (k0 < zg[0]) atomicMin(&zone[4 * t + 0], k0);
    if (k1 < zg[1]) atomicMin(&zone[4 * t + 1], k1);
    if (k2 > zg[2]) atomicMax(&zone[4 * t + 2], k2);
    if (k3 > zg[3]) atomicMax(&zone[4 * t + 3], k3);
}
