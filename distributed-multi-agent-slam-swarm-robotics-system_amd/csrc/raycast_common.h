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

struct QsRay { double ex, ey; bool valid; };

// sin and cos of a ray heading.  QS_FAST_SINCOS: for |a| < 1e5 (every real packet: yaw is a float near
// [-pi, pi]) Cody-Waite reduction by pi/2 in three pieces (33 + 33 + 53 bits) and the fdlibm
// degree-13/14 kernels -- < 1 ulp, like the C library behind the reference's math.cos / math.sin, at
// a third of the instructions of two general-range library calls; larger arguments take the
// library's Payne-Hanek path.  The value decides a cell only when (w - o) / res lands within an ulp
// of an integer.
#ifndef QS_FAST_SINCOS
#define QS_FAST_SINCOS 0
#endif
// out of line: the Payne-Hanek reduction needs ~40 VGPRs that the common path must not pay for
__device__ __attribute__((noinline)) static void qs_sincos_library(double x, double *sn, double *cs)
{
    *sn = sin(x); *cs = cos(x);
}
__device__ inline void qs_sincos(double x, double *sn, double *cs)
{
#if QS_FAST_SINCOS
    if (!(fabs(x) < 1.0e5)) { qs_sincos_library(x, sn, cs); return; }
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-n, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10,
                      -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                      8.33333333332248946124e-03);
    const double s = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                      -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                      2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c = 1.0 - (0.5 * z - z * (z * pc));
    const int q = (int)n & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
#else
    *sn = sin(x); *cs = cos(x);
#endif
}

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
// exact min / max in any order.  A thread accumulates the box of the bot it is currently seeing in
// registers (its rays are a fixed number of packets apart, which for round-robin streams is the same
// bot every time) and folds it into the workgroup's LDS copy -- order-preserving u64 keys, atomics
// only from lanes that still move an edge -- when the bot changes and once at the end.
struct QsZoneAcc { int agent; double mnx, mny, mxx, mxy; };

__device__ inline void qs_zone_flush(unsigned long long (*s_zone)[4], const QsZoneAcc &a)
{
    if (a.agent < 0) return;
    unsigned long long *z = s_zone[a.agent];
    const volatile unsigned long long *zv = z;          // a stale (less extreme) value only costs a redundant atomic
    const unsigned long long k0 = qs_ord_from_double(a.mnx), k1 = qs_ord_from_double(a.mny);
    const unsigned long long k2 = qs_ord_from_double(a.mxx), k3 = qs_ord_from_double(a.mxy);
    if (k0 < zv[0]) atomicMin(&z[0], k0);
    if (k1 < zv[1]) atomicMin(&z[1], k1);
    if (k2 > zv[2]) atomicMax(&z[2], k2);
    if (k3 > zv[3]) atomicMax(&z[3], k3);
}

__device__ inline void qs_zone_add(unsigned long long (*s_zone)[4], QsZoneAcc &a, int agent, double x, double y)
{
    if (agent != a.agent) {
        qs_zone_flush(s_zone, a);
        a.agent = agent; a.mnx = x; a.mxx = x; a.mny = y; a.mxy = y;
        return;
    }
    a.mnx = x < a.mnx ? x : a.mnx; a.mxx = x > a.mxx ? x : a.mxx;
    a.mny = y < a.mny ? y : a.mny; a.mxy = y > a.mxy ? y : a.mxy;
}
