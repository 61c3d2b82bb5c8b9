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

struct QsRay { double ex, ey; bool valid; };

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
    r.ex = rx + range * cos(a);                                                       // :890 / :901
    r.ey = ry + range * sin(a);                                                       // :891 / :902
    return r;
}

struct QsLine { int x0, y0, x1, y1, dx, dy, sx, sy; };

// Grid end points of a ray and the Bresenham set-up of :158-165.  Returns false when no cell
// of the ray can be inside the grid (or the ray exceeds the ABI length limit): the reference
// would walk the same cells and skip every one of them (:149, :155).
__device__ inline bool qs_line_setup(const QsRay &ray, double rx, double ry, const QsGeom &geo, QsLine &ln)
{
    const long long x0 = qs_w2g_ll(rx, geo.ox, geo.res), y0 = qs_w2g_ll(ry, geo.oy, geo.res);       // :142
    const long long x1 = qs_w2g_ll(ray.ex, geo.ox, geo.res), y1 = qs_w2g_ll(ray.ey, geo.oy, geo.res); // :143
    if (x0 == QS_LL_BAD || y0 == QS_LL_BAD || x1 == QS_LL_BAD || y1 == QS_LL_BAD) return false;
    const long long xlo = x0 < x1 ? x0 : x1, xhi = x0 < x1 ? x1 : x0;
    const long long ylo = y0 < y1 ? y0 : y1, yhi = y0 < y1 ? y1 : y0;
    if (xhi < 0 || yhi < 0 || xlo >= geo.size || ylo >= geo.size) return false;
    if (xhi - xlo > QS_MAX_RAY_CELLS || yhi - ylo > QS_MAX_RAY_CELLS) return false;
    ln.x0 = (int)x0; ln.y0 = (int)y0; ln.x1 = (int)x1; ln.y1 = (int)y1;
    ln.dx = (int)(xhi - xlo); ln.dy = (int)(yhi - ylo);                                              // :161-162
    ln.sx = x0 < x1 ? 1 : -1; ln.sy = y0 < y1 ? 1 : -1;                                             // :163-164
    return true;
}
