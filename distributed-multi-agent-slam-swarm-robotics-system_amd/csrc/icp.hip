// icp.hip -- point-to-point ICP and voxel down-sampling for the map merger.
// Call sites in the reference: server_nodes/map_merger.py:45-60
//     registration_icp(local, global, 1.0, I4, TransformationEstimationPointToPoint(),
//                      ICPConvergenceCriteria(max_iteration=30));  fitness < 0.6 -> reject;
//     local.transform(T); global += local; global.voxel_down_sample(voxel_size=res)
// The arithmetic lives in Open3D (absent here, no version pinned, no reference test) => PARITY
// UNPINNED.  Restated from Open3D's published algorithm:
//   correspondences  : for every source point its nearest target point if closer than the
//                      threshold (hybrid search, max_nn = 1); fitness = #corr / #source,
//                      inlier_rmse = sqrt(sum d^2 / #corr);
//   update           : Umeyama without scaling on the correspondences; for planar clouds (z = 0,
//                      as grid_to_pcd produces) that is the closed-form 2-D Kabsch rotation
//                      theta = atan2(sum(ax*by - ay*bx), sum(ax*bx + ay*by)) on demeaned pairs;
//   loop             : at most max_iteration updates; stop when |d fitness| < 1e-6 and |d rmse| < 1e-6.
// Nearest neighbours are exact brute force in fp64 (targets staged through LDS tiles; ties -> lowest
// target index); sums are two-level and fixed-order, so results are reproducible run to run.
#include "qs_internal.h"

#define ICP_BLOCK 256

// ---- nearest target of every source point --------------------------------------------------------
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_nn_kernel(const double2 *__restrict__ src, size_t n_src, const double2 *__restrict__ dst, size_t n_dst,
                 double max_d2, int *__restrict__ corr, double *__restrict__ d2_out)
{
    __shared__ double2 s_t[ICP_BLOCK];
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    const double2 p = i < n_src ? src[i] : make_double2(0, 0);
    double best = INFINITY;
    int best_j = -1;
    for (size_t base = 0; base < n_dst; base += ICP_BLOCK) {
        const size_t j = base + threadIdx.x;
        s_t[threadIdx.x] = j < n_dst ? dst[j] : make_double2(INFINITY, INFINITY);
        __syncthreads();
        const int lim = (int)((n_dst - base < ICP_BLOCK) ? n_dst - base : ICP_BLOCK);
        for (int k = 0; k < lim; k++) {
            const double dx = p.x - s_t[k].x, dy = p.y - s_t[k].y;
            const double d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; best_j = (int)(base + k); }
        }
        __syncthreads();
    }
    if (i < n_src) {
        const bool ok = best_j >= 0 && best < max_d2;
        corr[i] = ok ? best_j : -1;
        d2_out[i] = ok ? best : 0.0;
    }
}

// ---- fixed-order two-level sums ---------------------------------------------------------------------
// pass 0: per block {n, sum d2, sum ax, sum ay, sum bx, sum by}; pass 1 (means known):
// {sum (ax-am)(bx-bm) + (ay..)(by..), sum (ax-am)(by-bm) - (ay-am)(bx-bm)}
#define ICP_NSUM 6
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_sums_kernel(const double2 *__restrict__ src, size_t n_src, const double2 *__restrict__ dst,
                   const int *__restrict__ corr, const double *__restrict__ d2, int pass,
                   double amx, double amy, double bmx, double bmy, double *__restrict__ partial)
{
    __shared__ double s[ICP_NSUM][ICP_BLOCK];
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    double v[ICP_NSUM] = {0, 0, 0, 0, 0, 0};
    if (i < n_src && corr[i] >= 0) {
        const double2 a = src[i], b = dst[corr[i]];
        if (pass == 0) { v[0] = 1.0; v[1] = d2[i]; v[2] = a.x; v[3] = a.y; v[4] = b.x; v[5] = b.y; }
        else {
            const double ax = a.x - amx, ay = a.y - amy, bx = b.x - bmx, by = b.y - bmy;
            v[0] = ax * bx + ay * by;
            v[1] = ax * by - ay * bx;
        }
    }
    #pragma unroll
    for (int q = 0; q < ICP_NSUM; q++) s[q][threadIdx.x] = v[q];
    __syncthreads();
    for (int off = ICP_BLOCK / 2; off > 0; off >>= 1) {       // fixed tree order
        if (threadIdx.x < off) {
            #pragma unroll
            for (int q = 0; q < ICP_NSUM; q++) s[q][threadIdx.x] += s[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < ICP_NSUM) partial[(size_t)blockIdx.x * ICP_NSUM + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void qs_icp_final_kernel(const double *__restrict__ partial, size_t n_blocks, double *__restrict__ out)
{
    if (threadIdx.x < ICP_NSUM) {
        double acc = 0.0;
        for (size_t b = 0; b < n_blocks; b++) acc += partial[b * ICP_NSUM + threadIdx.x];   // fixed order
        out[threadIdx.x] = acc;
    }
}

// p <- R p + t  (PointCloud::Transform with a planar rigid transform)
__global__ void __launch_bounds__(ICP_BLOCK)
qs_icp_transform_kernel(double2 *__restrict__ pts, size_t n, double c, double s, double tx, double ty)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i < n) {
        const double2 p = pts[i];
        pts[i] = make_double2(c * p.x - s * p.y + tx, s * p.x + c * p.y + ty);
    }
}

// ---- voxel down-sampling (PointCloud::VoxelDownSample): points are averaged per voxel; the voxel
// of p is floor((p - (min_bound - voxel/2)) / voxel).  Output is in ascending voxel order (Open3D's
// order is a hash-map's: unspecified). -----------------------------------------------------------------
__global__ void __launch_bounds__(ICP_BLOCK)
qs_voxel_key_kernel(const double2 *__restrict__ pts, size_t n, double minx, double miny, double voxel,
                    unsigned long long *__restrict__ keys)
{
    const size_t i = (size_t)blockIdx.x * ICP_BLOCK + threadIdx.x;
    if (i < n) {
        const long long vx = (long long)floor((pts[i].x - minx) / voxel), vy = (long long)floor((pts[i].y - miny) / voxel);
        keys[i] = ((unsigned long long)(vy & 0xffffffffll) << 32) | (unsigned long long)(vx & 0xffffffffll);
    }
}

hipError_t qs_launch_icp_nn(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, size_t n_dst,
                            double max_d2, int *corr, double *d2)
{
    hipLaunchKernelGGL(qs_icp_nn_kernel, dim3((unsigned int)((n_src + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, src, n_src, dst, n_dst, max_d2, corr, d2);
    return hipGetLastError();
}

hipError_t qs_launch_icp_sums(qs_ctx *c, const double2 *src, size_t n_src, const double2 *dst, const int *corr,
                              const double *d2, int pass, const double means[4], double *partial, double *out6)
{
    const size_t nb = (n_src + ICP_BLOCK - 1) / ICP_BLOCK;
    hipLaunchKernelGGL(qs_icp_sums_kernel, dim3((unsigned int)nb), dim3(ICP_BLOCK), 0, c->stream, src, n_src, dst, corr, d2,
                       pass, means[0], means[1], means[2], means[3], partial);
    hipLaunchKernelGGL(qs_icp_final_kernel, dim3(1), dim3(64), 0, c->stream, partial, nb, out6);
    return hipGetLastError();
}

hipError_t qs_launch_icp_transform(qs_ctx *c, double2 *pts, size_t n, double cs, double sn, double tx, double ty)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_icp_transform_kernel, dim3((unsigned int)((n + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, pts, n, cs, sn, tx, ty);
    return hipGetLastError();
}

hipError_t qs_launch_voxel_keys(qs_ctx *c, const double2 *pts, size_t n, double minx, double miny, double voxel,
                                unsigned long long *keys)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(qs_voxel_key_kernel, dim3((unsigned int)((n + ICP_BLOCK - 1) / ICP_BLOCK)), dim3(ICP_BLOCK), 0,
                       c->stream, pts, n, minx, miny, voxel, keys);
    return hipGetLastError();
}
