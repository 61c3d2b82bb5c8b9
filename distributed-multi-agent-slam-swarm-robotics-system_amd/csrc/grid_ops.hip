// grid_ops.hip -- streaming kernels over whole grids: K2 tri-state / log-odds views of the
// stamp grid, K3 grid fuse, stamp rebase, and map_merger's grid_to_pcd / rasterise.
// All of them are HBM-streaming: 16-byte accesses per lane, consecutive lanes on consecutive
// addresses, grid-stride loops capped at 2048 workgroups.
#include "qs_internal.h"

#define GO_BLOCK 256
#define GO_MAX_BLOCKS 2048

static inline unsigned int go_blocks(size_t items)
{
    size_t b = (items + GO_BLOCK - 1) / GO_BLOCK;
    return (unsigned int)(b < 1 ? 1 : (b > GO_MAX_BLOCKS ? GO_MAX_BLOCKS : b));
}

// ---- K2: stamp -> OccupancyGrid.grid values (CELL_UNKNOWN/-1, CELL_FREE/0, CELL_OCCUPIED/100;
// dual_bot_mapper.py:92-94).  16 cells per lane: 4 x uint4 in, one uint4 (16 int8) out.
__device__ inline unsigned int go_tri(unsigned int s)
{
    return s == 0 ? 0xffu : ((s & 1u) ? 100u : 0u);
}
__global__ void __launch_bounds__(GO_BLOCK)
qs_view_i8_kernel(const uint4 *__restrict__ stamps4, size_t n16, uint4 *__restrict__ out16,
                  const unsigned int *__restrict__ stamps, size_t cells, signed char *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t k = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; k < n16; k += stride) {
        uint4 o;
        unsigned int w[4];
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = stamps4[4 * k + q];
            w[q] = go_tri(v.x) | (go_tri(v.y) << 8) | (go_tri(v.z) << 16) | (go_tri(v.w) << 24);
        }
        o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
        out16[k] = o;
    }
    // tail (cells not a multiple of 16)
    for (size_t c = n16 * 16 + (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; c < cells; c += stride)
        out[c] = (signed char)go_tri(stamps[c]);
}

hipError_t qs_launch_view_i8(qs_ctx *c, signed char *out_dev)
{
    const size_t n16 = c->cells / 16;
    hipLaunchKernelGGL(qs_view_i8_kernel, dim3(go_blocks(n16 ? n16 : c->cells)), dim3(GO_BLOCK), 0, c->stream,
                       (const uint4 *)c->d_stamps, n16, (uint4 *)out_dev, c->d_stamps, c->cells, out_dev);
    return hipGetLastError();
}

// ---- log-odds view from the integer counters (build extension) ---------------------------
__global__ void __launch_bounds__(GO_BLOCK)
qs_logodds_kernel(const unsigned long long *__restrict__ counts, size_t cells, float l_occ, float l_free,
                  float lmin, float lmax, float *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t c = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; c < cells; c += stride) {
        const unsigned long long v = counts[c];
        const float hits = (float)(unsigned int)(v >> 32), misses = (float)(unsigned int)(v & 0xffffffffu);
        float l = hits * l_occ - misses * l_free;
        l = l < lmin ? lmin : (l > lmax ? lmax : l);
        out[c] = l;
    }
}
hipError_t qs_launch_logodds(qs_ctx *c, float l_occ, float l_free, float lmin, float lmax, float *out_dev)
{
    hipLaunchKernelGGL(qs_logodds_kernel, dim3(go_blocks(c->cells)), dim3(GO_BLOCK), 0, c->stream,
                       c->counts_view_fused ? c->d_counts_fused : c->d_counts, c->cells, l_occ, l_free, lmin, lmax, out_dev);
    return hipGetLastError();
}

__global__ void __launch_bounds__(GO_BLOCK)
qs_split_counts_kernel(const unsigned long long *__restrict__ counts, size_t cells, int *__restrict__ hits,
                       int *__restrict__ misses)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t c = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; c < cells; c += stride) {
        const unsigned long long v = counts[c];
        hits[c] = (int)(unsigned int)(v >> 32);
        misses[c] = (int)(unsigned int)(v & 0xffffffffu);
    }
}
hipError_t qs_launch_split_counts(qs_ctx *c, int *hits_dev, int *misses_dev)
{
    hipLaunchKernelGGL(qs_split_counts_kernel, dim3(go_blocks(c->cells)), dim3(GO_BLOCK), 0, c->stream,
                       c->counts_view_fused ? c->d_counts_fused : c->d_counts, c->cells, hits_dev, misses_dev);
    return hipGetLastError();
}

// ---- stamp rebase: collapse every written cell to ordinal 1 so the 30-bit ordinal space can
// start over; relative order against all FUTURE writes is preserved (they are all larger).
__global__ void __launch_bounds__(GO_BLOCK)
qs_rebase_kernel(uint4 *__restrict__ stamps4, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t k = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; k < n4; k += stride) {
        uint4 v = stamps4[k];
        v.x = v.x ? (2u | (v.x & 1u)) : 0u; v.y = v.y ? (2u | (v.y & 1u)) : 0u;
        v.z = v.z ? (2u | (v.z & 1u)) : 0u; v.w = v.w ? (2u | (v.w & 1u)) : 0u;
        stamps4[k] = v;
    }
}
hipError_t qs_launch_rebase(qs_ctx *c)
{
    hipLaunchKernelGGL(qs_rebase_kernel, dim3(go_blocks(c->cells / 4)), dim3(GO_BLOCK), 0, c->stream,
                       (uint4 *)c->d_stamps, c->cells / 4);
    return hipGetLastError();
}

// ---- K3: grid fuse.  Both bots write one shared grid in the reference (dual_bot_mapper.py:785,
// :851-852); with per-context / per-GPU grids the same result is the cell-wise latest stamp
// (max) and the sum of the counters.  One pass: (n_src + 1) reads + 1 write per cell.
// A thread keeps FUSE_UNROLL source loads (16 B each) in flight before it folds them: the sources are
// independent streams, and a dependent load-fold-load chain per source left the HBM queues a source deep.
// Sources are read once (non-temporal); the destination stays in the cache hierarchy for the views.
#define FUSE_MAX_SRC 64
#define FUSE_UNROLL 8
struct FuseSrcs { const uint4 *s[FUSE_MAX_SRC]; };
struct FuseCnts { const ulonglong2 *s[FUSE_MAX_SRC]; };

__device__ inline uint4 go_ld_nt(const uint4 *p)
{
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load((const v4u *)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline ulonglong2 go_ld_nt(const ulonglong2 *p)
{
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    const v2u v = __builtin_nontemporal_load((const v2u *)p);
    return make_ulonglong2(v.x, v.y);
}
__device__ inline void go_fold(uint4 &v, const uint4 u)
{
    v.x = u.x > v.x ? u.x : v.x; v.y = u.y > v.y ? u.y : v.y;
    v.z = u.z > v.z ? u.z : v.z; v.w = u.w > v.w ? u.w : v.w;
}
__device__ inline void go_fold(ulonglong2 &v, const ulonglong2 u)
{
    v.x += u.x; v.y += u.y;      // hi32/lo32 halves add independently (no carry below 2^32 writes)
}

template <typename V, typename S>
__global__ void __launch_bounds__(GO_BLOCK)
qs_fuse_kernel(V *__restrict__ dst, S src, int n_src, size_t nv)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t k = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; k < nv; k += stride) {
        V v = dst[k];
        int q = 0;
        for (; q + FUSE_UNROLL <= n_src; q += FUSE_UNROLL) {
            V u[FUSE_UNROLL];
            #pragma unroll
            for (int j = 0; j < FUSE_UNROLL; j++) u[j] = go_ld_nt(src.s[q + j] + k);
            #pragma unroll
            for (int j = 0; j < FUSE_UNROLL; j++) go_fold(v, u[j]);
        }
        if (q < n_src) {                       // the last, partial group: still all loads before the first fold
            V u[FUSE_UNROLL];
            #pragma unroll
            for (int j = 0; j < FUSE_UNROLL; j++) if (q + j < n_src) u[j] = go_ld_nt(src.s[q + j] + k);
            #pragma unroll
            for (int j = 0; j < FUSE_UNROLL; j++) if (q + j < n_src) go_fold(v, u[j]);
        }
        dst[k] = v;
    }
}

// dst cells [cell_off, cell_off + n_cells) <- fuse(dst, sources); every source pointer names the source's
// first cell OF THAT RANGE (whole grids: cell_off = 0, n_cells = cells).  cell_off and n_cells are multiples of 4.
hipError_t qs_launch_fuse(qs_ctx *c, const unsigned int *const *src_stamps,
                          const unsigned long long *const *src_counts, size_t n_src, size_t cell_off, size_t n_cells,
                          unsigned long long *dst_counts)
{
    for (size_t base = 0; base < n_src; base += FUSE_MAX_SRC) {
        const int m = (int)((n_src - base < FUSE_MAX_SRC) ? n_src - base : FUSE_MAX_SRC);
        FuseSrcs fs{}; FuseCnts fc{};
        bool have_counts = dst_counts != nullptr && src_counts != nullptr;
        bool have_stamps = src_stamps != nullptr;
        for (int q = 0; q < m; q++) {
            if (have_stamps) { fs.s[q] = (const uint4 *)src_stamps[base + q]; if (!fs.s[q]) have_stamps = false; }
            if (have_counts) { fc.s[q] = (const ulonglong2 *)src_counts[base + q]; if (!fc.s[q]) have_counts = false; }
        }
        if (have_stamps)
            hipLaunchKernelGGL((qs_fuse_kernel<uint4, FuseSrcs>), dim3(go_blocks(n_cells / 4)), dim3(GO_BLOCK), 0, c->stream,
                               (uint4 *)(c->d_stamps + cell_off), fs, m, n_cells / 4);
        if (have_counts)
            hipLaunchKernelGGL((qs_fuse_kernel<ulonglong2, FuseCnts>), dim3(go_blocks(n_cells / 2)), dim3(GO_BLOCK), 0, c->stream,
                               (ulonglong2 *)(dst_counts + cell_off), fc, m, n_cells / 2);
    }
    return hipGetLastError();
}

__global__ void qs_zone_identity_kernel(unsigned long long *zone, int n_bots)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_bots) {
        zone[4 * t + 0] = QS_ORD_MIN_IDENT; zone[4 * t + 1] = QS_ORD_MIN_IDENT;
        zone[4 * t + 2] = QS_ORD_MAX_IDENT; zone[4 * t + 3] = QS_ORD_MAX_IDENT;
    }
}
// qs_reset's small state in ONE launch (it was six memsets and a kernel, each a gap on the stream of a 2 ms step): drift,
// zone boxes (identity), counters, per-graph batch counts, EKF state, flags
__global__ void __launch_bounds__(256)
qs_reset_small_kernel(double *__restrict__ drift, int n_drift, unsigned long long *__restrict__ zone, int n_bots,
                      unsigned long long *__restrict__ counters, unsigned long long *__restrict__ graph_batch, int n_gb,
                      double *__restrict__ ekf, int n_ekf, double *__restrict__ ekf_prev, int n_prev, unsigned int *__restrict__ flags)
{
    const int stride = gridDim.x * 256;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_ekf; t += stride) ekf[t] = 0.0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_prev; t += stride) ekf_prev[t] = 0.0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_drift; t += stride) drift[t] = 0.0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_gb; t += stride) graph_batch[t] = 0ull;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_bots; t += stride) {
        zone[4 * t + 0] = QS_ORD_MIN_IDENT; zone[4 * t + 1] = QS_ORD_MIN_IDENT;
        zone[4 * t + 2] = QS_ORD_MAX_IDENT; zone[4 * t + 3] = QS_ORD_MAX_IDENT;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < QS_CNT_N) counters[threadIdx.x] = 0ull;
        if (threadIdx.x < 4) flags[threadIdx.x] = 0u;
    }
}
hipError_t qs_launch_reset_small(qs_ctx *c)
{
    const int nb = c->cfg.max_agent + 1;
    const int n_ekf = nb * 44;
    hipLaunchKernelGGL(qs_reset_small_kernel, dim3((n_ekf + 255) / 256), dim3(256), 0, c->stream, c->d_drift, nb * 2, c->d_zone, nb,
                       c->d_counters, c->d_graph_batch, c->n_graphs * 2, c->d_ekf, n_ekf, c->d_ekf_prev, nb * 4, c->d_flags);
    return hipGetLastError();
}

hipError_t qs_launch_fill_zone_identity(qs_ctx *c)
{
    const int nb = c->cfg.max_agent + 1;
    hipLaunchKernelGGL(qs_zone_identity_kernel, dim3((nb + 255) / 256), dim3(256), 0, c->stream, c->d_zone, nb);
    return hipGetLastError();
}

// ---- MapMerger.grid_to_pcd  server_nodes/map_merger.py:64-85 -----------------------------
// np.argwhere(data > 50) is row-major, so the points are an order-preserving compaction:
// per-chunk counts, one scan, ranked writes.  Chunk = 1024 cells.
#define PCD_CHUNK 1024
__global__ void __launch_bounds__(GO_BLOCK)
qs_pcd_count_kernel(const signed char *__restrict__ grid, size_t cells, unsigned int *__restrict__ chunk_count)
{
    __shared__ unsigned int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * PCD_CHUNK;
    unsigned int m = 0;
    for (int q = 0; q < PCD_CHUNK / GO_BLOCK; q++) {
        const size_t c = base + q * GO_BLOCK + threadIdx.x;
        if (c < cells && grid[c] > 50) m++;
    }
    if (m) atomicAdd(&s, m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = s;
}
// single-workgroup exclusive scan of the chunk counts (<= a few 10^5 entries)
__global__ void __launch_bounds__(1024)
qs_pcd_scan_kernel(unsigned int *__restrict__ chunk_count, size_t n_chunks, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_part[1024];
    const int tid = threadIdx.x;
    const size_t per = (n_chunks + 1023) / 1024;
    const size_t lo = (size_t)tid * per, hi = (lo + per < n_chunks) ? lo + per : n_chunks;
    unsigned long long sum = 0;
    for (size_t k = lo; k < hi; k++) sum += chunk_count[k];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; t++) { const unsigned long long v = s_part[t]; s_part[t] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[tid];
    for (size_t k = lo; k < hi; k++) { const unsigned int v = chunk_count[k]; chunk_count[k] = (unsigned int)run; run += v; }
}
__global__ void __launch_bounds__(GO_BLOCK)
qs_pcd_write_kernel(const signed char *__restrict__ grid, size_t cells, int w, double res, double ox, double oy,
                    const unsigned int *__restrict__ chunk_off, double *__restrict__ xy, size_t cap)
{
    __shared__ unsigned int s_wave[GO_BLOCK / QS_WAVE];
    __shared__ unsigned int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = chunk_off[blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * PCD_CHUNK;
    for (int q = 0; q < PCD_CHUNK / GO_BLOCK; q++) {
        const size_t c = base + q * GO_BLOCK + tid;
        const bool occ = c < cells && grid[c] > 50;                      // data > 50   :72
        const unsigned long long m = __ballot(occ);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        unsigned int off = s_run;
        for (int v = 0; v < wave; v++) off += s_wave[v];
        if (occ) {
            const size_t slot = off + __popcll(m & ((1ull << lane) - 1));
            if (slot < cap) {
                const long long row = (long long)(c / w), col = (long long)(c % w);
                xy[2 * slot] = (double)col * res + ox;                  // x = col*res + origin_x  :77
                xy[2 * slot + 1] = (double)row * res + oy;              // y = row*res + origin_y  :76
            }
        }
        __syncthreads();
        if (tid == 0) { unsigned int t = 0; for (int v = 0; v < GO_BLOCK / QS_WAVE; v++) t += s_wave[v]; s_run += t; }
        __syncthreads();
    }
}
hipError_t qs_launch_grid_to_pcd(qs_ctx *c, const signed char *d_grid, int h, int w, double res, double ox,
                                 double oy, double *d_xy, size_t cap, unsigned long long *d_count,
                                 unsigned int *d_chunk)
{
    const size_t cells = (size_t)h * w;
    const size_t n_chunks = (cells + PCD_CHUNK - 1) / PCD_CHUNK;
    if (d_xy == nullptr) {
        hipLaunchKernelGGL(qs_pcd_count_kernel, dim3((unsigned int)n_chunks), dim3(GO_BLOCK), 0, c->stream, d_grid,
                           cells, d_chunk);
        hipLaunchKernelGGL(qs_pcd_scan_kernel, dim3(1), dim3(1024), 0, c->stream, d_chunk, n_chunks, d_count);
    } else {
        hipLaunchKernelGGL(qs_pcd_write_kernel, dim3((unsigned int)n_chunks), dim3(GO_BLOCK), 0, c->stream, d_grid,
                           cells, w, res, ox, oy, d_chunk, d_xy, cap);
    }
    return hipGetLastError();
}

// ---- MapMerger.publish_global_map  map_merger.py:87-127 -----------------------------------
__global__ void __launch_bounds__(GO_BLOCK)
qs_bbox_kernel(const double *__restrict__ xy, size_t n, unsigned long long *__restrict__ box4)
{
    __shared__ unsigned long long s[4];
    if (threadIdx.x == 0) { s[0] = s[1] = QS_ORD_MIN_IDENT; s[2] = s[3] = QS_ORD_MAX_IDENT; }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    unsigned long long mnx = QS_ORD_MIN_IDENT, mny = QS_ORD_MIN_IDENT, mxx = QS_ORD_MAX_IDENT, mxy = QS_ORD_MAX_IDENT;
    for (size_t k = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; k < n; k += stride) {
        const unsigned long long x = qs_ord_from_double(xy[2 * k]), y = qs_ord_from_double(xy[2 * k + 1]);
        mnx = x < mnx ? x : mnx; mxx = x > mxx ? x : mxx;
        mny = y < mny ? y : mny; mxy = y > mxy ? y : mxy;
    }
    atomicMin(&s[0], mnx); atomicMin(&s[1], mny); atomicMax(&s[2], mxx); atomicMax(&s[3], mxy);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&box4[0], s[0]); atomicMin(&box4[1], s[1]); atomicMax(&box4[2], s[2]); atomicMax(&box4[3], s[3]);
    }
}
hipError_t qs_launch_bbox(qs_ctx *c, const double *d_xy, size_t n, unsigned long long *d_box4)
{
    hipLaunchKernelGGL(qs_bbox_kernel, dim3(go_blocks(n)), dim3(GO_BLOCK), 0, c->stream, d_xy, n, d_box4);
    return hipGetLastError();
}
__global__ void __launch_bounds__(GO_BLOCK)
qs_rasterise_kernel(const double *__restrict__ xy, size_t n, double res, double minx, double miny, int h, int w,
                    signed char *__restrict__ grid)
{
    const size_t stride = (size_t)gridDim.x * GO_BLOCK;
    for (size_t k = (size_t)blockIdx.x * GO_BLOCK + threadIdx.x; k < n; k += stride) {
        long long xi = (long long)((xy[2 * k] - minx) / res);          // .astype(int)  :109-110
        long long yi = (long long)((xy[2 * k + 1] - miny) / res);
        xi = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi);                   // np.clip  :112-113
        yi = yi < 0 ? 0 : (yi > h - 1 ? h - 1 : yi);
        grid[(size_t)yi * w + xi] = 100;                               // :115 (idempotent store)
    }
}
hipError_t qs_launch_rasterise(qs_ctx *c, const double *d_xy, size_t n, double res, double minx, double miny,
                               int h, int w, signed char *d_grid)
{
    hipError_t e = hipMemsetAsync(d_grid, 0xff, (size_t)h * w, c->stream);   // np.full(-1)  :107
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qs_rasterise_kernel, dim3(go_blocks(n)), dim3(GO_BLOCK), 0, c->stream, d_xy, n, res, minx,
                       miny, h, w, d_grid);
    return hipGetLastError();
}
