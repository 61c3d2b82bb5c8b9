// frontier.hip -- frontier detection and clustering on the device grid.
// Semantics: OccupancyGrid.get_frontiers / cluster_frontiers / cluster_centroid_world
// (server_nodes/dual_bot_mapper.py:181-237), called every 3 s from main() (:948-956).
//
// The reference walks the grid in a Python double loop and flood-fills with a BFS.  Here:
//   1. stencil  : interior cell is a frontier iff FREE and a 4-neighbour is UNKNOWN (:187-195),
//                 read straight from the stamp grid (stamp 0 = UNKNOWN, even = FREE);
//   2. union    : 4-connected components by union-find with atomicMin, always linking the larger
//                 root under the smaller, so a component's root is its LOWEST linear index -- the
//                 first cell the reference's row-major seed loop meets (:210), which is also what
//                 orders the reference's cluster list;
//   3. flatten + integer sums per root (count, sum gx, sum gy: what cluster_centroid_world divides);
//   4. order-preserving compaction of the roots -> clusters in the reference's order.
// The centroid itself (two divisions per cluster) is left to the host: it is exact integer/fp64
// arithmetic on these sums.  BFS visiting order inside a cluster is not reproduced (it only orders
// the membership lists, which nothing downstream reads).
#include "qs_internal.h"

#define FR_BLOCK 256
#define FR_CHUNK 1024
#define FR_NONE 0xffffffffu

__device__ inline bool fr_is_free(unsigned int s) { return s != 0 && !(s & 1u); }

__global__ void __launch_bounds__(FR_BLOCK)
qs_frontier_mask_kernel(const unsigned int *__restrict__ stamps, int size, unsigned int *__restrict__ label)
{
    const size_t cells = (size_t)size * size;
    const size_t stride = (size_t)gridDim.x * FR_BLOCK;
    for (size_t i = (size_t)blockIdx.x * FR_BLOCK + threadIdx.x; i < cells; i += stride) {
        const int x = (int)(i % size), y = (int)(i / size);
        unsigned int l = FR_NONE;
        if (x >= 1 && x < size - 1 && y >= 1 && y < size - 1 && fr_is_free(stamps[i])) {      // :187-190
            if (stamps[i - 1] == 0 || stamps[i + 1] == 0 || stamps[i - size] == 0 || stamps[i + size] == 0)   // :192-193
                l = (unsigned int)i;
        }
        label[i] = l;
    }
}

__device__ inline unsigned int fr_find(unsigned int *label, unsigned int x)
{
    unsigned int p = label[x];
    while (p != x) { x = p; p = label[x]; }
    return x;
}

__device__ inline void fr_unite(unsigned int *label, unsigned int a, unsigned int b)
{
    for (;;) {
        a = fr_find(label, a); b = fr_find(label, b);
        if (a == b) return;
        if (a > b) { const unsigned int t = a; a = b; b = t; }     // a < b: link b under a
        const unsigned int old = atomicMin(&label[b], a);
        if (old == b) return;
        b = old;                                                   // somebody relinked b meanwhile
    }
}

__global__ void __launch_bounds__(FR_BLOCK)
qs_frontier_union_kernel(int size, unsigned int *__restrict__ label)
{
    const size_t cells = (size_t)size * size;
    const size_t stride = (size_t)gridDim.x * FR_BLOCK;
    for (size_t i = (size_t)blockIdx.x * FR_BLOCK + threadIdx.x; i < cells; i += stride) {
        if (label[i] == FR_NONE) continue;
        // frontier cells are interior, so i + 1 and i + size exist
        if (label[i + 1] != FR_NONE) fr_unite(label, (unsigned int)i, (unsigned int)(i + 1));
        if (label[i + size] != FR_NONE) fr_unite(label, (unsigned int)i, (unsigned int)(i + size));
    }
}

__global__ void __launch_bounds__(FR_BLOCK)
qs_frontier_stats_kernel(int size, unsigned int *__restrict__ label, unsigned int *__restrict__ cnt,
                         unsigned long long *__restrict__ sumx, unsigned long long *__restrict__ sumy)
{
    const size_t cells = (size_t)size * size;
    const size_t stride = (size_t)gridDim.x * FR_BLOCK;
    for (size_t i = (size_t)blockIdx.x * FR_BLOCK + threadIdx.x; i < cells; i += stride) {
        if (label[i] == FR_NONE) continue;
        const unsigned int r = fr_find(label, (unsigned int)i);
        atomicAdd(&cnt[r], 1u);
        atomicAdd(&sumx[r], (unsigned long long)(i % size));
        atomicAdd(&sumy[r], (unsigned long long)(i / size));
    }
}

// order-preserving compaction (chunk counts -> scan -> ranked writes); MODE 0: every frontier
// cell (get_frontiers), MODE 1: component roots (one per cluster, in first-cell order)
template <int MODE>
__device__ inline bool fr_pred(const unsigned int *label, const unsigned int *cnt, size_t i)
{
    if (MODE == 0 || MODE == 2) return label[i] != FR_NONE;
    return cnt[i] != 0;
}

template <int MODE>
__global__ void __launch_bounds__(FR_BLOCK)
qs_frontier_count_kernel(const unsigned int *__restrict__ label, const unsigned int *__restrict__ cnt, size_t cells,
                         unsigned int *__restrict__ chunk_count)
{
    __shared__ unsigned int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * FR_CHUNK;
    unsigned int m = 0;
    for (int q = 0; q < FR_CHUNK / FR_BLOCK; q++) {
        const size_t i = base + q * FR_BLOCK + threadIdx.x;
        if (i < cells && fr_pred<MODE>(label, cnt, i)) m++;
    }
    if (m) atomicAdd(&s, m);
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = s;
}

__global__ void __launch_bounds__(1024)
qs_frontier_scan_kernel(unsigned int *__restrict__ chunk_count, size_t n_chunks, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_part[1024];
    const int tid = threadIdx.x;
    const size_t per = (n_chunks + 1023) / 1024;
    const size_t lo = min((size_t)tid * per, n_chunks), hi = min(lo + per, n_chunks);
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

template <int MODE>
__global__ void __launch_bounds__(FR_BLOCK)
qs_frontier_write_kernel(const unsigned int *__restrict__ label, const unsigned int *__restrict__ cnt,
                         const unsigned long long *__restrict__ sumx, const unsigned long long *__restrict__ sumy,
                         size_t cells, int size, const unsigned int *__restrict__ chunk_off,
                         int *__restrict__ out_xy, long long *__restrict__ out_stats, size_t cap)
{
    __shared__ unsigned int s_wave[FR_BLOCK / QS_WAVE];
    __shared__ unsigned int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = chunk_off[blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * FR_CHUNK;
    for (int q = 0; q < FR_CHUNK / FR_BLOCK; q++) {
        const size_t i = base + q * FR_BLOCK + tid;
        const bool on = i < cells && fr_pred<MODE>(label, cnt, i);
        const unsigned long long m = __ballot(on);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        unsigned int off = s_run;
        for (int v = 0; v < wave; v++) off += s_wave[v];
        if (on) {
            const size_t slot = off + __popcll(m & ((1ull << lane) - 1));
            if (slot < cap) {
                if (MODE == 0) { out_xy[2 * slot] = (int)(i % size); out_xy[2 * slot + 1] = (int)(i / size); }
                else if (MODE == 2) {
                    unsigned int r = (unsigned int)i;                       // the cluster's first cell: walk to the root
                    for (unsigned int p = label[r]; p != r; p = label[r]) r = p;
                    out_xy[3 * slot] = (int)(i % size); out_xy[3 * slot + 1] = (int)(i / size); out_xy[3 * slot + 2] = (int)r;
                }
                else {
                    out_stats[5 * slot] = cnt[i];
                    out_stats[5 * slot + 1] = (long long)(i % size); out_stats[5 * slot + 2] = (long long)(i / size);
                    out_stats[5 * slot + 3] = (long long)sumx[i]; out_stats[5 * slot + 4] = (long long)sumy[i];
                }
            }
        }
        __syncthreads();
        if (tid == 0) { unsigned int t = 0; for (int v = 0; v < FR_BLOCK / QS_WAVE; v++) t += s_wave[v]; s_run += t; }
        __syncthreads();
    }
}

static inline unsigned int fr_blocks(size_t items)
{
    size_t b = (items + FR_BLOCK - 1) / FR_BLOCK;
    return (unsigned int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// Workspace layout (bytes): label u32[cells] | cnt u32[cells] | sumx u64[cells] | sumy u64[cells] |
// chunk u32[n_chunks] | total u64
size_t qs_frontier_workspace_bytes(const qs_ctx *c)
{
    const size_t cells = c->cells, n_chunks = (cells + FR_CHUNK - 1) / FR_CHUNK;
    return cells * (4 + 4 + 8 + 8) + ((n_chunks * 4 + 15) & ~(size_t)15) + 16;
}

hipError_t qs_launch_frontier_label(qs_ctx *c, void *ws, bool with_clusters)
{
    const size_t cells = c->cells;
    unsigned int *label = (unsigned int *)ws;
    unsigned int *cnt = label + cells;
    unsigned long long *sumx = (unsigned long long *)(cnt + cells), *sumy = sumx + cells;
    hipLaunchKernelGGL(qs_frontier_mask_kernel, dim3(fr_blocks(cells)), dim3(FR_BLOCK), 0, c->stream, c->d_stamps,
                       c->cfg.size, label);
    if (with_clusters) {
        hipError_t e = hipMemsetAsync(cnt, 0, cells * (4 + 8 + 8), c->stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(qs_frontier_union_kernel, dim3(fr_blocks(cells)), dim3(FR_BLOCK), 0, c->stream, c->cfg.size, label);
        hipLaunchKernelGGL(qs_frontier_stats_kernel, dim3(fr_blocks(cells)), dim3(FR_BLOCK), 0, c->stream, c->cfg.size, label,
                           cnt, sumx, sumy);
    }
    return hipGetLastError();
}

// phase 0: count + scan (total -> *d_total); phase 1: ranked write
hipError_t qs_launch_frontier_compact(qs_ctx *c, void *ws, int mode, int phase, int *d_xy, long long *d_stats, size_t cap)
{
    const size_t cells = c->cells, n_chunks = (cells + FR_CHUNK - 1) / FR_CHUNK;
    unsigned int *label = (unsigned int *)ws;
    unsigned int *cnt = label + cells;
    unsigned long long *sumx = (unsigned long long *)(cnt + cells), *sumy = sumx + cells;
    unsigned int *chunk = (unsigned int *)(sumy + cells);
    unsigned long long *total = (unsigned long long *)((char *)chunk + ((n_chunks * 4 + 15) & ~(size_t)15));
    if (phase == 0) {
        if (mode == 0) hipLaunchKernelGGL(qs_frontier_count_kernel<0>, dim3((unsigned int)n_chunks), dim3(FR_BLOCK), 0, c->stream, label, cnt, cells, chunk);
        else hipLaunchKernelGGL(qs_frontier_count_kernel<1>, dim3((unsigned int)n_chunks), dim3(FR_BLOCK), 0, c->stream, label, cnt, cells, chunk);
        hipLaunchKernelGGL(qs_frontier_scan_kernel, dim3(1), dim3(1024), 0, c->stream, chunk, n_chunks, total);
    } else {
        if (mode == 0) hipLaunchKernelGGL(qs_frontier_write_kernel<0>, dim3((unsigned int)n_chunks), dim3(FR_BLOCK), 0, c->stream, label, cnt, sumx, sumy, cells, c->cfg.size, chunk, d_xy, d_stats, cap);
        else if (mode == 2) hipLaunchKernelGGL(qs_frontier_write_kernel<2>, dim3((unsigned int)n_chunks), dim3(FR_BLOCK), 0, c->stream, label, cnt, sumx, sumy, cells, c->cfg.size, chunk, d_xy, d_stats, cap);
        else hipLaunchKernelGGL(qs_frontier_write_kernel<1>, dim3((unsigned int)n_chunks), dim3(FR_BLOCK), 0, c->stream, label, cnt, sumx, sumy, cells, c->cfg.size, chunk, d_xy, d_stats, cap);
    }
    return hipGetLastError();
}

unsigned long long *qs_frontier_total_ptr(const qs_ctx *c, void *ws)
{
    const size_t cells = c->cells, n_chunks = (cells + FR_CHUNK - 1) / FR_CHUNK;
    char *chunk = (char *)ws + cells * 24;
    return (unsigned long long *)(chunk + ((n_chunks * 4 + 15) & ~(size_t)15));
}
