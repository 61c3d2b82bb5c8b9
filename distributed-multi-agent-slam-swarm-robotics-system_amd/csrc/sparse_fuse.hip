// sparse_fuse.hip -- the grid fuse of a sharded deployment without moving the whole map.
//
// The reference keeps ONE grid that every bot writes (dual_bot_mapper.py:785).  Sharded by agent over N GPUs that grid
// exists N times, and the dense fuse (MAX all-reduce of 64 MiB of stamps + SUM of 128 MiB of counters at 4096^2) moves
// all of it after every batch although a shard's bots have written a few rooms: 3 % of the cells on configs[3].  Here
// every writer of the grid sets one bit per 4 x 16-cell block it touches (qs_internal.h: QsGeom::dirty), and a fuse moves
// only those blocks:
//   lists   every rank's bitmap (all-gathered by the caller) -> ascending block ids + count      qs_sf_lists_kernel
//   pack    this rank's blocks: 64 stamps + 64 counter deltas since its previous fuse, 768 B     qs_sf_pack_kernel
//   (the caller sends the packed segment to every peer: point-to-point, all xGMI links at once)
//   apply   every rank's segment folded in: stamps atomicMax, counter deltas atomicAdd            qs_sf_apply_kernel
// All three are HBM-streaming over the dirty blocks only: a block is 4 grid rows of 64 B (stamps) / 128 B (counters) and
// one 256 B / 512 B row of the payload per wave-instruction.
#include "qs_internal.h"

#define SF_BW QS_DIRTY_BLOCK_W
#define SF_BH QS_DIRTY_BLOCK_H
#define SF_CELLS (SF_BW * SF_BH)             // 64: one lane per cell
#define SF_LIST_BLOCK 1024

size_t qs_sf_block_bytes(const qs_ctx *c) { return SF_CELLS * (sizeof(unsigned int) + (c->d_counts ? sizeof(unsigned long long) : 0)); }

// ---- mark a cell range dirty (qs_fuse_buffers*: a local fold writes the grid without going through a raycast) ---------
__global__ void qs_sf_mark_rows_kernel(unsigned int *__restrict__ dirty, int pitch, int blocks_x, int by_lo, int by_hi)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = (by_hi - by_lo) * pitch;
    if (w >= n) return;
    const int col = w % pitch;
    const int left = blocks_x - 32 * col;                    // valid blocks in this word
    const unsigned int m = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
    if (m) atomicOr(&dirty[(size_t)by_lo * pitch + w], m);
}
hipError_t qs_launch_sf_mark_range(qs_ctx *c, size_t cell_off, size_t n_cells)
{
    if (!c->d_dirty || n_cells == 0) return hipSuccess;
    const int by_lo = (int)(cell_off / c->cfg.size) / SF_BH;
    const int by_hi = (int)((cell_off + n_cells - 1) / c->cfg.size) / SF_BH + 1;
    const int n = (by_hi - by_lo) * c->geom.dirty_pitch;
    hipLaunchKernelGGL(qs_sf_mark_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_dirty, c->geom.dirty_pitch,
                       c->blocks_x, by_lo, by_hi);
    return hipGetLastError();
}

// ---- lists: bitmap -> ascending block ids.  One workgroup per rank's bitmap; order-preserving compaction ---------------
__global__ void __launch_bounds__(SF_LIST_BLOCK)
qs_sf_lists_kernel(const unsigned int *__restrict__ bitmaps, size_t words, int pitch, int blocks_x,
                   unsigned int *__restrict__ lists, unsigned int *__restrict__ counts)
{
    __shared__ unsigned int s_wave[SF_LIST_BLOCK / QS_WAVE];
    const int tid = threadIdx.x, lane = tid & (QS_WAVE - 1), wave = tid >> 6;
    const unsigned int *bm = bitmaps + (size_t)blockIdx.x * words;
    unsigned int *out = lists + (size_t)blockIdx.x * words * 32;
    const size_t per = (words + SF_LIST_BLOCK - 1) / SF_LIST_BLOCK;
    const size_t lo = min((size_t)tid * per, words), hi = min(lo + per, words);
    // bits beyond blocks_x in a row's last word are never blocks
    auto valid = [&](size_t w) -> unsigned int {
        const int left = blocks_x - 32 * (int)(w % pitch);
        return left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
    };
    unsigned int mine = 0;
    for (size_t w = lo; w < hi; w++) mine += __popc(bm[w] & valid(w));
    unsigned int inc = mine;
    #pragma unroll
    for (int off = 1; off < QS_WAVE; off <<= 1) { const unsigned int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    if (lane == QS_WAVE - 1) s_wave[wave] = inc;
    __syncthreads();
    unsigned int run = inc - mine;
    for (int v = 0; v < wave; v++) run += s_wave[v];
    for (size_t w = lo; w < hi; w++) {
        unsigned int m = bm[w] & valid(w);
        while (m) { const int b = __ffs(m) - 1; m &= m - 1; out[run++] = (unsigned int)(w * 32 + b); }
    }
    if (tid == SF_LIST_BLOCK - 1) counts[blockIdx.x] = run;
}
hipError_t qs_launch_sf_lists(qs_ctx *c)
{
    hipLaunchKernelGGL(qs_sf_lists_kernel, dim3(c->sf_world), dim3(SF_LIST_BLOCK), 0, c->stream, c->d_sf_bitmaps, c->dirty_words,
                       c->geom.dirty_pitch, c->blocks_x, c->d_sf_lists, c->d_sf_counts);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
qs_sf_popcount_kernel(const unsigned int *__restrict__ bm, size_t words, int pitch, int blocks_x, unsigned long long *__restrict__ out)
{
    unsigned int n = 0;
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (size_t)gridDim.x * 256) {
        const int left = blocks_x - 32 * (int)(w % pitch);
        const unsigned int m = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
        n += __popc(bm[w] & m);
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, (unsigned long long)n);
}
hipError_t qs_launch_sf_popcount(qs_ctx *c, unsigned long long *d_out)
{
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long), c->stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qs_sf_popcount_kernel, dim3(64), dim3(256), 0, c->stream, c->d_dirty, c->dirty_words, c->geom.dirty_pitch,
                       c->blocks_x, d_out);
    return hipGetLastError();
}

// cell of lane `lane` of block `bid` (bit index in the bitmap: row by, column bx): false beyond the grid's right edge
__device__ inline bool sf_cell(unsigned int bid, int lane, int pitch, int size, size_t &cell)
{
    const int by = (int)(bid / (unsigned int)(pitch * 32)), bx = (int)(bid % (unsigned int)(pitch * 32));
    const int x = bx * SF_BW + (lane & (SF_BW - 1)), y = by * SF_BH + (lane >> 4);
    cell = (size_t)y * size + x;
    return x < size && y < size;
}

// ---- pack: one wave per block, one lane per cell ---------------------------------------------------------------------
template <bool COUNTS>
__global__ void __launch_bounds__(256)
qs_sf_pack_kernel(const unsigned int *__restrict__ list, unsigned int n_blocks, int pitch, int size,
                  const unsigned int *__restrict__ stamps, const unsigned long long *__restrict__ counts,
                  unsigned long long *__restrict__ sent, unsigned char *__restrict__ dst)
{
    const int lane = threadIdx.x & (QS_WAVE - 1);
    const unsigned int wave = blockIdx.x * (256 / QS_WAVE) + (threadIdx.x >> 6), n_waves = gridDim.x * (256 / QS_WAVE);
    constexpr size_t BB = SF_CELLS * (4 + (COUNTS ? 8 : 0));
    for (unsigned int k = wave; k < n_blocks; k += n_waves) {
        size_t cell;
        const bool in = sf_cell(list[k], lane, pitch, size, cell);
        unsigned char *blk = dst + (size_t)k * BB;
        ((unsigned int *)blk)[lane] = in ? stamps[cell] : 0u;
        if (COUNTS) {
            unsigned long long d = 0;
            if (in) {
                const unsigned long long cur = counts[cell], old = sent[cell];
                // hi32 hits, lo32 misses: the halves are independent counters, so the delta is taken per half
                d = ((unsigned long long)((unsigned int)(cur >> 32) - (unsigned int)(old >> 32)) << 32) |
                    (unsigned long long)((unsigned int)cur - (unsigned int)old);
                if (d) sent[cell] = cur;
            }
            ((unsigned long long *)(blk + SF_CELLS * 4))[lane] = d;
        }
    }
}
hipError_t qs_launch_sf_pack(qs_ctx *c, unsigned int n_own, unsigned char *dst)
{
    if (n_own == 0) return hipSuccess;
    const unsigned int *list = c->d_sf_lists + (size_t)c->sf_rank * c->dirty_words * 32;
    const unsigned int blocks = (n_own + 3) / 4 < 4096 ? (n_own + 3) / 4 : 4096;
    if (c->d_counts)
        hipLaunchKernelGGL(qs_sf_pack_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, list, n_own, c->geom.dirty_pitch, c->cfg.size,
                           c->d_stamps, c->d_counts, c->d_counts_sent, dst);
    else
        hipLaunchKernelGGL(qs_sf_pack_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, list, n_own, c->geom.dirty_pitch, c->cfg.size,
                           c->d_stamps, c->d_counts, c->d_counts_sent, dst);
    return hipGetLastError();
}

// ---- apply: every rank's segment.  Blocks of different ranks may coincide (rooms that share a block): atomics ------------
struct SfPlan { unsigned int first[QS_SPARSE_MAX_WORLD + 1]; size_t off[QS_SPARSE_MAX_WORLD]; int world, rank; };

template <bool COUNTS>
__global__ void __launch_bounds__(256)
qs_sf_apply_kernel(SfPlan pl, const unsigned int *__restrict__ lists, size_t list_stride, int pitch, int size,
                   const unsigned char *__restrict__ payload, unsigned int *__restrict__ stamps,
                   unsigned long long *__restrict__ fused)
{
    const int lane = threadIdx.x & (QS_WAVE - 1);
    const unsigned int wave = blockIdx.x * (256 / QS_WAVE) + (threadIdx.x >> 6), n_waves = gridDim.x * (256 / QS_WAVE);
    constexpr size_t BB = SF_CELLS * (4 + (COUNTS ? 8 : 0));
    const unsigned int total = pl.first[pl.world];
    int s = 0;
    for (unsigned int t = wave; t < total; t += n_waves) {
        while (pl.first[s + 1] <= t) s++;                                  // (t ascends: the search never goes back)
        const unsigned int k = t - pl.first[s];
        size_t cell;
        const bool in = sf_cell(lists[(size_t)s * list_stride + k], lane, pitch, size, cell);
        const unsigned char *blk = payload + pl.off[s] + (size_t)k * BB;
        if (s != pl.rank) {                                                  // own stamps are in place
            const unsigned int v = ((const unsigned int *)blk)[lane];
            if (in && v) atomicMax(&stamps[cell], v);
        }
        if (COUNTS) {
            const unsigned long long d = ((const unsigned long long *)(blk + SF_CELLS * 4))[lane];
            if (in && d) atomicAdd(&fused[cell], d);
        }
    }
}
hipError_t qs_launch_sf_apply(qs_ctx *c)
{
    SfPlan pl{};
    pl.world = c->sf_world; pl.rank = c->sf_rank;
    unsigned int run = 0;
    for (int s = 0; s < c->sf_world; s++) { pl.first[s] = run; run += c->sf_n[s]; pl.off[s] = c->sf_off[s]; }
    pl.first[c->sf_world] = run;
    if (run == 0) return hipSuccess;
    const unsigned int blocks = (run + 3) / 4 < 8192 ? (run + 3) / 4 : 8192;
    if (c->d_counts)
        hipLaunchKernelGGL(qs_sf_apply_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, pl, c->d_sf_lists, c->dirty_words * 32,
                           c->geom.dirty_pitch, c->cfg.size, c->d_sf_payload, c->d_stamps, c->d_counts_fused);
    else
        hipLaunchKernelGGL(qs_sf_apply_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, pl, c->d_sf_lists, c->dirty_words * 32,
                           c->geom.dirty_pitch, c->cfg.size, c->d_sf_payload, c->d_stamps, c->d_counts_fused);
    return hipGetLastError();
}
