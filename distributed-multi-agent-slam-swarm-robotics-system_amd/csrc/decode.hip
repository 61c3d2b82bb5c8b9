// decode.hip -- K0: QuasarPacket datagrams -> validated SoA batch.
// Semantics: server_nodes/dual_bot_mapper.py:826-857 (unpack, length/magic/agent filters,
// bot-2 offset).  The drift correction of :855-857 is applied by the SLAM stage because it
// depends on earlier loop closures.
#include "qs_internal.h"

#define DEC_BLOCK 256
#define DEC_MAX_STRIDE 64   // records are staged through LDS when stride <= 64 bytes
#define DEC_TILES 8         // tiles of DEC_BLOCK records per workgroup: the per-graph / per-bot counts reach HBM once per
                            // workgroup (64 bots: half a million same-address atomics per 2^20 packets otherwise)

// One workgroup stages DEC_BLOCK consecutive records (DEC_BLOCK*stride contiguous bytes) into
// LDS with coalesced dword loads, then each lane parses its own record from LDS: the 42-byte
// packed layout puts every float at an odd offset, so a direct per-lane global read would be
// ten unaligned loads per lane.
__global__ void __launch_bounds__(DEC_BLOCK)
qs_decode_kernel(const unsigned char *__restrict__ pkts, size_t n, size_t stride,
                 const unsigned short *__restrict__ lens, const double *__restrict__ offset,
                 int max_agent, int bots_per_graph, int n_graphs, QsBatch b,
                 unsigned long long *__restrict__ graph_batch, unsigned int *__restrict__ agent_ev,
                 unsigned long long *__restrict__ counters)
{
    __shared__ unsigned int s_raw[DEC_BLOCK * DEC_MAX_STRIDE / 4 + 2];
    __shared__ unsigned int s_agent_ev[QS_MAX_AGENT + 1];
    __shared__ unsigned int s_acc, s_hist_small[64][2];
    const size_t block_base = (size_t)blockIdx.x * DEC_BLOCK * DEC_TILES;
    const int tid = threadIdx.x;
    const bool small_g = n_graphs <= 64;

    if (tid == 0) s_acc = 0;
    if (small_g && tid < 64) { s_hist_small[tid][0] = 0; s_hist_small[tid][1] = 0; }
    for (int t = tid; t <= QS_MAX_AGENT; t += DEC_BLOCK) s_agent_ev[t] = 0;

  for (int tile = 0; tile < DEC_TILES; tile++) {
    const size_t base = block_base + (size_t)tile * DEC_BLOCK;
    if (base >= n) break;                                       // (uniform)
    const size_t nrec = (n - base < DEC_BLOCK) ? (n - base) : DEC_BLOCK;
    if (tile) __syncthreads();                                  // the previous tile is parsed: its bytes may go

    // stage: the byte range [base*stride, (base+nrec)*stride) widened to dword boundaries
    const size_t byte0 = base * stride;
    const size_t byte1 = byte0 + nrec * stride;
    const size_t total = n * stride;
    const unsigned long long addr0 = (unsigned long long)pkts + byte0;
    const unsigned int mis = (unsigned int)(addr0 & 3);          // misalignment of the range
    const unsigned int *src = (const unsigned int *)(addr0 - mis);
    const size_t ndw = (mis + (byte1 - byte0) + 3) / 4;
    // the last dword may reach past the caller's buffer by <4 bytes: read it bytewise
    const unsigned long long buf_end = (unsigned long long)pkts + total;
    if ((unsigned long long)src >= (unsigned long long)pkts && (unsigned long long)(src + ndw) <= buf_end) {
        // every dword of the range lies inside the caller's buffer (all tiles but the batch's first and last): the loads of four
        // rounds in flight before the first LDS store
        for (unsigned int k0 = 0; k0 < (unsigned int)ndw; k0 += 4 * DEC_BLOCK) {
            unsigned int v[4];
            #pragma unroll
            for (int q = 0; q < 4; q++) { const unsigned int k = k0 + q * DEC_BLOCK + tid; v[q] = k < (unsigned int)ndw ? src[k] : 0u; }
            #pragma unroll
            for (int q = 0; q < 4; q++) { const unsigned int k = k0 + q * DEC_BLOCK + tid; if (k < (unsigned int)ndw) s_raw[k] = v[q]; }
        }
    } else
    for (size_t k = tid; k < ndw; k += DEC_BLOCK) {
        const unsigned long long a = (unsigned long long)(src + k);
        unsigned int v;
        if (a + 4 <= buf_end && a >= (unsigned long long)pkts) {
            v = src[k];
        } else {
            v = 0;
            const unsigned char *p = (const unsigned char *)a;
            for (int q = 0; q < 4; q++)
                if (a + q >= (unsigned long long)pkts && a + q < buf_end) v |= (unsigned int)p[q] << (8 * q);
        }
        s_raw[k] = v;
    }
    __syncthreads();

    bool ok = false;
    int agent = 0, lmk = 0;
    if ((size_t)tid < nrec) {
        const size_t i = base + tid;
        const int len = lens ? (int)lens[i] : (int)stride;
        // :826-838 version by length
        if ((len == QS_PACKET_SIZE || len == QS_PACKET_SIZE_V1) && (size_t)len <= stride) {
            // the record's 42 bytes as eleven dwords: twelve aligned LDS reads shifted into place (v_alignbyte; the record starts
            // at any byte), then every field by a shift of two of them -- not 42 byte reads.  Bytes past the record's length are
            // never looked at (v1: the landmark byte, which is 0 then).
            const unsigned int off = mis + (unsigned int)tid * (unsigned int)stride;
            const unsigned int w0 = off >> 2, sh = off & 3u;
            unsigned int d[12], r[11];
            #pragma unroll
            for (int q = 0; q < 12; q++) d[q] = s_raw[w0 + q];
            #pragma unroll
            for (int q = 0; q < 11; q++) r[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
            #define DEC_U32(p) ((p) % 4 == 0 ? r[(p) / 4] : __builtin_amdgcn_alignbyte(r[(p) / 4 + 1], r[(p) / 4], (p) % 4))
            agent = (int)(r[1] & 0xffu);                                        // byte 4
            lmk = (len == QS_PACKET_SIZE) ? (int)((r[10] >> 8) & 0xffu) : 0;    // byte 41
            const float x = __uint_as_float(DEC_U32(5)), y = __uint_as_float(DEC_U32(9)), yaw = __uint_as_float(DEC_U32(13));
            const int enc = (int)DEC_U32(17);
            const float d0 = __uint_as_float(DEC_U32(25)), d1 = __uint_as_float(DEC_U32(29)), d2 = __uint_as_float(DEC_U32(33)),
                        d3 = __uint_as_float(DEC_U32(37));
            #undef DEC_U32
            ok = r[0] == 0x4c525351u                                            // 'Q','S','R','L'  :840
                 && agent >= 1 && agent <= max_agent;                           // :842
            // CPython's int() raises on a non-finite pose (:123); the build drops the packet
            ok = ok && isfinite(x) && isfinite(y) && isfinite(yaw);
            if (ok) {
                b.agent[i] = (unsigned char)agent;
                b.lm[i] = (unsigned char)lmk;
                b.px[i] = (double)x + offset[agent];                            // :851-852
                b.py[i] = (double)y;
                b.yaw[i] = (double)yaw;
                b.dist[i] = make_float4(d0, d1, d2, d3);
                b.enc[i] = enc;
            }
        }
        b.accept[i] = ok ? 1 : 0;
        if (b.map_ok != b.accept) b.map_ok[i] = (ok && agent >= b.own_lo && agent <= b.own_hi) ? 1 : 0;
    }
    // per-graph accepted / landmark-event counts of the batch (capacity planning + counters)
    if (ok) {
        const int g = (agent - 1) / bots_per_graph;
        if (small_g) {
            atomicAdd(&s_hist_small[g][0], 1u);
            if (lmk) atomicAdd(&s_hist_small[g][1], 1u);
        } else {
            atomicAdd(&graph_batch[2 * g], 1ull);
            if (lmk) atomicAdd(&graph_batch[2 * g + 1], 1ull);
        }
        atomicAdd(&s_acc, 1u);
        if (lmk) atomicAdd(&s_agent_ev[agent], 1u);
    }
  }
    __syncthreads();
    for (int t = tid; t <= max_agent; t += DEC_BLOCK)
        if (s_agent_ev[t]) atomicAdd(&agent_ev[t], s_agent_ev[t]);
    if (small_g && tid < n_graphs) {
        if (s_hist_small[tid][0]) atomicAdd(&graph_batch[2 * tid], (unsigned long long)s_hist_small[tid][0]);
        if (s_hist_small[tid][1]) atomicAdd(&graph_batch[2 * tid + 1], (unsigned long long)s_hist_small[tid][1]);
    }
    if (tid == 0) {
        const size_t left = n - block_base;
        atomicAdd(&counters[QS_CNT_DATAGRAMS], (unsigned long long)(left < (size_t)DEC_BLOCK * DEC_TILES ? left : (size_t)DEC_BLOCK * DEC_TILES));
        if (s_acc) atomicAdd(&counters[QS_CNT_ACCEPTED], (unsigned long long)s_acc);
    }
}

// Fallback for strides the LDS stage does not cover (> 64 bytes): per-lane byte reads.
__global__ void __launch_bounds__(DEC_BLOCK)
qs_decode_wide_kernel(const unsigned char *__restrict__ pkts, size_t n, size_t stride,
                      const unsigned short *__restrict__ lens, const double *__restrict__ offset,
                      int max_agent, int bots_per_graph, QsBatch b,
                      unsigned long long *__restrict__ graph_batch, unsigned int *__restrict__ agent_ev,
                      unsigned long long *__restrict__ counters)
{
    const size_t i = (size_t)blockIdx.x * DEC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned char *r = pkts + i * stride;
    const int len = lens ? (int)lens[i] : (int)stride;
    bool ok = false;
    if ((len == QS_PACKET_SIZE || len == QS_PACKET_SIZE_V1) && (size_t)len <= stride) {
        unsigned char f[QS_PACKET_SIZE];
        for (int q = 0; q < QS_PACKET_SIZE; q++) f[q] = r[q < len ? q : 0];
        const int agent = f[4];
        const int lmk = (len == QS_PACKET_SIZE) ? f[41] : 0;
        float x, y, yaw, d0, d1, d2, d3; int enc;
        __builtin_memcpy(&x, f + 5, 4);  __builtin_memcpy(&y, f + 9, 4);
        __builtin_memcpy(&yaw, f + 13, 4); __builtin_memcpy(&enc, f + 17, 4);
        __builtin_memcpy(&d0, f + 25, 4); __builtin_memcpy(&d1, f + 29, 4);
        __builtin_memcpy(&d2, f + 33, 4); __builtin_memcpy(&d3, f + 37, 4);
        ok = f[0] == 'Q' && f[1] == 'S' && f[2] == 'R' && f[3] == 'L' && agent >= 1 &&
             agent <= max_agent && isfinite(x) && isfinite(y) && isfinite(yaw);
        if (ok) {
            b.agent[i] = (unsigned char)agent; b.lm[i] = (unsigned char)lmk;
            b.px[i] = (double)x + offset[agent]; b.py[i] = (double)y; b.yaw[i] = (double)yaw;
            b.dist[i] = make_float4(d0, d1, d2, d3); b.enc[i] = enc;
            const int g = (agent - 1) / bots_per_graph;
            atomicAdd(&graph_batch[2 * g], 1ull);
            if (lmk) { atomicAdd(&graph_batch[2 * g + 1], 1ull); atomicAdd(&agent_ev[agent], 1u); }
            atomicAdd(&counters[QS_CNT_ACCEPTED], 1ull);
        }
    }
    b.accept[i] = ok ? 1 : 0;
    if (b.map_ok != b.accept) b.map_ok[i] = (ok && r[4] >= b.own_lo && r[4] <= b.own_hi) ? 1 : 0;
    atomicAdd(&counters[QS_CNT_DATAGRAMS], 1ull);
}

hipError_t qs_launch_decode(qs_ctx *c, const unsigned char *d_pkts, size_t n, size_t stride,
                            const unsigned short *d_lens)
{
    if (n == 0) return hipSuccess;
    const unsigned int blocks = (unsigned int)((n + DEC_BLOCK - 1) / DEC_BLOCK);
    if (stride <= DEC_MAX_STRIDE)
        hipLaunchKernelGGL(qs_decode_kernel, dim3((blocks + DEC_TILES - 1) / DEC_TILES), dim3(DEC_BLOCK), 0, c->stream, d_pkts, n,
                           stride, d_lens, c->d_offset, c->cfg.max_agent, c->bots_per_graph,
                           c->n_graphs, c->b, c->d_graph_batch, c->sb.agent_ev, c->d_counters);
    else
        hipLaunchKernelGGL(qs_decode_wide_kernel, dim3(blocks), dim3(DEC_BLOCK), 0, c->stream, d_pkts,
                           n, stride, d_lens, c->d_offset, c->cfg.max_agent, c->bots_per_graph, c->b,
                           c->d_graph_batch, c->sb.agent_ev, c->d_counters);
    return hipGetLastError();
}
