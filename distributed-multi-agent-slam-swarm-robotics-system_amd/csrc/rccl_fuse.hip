// rccl_fuse.hip -- the sparse grid fuse (sparse_fuse.hip) with its two exchanges on RCCL, behind the C ABI.
//
// dist.py drives the same three device steps from Python with torch.distributed (backend "nccl" = RCCL); a host that is not
// torch -- the reference's own process, a C++ service -- calls qs_sparse_fuse_rccl instead: one process per GPU, one RCCL
// communicator over the node's xGMI links, and per fuse
//     qs_sparse_fuse_begin  ->  ncclAllGather of the ranks' block bitmaps (in place)
//     qs_sparse_fuse_plan   ->  ncclSend of this rank's packed blocks to every peer + ncclRecv of theirs, one group:
//                               point to point, all links at once (xGMI is a full mesh: a ring would be bound by one link)
//     qs_sparse_fuse_apply
// RCCL is loaded with dlopen at first use: the library itself does not depend on it.
#include <dlfcn.h>
#include <string.h>

#include "qs_internal.h"

// the few RCCL entry points and enum values used (rccl.h: ncclUint8 = 1, ncclInt32 = 2; ncclSuccess = 0)
typedef void *qs_nccl_comm;
struct QsRccl {
    void *h;
    int (*GetUniqueId)(void *id);
    int (*CommInitRank)(qs_nccl_comm *comm, int nranks, QsNcclId id, int rank);
    int (*CommDestroy)(qs_nccl_comm comm);
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, qs_nccl_comm comm, hipStream_t st);
    int (*Send)(const void *buf, size_t count, int dtype, int peer, qs_nccl_comm comm, hipStream_t st);
    int (*Recv)(void *buf, size_t count, int dtype, int peer, qs_nccl_comm comm, hipStream_t st);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char *(*GetErrorString)(int);
};
static QsRccl g_rccl;

static const char *rccl_load()
{
    if (g_rccl.h) return nullptr;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return "librccl.so not found (RCCL is loaded on demand)";
#define QS_SYM(field, name) *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return "librccl.so lacks " name
    QS_SYM(GetUniqueId, "ncclGetUniqueId"); QS_SYM(CommInitRank, "ncclCommInitRank"); QS_SYM(CommDestroy, "ncclCommDestroy");
    QS_SYM(AllGather, "ncclAllGather"); QS_SYM(Send, "ncclSend"); QS_SYM(Recv, "ncclRecv");
    QS_SYM(GroupStart, "ncclGroupStart"); QS_SYM(GroupEnd, "ncclGroupEnd"); QS_SYM(GetErrorString, "ncclGetErrorString");
#undef QS_SYM
    g_rccl.h = h;
    return nullptr;
}

static int rccl_fail(qs_ctx *c, const char *what, int rc)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, (rc && g_rccl.GetErrorString) ? g_rccl.GetErrorString(rc) : "RCCL unavailable");
    if (c) c->err = buf;
    return QS_E_HIP;
}

extern "C" int qs_rccl_unique_id(uint8_t out[QS_RCCL_ID_BYTES])
{
    if (!out) return QS_E_INVAL;
    if (rccl_load()) return QS_E_NODEV;
    QsNcclId id;
    if (g_rccl.GetUniqueId(&id) != 0) return QS_E_HIP;
    memcpy(out, id.internal, QS_RCCL_ID_BYTES);
    return QS_OK;
}

extern "C" int qs_rccl_comm_init(qs_ctx *c, const uint8_t id_bytes[QS_RCCL_ID_BYTES], int32_t world, int32_t rank, void **comm)
{
    if (!c || !id_bytes || !comm || world < 1 || rank < 0 || rank >= world) return QS_E_INVAL;
    const char *e = rccl_load();
    if (e) { c->err = e; return QS_E_NODEV; }
    if (hipSetDevice(c->device) != hipSuccess) return QS_E_HIP;
    QsNcclId id;
    memcpy(id.internal, id_bytes, QS_RCCL_ID_BYTES);
    qs_nccl_comm cm = nullptr;
    const int rc = g_rccl.CommInitRank(&cm, world, id, rank);
    if (rc != 0) return rccl_fail(c, "ncclCommInitRank", rc);
    *comm = cm;
    return QS_OK;
}

extern "C" int qs_rccl_comm_destroy(void *comm)
{
    if (!comm) return QS_OK;
    if (rccl_load()) return QS_E_NODEV;
    return g_rccl.CommDestroy((qs_nccl_comm)comm) == 0 ? QS_OK : QS_E_HIP;
}

extern "C" int qs_sparse_fuse_rccl(qs_ctx *c, void *comm, int32_t world, int32_t rank, uint64_t stats[4])
{
    if (!c || !comm || world < 1 || world > QS_SPARSE_MAX_WORLD || rank < 0 || rank >= world) return QS_E_INVAL;
    const char *e = rccl_load();
    if (e) { c->err = e; return QS_E_NODEV; }
    void *bitmaps = nullptr; size_t bm_bytes = 0;
    int rc = qs_sparse_fuse_begin(c, world, rank, &bitmaps, &bm_bytes);
    if (rc != QS_OK) return rc;
    qs_nccl_comm cm = (qs_nccl_comm)comm;
    if (world > 1) {
        // in place: this rank's bitmap already sits in slot `rank` of the receive buffer
        const int r = g_rccl.AllGather((const char *)bitmaps + (size_t)rank * bm_bytes, bitmaps, bm_bytes / 4, 2 /* ncclInt32 */, cm, c->stream);
        if (r != 0) return rccl_fail(c, "ncclAllGather", r);
    }
    uint32_t n_blocks[QS_SPARSE_MAX_WORLD]; size_t off[QS_SPARSE_MAX_WORLD + 1];
    void *payload = nullptr; size_t bb = 0;
    rc = qs_sparse_fuse_plan(c, n_blocks, off, &payload, &bb);          // (waits for the stream: the all-gather is done)
    if (rc != QS_OK) return rc;
    if (world > 1) {
        const size_t mine = off[rank + 1] - off[rank];
        int r = g_rccl.GroupStart();
        for (int p = 0; p < world && r == 0; p++) {
            if (p == rank) continue;
            if (mine) r = g_rccl.Send((const char *)payload + off[rank], mine, 1 /* ncclUint8 */, p, cm, c->stream);
            if (r == 0 && off[p + 1] > off[p]) r = g_rccl.Recv((char *)payload + off[p], off[p + 1] - off[p], 1, p, cm, c->stream);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r != 0 || r2 != 0) return rccl_fail(c, "ncclSend / ncclRecv", r ? r : r2);
    }
    rc = qs_sparse_fuse_apply(c);
    if (rc != QS_OK) return rc;
    if (stats) {
        const uint64_t own = off[rank + 1] - off[rank];
        stats[0] = n_blocks[rank]; stats[1] = own;                                   // blocks, bytes this rank packed
        stats[2] = own * (uint64_t)(world - 1) + (uint64_t)(world - 1) * bm_bytes;   // bytes sent
        stats[3] = (off[world] - own) + (uint64_t)(world - 1) * bm_bytes;            // bytes received
    }
    return QS_OK;
}
