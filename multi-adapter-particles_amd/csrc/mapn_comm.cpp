// mapn_comm.cpp -- see mapn_comm.h.  The reference has no collective at all (its two adapters
// exchange positions through a D3D12 cross-adapter heap in system memory, Compute.cpp:163-201,
// Render.cpp:789-831); this is the data-sharded replacement: one ncclAllGather of the new
// float4 position slices per step.
#include "mapn_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace mapn {

namespace {

thread_local std::string g_err;

struct Api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Api g_api;
std::once_flag g_once;

void load_api()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.handle) break;
    }
    if (!g_api.handle) return;
    g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(g_api.handle, "ncclGetUniqueId"));
    g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(g_api.handle, "ncclCommInitRank"));
    g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(g_api.handle, "ncclCommDestroy"));
    g_api.AllGather = reinterpret_cast<decltype(g_api.AllGather)>(dlsym(g_api.handle, "ncclAllGather"));
    g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(g_api.handle, "ncclGetErrorString"));
    g_api.Send = reinterpret_cast<decltype(g_api.Send)>(dlsym(g_api.handle, "ncclSend"));
    g_api.Recv = reinterpret_cast<decltype(g_api.Recv)>(dlsym(g_api.handle, "ncclRecv"));
    g_api.GroupStart = reinterpret_cast<decltype(g_api.GroupStart)>(dlsym(g_api.handle, "ncclGroupStart"));
    g_api.GroupEnd = reinterpret_cast<decltype(g_api.GroupEnd)>(dlsym(g_api.handle, "ncclGroupEnd"));
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllGather && g_api.GetErrorString &&
               g_api.Send && g_api.Recv && g_api.GroupStart && g_api.GroupEnd;
}

bool api()
{
    std::call_once(g_once, load_api);
    if (!g_api.ok) g_err = "librccl.so.1 could not be loaded (dlopen/dlsym failed): RCCL is required for sharded mode";
    return g_api.ok;
}

int check(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return 0;
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_api.GetErrorString(r));
    g_err = buf;
    return -1;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

static_assert(sizeof(ncclUniqueId) == 128, "mapn.h MAPN_UNIQUE_ID_BYTES must match ncclUniqueId");

int comm_get_unique_id(void *out_id128)
{
    if (!api()) return -1;
    ncclUniqueId id;
    if (check(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return -1;
    memcpy(out_id128, &id, sizeof id);
    return 0;
}

Comm *comm_create(const void *id128, int rank, int nranks)
{
    if (!api()) return nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    if (check(g_api.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank")) {
        delete c;
        return nullptr;
    }
    return c;
}

void comm_destroy(Comm *c)
{
    if (!c) return;
    if (c->comm && g_api.ok) g_api.CommDestroy(c->comm);
    delete c;
}

int comm_all_gather_inplace(Comm *c, void *buf, size_t floats_per_rank, hipStream_t stream)
{
    if (!c || !g_api.ok) { g_err = "communicator not initialised"; return -1; }
    const float *send = static_cast<const float *>(buf) + (size_t)c->rank * floats_per_rank;
    return check(g_api.AllGather(send, buf, floats_per_rank, ncclFloat, c->comm, stream), "ncclAllGather");
}

// The same exchange as one group of point-to-point transfers: every rank sends its slice to
// each peer and receives each peer's slice in place -- one direct xGMI hop per pair (the 8 GPUs
// of a node are fully connected, 7 links each) instead of the P-1 sequential steps of a ring
// all-gather.  Which of the two is faster for 128 KiB slices is measured at run time (bench.py).
int comm_gather_sendrecv_inplace(Comm *c, void *buf, size_t floats_per_rank, hipStream_t stream)
{
    if (!c || !g_api.ok) { g_err = "communicator not initialised"; return -1; }
    float *base = static_cast<float *>(buf);
    const float *send = base + (size_t)c->rank * floats_per_rank;
    if (check(g_api.GroupStart(), "ncclGroupStart")) return -1;
    for (int d = 1; d < c->nranks; d++) {
        const int to = (c->rank + d) % c->nranks, from = (c->rank - d + c->nranks) % c->nranks;
        if (check(g_api.Send(send, floats_per_rank, ncclFloat, to, c->comm, stream), "ncclSend")) { g_api.GroupEnd(); return -1; }
        if (check(g_api.Recv(base + (size_t)from * floats_per_rank, floats_per_rank, ncclFloat, from, c->comm, stream), "ncclRecv")) { g_api.GroupEnd(); return -1; }
    }
    return check(g_api.GroupEnd(), "ncclGroupEnd");
}

// Gather algorithm 6: what sym_shard_exchange_kernel does with remote stores and counters, as RCCL transfers -- for nodes where
// peers' memory cannot be mapped (hipIpc blocked in a container).  The reference analogue of the transport being replaced:
// the cross-adapter heap of Compute.cpp:163-201 / Render.cpp:789-831.
int comm_exchange_rows(Comm *c, const void *send, void *recv, size_t floats_per_rank, unsigned send_mask, unsigned recv_mask, hipStream_t stream)
{
    if (!c || !g_api.ok) { g_err = "communicator not initialised"; return -1; }
    const float *s = static_cast<const float *>(send);
    float *r = static_cast<float *>(recv);
    bool any = false;
    for (int q = 0; q < c->nranks; q++) any = any || (q != c->rank && (((send_mask | recv_mask) >> q) & 1u));
    if (!any) return 0;
    if (check(g_api.GroupStart(), "ncclGroupStart")) return -1;
    for (int d = 1; d < c->nranks; d++) {
        const int to = (c->rank + d) % c->nranks, from = (c->rank - d + c->nranks) % c->nranks;
        if ((send_mask >> to) & 1u)
            if (check(g_api.Send(s + (size_t)to * floats_per_rank, floats_per_rank, ncclFloat, to, c->comm, stream), "ncclSend")) { g_api.GroupEnd(); return -1; }
        if ((recv_mask >> from) & 1u)
            if (check(g_api.Recv(r + (size_t)from * floats_per_rank, floats_per_rank, ncclFloat, from, c->comm, stream), "ncclRecv")) { g_api.GroupEnd(); return -1; }
    }
    return check(g_api.GroupEnd(), "ncclGroupEnd");
}

const char *comm_last_error() { return g_err.c_str(); }

}  // namespace mapn
