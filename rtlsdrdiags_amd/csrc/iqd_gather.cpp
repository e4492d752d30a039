// PCM gather over RCCL / xGMI for hosts that run one engine per GPU (channels are the shard, the data
// path has no collective; what a node may want is the ranks' PCM in one place).  Plain C ABI (include/iqdemod.h), no
// torch: the C++ host of north_star calls this directly; bench.py --gather uses it too.
//
// One communicator per gatherer.  A gather is a group of point-to-point transfers on the ENGINE's stream - the root
// posts one ncclRecv per peer into its row of the destination, every other rank one ncclSend - so it is ordered behind
// the accept that produced the PCM without a host synchronisation, and the next accept is ordered behind it.  PCM is
// 1/32 of the input volume and goes to one rank: a direct gather, not a ring (xGMI is point to point).
//
// librccl.so is opened when the first gatherer is created, not when libiqdemod.so is loaded: a single-GPU host never
// pays for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// The handful of RCCL declarations this file uses (rccl.h: the NCCL 2 API), so that the library builds on a host without
// the RCCL headers; the library itself is opened at run time either way.
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#include <mutex>
#include <new>
#include <vector>

#include "iqdemod.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    bool ok = false;
    bool reused = false;   // the process had an RCCL loaded already (torch's): that copy is used, not a second one
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy the process has loaded already (bench.py: torch brings its own) is the one to use - two RCCL instances
        // in one process each set up their own transports and proxy threads.
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.lib) { r.reused = true; break; }
        }
        if (!r.lib)
            for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        if (!r.lib) return;
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
}

}  // namespace

struct iqd_gather {
    iqd_t *e = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1, root = 0;
    int device = 0;
};

extern "C" {

int iqd_gather_unique_id(uint8_t *id128)
{
    if (!id128) return IQD_EINVAL;
    Rccl &r = rccl();
    if (!r.ok) return IQD_ENODEV;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return IQD_EHIP;
    static_assert(sizeof(id) == IQD_GATHER_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return IQD_OK;
}

int iqd_gather_create(iqd_t *e, const uint8_t *id128, uint32_t rank, uint32_t world, uint32_t root, iqd_gather_t **out)
{
    if (!e || !id128 || !out || world == 0 || rank >= world || root >= world) return IQD_EINVAL;
    Rccl &r = rccl();
    if (!r.ok) return IQD_ENODEV;
    iqd_gather *g = new (std::nothrow) iqd_gather;
    if (!g) return IQD_ENOMEM;
    g->e = e;
    g->rank = rank;
    g->world = world;
    g->root = root;
    // the communicator belongs to the ENGINE's device, whatever device the calling thread has current (like every iqd_* entry)
    if (iqd_get_device(e, &g->device) != IQD_OK || hipSetDevice(g->device) != hipSuccess) { delete g; return IQD_ENODEV; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (r.CommInitRank(&g->comm, (int)world, id, (int)rank) != ncclSuccess) {
        delete g;
        return IQD_EHIP;
    }
    *out = g;
    return IQD_OK;
}

void iqd_gather_destroy(iqd_gather_t *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)iqd_synchronize(g->e);
    if (g->comm) (void)rccl().CommDestroy(g->comm);
    delete g;
}

// Every rank's `bytes_per_rank[rank]` bytes at `send_dev` go to row `rank` of `recv_dev` on the root (rows `row_stride`
// bytes apart; recv_dev is only looked at on the root).  All ranks pass the same bytes_per_rank[0..world).  Queued on the
// engine's stream; returns at once.
int iqd_gather_pcm(iqd_gather_t *g, const void *send_dev, const size_t *bytes_per_rank, void *recv_dev, size_t row_stride)
{
    if (!g || !bytes_per_rank || (bytes_per_rank[g->rank] && !send_dev)) return IQD_EINVAL;
    if (g->rank == g->root && !recv_dev) return IQD_EINVAL;
    for (uint32_t r = 0; r < g->world; r++)
        if (bytes_per_rank[r] > row_stride) return IQD_EINVAL;
    Rccl &r = rccl();
    if (hipSetDevice(g->device) != hipSuccess) return IQD_EHIP;
    hipStream_t s = (hipStream_t)iqd_stream(g->e);
    if (g->rank == g->root) {
        // the root's own rows: a device copy on the same stream
        if (bytes_per_rank[g->rank] &&
            hipMemcpyAsync((char *)recv_dev + (size_t)g->rank * row_stride, send_dev, bytes_per_rank[g->rank], hipMemcpyDeviceToDevice, s) != hipSuccess)
            return IQD_EHIP;
        if (g->world == 1) return IQD_OK;
        bool ok = r.GroupStart() == ncclSuccess;
        for (uint32_t p = 0; p < g->world && ok; p++)
            if (p != g->root && bytes_per_rank[p])
                ok = r.Recv((char *)recv_dev + (size_t)p * row_stride, bytes_per_rank[p], ncclUint8, (int)p, g->comm, s) == ncclSuccess;
        ok = (r.GroupEnd() == ncclSuccess) && ok;
        return ok ? IQD_OK : IQD_EHIP;
    }
    if (!bytes_per_rank[g->rank]) return IQD_OK;
    bool ok = r.GroupStart() == ncclSuccess;
    ok = ok && r.Send(send_dev, bytes_per_rank[g->rank], ncclUint8, (int)g->root, g->comm, s) == ncclSuccess;
    ok = (r.GroupEnd() == ncclSuccess) && ok;
    return ok ? IQD_OK : IQD_EHIP;
}

// What the communicator really is: RCCL's version code (major * 10000 + minor * 100 + patch for 2.9 and later), the rank
// count the communicator itself reports (ncclCommCount), and whether the library was already in the process.
int iqd_gather_info(iqd_gather_t *g, int *rccl_version, int *comm_ranks, int *library_reused)
{
    Rccl &r = rccl();
    if (!r.ok) return IQD_ENODEV;
    if (rccl_version) {
        *rccl_version = 0;
        if (r.GetVersion) (void)r.GetVersion(rccl_version);
    }
    if (comm_ranks) {
        *comm_ranks = g ? (int)g->world : 0;
        if (g && g->comm && r.CommCount && r.CommCount(g->comm, comm_ranks) != ncclSuccess) return IQD_EHIP;
    }
    if (library_reused) *library_reused = r.reused ? 1 : 0;
    return IQD_OK;
}

}  // extern "C"
