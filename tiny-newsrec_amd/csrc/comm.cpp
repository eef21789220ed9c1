// tnr_comm_*: the gradient exchange of the data-parallel step behind the C ABI (replaces hvd.init,
// hvd.broadcast_parameters / broadcast_optimizer_state and hvd.DistributedOptimizer's all-reduce: Tiny-NewsRec/run.py:141-149,
// utils.py:43-60).  RCCL is bound at RUN time (dlopen of librccl.so.1 - inside a PyTorch-ROCm process that is the copy torch has
// already loaded), so the library has no link-time dependency on it: without RCCL every entry point returns TNR_EUNSUPPORTED.
// One process per GPU; the caller's current device is the communicator's device (the library never calls hipSetDevice); every
// collective is asynchronous on the stream it is given.  The unique id travels out of band (the host side uses the rendezvous
// store torch.distributed already has: dist.py).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "common.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) return;
#define TNR_SYM(field, sym) *(void**)(&r.field) = dlsym(r.h, sym)
        TNR_SYM(GetUniqueId, "ncclGetUniqueId");
        TNR_SYM(CommInitRank, "ncclCommInitRank");
        TNR_SYM(CommDestroy, "ncclCommDestroy");
        TNR_SYM(Broadcast, "ncclBroadcast");
        TNR_SYM(AllReduce, "ncclAllReduce");
        TNR_SYM(ReduceScatter, "ncclReduceScatter");
        TNR_SYM(AllGather, "ncclAllGather");
        TNR_SYM(GetErrorString, "ncclGetErrorString");
#undef TNR_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Broadcast && r.AllReduce && r.ReduceScatter && r.AllGather &&
               r.GetErrorString;
    });
    return &r;
}

int need(const char* who) {
    if (rccl()->ok) return TNR_OK;
    tnr_set_error("%s: RCCL is not available in this process (dlopen librccl.so.1: %s)", who, rccl()->h ? "symbols missing" : "not found");
    return TNR_EUNSUPPORTED;
}

int check(ncclResult_t rc, const char* who) {
    if (rc == ncclSuccess) return TNR_OK;
    tnr_set_error("%s: RCCL error %d (%s)", who, (int)rc, rccl()->GetErrorString(rc));
    return TNR_ELAUNCH;
}
}  // namespace

struct tnr_comm {
    ncclComm_t c;
    int world, rank;
};

extern "C" int tnr_comm_unique_id(void* id128) {
    if (int rc = need("tnr_comm_unique_id")) return rc;
    TNR_CHECK_ARG(id128, "tnr_comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (int rc = check(rccl()->GetUniqueId(&id), "tnr_comm_unique_id")) return rc;
    memcpy(id128, &id, sizeof(id));
    return TNR_OK;
}

extern "C" int tnr_comm_init(const void* id128, int world, int rank, tnr_comm** comm) {
    if (int rc = need("tnr_comm_init")) return rc;
    TNR_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world, "tnr_comm_init: bad arguments (world %d rank %d)", world, rank);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (int rc = check(rccl()->CommInitRank(&c, world, id, rank), "tnr_comm_init")) return rc;
    *comm = new tnr_comm{c, world, rank};
    return TNR_OK;
}

extern "C" int tnr_comm_destroy(tnr_comm* comm) {
    if (!comm) return TNR_OK;
    int rc = rccl()->ok ? check(rccl()->CommDestroy(comm->c), "tnr_comm_destroy") : TNR_OK;
    delete comm;
    return rc;
}

extern "C" int tnr_comm_broadcast(tnr_comm* comm, float* buf, int64_t n, int root, void* stream) {
    if (int rc = need("tnr_comm_broadcast")) return rc;
    TNR_CHECK_ARG(comm && buf && n >= 0 && root >= 0 && root < comm->world, "tnr_comm_broadcast: bad arguments");
    if (n == 0) return TNR_OK;
    return check(rccl()->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, comm->c, (hipStream_t)stream), "tnr_comm_broadcast");
}

extern "C" int tnr_comm_allreduce_avg(tnr_comm* comm, float* buf, int64_t n, int average, void* stream) {
    if (int rc = need("tnr_comm_allreduce_avg")) return rc;
    TNR_CHECK_ARG(comm && buf && n >= 0, "tnr_comm_allreduce_avg: bad arguments");
    if (n == 0) return TNR_OK;
    return check(rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, comm->c, (hipStream_t)stream),
                 "tnr_comm_allreduce_avg");
}

// direct exchange on the fully connected xGMI mesh: reduce-scatter into `shard` (n / world floats, caller-owned), all-gather back
extern "C" int tnr_comm_reduce_scatter_allgather(tnr_comm* comm, float* buf, float* shard, int64_t n, int average, void* stream) {
    if (int rc = need("tnr_comm_reduce_scatter_allgather")) return rc;
    TNR_CHECK_ARG(comm && buf && shard && n >= 0 && (n % comm->world) == 0, "tnr_comm_reduce_scatter_allgather: n must be a multiple of the world size");
    if (n == 0) return TNR_OK;
    const size_t per = (size_t)(n / comm->world);
    if (int rc = check(rccl()->ReduceScatter(buf, shard, per, ncclFloat32, average ? ncclAvg : ncclSum, comm->c, (hipStream_t)stream),
                       "tnr_comm_reduce_scatter_allgather")) return rc;
    return check(rccl()->AllGather(shard, buf, per, ncclFloat32, comm->c, (hipStream_t)stream), "tnr_comm_reduce_scatter_allgather");
}

extern "C" int tnr_comm_world(const tnr_comm* comm, int* world, int* rank) {
    TNR_CHECK_ARG(comm && world && rank, "tnr_comm_world: null pointer");
    *world = comm->world;
    *rank = comm->rank;
    return TNR_OK;
}
