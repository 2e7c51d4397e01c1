// Data-parallel exchange steps of the hot path over RCCL / xGMI (SURVEY.md §8b minimum set, §8e):
//   gradients            one all-reduce(AVG) of the flat fp32 gradient buffer per optimizer step
//                        (cusrl/utils/distributed.py:145-172, caller cusrl/template/actor_critic.py:314)
//   advantage statistics one all-gather of cat(mean, var) per update (cusrl/utils/distributed.py:101-110,175-183)
//   parameters           one broadcast from rank 0 at agent construction (cusrl/utils/distributed.py:58-63)
// Every call is ENQUEUED on the caller's hipStream_t and returns; RCCL kernels launched this way are legal inside
// hipGraph capture, so the gradient all-reduce becomes a node of the captured minibatch step instead of an eager call
// between two graphs.  All messages are latency-bound (<= 5 MB), one communicator per process (one process per GPU).
//
// RCCL is resolved at RUN time from the copy already loaded into the process (PyTorch-ROCm ships its own librccl.so;
// linking a second one at build time would put two RCCL instances into one process): dlopen(RTLD_NOLOAD) first, the
// system library second.  Nothing here allocates device memory or synchronises a stream.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.hpp"

namespace {

// The handful of RCCL declarations used (rccl.h, NCCL 2.x ABI: stable enums / struct sizes).
constexpr int kUniqueIdBytes = 128;
struct UniqueId {
    char internal[kUniqueIdBytes];
};
using Comm = void *;
enum { kNcclSuccess = 0 };
enum { kNcclSum = 0, kNcclAvg = 4 };
enum { kNcclUint8 = 1, kNcclFloat32 = 7 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local int g_last_rccl_result = 0;

template <typename F>
bool resolve(void *handle, const char *name, F &slot) {
    slot = reinterpret_cast<F>(dlsym(handle, name));
    return slot != nullptr;
}

void load_rccl() {
    const char *override_path = getenv("CUSRL_RCCL_LIBRARY");
    void *handle = nullptr;
    if (override_path && *override_path) {
        // an explicit library is used or nothing is: silently loading another RCCL than the one the user named would hide
        // exactly the misconfiguration the variable exists to test (the job then takes torch.distributed's collectives)
        handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!handle) return;
    }
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
        if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // the instance PyTorch already loaded
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!handle) return;
    Rccl &r = g_rccl;
    r.handle = handle;
    r.ok = resolve(handle, "ncclGetUniqueId", r.GetUniqueId) && resolve(handle, "ncclCommInitRank", r.CommInitRank) &&
           resolve(handle, "ncclCommDestroy", r.CommDestroy) && resolve(handle, "ncclAllReduce", r.AllReduce) &&
           resolve(handle, "ncclAllGather", r.AllGather) && resolve(handle, "ncclBroadcast", r.Broadcast) &&
           resolve(handle, "ncclGetErrorString", r.GetErrorString);
    resolve(handle, "ncclCommAbort", r.CommAbort);  // optional: destroy is the fallback
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int comm_status(int result) {
    g_last_rccl_result = result;
    return result == kNcclSuccess ? 0 : CUSRL_E_COMM;
}

}  // namespace

struct cusrl_comm {
    Comm comm;
    int world, rank;
};

extern "C" int cusrl_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" const char *cusrl_comm_last_error(void) {
    const Rccl *r = rccl();
    if (!r) return "RCCL could not be loaded (librccl.so not found in the process or on the library path)";
    return r->GetErrorString(g_last_rccl_result);
}

extern "C" int cusrl_comm_unique_id(void *id_out) {
    const Rccl *r = rccl();
    if (!r) return CUSRL_E_COMM;
    if (!id_out) return CUSRL_E_INVALID;
    UniqueId id;
    if (int rc = comm_status(r->GetUniqueId(&id))) return rc;
    std::memcpy(id_out, id.internal, kUniqueIdBytes);
    return 0;
}

extern "C" int cusrl_comm_create(const void *id_bytes, int world, int rank, cusrl_comm_t **comm_out) {
    const Rccl *r = rccl();
    if (!r) return CUSRL_E_COMM;
    if (!id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return CUSRL_E_INVALID;
    UniqueId id;
    std::memcpy(id.internal, id_bytes, kUniqueIdBytes);
    Comm comm = nullptr;
    if (int rc = comm_status(r->CommInitRank(&comm, world, id, rank))) return rc;  // binds to the CURRENT hip device
    *comm_out = new cusrl_comm{comm, world, rank};
    return 0;
}

extern "C" int cusrl_comm_destroy(cusrl_comm_t *comm) {
    const Rccl *r = rccl();
    if (!comm) return 0;
    int rc = r ? comm_status(r->CommDestroy(comm->comm)) : CUSRL_E_COMM;
    delete comm;
    return rc;
}

// Abandon a communicator whose collectives may never complete (a peer failed before enqueueing its half): ncclCommAbort
// stops the communicator's in-flight kernels instead of waiting for them like ncclCommDestroy.
extern "C" int cusrl_comm_abort(cusrl_comm_t *comm) {
    const Rccl *r = rccl();
    if (!comm) return 0;
    int rc = CUSRL_E_COMM;
    if (r) rc = comm_status(r->CommAbort ? r->CommAbort(comm->comm) : r->CommDestroy(comm->comm));
    delete comm;
    return rc;
}

extern "C" int cusrl_comm_world_size(const cusrl_comm_t *comm) { return comm ? comm->world : 0; }

extern "C" int cusrl_allreduce_mean(float *buffer, int64_t count, cusrl_comm_t *comm, void *stream) {
    const Rccl *r = rccl();
    if (!r) return CUSRL_E_COMM;
    if (!comm || count < 0 || (count > 0 && !buffer)) return CUSRL_E_INVALID;
    if (count == 0) return 0;
    return comm_status(r->AllReduce(buffer, buffer, size_t(count), kNcclFloat32, kNcclAvg, comm->comm,
                                    cusrl::as_stream(stream)));
}

extern "C" int cusrl_allgather(const void *input, void *output, int64_t bytes_per_rank, cusrl_comm_t *comm,
                               void *stream) {
    const Rccl *r = rccl();
    if (!r) return CUSRL_E_COMM;
    if (!comm || bytes_per_rank < 0 || (bytes_per_rank > 0 && (!input || !output))) return CUSRL_E_INVALID;
    if (bytes_per_rank == 0) return 0;
    return comm_status(r->AllGather(input, output, size_t(bytes_per_rank), kNcclUint8, comm->comm,
                                    cusrl::as_stream(stream)));
}

extern "C" int cusrl_broadcast(void *buffer, int64_t bytes, int root, cusrl_comm_t *comm, void *stream) {
    const Rccl *r = rccl();
    if (!r) return CUSRL_E_COMM;
    if (!comm || bytes < 0 || (bytes > 0 && !buffer) || root < 0 || root >= comm->world) return CUSRL_E_INVALID;
    if (bytes == 0) return 0;
    return comm_status(r->Broadcast(buffer, buffer, size_t(bytes), kNcclUint8, root, comm->comm,
                                    cusrl::as_stream(stream)));
}
