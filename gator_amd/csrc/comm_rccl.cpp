// gator_allgather_verts: the path's ONE collective (all-gather of the predicted vertices [B/N, 6890, 3] and pose3d
// [B/N, J, 3] over xGMI) on RCCL directly, for hosts that do not bring torch.distributed.  The reference has no counterpart
// (no distributed code at all).
//
// RCCL is resolved at run time, never linked: a PyTorch process already carries its own librccl (torch/lib/librccl.so), and a
// second copy with the same SONAME must not be pulled in beside it.  dlopen(RTLD_NOLOAD) picks up whatever is loaded; only a
// process without one loads the system library.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "internal.h"

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);            // the copy the process already uses (PyTorch's), if any
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    });
    return (r.h && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd) ? &r : nullptr;
}

#define GATOR_NCCL_CHECK(R, expr)                                                                                       \
    do {                                                                                                                \
        ncclResult_t _e = (expr);                                                                                       \
        if (_e != ncclSuccess)                                                                                          \
            return gator::fail(GATOR_EHIP, "%s failed: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(_e) : "?"); \
    } while (0)

}  // namespace

struct gator_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

static_assert(sizeof(ncclUniqueId) == GATOR_COMM_ID_BYTES, "gator_comm_unique_id: id size");

extern "C" int gator_comm_unique_id(uint8_t* id) {
    if (!id) return gator::fail(GATOR_EINVAL, "gator_comm_unique_id: null id");
    Rccl* R = rccl();
    if (!R) {
        const char* why = dlerror();                  // (a second call would return NULL: the message is consumed by the first)
        return gator::fail(GATOR_EUNSUPPORTED, "gator_comm_unique_id: librccl not found (%s)", why ? why : "no error");
    }
    ncclUniqueId u;
    GATOR_NCCL_CHECK(R, R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return GATOR_OK;
}

extern "C" int gator_comm_create(const uint8_t* id, int32_t rank, int32_t world, gator_comm** out) {
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return gator::fail(GATOR_EINVAL, "gator_comm_create: bad arguments");
    Rccl* R = rccl();
    if (!R) return gator::fail(GATOR_EUNSUPPORTED, "gator_comm_create: librccl not found");
    gator_comm* c = new gator_comm();
    c->rank = rank; c->world = world;
    (void)hipGetDevice(&c->device);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t e = R->CommInitRank(&c->comm, world, u, rank);            // one communicator per process, on the current device
    if (e != ncclSuccess) { delete c; return gator::fail(GATOR_EHIP, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?"); }
    *out = c;
    return GATOR_OK;
}

extern "C" int gator_comm_destroy(gator_comm* c) {
    if (!c) return GATOR_OK;
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return GATOR_OK;
}

extern "C" int gator_allgather_verts(gator_comm* c, const float* verts_local, const float* pose3d_local, int32_t batch_local,
                                     int32_t num_joint, float* verts_all, float* pose3d_all, void* stream) {
    if (!c || !verts_local || !verts_all || batch_local <= 0) return gator::fail(GATOR_EINVAL, "gator_allgather_verts: bad arguments");
    if ((pose3d_local == nullptr) != (pose3d_all == nullptr)) return gator::fail(GATOR_EINVAL, "gator_allgather_verts: pose3d_local and pose3d_all go together");
    Rccl* R = rccl();
    if (!R) return gator::fail(GATOR_EUNSUPPORTED, "gator_allgather_verts: librccl not found");
    // rank-major concatenation: rank r's samples land at rows r*batch_local ... of the outputs on every rank
    GATOR_NCCL_CHECK(R, R->GroupStart());
    ncclResult_t e1 = R->AllGather(verts_local, verts_all, (size_t)batch_local * gator::kNV * 3, ncclFloat, c->comm, (hipStream_t)stream);
    ncclResult_t e2 = ncclSuccess;
    if (pose3d_local) e2 = R->AllGather(pose3d_local, pose3d_all, (size_t)batch_local * num_joint * 3, ncclFloat, c->comm, (hipStream_t)stream);
    GATOR_NCCL_CHECK(R, R->GroupEnd());
    GATOR_NCCL_CHECK(R, e1);
    GATOR_NCCL_CHECK(R, e2);
    return GATOR_OK;
}
