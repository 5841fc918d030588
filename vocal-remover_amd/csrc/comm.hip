// Data-parallel exchange of the train step: one in-place SUM all-reduce of the single flat
// gradient arena per optimizer step, RCCL over xGMI, on the handle's own HIP stream (so the fused Adam that follows
// needs no host synchronisation), plus the rank-0 broadcast of weights / BatchNorm buffers / Adam state that makes
// the replicas start identical.  The reference has no multi-GPU path (train.py:211-213: a single --gpu).
//
// RCCL is bound at run time (dlopen), not at link time: a process that already has an RCCL mapped -- torch ships its
// own librccl.so.1 -- must use THAT copy (two RCCLs in one process would each keep their own device state), and
// single-GPU users of libvr_mi355.so never load it at all.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "model.h"

namespace vr {

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char* forced = getenv("VR_RCCL_LIB");      // this library and nothing else (deployments with several RCCLs; the tests'
        if (forced && *forced) {                         // "no RCCL on this machine" case)
            api.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* n : names) {      // 1. whatever RCCL this process already mapped (torch's)
                api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
                if (api.lib) break;
            }
            for (int i = 0; !api.lib && i < 3; ++i) api.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);   // 2. the system's
        }
        if (!api.lib) {
            const char* why = dlerror();                  // (one call: dlerror() clears the message it returns)
            api.error = std::string("cannot load RCCL: ") + (why ? why : "?");
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(api.lib, s);
            if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!api.error.empty()) throw Error(-8, api.error);
    return api;
}

void nccl_check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return;
    RcclApi& a = rccl();
    throw Error(-8, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == 128, "vr_comm_unique_id hands out 128 bytes");

void comm_unique_id(void* out128) {
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof id);
}

void Model::comm_init(int rank, int world, const void* id128) {
    DeviceGuard dev_guard(device);
    VR_CHECK(world >= 1 && rank >= 0 && rank < world, -2, "comm: rank / world out of range");
    VR_CHECK(id128 != nullptr, -2, "comm: null unique id");
    comm_destroy();
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    nccl_check(rccl().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    comm = c; comm_rank = rank; comm_world = world;
}

void Model::comm_destroy() {
    if (!comm) return;
    DeviceGuard dev_guard(device);
    hipStreamSynchronize(stream);
    rccl().CommDestroy(static_cast<ncclComm_t>(comm));
    comm = nullptr; comm_rank = 0; comm_world = 1;
}

// SUM over ranks, in place, of the flat gradient arena (14,740,834 trainable elements + layout padding for the
// default net; `aux_out.weight` never receives a gradient -- lib/nets.py:80 -- and contributes zeros).  Enqueued on
// the handle's stream: vr_adam_step(grad_scale = 1/world) follows on the same stream with no host round trip.
// dtype 1 = bf16 wire format: the bucket is rounded to bf16, summed by RCCL in bf16 (half the xGMI bytes) and widened
// back into the fp32 arena; 0 = fp32 (exact sum order aside, what gradient accumulation computes).
void Model::allreduce_grads(int wire_dtype) {
    DeviceGuard dev_guard(device);
    VR_CHECK(comm != nullptr, -2, "vr_allreduce_grads: call vr_comm_init first");
    VR_CHECK(wire_dtype == 0 || wire_dtype == 1, -2, "vr_allreduce_grads: dtype must be 0 (fp32) or 1 (bf16)");
    ensure_train_state();
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    if (wire_dtype == 0) {
        nccl_check(rccl().AllReduce(g_arena, g_arena, p_floats, ncclFloat32, ncclSum, c, stream), "ncclAllReduce(fp32 bucket)");
    } else {
        if (!wire_buf) VR_HIP(hipMalloc(&wire_buf, p_floats * sizeof(unsigned short)));
        launch_f32_to_bf16(g_arena, static_cast<unsigned short*>(wire_buf), (long long)p_floats, stream);
        nccl_check(rccl().AllReduce(wire_buf, wire_buf, p_floats, ncclBfloat16, ncclSum, c, stream), "ncclAllReduce(bf16 bucket)");
        launch_bf16_to_f32(static_cast<const unsigned short*>(wire_buf), g_arena, (long long)p_floats, stream);
    }
}

// Rank `root`'s trainable parameters, BatchNorm buffers (running statistics) and
// num_batches_tracked counters replace everyone's; `with_optimizer`: the Adam moments and step counter too.
void Model::broadcast_params(int root, bool with_optimizer) {
    DeviceGuard dev_guard(device);
    VR_CHECK(comm != nullptr, -2, "vr_broadcast_params: call vr_comm_init first");
    VR_CHECK(root >= 0 && root < comm_world, -2, "vr_broadcast_params: root out of range");
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    RcclApi& R = rccl();
    std::vector<Param*> counters;
    for (auto& p : params) if (p.kind == PK_NBT) counters.push_back(&p);
    std::vector<long long> host(counters.size() + 1);
    for (size_t i = 0; i < counters.size(); ++i) host[i] = counters[i]->nbt;
    host.back() = adam_step;
    long long* dcnt = nullptr;
    VR_HIP(hipMalloc(&dcnt, host.size() * sizeof(long long)));
    struct Free { void* p; ~Free() { hipFree(p); } } free_dcnt{dcnt};
    VR_HIP(hipMemcpyAsync(dcnt, host.data(), host.size() * sizeof(long long), hipMemcpyHostToDevice, stream));
    if (with_optimizer) ensure_train_state();
    nccl_check(R.GroupStart(), "ncclGroupStart");
    nccl_check(R.Broadcast(p_arena, p_arena, p_floats, ncclFloat32, root, c, stream), "ncclBroadcast(parameters)");
    nccl_check(R.Broadcast(b_arena, b_arena, b_floats, ncclFloat32, root, c, stream), "ncclBroadcast(buffers)");
    nccl_check(R.Broadcast(dcnt, dcnt, host.size(), ncclInt64, root, c, stream), "ncclBroadcast(counters)");
    if (with_optimizer) {
        nccl_check(R.Broadcast(m_arena, m_arena, p_floats, ncclFloat32, root, c, stream), "ncclBroadcast(adam m)");
        nccl_check(R.Broadcast(v_arena, v_arena, p_floats, ncclFloat32, root, c, stream), "ncclBroadcast(adam v)");
    }
    nccl_check(R.GroupEnd(), "ncclGroupEnd");
    VR_HIP(hipMemcpyAsync(host.data(), dcnt, host.size() * sizeof(long long), hipMemcpyDeviceToHost, stream));
    VR_HIP(hipStreamSynchronize(stream));
    for (size_t i = 0; i < counters.size(); ++i) counters[i]->nbt = host[i];
    if (with_optimizer) adam_step = host.back();
    affine_dirty = true;
}

}  // namespace vr
