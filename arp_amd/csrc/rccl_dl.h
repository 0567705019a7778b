// RCCL bound at run time (shared by the policy step, arp_dt.hip, and the fine-tune head, arp_ft.hip).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: librccl is dlopen'ed, never linked

#include <string>

#include "common.h"

namespace arp {

// RCCL is bound at run time.  Linking it would make every process that loads libarp_hip.so also load
// /opt/rocm's librccl next to the copy PyTorch-ROCm bundles (same SONAME, different file) -- two RCCLs in one
// process abort in glibc at exit.  dlopen("librccl.so.1") returns whichever copy is already loaded, else the
// system one.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;  // optional: several all-reduces of one bucket as one RCCL group
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional: what the communicator itself says (comm_info below)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    bool ok = false;
};
inline RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.ok ? &api : nullptr;
    tried = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return nullptr;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(h, "ncclBroadcast"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(h, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    api.CommCuDevice = reinterpret_cast<decltype(api.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
    api.ok = api.GetErrorString && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Broadcast;
    return api.ok ? &api : nullptr;
}

inline int rccl_fail(const char* what, ncclResult_t r) { return fail(std::string(what) + ": " + rccl_api()->GetErrorString(r)); }

// What the COMMUNICATOR says about itself (not what the caller passed in): info5 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice,
// ncclGetVersion code, 1 if a communicator exists}.  Without a communicator (world 1): {1, 0, device, 0, 0}.  A multi-GPU bench line
// prints these so that "did RCCL see N ranks" can be read off it.
inline int rccl_comm_info(ncclComm_t comm, bool has_comm, int device, int32_t* info5) {
    info5[0] = 1; info5[1] = 0; info5[2] = device; info5[3] = 0; info5[4] = has_comm ? 1 : 0;
    if (!has_comm) return 0;
    RcclApi* a = rccl_api();
    if (!a) return fail("librccl.so.1 could not be loaded");
    int v = -1;
    if (a->CommCount) { if (ncclResult_t r = a->CommCount(comm, &v); r != ncclSuccess) return rccl_fail("ncclCommCount", r); info5[0] = v; } else info5[0] = -1;
    if (a->CommUserRank) { if (ncclResult_t r = a->CommUserRank(comm, &v); r != ncclSuccess) return rccl_fail("ncclCommUserRank", r); info5[1] = v; } else info5[1] = -1;
    if (a->CommCuDevice) { if (ncclResult_t r = a->CommCuDevice(comm, &v); r != ncclSuccess) return rccl_fail("ncclCommCuDevice", r); info5[2] = v; }
    if (a->GetVersion && a->GetVersion(&v) == ncclSuccess) info5[3] = v;
    return 0;
}
// One all-reduce(sum) of the scalar `rank + 1` through the communicator on `stream`: every rank must read world (world + 1) / 2.
// d_scratch: >= 8 bytes of device memory.  Without a communicator nothing is reduced and *sum = rank + 1 = 1.
inline int rccl_selfcheck(ncclComm_t comm, bool has_comm, hipStream_t stream, float* d_scratch, int rank, double* sum) {
    float v = (float)(rank + 1);
    if (has_comm) {
        ARP_HIP_OK(hipMemcpyAsync(d_scratch, &v, 4, hipMemcpyHostToDevice, stream));
        if (ncclResult_t r = rccl_api()->AllReduce(d_scratch, d_scratch, 1, ncclFloat, ncclSum, comm, stream); r != ncclSuccess) return rccl_fail("ncclAllReduce(selfcheck)", r);
        ARP_HIP_OK(hipMemcpyAsync(&v, d_scratch, 4, hipMemcpyDeviceToHost, stream));
        ARP_HIP_OK(hipStreamSynchronize(stream));
    }
    *sum = (double)v;
    return 0;
}

}  // namespace arp
