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
    api.ok = api.GetErrorString && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Broadcast;
    return api.ok ? &api : nullptr;
}

inline int rccl_fail(const char* what, ncclResult_t r) { return fail(std::string(what) + ": " + rccl_api()->GetErrorString(r)); }

}  // namespace arp
