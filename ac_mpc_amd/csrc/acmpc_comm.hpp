// acmpc_comm.hpp — the one exchange step of the sharded aggregation: an in-place all-reduce (sum, int64)
// of the concatenated [product_qty | customer_total_cents] vector through RCCL over xGMI.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the symbols by name), not at link time: a process
// that never shards never loads it, and a host program that already carries an RCCL of its own (PyTorch
// does) shares that copy instead of mapping a second one.  Host code only; no kernels here.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

namespace acmpc {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = "";
};

// The process-wide binding; nullptr (with the reason in *why) when RCCL cannot be loaded.
inline const RcclApi* rccl_api(const char** why) {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      snprintf(api.why, sizeof(api.why), "librccl.so.1 cannot be loaded: %s", dlerror());
      return;
    }
    bool ok = true;
    auto bind = [&](auto& fn, const char* name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
      if (!fn) { ok = false; snprintf(api.why, sizeof(api.why), "librccl has no symbol %s", name); }
    };
    bind(api.GetUniqueId, "ncclGetUniqueId");
    bind(api.CommInitRank, "ncclCommInitRank");
    bind(api.CommInitAll, "ncclCommInitAll");
    bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.AllReduce, "ncclAllReduce");
    bind(api.GroupStart, "ncclGroupStart");
    bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.GetErrorString, "ncclGetErrorString");
    if (!ok) { dlclose(api.handle); api.handle = nullptr; }
  });
  if (!api.handle) { if (why) *why = api.why; return nullptr; }
  return &api;
}

}  // namespace acmpc

// One rank of a communicator: bound to the device of the context it was created for.
struct acmpc_comm {
  ncclComm_t comm = nullptr;
  int device = -1;
  int rank = 0;
  int world = 1;
};
