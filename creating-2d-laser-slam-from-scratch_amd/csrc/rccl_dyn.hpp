// RCCL called directly from the C++ host (the offline map build is the one path with a real exchange
// step).  librccl is dlopen'ed on first use so that liblslam_gpu.so itself loads without it (the match / update paths
// have no collective); types and enums come from <rccl/rccl.h>.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace lslam {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;

  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;  // GLOBAL: a torch in the same process shares this copy
      if (!r.lib) {
        r.error = std::string("cannot load librccl: ") + dlerror();
        return;
      }
      auto sym = [&](const char* s) {
        void* p = dlsym(r.lib, s);
        if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s;
        return p;
      };
      r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
      r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
      r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
      r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
      r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
      r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
      r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
      r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
  }
  bool ok() const { return lib && error.empty(); }
};

}  // namespace lslam
