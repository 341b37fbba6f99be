// comm.hip -- thin RCCL entry points of the C ABI (one process per GPU, the flat fp32
// gradient bucket summed by ONE all-reduce over xGMI; all-gather / reduce-scatter for the batch-coupled
// regularisers: beta-TC total correlation over the global batch, FactorVAE's global permute_dims).
// RCCL is bound lazily (dlopen of librccl, the library PyTorch-ROCm itself loads): libodin_hip.so has no
// link-time dependency on it, single-GPU use never touches it, and a missing library is a loud error
// from odin_comm_init -- there is no other transport behind these calls.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdint>
#include <cstdlib>
#ifndef ODIN_SIM
#include <dlfcn.h>
#endif

namespace {

struct OdinNcclId { char internal[128]; };  // ncclUniqueId (rccl.h)
typedef void* OdinNcclComm;                 // ncclComm_t
enum { ODIN_NCCL_FLOAT = 7, ODIN_NCCL_SUM = 0 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(OdinNcclId*) = nullptr;
  int (*CommInitRank)(OdinNcclComm*, int, OdinNcclId, int) = nullptr;
  int (*CommDestroy)(OdinNcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, OdinNcclComm, void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, OdinNcclComm, void*) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, OdinNcclComm, void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char origin[256] = "";  // which library was bound (odin_comm_library)
};
Rccl g_rccl;

int rccl_bind() {
#ifdef ODIN_SIM
  return odin_fail(-3, "RCCL is unavailable in the simulator build");
#else
  if (g_rccl.h != nullptr) return 0;
  // 1. an explicit ODIN_RCCL_LIB; 2. the RCCL the process has ALREADY mapped (PyTorch bundles its own librccl and
  // loads it with torch.distributed: RTLD_NOLOAD returns that handle instead of mapping a second copy of the library
  // -- two RCCLs in one process each run their own bootstrap and proxy threads); 3. the system library
  const char* explicit_lib = getenv("ODIN_RCCL_LIB");
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  if (explicit_lib != nullptr && *explicit_lib != 0) {
    h = dlopen(explicit_lib, RTLD_NOW | RTLD_LOCAL);
    if (h != nullptr) snprintf(g_rccl.origin, sizeof(g_rccl.origin), "ODIN_RCCL_LIB=%s", explicit_lib);
  }
  for (int pass = 0; h == nullptr && pass < 2; ++pass) {
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h != nullptr) {
        snprintf(g_rccl.origin, sizeof(g_rccl.origin), "%s (%s)", n, pass == 0 ? "already loaded by the process" : "dlopen");
        break;
      }
    }
  }
  if (h == nullptr) return odin_fail(-4, "cannot load librccl (set ODIN_RCCL_LIB to its path)");
#define ODIN_SYM(field, name)                                      \
  *(void**)(&g_rccl.field) = dlsym(h, name);                       \
  if (g_rccl.field == nullptr) return odin_fail(-4, "librccl lacks " name)
  ODIN_SYM(GetUniqueId, "ncclGetUniqueId");
  ODIN_SYM(CommInitRank, "ncclCommInitRank");
  ODIN_SYM(CommDestroy, "ncclCommDestroy");
  ODIN_SYM(AllReduce, "ncclAllReduce");
  ODIN_SYM(AllGather, "ncclAllGather");
  ODIN_SYM(ReduceScatter, "ncclReduceScatter");
  ODIN_SYM(GetErrorString, "ncclGetErrorString");
#undef ODIN_SYM
  g_rccl.h = h;
  return 0;
#endif
}

int rccl_rc(int rc, const char* what) {
  if (rc == 0) return 0;
  static thread_local char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  return odin_fail(-200 - rc, msg);
}

}  // namespace

extern "C" int odin_comm_unique_id(void* id128) {
  if (int rc = rccl_bind()) return rc;
  return rccl_rc(g_rccl.GetUniqueId((OdinNcclId*)id128), "ncclGetUniqueId");
}

extern "C" int odin_comm_init(void** comm_out, const void* id128, int rank, int world_size) {
  if (int rc = rccl_bind()) return rc;
  OdinNcclId id;
  memcpy(&id, id128, sizeof(id));
  OdinNcclComm c = nullptr;
  if (int rc = rccl_rc(g_rccl.CommInitRank(&c, world_size, id, rank), "ncclCommInitRank")) return rc;
  *comm_out = c;
  return 0;
}

// which RCCL the communicators of this process run on ("" before the first odin_comm_* call)
extern "C" const char* odin_comm_library(void) { return g_rccl.origin; }

extern "C" int odin_comm_destroy(void* comm) {
  if (comm == nullptr || g_rccl.h == nullptr) return 0;
  return rccl_rc(g_rccl.CommDestroy((OdinNcclComm)comm), "ncclCommDestroy");
}

extern "C" int odin_allreduce_flat(void* comm, float* buf, size_t n, void* stream) {
  if (g_rccl.h == nullptr) return odin_fail(-4, "odin_allreduce_flat: no communicator (odin_comm_init first)");
  return rccl_rc(g_rccl.AllReduce(buf, buf, n, ODIN_NCCL_FLOAT, ODIN_NCCL_SUM, (OdinNcclComm)comm, stream),
                 "ncclAllReduce");
}

extern "C" int odin_allgather_flat(void* comm, const float* send, float* recv, size_t n_per_rank, void* stream) {
  if (g_rccl.h == nullptr) return odin_fail(-4, "odin_allgather_flat: no communicator (odin_comm_init first)");
  return rccl_rc(g_rccl.AllGather(send, recv, n_per_rank, ODIN_NCCL_FLOAT, (OdinNcclComm)comm, stream),
                 "ncclAllGather");
}

extern "C" int odin_reduce_scatter_flat(void* comm, const float* send, float* recv, size_t n_per_rank,
                                        void* stream) {
  if (g_rccl.h == nullptr) return odin_fail(-4, "odin_reduce_scatter_flat: no communicator (odin_comm_init first)");
  return rccl_rc(g_rccl.ReduceScatter(send, recv, n_per_rank, ODIN_NCCL_FLOAT, ODIN_NCCL_SUM, (OdinNcclComm)comm,
                                      stream),
                 "ncclReduceScatter");
}
