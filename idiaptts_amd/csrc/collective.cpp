// Gradient exchange of the data-parallel training step behind the C ABI (SURVEY.md section 8(b):
// `allreduce_flat`).  Replaces what torch.nn.DataParallel does between replicas in
// idiaptts/src/neural_networks/pytorch/ModularModelHandlerPyTorch.py:732-735 / :757-763 (scatter,
// per-replica backward, gradient reduction to device 0): one process per GPU, the flat gradient
// arena summed in place over RCCL on the caller's stream.
//
// RCCL is resolved at run time from the copy the process already holds (PyTorch ships its own
// librccl.so.1; two copies in one process would not share communicators), falling back to the
// system library.  Nothing here is needed, or loaded, on a single GPU.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "../../include/idiaptts_amd.h"

namespace itts {
void set_error(const std::string& msg);
}

namespace {

// the few RCCL declarations used, by their documented ABI (rccl.h: ncclResult_t is an int enum with
// ncclSuccess = 0; ncclFloat32 = 7, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2, ncclAvg = 4; a unique id is
// 128 opaque bytes passed by value)
struct UniqueId { char internal[128]; };
using comm_t = void*;
using fn_get_id = int (*)(UniqueId*);
using fn_init_rank = int (*)(comm_t*, int, UniqueId, int);
using fn_destroy = int (*)(comm_t);
using fn_allreduce = int (*)(const void*, void*, size_t, int, int, comm_t, void*);
using fn_errstr = const char* (*)(int);
using fn_version = int (*)(int*);

struct Rccl {
  void* handle = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_errstr errstr = nullptr;
  int version = 0;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already mapped (same soname) wins; otherwise the loader's search path
    r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) {
      const char* e = dlerror();
      r.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?");
      return;
    }
    r.get_id = reinterpret_cast<fn_get_id>(dlsym(r.handle, "ncclGetUniqueId"));
    r.init_rank = reinterpret_cast<fn_init_rank>(dlsym(r.handle, "ncclCommInitRank"));
    r.destroy = reinterpret_cast<fn_destroy>(dlsym(r.handle, "ncclCommDestroy"));
    r.allreduce = reinterpret_cast<fn_allreduce>(dlsym(r.handle, "ncclAllReduce"));
    r.errstr = reinterpret_cast<fn_errstr>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.get_id || !r.init_rank || !r.destroy || !r.allreduce) { r.why = "librccl.so.1 lacks the nccl* entry points"; return; }
    // The enum values and the by-value unique id above are the ABI of NCCL / RCCL 2.10 and later (ncclAvg = 4
    // appeared in 2.10; version codes are 10000 major + 100 minor + patch from 2.9 on): anything else is
    // refused rather than called with constants it may read differently.
    const fn_version get_version = reinterpret_cast<fn_version>(dlsym(r.handle, "ncclGetVersion"));
    if (!get_version || get_version(&r.version) != 0 || r.version < 21000 || r.version >= 30000)
      r.why = "librccl.so.1 reports version code " + std::to_string(r.version) +
              "; this binding was written against the 2.10 ... 2.x ABI (rccl.h)";
  });
  return &r;
}

int fail(const Rccl* r, const char* what, int code) {
  std::string m = std::string(what) + " failed";
  if (code && r->errstr) m += std::string(": ") + r->errstr(code);
  itts::set_error(m);
  return ITTS_E_HIP;
}

#define RCCL_OR_RETURN(r)                 \
  Rccl* r = rccl();                       \
  if (!r->why.empty()) {                  \
    itts::set_error(r->why);              \
    return ITTS_E_UNSUPPORTED;            \
  }

}  // namespace

extern "C" int itts_comm_version(void) {
  Rccl* r = rccl();
  return r->why.empty() ? r->version : 0;
}

extern "C" int itts_comm_unique_id(void* id128) {
  if (!id128) { itts::set_error("itts_comm_unique_id: null pointer"); return ITTS_E_INVALID; }
  RCCL_OR_RETURN(r);
  const int rc = r->get_id(reinterpret_cast<UniqueId*>(id128));
  return rc ? fail(r, "ncclGetUniqueId", rc) : ITTS_OK;
}

extern "C" int itts_comm_init_rank(const void* id128, int n_ranks, int rank, void** comm_out) {
  if (!id128 || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    itts::set_error("itts_comm_init_rank: bad arguments");
    return ITTS_E_INVALID;
  }
  RCCL_OR_RETURN(r);
  UniqueId id = *reinterpret_cast<const UniqueId*>(id128);
  comm_t c = nullptr;
  const int rc = r->init_rank(&c, n_ranks, id, rank);
  if (rc) return fail(r, "ncclCommInitRank", rc);
  *comm_out = c;
  return ITTS_OK;
}

extern "C" int itts_comm_destroy(void* comm) {
  if (!comm) return ITTS_OK;
  RCCL_OR_RETURN(r);
  const int rc = r->destroy(comm);
  return rc ? fail(r, "ncclCommDestroy", rc) : ITTS_OK;
}

extern "C" int itts_allreduce_flat(void* d_buf, int64_t n, int dtype, int op, void* comm, void* stream) {
  if (n < 0 || (n > 0 && !d_buf) || !comm) { itts::set_error("itts_allreduce_flat: bad arguments"); return ITTS_E_INVALID; }
  if (dtype != ITTS_F32 && dtype != ITTS_F64) { itts::set_error("itts_allreduce_flat: dtype must be ITTS_F32 or ITTS_F64"); return ITTS_E_INVALID; }
  if (op != ITTS_REDUCE_SUM && op != ITTS_REDUCE_MAX && op != ITTS_REDUCE_AVG) {
    itts::set_error("itts_allreduce_flat: unknown reduction");
    return ITTS_E_INVALID;
  }
  if (n == 0) return ITTS_OK;
  RCCL_OR_RETURN(r);
  const int nccl_dtype = dtype == ITTS_F32 ? 7 : 8;
  const int nccl_op = op == ITTS_REDUCE_SUM ? 0 : (op == ITTS_REDUCE_MAX ? 2 : 4);
  const int rc = r->allreduce(d_buf, d_buf, (size_t)n, nccl_dtype, nccl_op, comm, stream);
  return rc ? fail(r, "ncclAllReduce", rc) : ITTS_OK;
}
