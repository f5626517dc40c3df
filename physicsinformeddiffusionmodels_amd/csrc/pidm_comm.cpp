// Gradient all-reduce behind the C ABI (include/pidm.h: pidm_comm_*, pidm_allreduce_f32): RCCL over xGMI, one communicator per
// process (= per GPU).  SURVEY 8(b) / 8(e): the data-parallel exchange of the flat gradient buffer is the one collective of the
// hot path; the reference has no counterpart (main.py:157-166 is single-device).
//
// RCCL is bound at FIRST USE with dlopen (librccl.so is already mapped when torch.distributed initialised the "nccl" backend; a
// process that never calls these entries does not need it, and the library loads on boxes without it).  No torch types here: the
// host side hands over plain pointers, counts and a hipStream_t; the 128-byte unique id travels between ranks by whatever the
// caller has (parallel.py broadcasts it over the process group it already owns).
#include <dlfcn.h>
#include <mutex>
#include <string>

#include "pidm_common.h"

namespace {
// the part of the RCCL C API used here (nccl.h of ROCm 7: values are ABI)
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
typedef int ncclResult_i;
enum { kNcclFloat = 7, kNcclSum = 0, kNcclAvg = 4 };
struct Rccl {
  void* so = nullptr;
  ncclResult_i (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
  ncclResult_i (*CommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int) = nullptr;
  ncclResult_i (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  ncclResult_i (*CommDestroy)(ncclComm_p) = nullptr;
  const char* (*GetErrorString)(ncclResult_i) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_why;        // why the binding failed (dlerror() is read once, right after the failing call: reading it clears it)

const Rccl& rccl() {
  std::call_once(g_rccl_once, [] {
#ifndef PIDM_BACKEND_NAME      // (the host-emulated test build has no device and no RCCL)
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      g_rccl.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.so) break;
      if (const char* e = dlerror()) g_rccl_why = e;
    }
    if (!g_rccl.so) return;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.so, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.so, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(g_rccl.so, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.so, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.so, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
    if (!g_rccl.ok) g_rccl_why = "ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy: symbol missing";
#endif
  });
  return g_rccl;
}
int need(const Rccl& r) {
  if (r.ok) return 0;
#ifdef PIDM_BACKEND_NAME
  return pidm::fail("pidm_comm: the host-emulated build has no RCCL (gloo carries the multi-rank tests)");
#else
  return pidm::fail("pidm_comm: librccl.so could not be bound (%s)", g_rccl_why.empty() ? "unknown reason" : g_rccl_why.c_str());
#endif
}
int check(const Rccl& r, ncclResult_i rc, const char* what) {
  if (rc == 0) return 0;
  return pidm::fail("%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
}
struct Comm { ncclComm_p c; int rank, world; };
}  // namespace

// 0: RCCL is bound (dlopen + the four symbols) and the other entries can be used; -1 + message otherwise.  Starts nothing: no
// bootstrap thread, no socket (ncclGetUniqueId does - only the rank whose id is used should call pidm_comm_unique_id).
extern "C" int pidm_comm_available(void) {
  return need(rccl()) ? -1 : 0;
}

extern "C" int pidm_comm_unique_id(void* out128) {
  const Rccl& r = rccl();
  if (need(r)) return -1;
  if (!out128) return pidm::fail("pidm_comm_unique_id: null argument");
  ncclUniqueId_t id;
  if (check(r, r.GetUniqueId(&id), "ncclGetUniqueId")) return -1;
  memcpy(out128, id.internal, 128);
  return 0;
}

extern "C" int pidm_comm_init(int rank, int world, const void* unique_id128, void** comm) {
  const Rccl& r = rccl();
  if (need(r)) return -1;
  if (!unique_id128 || !comm || world < 1 || rank < 0 || rank >= world) return pidm::fail("pidm_comm_init: bad arguments (rank %d of %d)", rank, world);
  ncclUniqueId_t id;
  memcpy(id.internal, unique_id128, 128);
  Comm* c = new Comm{nullptr, rank, world};
  if (check(r, r.CommInitRank(&c->c, world, id, rank), "ncclCommInitRank")) {     // uses the calling thread's current device
    delete c;
    return -1;
  }
  *comm = c;
  return 0;
}

// in place; average != 0: the mean over ranks (ncclAvg), else the sum.  Enqueued on `stream`; RCCL orders it with the stream's work.
extern "C" int pidm_allreduce_f32(void* comm, float* buf, size_t count, int average, void* stream) {
  const Rccl& r = rccl();
  if (need(r)) return -1;
  if (!comm || (!buf && count)) return pidm::fail("pidm_allreduce_f32: null argument");
  Comm* c = static_cast<Comm*>(comm);
  return check(r, r.AllReduce(buf, buf, count, kNcclFloat, average ? kNcclAvg : kNcclSum, c->c, pidm::as_stream(stream)), "ncclAllReduce");
}

extern "C" int pidm_comm_destroy(void* comm) {
  if (!comm) return 0;
  const Rccl& r = rccl();
  Comm* c = static_cast<Comm*>(comm);
  int rc = 0;
  if (r.ok && c->c) rc = check(r, r.CommDestroy(c->c), "ncclCommDestroy");
  delete c;
  return rc;
}
