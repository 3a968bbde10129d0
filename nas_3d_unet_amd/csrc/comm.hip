// RCCL exchange step of data-parallel training behind the C ABI (the reference has no collective:
// config.yml:54 `multi_gpus` is never read).  One process per GPU; the only exchange of the hot path is a SUM all-reduce of
// the flat fp32 gradient buffer, in place, on the caller's stream (so it can run on a side HIP stream under the backward
// kernels of the next bucket).  RCCL is bound at run time with dlopen -- the copy torch already loaded wins (same soname),
// and a CPU-only host can still load libn3d.so and see every symbol of include/n3d.h.
#include <dlfcn.h>

#include "n3d_common.h"

namespace n3d {

// the six entry points used, with RCCL's (= NCCL's) C signatures; enums passed as ints (ncclFloat32 = 7, ncclSum = 0)
struct Id128 { char b[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

static Rccl load_rccl() {
  Rccl r;
  {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // torch's copy, if torch.distributed brought it in
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      r.h = h;
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
      r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
      r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Broadcast || !r.CommDestroy) r.h = nullptr;
    }
  }
  return r;
}

static Rccl* rccl() {
  static Rccl r = load_rccl();   // function-local static: initialised once, also when two threads make the first call
  return r.h ? &r : nullptr;
}

static int rccl_fail(const char* what, int code) {
  Rccl* r = rccl();
  set_error("%s: RCCL error %d (%s)", what, code, (r && r->GetErrorString) ? r->GetErrorString(code) : "?");
  return N3D_ERR_HIP;
}

}  // namespace n3d

using namespace n3d;

extern "C" {

int n3d_comm_available(void) { return rccl() ? 1 : 0; }

int n3d_comm_unique_id(void* id_out) {
  N3D_CHECK_ARG(id_out, "comm_unique_id: null output");
  Rccl* r = rccl();
  if (!r) {
    const char* why = dlerror();    // one call: dlerror() clears the message it returns
    N3D_UNSUPPORTED("comm_unique_id: librccl.so.1 could not be loaded (%s)", why ? why : "?");
  }
  if (int e = r->GetUniqueId(id_out)) return rccl_fail("ncclGetUniqueId", e);
  return N3D_OK;
}

int n3d_comm_init(const void* id, int world, int rank, void** comm_out) {
  N3D_CHECK_ARG(id && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
  Rccl* r = rccl();
  if (!r) N3D_UNSUPPORTED("comm_init: librccl.so.1 could not be loaded");
  Id128 u;
  memcpy(u.b, id, sizeof(u.b));
  void* c = nullptr;
  if (int e = r->CommInitRank(&c, world, u, rank)) return rccl_fail("ncclCommInitRank", e);
  *comm_out = c;
  return N3D_OK;
}

int n3d_comm_allreduce_sum(void* comm, float* buf, int64_t n, void* stream) {
  N3D_CHECK_ARG(comm && buf && n > 0, "comm_allreduce_sum: bad args");
  Rccl* r = rccl();
  if (!r) N3D_UNSUPPORTED("comm_allreduce_sum: RCCL not loaded");
  if (int e = r->AllReduce(buf, buf, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, (hipStream_t)stream)) return rccl_fail("ncclAllReduce", e);
  return N3D_OK;
}

int n3d_comm_broadcast(void* comm, float* buf, int64_t n, int root, void* stream) {
  N3D_CHECK_ARG(comm && buf && n > 0 && root >= 0, "comm_broadcast: bad args");
  Rccl* r = rccl();
  if (!r) N3D_UNSUPPORTED("comm_broadcast: RCCL not loaded");
  if (int e = r->Broadcast(buf, buf, (size_t)n, /* ncclFloat32 */ 7, root, comm, (hipStream_t)stream)) return rccl_fail("ncclBroadcast", e);
  return N3D_OK;
}

int n3d_comm_destroy(void* comm) {
  if (!comm) return N3D_OK;
  Rccl* r = rccl();
  if (!r) return N3D_OK;
  if (int e = r->CommDestroy(comm)) return rccl_fail("ncclCommDestroy", e);
  return N3D_OK;
}

}  // extern "C"
