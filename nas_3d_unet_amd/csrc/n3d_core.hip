// Error reporting, version and device probe for libn3d.
#include <stdarg.h>

#include "n3d_common.h"

namespace n3d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace n3d

// ---- device-side stream hand-off (include/n3d.h, "stream hand-off") --------------------------------------------------------
// One lane publishes / polls a 32-bit step number in device memory.  The producing kernels of the hand-off are ordinary
// predecessors of the signal kernel on its stream (a kernel boundary releases their stores at agent scope); the consumer
// kernels are ordinary successors of the wait kernel on the other stream (a kernel boundary acquires).  The poll is a
// relaxed agent-scope load with s_sleep between tries and is BOUNDED: a wait that gives up adds 1 to *timeouts and lets the
// stream go on (wrong results, never a hung GPU); callers check the counter.
namespace {
__global__ void sync_signal_kernel(unsigned* flag, unsigned* step, int bump) {
  unsigned s = *step;
  __hip_atomic_store(flag, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (bump) *step = s + 1;
}
__global__ void sync_wait_kernel(const unsigned* flag, unsigned* step, unsigned* timeouts, int bump, long max_polls) {
  unsigned want = *step;
  bool ok = false;
  for (long it = 0; it < max_polls; ++it) {
    // steps are compared modulo 2^32 (the counters wrap after 4e9 steps)
    if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) { ok = true; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (!ok) atomicAdd(timeouts, 1u);
  if (bump) *step = want + 1;
}
__global__ void stamp_kernel(unsigned long long* out) { *out = __builtin_amdgcn_s_memrealtime(); }
}  // namespace

extern "C" {
int n3d_stamp(void* out, void* stream) {
  N3D_CHECK_ARG(out && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "n3d_stamp: needs an 8-byte aligned device word");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
int n3d_sync_signal(void* flag, void* step, int bump, void* stream) {
  N3D_CHECK_ARG(flag && step, "n3d_sync_signal: null pointer");
  hipLaunchKernelGGL(sync_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)flag, (unsigned*)step, bump);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
int n3d_sync_wait(const void* flag, void* step, void* timeouts, int bump, int64_t max_polls, void* stream) {
  N3D_CHECK_ARG(flag && step && timeouts, "n3d_sync_wait: null pointer");
  N3D_CHECK_ARG(max_polls > 0, "n3d_sync_wait: max_polls must be positive (the poll is bounded by construction)");
  hipLaunchKernelGGL(sync_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)flag, (unsigned*)step,
                     (unsigned*)timeouts, bump, (long)max_polls);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
const char* n3d_last_error(void) { return n3d::g_err; }
int n3d_version(void) { return 1; }
int n3d_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { n3d::set_error("no HIP device visible"); return 0; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { n3d::set_error("hipGetDeviceProperties failed"); return 0; }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) { n3d::set_error("libn3d is built for gfx950, found %s", p.gcnArchName); return 0; }
  return 1;
}
}
