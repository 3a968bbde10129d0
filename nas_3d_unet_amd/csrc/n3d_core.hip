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
// two flags, one launch: where a stream has two joins in a row (a cell's own side work and the late writer of one of its gradient
// buffers) the second wait kernel and its boundary (~3.3 us on the chain) go
__global__ void sync_wait2_kernel(const unsigned* flag0, const unsigned* flag1, unsigned* step, unsigned* timeouts, int bump, long max_polls) {
  unsigned want = *step;
  bool ok0 = false, ok1 = false;
  for (long it = 0; it < max_polls; ++it) {
    if (!ok0 && (int)(__hip_atomic_load(flag0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) ok0 = true;
    if (!ok1 && (int)(__hip_atomic_load(flag1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) ok1 = true;
    if (ok0 && ok1) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (!(ok0 && ok1)) atomicAdd(timeouts, 1u);
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
int n3d_sync_wait2(const void* flag0, const void* flag1, void* step, void* timeouts, int bump, int64_t max_polls, void* stream) {
  N3D_CHECK_ARG(flag0 && flag1 && step && timeouts, "n3d_sync_wait2: null pointer");
  N3D_CHECK_ARG(max_polls > 0, "n3d_sync_wait2: max_polls must be positive (the poll is bounded by construction)");
  hipLaunchKernelGGL(sync_wait2_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)flag0, (const unsigned*)flag1, (unsigned*)step,
                     (unsigned*)timeouts, bump, (long)max_polls);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
// ---- host-visible word, side streams and raw stream capture (include/n3d.h, "stream hand-off") --------------------------------
// These go through the HIP runtime libn3d itself is linked against -- the one that launches the kernels -- instead of a second
// copy a Python-side dlopen("libamdhip64.so") might resolve to.
int n3d_host_word_alloc(void** host_ptr) {
  N3D_CHECK_ARG(host_ptr, "n3d_host_word_alloc: null pointer");
  void* p = nullptr;
  // pinned, mapped, coherent: the device stores to it (system scope), the host reads it without any HIP call
  if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !p) {
    n3d::set_error("n3d_host_word_alloc: hipHostMalloc failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  memset(p, 0, 64);
  *host_ptr = p;
  return N3D_OK;
}
int n3d_host_word_free(void* host_ptr) {
  if (host_ptr && hipHostFree(host_ptr) != hipSuccess) { (void)hipGetLastError(); return N3D_ERR_HIP; }
  return N3D_OK;
}
int n3d_stream_create_low_priority(void** stream_out) {
  N3D_CHECK_ARG(stream_out, "n3d_stream_create_low_priority: null pointer");
  int lo = 0, hi = 0;
  hipStream_t s = nullptr;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess || !s) {
    n3d::set_error("n3d_stream_create_low_priority: HIP stream creation failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  *stream_out = (void*)s;
  return N3D_OK;
}
int n3d_stream_capture_begin(void* stream) {
  if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    n3d::set_error("n3d_stream_capture_begin: hipStreamBeginCapture failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  return N3D_OK;
}
int n3d_stream_capture_end(void* stream, void** graph_exec_out) {
  N3D_CHECK_ARG(graph_exec_out, "n3d_stream_capture_end: null pointer");
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) {
    n3d::set_error("n3d_stream_capture_end: hipStreamEndCapture failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);      // the executable graph is self-contained
  if (e != hipSuccess || !ex) {
    n3d::set_error("n3d_stream_capture_end: hipGraphInstantiate failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  *graph_exec_out = (void*)ex;
  return N3D_OK;
}
int n3d_graph_launch(void* graph_exec, void* stream) {
  N3D_CHECK_ARG(graph_exec, "n3d_graph_launch: null graph");
  if (hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) != hipSuccess) {
    n3d::set_error("n3d_graph_launch: hipGraphLaunch failed");
    (void)hipGetLastError();
    return N3D_ERR_HIP;
  }
  return N3D_OK;
}
int n3d_graph_destroy(void* graph_exec) {
  if (graph_exec && hipGraphExecDestroy((hipGraphExec_t)graph_exec) != hipSuccess) { (void)hipGetLastError(); return N3D_ERR_HIP; }
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
