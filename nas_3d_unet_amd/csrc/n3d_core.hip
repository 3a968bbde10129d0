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

extern "C" {
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
