/* LD_PRELOAD helper for tools/dp_capture_loop.py: when the process receives SIGABRT (abort(), std::terminate, a runtime
 * `guarantee`), print WHICH thread it was (its kernel name, e.g. "pt_nccl_watchdg") and its native backtrace as
 * module+offset lines (resolve with addr2line / llvm-symbolizer), to stderr and to $N3D_ABORT_TRACE_FILE, then die with the
 * default action.  Test tooling only; the product never loads it.
 *   gcc -O1 -g -shared -fPIC -o tools/bin/libabort_trace.so tools/abort_trace.c -ldl -lpthread */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static char g_path[512];

static void emit(int fd, const char* name, void** bt, int n) {
  char line[160];
  int k = snprintf(line, sizeof line, "\n=== abort_trace: SIGABRT on thread '%s' (tid %d), native backtrace:\n", name, (int)gettid());
  if (write(fd, line, k) < 0) return;
  backtrace_symbols_fd(bt, n, fd);
}

static void on_abort(int sig) {
  void* bt[96];
  char name[32] = "?";
  int n = backtrace(bt, 96);
  pthread_getname_np(pthread_self(), name, sizeof name);
  emit(2, name, bt, n);
  if (g_path[0]) {
    int fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) { emit(fd, name, bt, n); close(fd); }
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  const char* p = getenv("N3D_ABORT_TRACE_FILE");
  void* warm[4];
  if (p) strncpy(g_path, p, sizeof g_path - 1);
  backtrace(warm, 4);            /* loads libgcc now: not async-signal-safe to do so inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abort;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGABRT, &sa, NULL);
}
