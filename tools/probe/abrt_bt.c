/* LD_PRELOAD helper for debugging on the GPU box: prints the native call stack when the process aborts. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abrt(int sig) {
  void* fr[64];
  int n = backtrace(fr, 64);
  backtrace_symbols_fd(fr, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) {
  signal(SIGABRT, on_abrt);
  signal(SIGSEGV, on_abrt);
}
