// LD_PRELOAD helper: native backtrace on SIGABRT / SIGSEGV (development tool: where does a crash inside the GPU suite come from?)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void on_sig(int s) {
  void* bt[64];
  int n = backtrace(bt, 64);
  dprintf(2, "\n=== native backtrace (signal %d) ===\n", s);
  backtrace_symbols_fd(bt, n, 2);
  signal(s, SIG_DFL);
  raise(s);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_sig); signal(SIGSEGV, on_sig); }
