/* LD_PRELOAD helper: print a native backtrace on SIGSEGV / SIGABRT (no gdb on the GPU box).
 * gcc -shared -fPIC -O1 -o /tmp/segv_trace.so tools/debug/segv_trace.c ; LD_PRELOAD=/tmp/segv_trace.so python3 ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void on_signal(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char* msg = sig == SIGSEGV ? "\n==== SIGSEGV: native backtrace ====\n" : "\n==== SIGABRT: native backtrace ====\n";
  if (write(2, msg, strlen(msg)) < 0) {}
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  signal(SIGSEGV, on_signal);
  signal(SIGABRT, on_signal);
}
