// LD_PRELOAD shim for hunting a silent abort(): prints the C backtrace of whoever calls abort() or raises SIGABRT / SIGSEGV, then dies as before.
//   gcc -shared -fPIC -O1 -o abort_bt.so abort_bt.c ; LD_PRELOAD=./abort_bt.so python -m pytest -p no:faulthandler ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static int out_fd = 2;      // the stderr of process start (pytest swaps fd 2 for a capture file while a test runs)
static void dump(const char* why) {
  void* bt[96];
  const int n = backtrace(bt, 96);
  if (write(out_fd, "\n=== ", 5) < 0 || write(out_fd, why, strlen(why)) < 0 || write(out_fd, " ===\n", 5) < 0) return;
  backtrace_symbols_fd(bt, n, out_fd);
}
static void on_signal(int sig) {
  dump(sig == SIGABRT ? "SIGABRT handler" : "SIGSEGV handler");
  signal(sig, SIG_DFL);
  raise(sig);
}
void abort(void) {
  dump("abort() called");
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
  _exit(134);
}
__attribute__((constructor)) static void init(void) {
  const int d = dup(2);
  if (d >= 0) out_fd = d;
  struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = on_signal;
  sigaction(SIGABRT, &sa, 0); sigaction(SIGSEGV, &sa, 0);
}
