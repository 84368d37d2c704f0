/* tests/native/native_bt.c -- TEST INFRASTRUCTURE: prints a native (C-level) backtrace when the process receives
 * SIGABRT / SIGSEGV / SIGBUS / SIGFPE, then hands over to whichever handler was installed before (pytest's faulthandler prints the
 * Python stack).  Round 1's driver-side `pytest -m gpu` died with SIGABRT inside a C-ABI call and left no native
 * evidence; with this loaded (tests/conftest.py) the log names the aborting frame.  Not part of the product. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[65];
static int g_fd = 2;

static void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }

static void on_fatal(int sig, siginfo_t* info, void* ctx) {
  void* frames[96];
  (void)info;
  (void)ctx;
  put("\n=== native backtrace (tests/native/native_bt.c), signal ");
  char num[4] = {(char)('0' + sig / 10), (char)('0' + sig % 10), '\n', 0};
  put(num);
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, g_fd);
  put("=== end native backtrace\n");
  sigaction(sig, &g_prev[sig], NULL); /* chain: faulthandler (Python stack), then the default action */
  raise(sig);
}

int native_bt_install(int fd) {
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc's unwinder now, not inside the handler */
  if (fd >= 0) g_fd = fd;
  const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE};
  for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) return -1;
  }
  return 0;
}
