/* Diagnostic only (tools/stress/team_stress.sh): a SIGSEGV / SIGBUS handler that prints the fault address, the
 * mappings around it and the native backtrace of the faulting thread, then hands over to the handler that was
 * installed before it (Python's faulthandler: the Python stacks of every thread).  Loaded with ctypes. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static struct sigaction prev_segv, prev_bus;

static void put(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void handler(int sig, siginfo_t* si, void* uc) {
  char line[512];
  uintptr_t a = (uintptr_t)si->si_addr;
  snprintf(line, sizeof line, "\n[segv_bt] signal %d code %d fault address %p\n", sig, si->si_code, si->si_addr);
  put(line);
  FILE* f = fopen("/proc/self/maps", "r");
  if (f) {
    char prev[512] = "";
    while (fgets(line, sizeof line, f)) {
      unsigned long lo = 0, hi = 0;
      if (sscanf(line, "%lx-%lx", &lo, &hi) == 2) {
        if (a >= lo && a < hi) { put("[segv_bt] inside: "); put(line); }
        else if (a < lo && prev[0]) { put("[segv_bt] below : "); put(prev); put("[segv_bt] above : "); put(line); prev[0] = 0; break; }
      }
      if (a >= hi) { strncpy(prev, line, sizeof prev - 1); prev[sizeof prev - 1] = 0; }
    }
    fclose(f);
  }
  void* bt[64];
  int n = backtrace(bt, 64);
  put("[segv_bt] native backtrace:\n");
  backtrace_symbols_fd(bt, n, 2);
  struct sigaction* p = sig == SIGBUS ? &prev_bus : &prev_segv;
  if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) { p->sa_sigaction(sig, si, uc); return; }
  if (p->sa_handler && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) { p->sa_handler(sig); return; }
  signal(sig, SIG_DFL);
  raise(sig);
}

int segv_bt_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  if (sigaction(SIGSEGV, &sa, &prev_segv)) return -1;
  if (sigaction(SIGBUS, &sa, &prev_bus)) return -1;
  return 0;
}
