/* LD_PRELOAD shim (debugging aid, tools only): native backtrace on SIGSEGV. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t* si, void* uc) {
    void* bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "=== SIGSEGV at address %p, native backtrace (%d frames) ===\n", si->si_addr, n);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
__attribute__((constructor)) static void init(void) {
    struct sigaction sa;
    sa.sa_sigaction = on_segv;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, NULL);
}
