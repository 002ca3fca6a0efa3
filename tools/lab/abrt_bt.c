/* LD_PRELOAD shim (debugging aid, tools only): native backtrace on SIGABRT / SIGSEGV -- glibc's "free(): invalid pointer" aborts
 * from inside free(), so the frames above abort() name the caller that handed it the pointer.  Built and used by
 * tools/lab/heap_hunt.sh. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    void* bt[96];
    int n = backtrace(bt, 96);
    dprintf(2, "=== signal %d (%s) at address %p, pid %d, native backtrace (%d frames) ===\n", sig, sig == SIGABRT ? "SIGABRT" : "SIGSEGV", si->si_addr, (int)getpid(), n);
    backtrace_symbols_fd(bt, n, 2);
    {   /* the mappings, so that frames without symbols can be placed */
        char line[512];
        FILE* f = fopen("/proc/self/maps", "r");
        if (f) {
            while (fgets(line, sizeof line, f))
                if (strstr(line, "r-xp") || strstr(line, "r-x")) dprintf(2, "MAP %s", line);
            fclose(f);
        }
    }
    _exit(128 + sig);
}
__attribute__((constructor)) static void init(void) {
    struct sigaction sa;
    {   /* backtrace()'s first call loads libgcc_s (dlopen: malloc) -- done HERE, not inside a handler that runs while free() holds the
         * allocator's lock (round 6's first hunt reproduced the abort and then hung exactly there) */
        void* warm[4];
        (void)backtrace(warm, 4);
    }
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_sig;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGABRT, &sa, NULL);
}
