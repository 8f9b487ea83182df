/* LD_PRELOAD helper: on SIGABRT write a C backtrace to $ABORT_BT_FILE (default /tmp/abort_bt.txt).
   gcc -shared -fPIC -o /tmp/abort_bt.so tools/dbg/abort_bt.c ; run python with -p no:faulthandler */
#define _GNU_SOURCE
#include <stdio.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <dlfcn.h>
static void on_abort(int sig) {
    const char* path = getenv("ABORT_BT_FILE");
    int fd = open(path ? path : "/tmp/abort_bt.txt", O_WRONLY | O_CREAT | O_TRUNC, 0644);
    void* bt[64];
    int n = backtrace(bt, 64);
    if (fd >= 0) {
        backtrace_symbols_fd(bt, n, fd);
        /* what the process wrote to stderr / stdout lately, when those are regular files (pytest's capture files) */
        for (int src = 2; src >= 1; src--) {
            static char buf[1 << 16];
            off_t end = lseek(src, 0, SEEK_END);
            if (end > 0) {
                off_t from = end > (off_t)sizeof buf ? end - (off_t)sizeof buf : 0;
                ssize_t got = pread(src, buf, sizeof buf, from);
                dprintf(fd, "---- last %ld bytes of fd %d ----\n", (long)(got > 0 ? got : 0), src);
                if (got > 0) (void)!write(fd, buf, (size_t)got);
            }
        }
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
/* nobody else (Python's faulthandler) may replace the SIGABRT handler */
static int installed;
int sigaction(int sig, const struct sigaction* act, struct sigaction* old) {
    static int (*real)(int, const struct sigaction*, struct sigaction*);
    if (!real) real = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
    if (sig == SIGABRT && installed && act) { if (old) memset(old, 0, sizeof *old); return 0; }
    return real(sig, act, old);
}
__attribute__((constructor)) static void init(void) {
    void* bt[2];
    backtrace(bt, 2);              /* load libgcc now: not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, 0);
    installed = 1;
}
