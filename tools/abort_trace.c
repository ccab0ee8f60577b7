/* LD_PRELOAD helper for hunting a silent abort(): prints the C backtrace of the aborting thread to stderr.
 *   gcc -shared -fPIC -O1 -o tools/bin/libaborttrace.so tools/abort_trace.c
 *   LD_PRELOAD=tools/bin/libaborttrace.so python ... */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n==== SIGABRT: C backtrace of the aborting thread ====\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

/* abort() itself, so that nothing installed later (Python's faulthandler) hides the first trace */
void abort(void)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n==== abort() called: C backtrace ====\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}

__attribute__((constructor)) static void install(void) { signal(SIGABRT, on_abort); }
