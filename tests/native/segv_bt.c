/* Test infrastructure: print the NATIVE stack of the thread that takes a SIGSEGV (Python's faulthandler shows Python frames only).
 * Enabled by E2T_TEST_SEGV_BT=1 (tests/conftest.py); built by ecog2txt_amd/csrc/build.sh. */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
#include <fcntl.h>
#include <stdlib.h>
static int g_fd = 2;     /* E2T_TEST_SEGV_BT=<path>: the report goes to that file (pytest captures fd 2 while a test runs) */
static void on_segv(int sig, siginfo_t* si, void* ctx) {
    void* frames[96];
    const char msg[] = "\n==== native backtrace of the faulting thread ====\n";
    (void)sig; (void)ctx; (void)si;
    if (write(g_fd, msg, sizeof msg - 1) < 0) _exit(139);
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, g_fd);
    _exit(139);
}
void e2t_test_install_segv_bt(void) {
    struct sigaction sa;
    const char* path = getenv("E2T_TEST_SEGV_BT");
    if (g_fd == 2 && path && path[0] == '/') { int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) g_fd = fd; }
    static char alt[1 << 16];                 /* the calling thread's handler stack: a stack overflow can still be reported */
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
