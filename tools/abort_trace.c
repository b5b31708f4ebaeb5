/* Debug aid: LD_PRELOAD=tools/libabort_trace.so prints the native call stack when the process
 * receives SIGABRT (python's faulthandler only shows the interpreter's frames).
 *   gcc -shared -fPIC -o tools/libabort_trace.so tools/abort_trace.c */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n==== native stack at SIGABRT ====\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}
