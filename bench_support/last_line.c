/* Bench-side helper (NOT part of the product's C-ABI: nothing under genrl_amd/ or include/ knows about it).
 *
 * bench.py, under data parallelism over RCCL, first measures the 'cut' mode and then TRIES the collectives inside the hipGraph.  That
 * attempt first meets real peers in the driver's multi-GPU job; if a backend thread aborts the process there (SIGABRT from the process
 * group's watchdog, SIGSEGV), the line already measured would be lost.  bench_set_last_line() deposits that line; the handler writes it to
 * stdout with write(2) and ends the process.  Async-signal-safe (no allocation, no stdio), runs on its own stack (a stack overflow still
 * reaches it), and bench_clear_last_line() puts the PREVIOUS dispositions back (Python's faulthandler, torch's handlers).
 * Exit status: `status_with_line` when a line was deposited (bench.py passes 0 and says why inside the line's config.launch), 3 otherwise. */
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_line[1 << 16];
static volatile int g_len = 0;
static volatile int g_status = 0;
static struct sigaction g_prev[2];
static int g_installed = 0;
static char g_stack[1 << 16];
static stack_t g_prev_stack;
static const int g_sigs[2] = {SIGABRT, SIGSEGV};

static void on_signal(int sig) {
  (void)sig;
  if (g_len > 0) {
    ssize_t r = write(1, g_line, (size_t)g_len);
    (void)r;
    _exit(g_status);
  }
  _exit(3);
}

int bench_set_last_line(const char* line, int status_with_line) {
  int n = line ? (int)strlen(line) : 0;
  if (n > (int)sizeof(g_line) - 2) return 1;
  if (n > 0) { memcpy(g_line, line, (size_t)n); g_line[n++] = '\n'; }
  g_len = n;
  g_status = status_with_line;
  if (!g_installed) {
    stack_t st;
    st.ss_sp = g_stack; st.ss_size = sizeof(g_stack); st.ss_flags = 0;
    if (sigaltstack(&st, &g_prev_stack) != 0) return 2;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_signal;
    sa.sa_flags = SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    for (int i = 0; i < 2; ++i)
      if (sigaction(g_sigs[i], &sa, &g_prev[i]) != 0) return 2;
    g_installed = 1;
  }
  return 0;
}

int bench_clear_last_line(void) {
  if (g_installed) {
    for (int i = 0; i < 2; ++i) sigaction(g_sigs[i], &g_prev[i], 0);
    sigaltstack(&g_prev_stack, 0);
    g_installed = 0;
  }
  g_len = 0;
  return 0;
}
