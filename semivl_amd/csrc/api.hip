// Error plumbing + version for libsemivl_hip.so.
#include "svl_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = {0};

void svl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int svl_version(void) { return 100; }

extern "C" int svl_last_error(char* buf, size_t len) {
  const size_t n = strlen(g_err);
  if (buf && len > 0) {
    const size_t c = n < len - 1 ? n : len - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)n;
}

// ---- helper stream (svl_common.h) -------------------------------------------------------------------------------
static hipStream_t g_aux = nullptr;
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;

int svl_fork(hipStream_t st, hipStream_t* aux) {
  if (!g_aux) {
    SVL_HIP_CHECK(hipStreamCreateWithFlags(&g_aux, hipStreamNonBlocking));
    SVL_HIP_CHECK(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
    SVL_HIP_CHECK(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
  }
  SVL_HIP_CHECK(hipEventRecord(g_ev_fork, st));
  SVL_HIP_CHECK(hipStreamWaitEvent(g_aux, g_ev_fork, 0));
  *aux = g_aux;
  return SVL_OK;
}

int svl_join(hipStream_t st) {
  SVL_HIP_CHECK(hipEventRecord(g_ev_join, g_aux));
  SVL_HIP_CHECK(hipStreamWaitEvent(st, g_ev_join, 0));
  return SVL_OK;
}
