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
