// Error plumbing + version entry points of the C ABI (include/pidm.h).
#include "pidm_common.h"

#include <cstdarg>
#include <cstdio>

namespace pidm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
}  // namespace pidm

extern "C" int pidm_version(void) { return PIDM_ABI_VERSION; }
extern "C" const char* pidm_last_error(void) { return pidm::g_err; }
extern "C" const char* pidm_backend(void) {
#ifdef PIDM_BACKEND_NAME
  return PIDM_BACKEND_NAME;
#else
  return "hip";
#endif
}
