// Library-wide plumbing: error text and version.
#include "common.h"
#include <cstring>

static thread_local char g_last_error[512] = "";

void noise_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

NOISE_EXPORT const char* noise_last_error(void) { return g_last_error; }

NOISE_EXPORT int noise_version(void) { return 100; }
