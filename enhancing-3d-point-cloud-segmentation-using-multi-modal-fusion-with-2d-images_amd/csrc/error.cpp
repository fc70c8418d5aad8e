// Error string + ABI version of libmvkpconv.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mvkpconv.h"

static thread_local char g_err[512] = "";

extern "C" void mvk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mvk_last_error(void) { return g_err; }
extern "C" int mvk_abi_version(void) { return MVK_ABI_VERSION; }
