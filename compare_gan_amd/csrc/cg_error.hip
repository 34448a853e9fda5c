// Thread-local error string + ABI version.
#include <stdarg.h>
#include "cg_common.h"

static thread_local char g_err[512] = {0};

void cg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int cg_abi_version(void) { return 1; }
extern "C" const char* cg_last_error(void) { return g_err; }
