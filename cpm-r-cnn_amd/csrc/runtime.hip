// Library-wide state: last error string, ABI version.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace cpm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cpm

CPM_EXPORT int cpm_abi_version(void) { return 1; }
CPM_EXPORT const char* cpm_last_error(void) { return cpm::g_err; }
