// Error text and version of libmmtta.so.
#include "common.h"
#include <string.h>

namespace mmtta {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mmtta

extern "C" const char* mmtta_last_error(void) { return mmtta::g_err; }
extern "C" int mmtta_abi_version(void) { return MMTTA_ABI_VERSION; }
