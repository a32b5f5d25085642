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
int g_profile_main_only = 0;
}  // namespace mmtta

extern "C" int mmtta_set_option(int key, int value) {
  if (key != MMTTA_OPT_PROFILE_MAIN_KERNEL_ONLY) return MMTTA_ERR_INVALID;
  const int prev = mmtta::g_profile_main_only;
  mmtta::g_profile_main_only = value;
  return prev;
}

extern "C" const char* mmtta_last_error(void) { return mmtta::g_err; }
extern "C" int mmtta_abi_version(void) { return MMTTA_ABI_VERSION; }
