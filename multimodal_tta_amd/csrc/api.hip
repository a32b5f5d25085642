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
int g_igemm_pipeline = 1;
int g_epilogue_vec = 1;
int g_igemm_lean = 1;
int g_cls_fused_min = 128;
int g_thin_mfma = 2;
int g_wgrad_vec = 1;
// split-K below / target, weight-gradient workgroups, thin-layer slabs.  Swept with the lanes bound to their own hardware
// queues (profiles/r02_tuning_sweep.txt): four volumes in flight want half the splitting two did (96/128, 128 slabs)
int g_tune[4] = {96, 128, 128, 256};
}  // namespace mmtta

extern "C" int mmtta_set_option(int key, int value) {
  if (key == MMTTA_OPT_PROFILE_MAIN_KERNEL_ONLY) {
    const int prev = mmtta::g_profile_main_only;
    mmtta::g_profile_main_only = value;
    return prev;
  }
  if (key == MMTTA_OPT_WGRAD_VECTOR_STAGING) {
    const int prev = mmtta::g_wgrad_vec;
    mmtta::g_wgrad_vec = value ? 1 : 0;
    return prev;
  }
  if (key == MMTTA_OPT_THIN_MFMA) {
    const int prev = mmtta::g_thin_mfma;
    mmtta::g_thin_mfma = value < 0 ? 0 : (value > 3 ? 3 : value);
    return prev;
  }
  if (key == MMTTA_OPT_CLASS_FUSED_MIN_WORKGROUPS) {
    const int prev = mmtta::g_cls_fused_min;
    mmtta::g_cls_fused_min = value < 0 ? 0 : value;
    return prev;
  }
  if (key == MMTTA_OPT_IGEMM_LEAN) {
    const int prev = mmtta::g_igemm_lean;
    mmtta::g_igemm_lean = value ? 1 : 0;
    return prev;
  }
  if (key == MMTTA_OPT_EPILOGUE_VEC16) {
    const int prev = mmtta::g_epilogue_vec;
    mmtta::g_epilogue_vec = value ? 1 : 0;
    return prev;
  }
  if (key == MMTTA_OPT_IGEMM_PIPELINE) {
    const int prev = mmtta::g_igemm_pipeline;
    mmtta::g_igemm_pipeline = value ? 1 : 0;
    return prev;
  }
  if (key >= MMTTA_OPT_SPLITK_BELOW && key <= MMTTA_OPT_WGRAD_THIN_SLABS) {
    if (value < 1) return MMTTA_ERR_INVALID;
    const int prev = mmtta::g_tune[key - MMTTA_OPT_SPLITK_BELOW];
    mmtta::g_tune[key - MMTTA_OPT_SPLITK_BELOW] = value;
    return prev;
  }
  return MMTTA_ERR_INVALID;
}

extern "C" const char* mmtta_last_error(void) { return mmtta::g_err; }
extern "C" int mmtta_abi_version(void) { return MMTTA_ABI_VERSION; }
