// C-ABI plumbing: thread-local error string, launch-error translation, ABI version.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void ctclip_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* ctclip_last_error(void) { return g_err; }
extern "C" int ctclip_abi_version(void) { return 1; }
extern "C" const char* ctclip_target_arch(void) { return "gfx950"; }

int ctclip_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return CTCLIP_OK;
  char buf[400];
  snprintf(buf, sizeof(buf), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
  ctclip_set_error(buf);
  return -(1000 + (int)e);
}
