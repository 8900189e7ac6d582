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

// ---- device-resident step state (hipGraph replay of the training step)
static const unsigned long long* g_step_state = nullptr;
const unsigned long long* ctclip_step_state(void) { return g_step_state; }

__global__ void step_state_advance_kernel(unsigned long long* st) {
  st[0] += 0x9E3779B97F4A7C15ull;      // a new dropout seed offset (odd constant: 2^64 distinct offsets)
  st[1] += 1ull;                       // the optimiser step
}

// state: two 64-bit words in DEVICE memory { seed offset, optimiser step } or NULL (off, the default).  While set, every ctclip_dropout /
// ctclip_relu_dropout / ctclip_attn_fwd / ctclip_attn_bwd launch adds state[0] to the seed it was given and ctclip_adam_step takes its step count
// from state[1] (the `step` argument is ignored) -- read by the kernels when they RUN, so a captured hipGraph of the step replays with fresh
// dropout masks and the right Adam bias correction.  Process-wide; the caller keeps the memory alive.
extern "C" int ctclip_set_step_state(const void* state) { g_step_state = (const unsigned long long*)state; return CTCLIP_OK; }
// state[0] += an odd 64-bit constant, state[1] += 1 (one launch; put it first in the captured step).
extern "C" int ctclip_advance_step_state(void* state, hipStream_t s) {
  if (!state) { ctclip_set_error("advance_step_state: null state"); return CTCLIP_EBADARG; }
  hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)state);
  return ctclip_check_launch("advance_step_state");
}
