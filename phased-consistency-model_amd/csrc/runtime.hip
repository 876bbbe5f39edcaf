// Error reporting for the C ABI (thread-local message, no exceptions across the boundary).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pcm_common.h"

static thread_local char g_err[512] = "";

void pcm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pcm_post_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pcm_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return PCM_EHIP;
  }
  return PCM_OK;
}

extern "C" const char* pcm_last_error(void) { return g_err; }
extern "C" int pcm_abi_version(void) { return 1; }
