// Error plumbing + ABI version of the MI355X kernel library.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace sglm {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("HIP error in %s: %s", what, hipGetErrorString(e));
  return SGL_MI355_ERR_RUNTIME;
}

}  // namespace sglm

extern "C" int sgl_mi355_abi_version(void) { return SGL_MI355_ABI_VERSION; }

extern "C" size_t sgl_mi355_last_error(char* buf, size_t buf_size) {
  size_t n = strlen(sglm::g_err);
  if (buf && buf_size > 0) {
    size_t c = n < buf_size - 1 ? n : buf_size - 1;
    memcpy(buf, sglm::g_err, c);
    buf[c] = 0;
  }
  return n;
}
