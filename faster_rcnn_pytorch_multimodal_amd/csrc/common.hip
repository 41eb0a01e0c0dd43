#include "common.h"

namespace frcnn {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace frcnn

extern "C" int frcnn_version(void) { return 107; }
extern "C" const char* frcnn_last_error(void) { return frcnn::error_buffer(); }
