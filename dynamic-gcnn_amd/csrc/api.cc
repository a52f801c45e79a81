// api.cc -- error plumbing shared by every entry point of libdgcnn_hip.so.
#include "common.h"
#include <string.h>

namespace dg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dg

extern "C" int dgcnn_version(void) { return 100; }
extern "C" const char* dgcnn_last_error(void) { return dg::g_err; }
