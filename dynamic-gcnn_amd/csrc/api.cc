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
// per calling thread: a host that drives several streams from several threads gives each its own count (the count is read on
// the host when a launch is prepared and travels to the kernel as an argument; nothing on the device is shared)
static thread_local int g_stat_slots = DGCNN_STAT_SLOTS;
int stat_slots() { return g_stat_slots; }
}  // namespace dg

extern "C" int dgcnn_set_stat_slots(int n) {
  DG_REQUIRE(n >= DGCNN_STAT_SLOTS && n <= (1 << 16), DGCNN_EINVAL, "dgcnn_set_stat_slots: n must be in [%d, 65536] (got %d)", DGCNN_STAT_SLOTS, n);
  dg::g_stat_slots = n;
  return DGCNN_OK;
}
extern "C" int dgcnn_get_stat_slots(void) { return dg::g_stat_slots; }

extern "C" int dgcnn_version(void) { return 100; }
extern "C" const char* dgcnn_last_error(void) { return dg::g_err; }
