// api.hip — library-level entry points of the C ABI (include/factorizer_hip.h).
#include "fz_common.h"

namespace fz {
std::string& last_error() {
  static thread_local std::string s;
  return s;
}
std::atomic<int64_t>& launch_counter() {
  static std::atomic<int64_t> c{0};
  return c;
}
int& tile_order_ref() {
  static thread_local int v = 0;
  return v;
}
}  // namespace fz

extern "C" int fz_version(void) { return 500; }  // 0.5.0
extern "C" int fz_abi_version(void) { return FZ_ABI_VERSION; }
extern "C" int fz_set_tile_order(int descending) {
  const int prev = fz::tile_order_ref();
  if (descending >= 0) fz::tile_order_ref() = descending ? 1 : 0;
  return prev;
}
extern "C" const char* fz_last_error_string(void) { return fz::last_error().c_str(); }
extern "C" int64_t fz_launch_count(void) { return fz::launch_counter().load(); }
