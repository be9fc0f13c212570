// common.hip — error buffer and ABI version.
#include "mppo_common.h"

namespace mppo {
char* last_error_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace mppo

extern "C" const char* mppo_last_error(void) { return mppo::last_error_buf(); }
extern "C" int32_t mppo_abi_version(void) { return MPPO_ABI_VERSION; }
