// ABI plumbing: version, build info, thread-local error text.
#include "common.h"
#include <string.h>

namespace cdnet {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cdnet

extern "C" int cdnet_abi_version(void) { return CDNET_ABI_VERSION; }
extern "C" const char *cdnet_last_error(void) { return cdnet::g_err; }
extern "C" const char *cdnet_build_info(void) { return "cdnet_hip;gfx950;wave64;hipcc"; }
