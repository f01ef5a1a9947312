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
extern "C" size_t cdnet_abi_sizeof(const char *name) {
    if (!name) return 0;
#define CDNET_SZ(T) if (!strcmp(name, #T)) return sizeof(T)
    CDNET_SZ(cdnet_conv_src); CDNET_SZ(cdnet_conv_args); CDNET_SZ(cdnet_pack_job); CDNET_SZ(cdnet_head_feat); CDNET_SZ(cdnet_wgrad_reduce_desc);
    CDNET_SZ(cdnet_grad_in); CDNET_SZ(cdnet_bn_bwd_args); CDNET_SZ(cdnet_fuse_term); CDNET_SZ(cdnet_grad_term);
#undef CDNET_SZ
    return 0;
}

// a wave that only waits: the stream probe's load (see include/cdnet_hip.h)
__global__ void spin_kernel(long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int cdnet_spin(int microseconds, void *stream) {
    CDNET_REQUIRE(microseconds >= 0 && microseconds <= 1000000, "cdnet_spin: 0 .. 1 000 000 microseconds");
    spin_kernel<<<1, 64, 0, (hipStream_t)stream>>>((long long)microseconds * 100);
    return cdnet::check_launch("cdnet_spin");
}
