// core.hip — version / error plumbing of liblidarnerf_hip.so.
#include "common.h"
#include "wgrad.h"

static thread_local char g_err[512] = "";

void lnh_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int lnh_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

extern "C" {
int lnh_version(void) { return 100; }
const char *lnh_last_error(void) { return g_err; }
const char *lnh_arch(void) { return "gfx950"; }
const char *lnh_build_variant(void) { return LNH_VARIANT_TAG; }
uint64_t lnh_wgrad_workspace_bytes(void) { return wgrad_ws_bytes(); }
}
