// Small non-kernel part of the C ABI (include/zigma_hip.h).
#include "zigma_common.h"

namespace zigma {
static thread_local const char *g_last_kernel = "";
void set_last_kernel(const char *name) { g_last_kernel = name; }
}  // namespace zigma

extern "C" const char *zigma_strerror(int status) {
    switch (status) {
        case ZIGMA_OK: return "ok";
        case ZIGMA_ERR_NULL: return "required pointer is NULL";
        case ZIGMA_ERR_SHAPE: return "size out of the supported range";
        case ZIGMA_ERR_DTYPE: return "unsupported element type";
        case ZIGMA_ERR_STRIDE: return "layout not supported";
        case ZIGMA_ERR_LAUNCH: return "kernel launch failed";
        case ZIGMA_ERR_UNSUPPORTED: return "feature out of scope";
        default: return "unknown status";
    }
}
extern "C" int zigma_abi_version(void) { return ZIGMA_ABI_VERSION; }
extern "C" const char *zigma_last_kernel(void) { return zigma::g_last_kernel; }
