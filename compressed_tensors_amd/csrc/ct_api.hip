// ct_api.hip — error reporting and ABI version of libct_hip.so.
#include "ct_common.h"

namespace ct {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return CT_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return CT_ERR_HIP;
}

}  // namespace ct

extern "C" {

const char* ct_last_error(void) { return ct::g_err; }

int ct_abi_version(void) { return 1; }

}  // extern "C"
