// ct_api.hip — error reporting and ABI version of libct_hip.so.
#include "ct_common.h"

#include <cstring>

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

int ct_abi_version(void) { return 2; }  // 2: ct_marlin24_compress_w4_verdict takes the caller's workspace (round 6)

// ---- host mailbox: the two places where the reference's interface makes the HOST wait for a device result ----------------------
int ct_mailbox_alloc(int64_t bytes, void** host_ptr, void** dev_ptr) {
    using namespace ct;
    CT_REQUIRE(bytes > 0 && host_ptr && dev_ptr, "ct_mailbox_alloc: bad arguments");
    void* h = nullptr;
    int rc = hip_check(hipHostMalloc(&h, (size_t)bytes, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable), "hipHostMalloc");
    if (rc) return rc;
    void* d = nullptr;
    rc = hip_check(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
    if (rc) { (void)hipHostFree(h); return rc; }
    std::memset(h, 0, (size_t)bytes);
    *host_ptr = h;
    *dev_ptr = d;
    return CT_OK;
}

int ct_mailbox_free(void* host_ptr) {
    if (!host_ptr) return CT_OK;
    return ct::hip_check(hipHostFree(host_ptr), "hipHostFree");
}

int ct_stream_wait(ct_stream_t stream) {
    hipStream_t s = ct::as_stream(stream);
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return CT_OK;
        if (e != hipErrorNotReady) return ct::hip_check(e, "hipStreamQuery");
    }
}

int ct_mailbox_wait_i64(const int64_t* host_word, int64_t pending, ct_stream_t stream, int64_t* value) {
    using namespace ct;
    CT_REQUIRE(host_word && value, "ct_mailbox_wait_i64: bad arguments");
    const volatile int64_t* w = host_word;
    hipStream_t s = as_stream(stream);
    for (unsigned spin = 0;; ++spin) {
        const int64_t v = *w;
        if (v != pending) { *value = v; return CT_OK; }
        if ((spin & 63u) == 63u) {  // every 64 reads of the word: has the stream drained without the word arriving?
            const hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) { *value = *w; return CT_OK; }  // the work is complete: whatever the word holds now is final
            if (e != hipErrorNotReady) return hip_check(e, "hipStreamQuery");
        }
    }
}

}  // extern "C"
