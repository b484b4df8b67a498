// core.hip — version / error plumbing of libmdl_hip.so (include/mdl_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <unordered_map>

#include "mdl_common.h"

namespace mdl {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size class) instead of once per launch: the
// call takes the runtime's global lock, and with a deep launch queue (a training loop that runs ahead of the device) it
// cost the host more than the launch itself.  The attribute is monotone (we only ever raise it), so a small table of
// the largest value set so far is enough; it is per device because the attribute is.
hipError_t set_max_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)reinterpret_cast<uintptr_t>(kernel) ^ ((uint64_t)(unsigned)dev << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[key] = bytes;
    return e;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MDL_E_LAUNCH;
    }
    return MDL_OK;
}
}  // namespace mdl

extern "C" int mdl_version(void) { return MDL_VERSION; }
extern "C" const char* mdl_last_error_string(void) { return mdl::g_err; }
