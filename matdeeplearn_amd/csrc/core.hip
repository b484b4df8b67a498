// core.hip — version / error plumbing of libmdl_hip.so (include/mdl_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "mdl_common.h"

namespace mdl {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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
