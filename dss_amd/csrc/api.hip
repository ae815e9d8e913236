// Error reporting and version entry points of libdss_hip.so.
#include <stdarg.h>
#include "common.h"

namespace dss {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Launch-error check (reference: AT_CUDA_CHECK(cudaGetLastError()), rasterize_points.cu:665).
// Does not synchronise.
int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DSS_ERR_LAUNCH;
    }
    return DSS_OK;
}

}  // namespace dss

extern "C" int dss_version(void) { return DSS_HIP_VERSION; }
extern "C" const char *dss_last_error(void) { return dss::g_err; }
