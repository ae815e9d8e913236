// Error reporting and version entry points of libdss_hip.so.
#include <stdarg.h>
#include <atomic>
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

// Process-wide option table of dss_set_option: the ONLY mutable process state besides the thread-local error string.
// Relaxed atomics: an option is a tuning hint read once per call; callers change it between calls, not during one.
static std::atomic<int> g_opt[DSS_OPT_COUNT];
int option(int which) { return (which >= 0 && which < DSS_OPT_COUNT) ? g_opt[which].load(std::memory_order_relaxed) : 0; }

// Per-device cache of (CU count, resident workgroups per CU of the backward gather) -- see raster_backward.hip.
static std::atomic<int> g_dev_cache[DSS_MAX_DEVICES][DSS_DEV_CACHE_SLOTS];
std::atomic<int> *device_cache(int dev) { return (dev >= 0 && dev < DSS_MAX_DEVICES) ? g_dev_cache[dev] : nullptr; }

}  // namespace dss

extern "C" int dss_set_option(int option, int value)
{
    if (option < 0 || option >= DSS_OPT_COUNT) { dss::set_error("dss_set_option: unknown option %d", option); return DSS_ERR_INVALID_ARGUMENT; }
    if (option == DSS_OPT_BACKWARD_TPW && !(value == 0 || value == 1 || value == 2 || value == 4)) {
        dss::set_error("dss_set_option: DSS_OPT_BACKWARD_TPW takes 0 (automatic), 1, 2 or 4");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    dss::g_opt[option].store(value, std::memory_order_relaxed);
    return DSS_OK;
}
extern "C" int dss_get_option(int option) { return dss::option(option); }

// rows a band tensor has: the rows of [row0, row1) for a contiguous band; with row_cycle c > 1 the rows of every c-th
// 8-row tile row of [row0, row1), starting at row0
extern "C" int dss_band_rows(int row0, int row1, int row_cycle)
{
    if (row1 <= row0) return 0;
    if (row_cycle <= 1) return row1 - row0;
    const int span = row1 - row0, period = 8 * row_cycle;
    const int full = span / period, rem = span - full * period;
    return full * 8 + (rem < 8 ? rem : 8);
}

extern "C" int dss_version(void) { return DSS_HIP_VERSION; }
extern "C" const char *dss_last_error(void) { return dss::g_err; }
