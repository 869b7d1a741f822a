// liby5m.so: version / error plumbing.
#include "y5m_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";
thread_local int y5m_name_only = 0;
thread_local char y5m_name_buf[192] = "";

extern "C" void y5m_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y5m_version(void) { return "y5m-gfx950 0.1"; }
extern "C" const char* y5m_last_error(void) { return g_err; }

extern "C" int y5m_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        y5m_set_error("no HIP device visible");
        return Y5M_EINVAL;
    }
    hipDeviceProp_t p;
    int dev = 0;
    hipGetDevice(&dev);
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return Y5M_EINVAL;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        y5m_set_error("device is %s, liby5m.so is built for gfx950 only", p.gcnArchName);
        return Y5M_EINVAL;
    }
    return Y5M_OK;
}
