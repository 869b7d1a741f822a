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

// 32-bit fill as a KERNEL launch. The library never calls hipMemsetAsync: inside a captured hipGraph a memset is its own
// node type (not a kernel node), and this code base keeps every node of its graphs a kernel node (NOTES.md, "graph replay").
__global__ void y5m_fill32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n, int vec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (vec) {                                  // 16-byte aligned base: 16-byte stores + scalar tail
        const size_t n4 = n / 4;
        uint4* p4 = reinterpret_cast<uint4*>(p);
        const uint4 v4 = make_uint4(v, v, v, v);
        for (size_t j = i; j < n4; j += stride) p4[j] = v4;
        for (size_t j = n4 * 4 + i; j < n; j += stride) p[j] = v;
    } else {
        for (size_t j = i; j < n; j += stride) p[j] = v;
    }
}
int y5m_fill32(void* p, uint32_t v, size_t n_words, hipStream_t st) {
    if (n_words == 0) return Y5M_OK;
    if ((reinterpret_cast<uintptr_t>(p) & 3) != 0) { y5m_set_error("y5m_fill32: pointer must be 4-byte aligned"); return Y5M_EINVAL; }
    const int vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    size_t blocks = ((vec ? n_words / 4 : n_words) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(y5m_fill32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), v, n_words, vec);
    Y5M_CHECK_LAUNCH("y5m_fill32_kernel");
    return Y5M_OK;
}

extern "C" const char* y5m_version(void) { return "y5m-gfx950 0.1"; }
extern "C" const char* y5m_last_error(void) { return g_err; }

// workgroups a persistent (one per CU) launch uses: the device's CU count capped by Y5M_PERSIST_CUS (y5m_common.h)
extern "C" int y5m_persistent_cu_count(void) { return y5m_persistent_cus(); }

// the Y5M_R4_KERNELS mask as the library parsed it (once, atoi & 31: y5m_common.h) -- what the launchers actually dispatch on
extern "C" int y5m_r4_kernel_forms(void) { return y5m_r4_forms(); }

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
