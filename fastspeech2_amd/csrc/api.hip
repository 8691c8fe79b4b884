// Error reporting / introspection for libfs2_gfx950.so (C ABI in include/fs2_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void fs2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fs2_last_error(void) { return g_err; }

extern "C" int fs2_abi_version(void) { return 1; }

// Returns the gcnArchName of the current device into buf (e.g. "gfx950:sramecc+:xnack-").
extern "C" int fs2_device_arch(char* buf, int n) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        fs2_set_error("fs2_device_arch: no HIP device");
        return -3;
    }
    strncpy(buf, p.gcnArchName, n > 0 ? n - 1 : 0);
    if (n > 0) buf[n - 1] = 0;
    return 0;
}
