// Error plumbing and device queries for the C ABI (no compute here).
#include "common.h"
#include "ecog2txt_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int e2t_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char* e2t_last_error(void) { return g_err; }
extern "C" int e2t_abi_version(void) { return E2T_ABI_VERSION; }
extern "C" int e2t_device_cus(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return prop.multiProcessorCount;
}
