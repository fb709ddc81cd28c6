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
extern "C" int e2t_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(e2t_gemm_epilogue);
        case 1: return (int)sizeof(e2t_lstm_desc);
        case 2: return (int)sizeof(e2t_pack_desc);
        case 3: return (int)sizeof(e2t_adam_hyper);
        case 4: return (int)sizeof(e2t_dropout);
        default: return -1;
    }
}
extern "C" int e2t_device_cus(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return prop.multiProcessorCount;
}
