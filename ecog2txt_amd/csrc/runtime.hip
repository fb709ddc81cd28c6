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
        case 5: return (int)sizeof(e2t_gemm_call);
        case 6: return (int)sizeof(e2t_tile_desc);
        default: return -1;
    }
}
extern "C" int e2t_device_cus(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return prop.multiProcessorCount;
}

// CRC-32C of a host buffer with the SSE4.2 crc32 instruction (TFRecord / TF-checkpoint framing: rows a3, f1, f2)
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const unsigned char* p, size_t n, uint32_t crc) {
    unsigned long long c = crc ^ 0xFFFFFFFFu;
    while (n && ((uintptr_t)p & 7)) { c = __builtin_ia32_crc32qi((unsigned)c, *p++); --n; }
    for (; n >= 8; n -= 8, p += 8) c = __builtin_ia32_crc32di(c, *(const unsigned long long*)p);
    while (n--) c = __builtin_ia32_crc32qi((unsigned)c, *p++);
    return (uint32_t)c ^ 0xFFFFFFFFu;
}
static uint32_t crc32c_sw(const unsigned char* p, size_t n, uint32_t crc) {
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n--) {
        c ^= *p++;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    }
    return c ^ 0xFFFFFFFFu;
}
extern "C" uint32_t e2t_crc32c(const void* data, size_t n, uint32_t crc) {
    static const bool hw = __builtin_cpu_supports("sse4.2");
    return hw ? crc32c_hw((const unsigned char*)data, n, crc) : crc32c_sw((const unsigned char*)data, n, crc);
}
