// Probe: LDS-DMA throughput per CU for the access patterns of the LSTM step kernel.
//  pattern 0: linear 1 KiB per instruction;  pattern 1: 8 rows x 128 B (row stride 1600 B)
//  region: small (L2-hot, 2 MiB) or large (512 MiB, MALL/HBM).  256 workgroups x 512 threads, 124 instr per WG.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gvoid;
__global__ void rd(const char* src, float* out, int pattern, size_t region, int ninstr, int iter) {
    extern __shared__ uint4 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t span = (size_t)ninstr * 1600 * 8;
    size_t base = ((size_t)(blockIdx.x + 256 * iter) * 1000003ull * 4096ull) % (region - span);
    base &= ~(size_t)127;
    for (int i = wave; i < ninstr; i += nw) {
        const char* g;
        if (pattern == 0) g = src + base + (size_t)i * 1024 + lane * 16;
        else g = src + base + ((size_t)i * 8 + (lane >> 3)) * 1600 + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((gvoid*)g, (lds_void*)(sm + (i % 128) * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    uint4 v = sm[threadIdx.x];
    if (v.x == 0x12345678u) out[threadIdx.x] = 1.f;
}
int main() {
    const size_t big = 512ull << 20;
    char* src; float* out; hipMalloc(&src, big); hipMalloc(&out, 1 << 16); hipMemset(src, 1, big);
    hipFuncSetAttribute((const void*)rd, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pattern : {0, 1}) for (size_t region : {(size_t)4 << 20, big}) for (int ninstr : {64, 124}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(rd, dim3(200), dim3(512), 128 * 1024, s, src, out, pattern, region, ninstr, i);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float us = ms * 1e3f / 100;
        printf("pattern %d region %4zu MiB instr/WG %3d: %5.2f us/launch -> %5.1f cycles/instr/CU, %5.1f GB/s/CU\n", pattern, region >> 20, ninstr, us,
               (us - 1.7) * 2390 / ninstr, ninstr * 1024.0 / (us - 1.7) / 1e3);
    }
    return 0;
}
