// Probe: per-CU load throughput of a short kernel reading an L2/MALL-resident region
// (every workgroup reads `kb` KiB, 16 B per lane per instruction), as a function of waves per workgroup
// and of register loads vs LDS-DMA.  Graph-chained launches; reports us per launch and GB/s per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gvoid;
template <int MODE> __global__ void rd(const uint4* src, uint4* out, int kb, int region16) {
    extern __shared__ uint4 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int n = kb;                              // 1-KiB wave-instructions per workgroup
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int base = (blockIdx.x * 9973) % (region16 / 64 - n);
    for (int i = wave; i < n; i += nw) {
        const uint4* g = src + (size_t)(base + i) * 64 + lane;
        if (MODE == 0) { uint4 v = *g; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        else __builtin_amdgcn_global_load_lds((gvoid*)g, (lds_void*)(sm + (i % 128) * 64), 16, 0, 0);
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc = sm[threadIdx.x]; }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
}
template <int MODE> void run(hipStream_t s, const uint4* src, uint4* out, int kb, int region16, int threads, int blocks) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(rd<MODE>, dim3(blocks), dim3(threads), 128 * 1024, s, src, out, kb, region16);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float us = ms * 1e3f / 200;
    printf("  mode %s blocks %4d threads %4d KiB/WG %4d: %6.2f us/launch  -> %6.1f GB/s per WG, %5.2f TB/s chip\n", MODE ? "dma" : "reg", blocks, threads, kb,
           us, kb * 1024.0 / (us - 1.7) / 1e3, blocks * kb * 1024.0 / (us - 1.7) / 1e6);
}
int main() {
    const int region16 = (4 << 20) / 16;            // 4 MiB region (L2/MALL resident)
    uint4 *src, *out; hipMalloc(&src, 4 << 20); hipMalloc(&out, 1 << 16); hipMemset(src, 1, 4 << 20);
    hipFuncSetAttribute((const void*)rd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)rd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipStream_t s; hipStreamCreate(&s);
    for (int threads : {256, 512, 1024}) for (int kb : {64, 128}) { run<0>(s, src, out, kb, region16, threads, 256); run<1>(s, src, out, kb, region16, threads, 256); }
    run<0>(s, src, out, 128, region16, 256, 200);
    run<0>(s, src, out, 32, region16, 256, 1024);
    return 0;
}
