// Probe: effective shader clock and per-launch cost of a dependent chain of tiny kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long cycles, long long* out) {
    long long t0 = clock64(), w0 = wall_clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
}
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny_grid(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
    long long* d; hipMalloc(&d, 16); int* c; hipMalloc(&c, 4); hipMemset(c, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL, d);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
        printf("spin: %lld shader cycles, %lld wall ticks (rate %d kHz) => %.1f us by wallclock, %.1f us by events => shader clock %.0f MHz\n",
               h[0], h[1], wrate, h[1] * 1e3 / wrate, ms * 1e3, h[0] / (h[1] * 1e3 / wrate));
    }
    // chain of 2000 dependent tiny kernels: eager and graph
    for (int grid : {1, 200, 800}) {
        hipEventRecord(e0, s);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, c);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("eager chain grid=%d: %.2f us per launch\n", grid, ms * 1e3 / 2000);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, c);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("graph chain grid=%d: %.2f us per launch\n", grid, ms * 1e3 / 2000);
    }
    return 0;
}
