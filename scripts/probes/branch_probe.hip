// Probe: do parallel branches of a captured hipGraph run concurrently?  Each kernel = 50 workgroups spinning ~4 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long cycles, int* p) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
float run(int nbranch, int per_branch, bool graph) {
    hipStream_t s0, side[8]; hipStreamCreate(&s0);
    for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking);
    int* c; hipMalloc(&c, 64 * 4); hipMemset(c, 0, 256);
    hipEvent_t e0, e1, fork, done[8]; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&fork);
    for (int i = 0; i < 8; ++i) hipEventCreate(&done[i]);
    auto body = [&]() {
        hipEventRecord(fork, s0);
        for (int b = 0; b < nbranch; ++b) {
            hipStream_t st = nbranch == 1 ? s0 : side[b];
            if (nbranch > 1) hipStreamWaitEvent(st, fork, 0);
            for (int i = 0; i < per_branch; ++i) hipLaunchKernelGGL(spin, dim3(50), dim3(512), 0, st, 9500LL, c + b * 8);
            if (nbranch > 1) { hipEventRecord(done[b], st); hipStreamWaitEvent(s0, done[b], 0); }
        }
    };
    float ms;
    if (graph) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal); body(); hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s0); hipStreamSynchronize(s0);
        hipEventRecord(e0, s0); hipGraphLaunch(ge, s0); hipEventRecord(e1, s0); hipEventSynchronize(e1);
    } else {
        body(); hipDeviceSynchronize();
        hipEventRecord(e0, s0); body(); hipEventRecord(e1, s0); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    for (int graph = 0; graph < 2; ++graph)
        for (int nb : {1, 2, 4}) {
            float us = run(nb, 400 / nb, graph);
            printf("%s branches %d x %3d kernels (50 WGs, ~4 us each): total %8.1f us  -> %.2f us per kernel\n", graph ? "graph" : "eager", nb, 400 / nb, us, us / 400);
        }
    return 0;
}
