// Probe: cost per launch of a graph-chained kernel with D dependent global loads (L2-resident data),
// 1024 waves (256 blocks x 256 threads), to price one "memory round trip" inside a short kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int D> __global__ void chain(const int* idx, float* out, int n) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
#pragma unroll
    for (int d = 0; d < D; ++d) i = idx[i];
    if (D == 0 || i >= 0) out[blockIdx.x * blockDim.x + threadIdx.x] = (float)i;
}
template <int D> float run(hipStream_t s, const int* idx, float* out, int n, int blocks) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(chain<D>, dim3(blocks), dim3(256), 0, s, idx, out, n);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / 500;
}
int main() {
    const int n = 65536;
    int* h = (int*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (i * 7919 + 13) % n;
    int* idx; float* out; hipMalloc(&idx, n * 4); hipMalloc(&out, 1 << 22);
    hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    for (int blocks : {1, 256, 1024}) {
        printf("blocks %4d: D=0 %.2f  D=1 %.2f  D=2 %.2f  D=3 %.2f  D=6 %.2f us/launch\n", blocks, run<0>(s, idx, out, n, blocks),
               run<1>(s, idx, out, n, blocks), run<2>(s, idx, out, n, blocks), run<3>(s, idx, out, n, blocks), run<6>(s, idx, out, n, blocks));
    }
    return 0;
}
