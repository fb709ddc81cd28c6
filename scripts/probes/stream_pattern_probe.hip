// HBM read bandwidth as a function of the access pattern of the fused conv front-end: a workgroup of 256 threads reads, per step,
// ROWS runs of RUN bytes (ROWS * RUN = 16 KiB), the runs `stride` bytes apart, then moves RUN bytes forward in every run.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe stream_pattern_probe.hip ; run: /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int RUN, int DEPTH>
__global__ __launch_bounds__(256) void k_read(const char* x, size_t block_bytes, size_t stride, int steps, float* out) {
    constexpr int TPR = RUN / 16, ROWS = 256 / TPR * (16384 / 4096);       // threads per run; 16 KiB per step = 4 loads per thread
    const int tid = threadIdx.x;
    const char* base = x + (size_t)blockIdx.x * block_bytes;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; s += DEPTH) {
        f32x4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 256 + tid) / TPR, c = (i * 256 + tid) % TPR;
                v[d][i] = __builtin_nontemporal_load((const f32x4*)(base + (size_t)row * stride + (size_t)(s + d) * RUN + c * 16));
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += v[d][i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
template <int RUN, int DEPTH>
static void run(const char* x, size_t total, float* out, int wgs_per_cu_hint) {
    constexpr int ROWS = 16384 / RUN;
    // every workgroup owns ROWS runs of 48 KiB (the conv front-end: 12 taps x 1024 channels x 4 B), contiguous in memory: block = ROWS * 48 KiB
    const size_t runlen = 49152, block = (size_t)ROWS * runlen;
    const int nblocks = (int)(total / block);
    const int steps = (int)(runlen / RUN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_read<RUN, DEPTH>), dim3(nblocks), dim3(256), 0, 0, x, block, runlen, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("run %5d B x %3d rows, depth %d: %7.1f us  %.2f TB/s\n", RUN, ROWS, DEPTH, ms * 1e3, (double)nblocks * block / ms / 1e9);
}
__global__ void k_fill(unsigned* x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)i * 2654435761u; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        x[i] = (v & 0x007FFFFF) | 0x3F000000;          // floats in [0.5, 1)
    }
}
// the front-end's own pattern: a workgroup = 64 utterances at one decimated step; per step, run of 256 B per utterance; the 12
// taps of a row are walked BACKWARDS in memory (tap w = sample len-1-(t'*12+w)), 16 steps of 256 B forward inside a tap
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_front(const char* x, int B, int T, int S, float* out) {
    const int tiles_b = B / 64;
    const int tp = blockIdx.x / tiles_b, b0 = (blockIdx.x % tiles_b) * 64;
    const int tid = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    const size_t rowbytes = 4096;
    for (int s = 0; s < 192; s += DEPTH) {
        f32x4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 256 + tid) / 16, c = (i * 256 + tid) % 16;
                const int st = s + d, w = st / 16, c0 = st % 16;
                const size_t t_idx = (size_t)(T - 1 - (tp * 12 + w));
                v[d][i] = __builtin_nontemporal_load((const f32x4*)(x + ((size_t)(b0 + row) * T + t_idx) * rowbytes + c0 * 256 + c * 16));
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += v[d][i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
template <int DEPTH>
static void run_front(const char* x, float* out) {
    const int B = 256, T = 2000, S = 166;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_read_front<DEPTH>), dim3(S * (B / 64)), dim3(256), 0, 0, x, B, T, S, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("front-end pattern (64 utterances x 256 B, taps backwards), depth %d: %7.1f us  %.2f TB/s\n", DEPTH, ms * 1e3, (double)S * B * 49152 / ms / 1e9);
}
int main() {
    const size_t total = (size_t)2100 << 20;
    char* x; float* out;
    hipMalloc(&x, total); hipMalloc(&out, 4);
    hipMemset(x, 1, total);
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)x, total / 4); hipDeviceSynchronize(); printf("-- random data\n"); }
    run<256, 1>(x, total, out, 0); run<256, 4>(x, total, out, 0);
    run<1024, 1>(x, total, out, 0);
    run<4096, 1>(x, total, out, 0); run<4096, 4>(x, total, out, 0);
    run_front<1>(x, out); run_front<2>(x, out); run_front<4>(x, out);
    }
    return 0;
}
