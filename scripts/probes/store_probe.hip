// Store-pattern probe for the GEMM epilogue: how fast can a 256x256-tile grid write an fp32 [M][N] matrix when every
// wave store instruction covers (a) 16 rows x 64 B (the MFMA accumulator layout with 4 consecutive columns per lane),
// (b) 8 rows x 128 B (full lines, after a lane-pair exchange), (c) linear 1 KiB per instruction?
// build: hipcc -O3 --offload-arch=gfx950 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* C, int M, int N, int ntn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int m0 = tm * 256 + (wave / 4) * 128, n0 = tn * 256 + (wave % 4) * 64;
    const int frow = lane & 15, fq = lane >> 4;
    const float val = (float)lane;
    if (MODE == 2) {       // linear: wave writes its 128x64 region row by row?  no: a flat 32 KiB region
        float4* base = (float4*)(C + ((size_t)blockIdx.x * 8 + wave) * 8192);
        if (((size_t)blockIdx.x * 8 + wave + 1) * 8192 > (size_t)M * N) return;
#pragma unroll
        for (int i = 0; i < 32; ++i) base[i * 64 + lane] = make_float4(val, val, val, val);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gm, gn;
            if (MODE == 0) { gm = m0 + i * 16 + frow; gn = n0 + j * 16 + fq * 4; }
            else {          // pair (j, j^1): instruction j&1 covers rows of that parity with full 128-B lines
                const int jj = j & ~1, par = j & 1;
                gm = m0 + i * 16 + (frow & ~1) + par;
                gn = n0 + jj * 16 + (frow & 1) * 16 + fq * 4;
            }
            if (gm < M && gn < N) *(float4*)(C + (size_t)gm * N + gn) = make_float4(val, val, val, val);
        }
}
int main() {
    const int M = 8704, N = 3200;
    float* C; hipMalloc(&C, (size_t)M * N * 4 + (1 << 20));
    const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) {
                if (mode == 0) k<0><<<ntm * ntn, 512>>>(C, M, N, ntn);
                if (mode == 1) k<1><<<ntm * ntn, 512>>>(C, M, N, ntn);
                if (mode == 2) k<2><<<ntm * ntn, 512>>>(C, M, N, ntn);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.1f us per launch, %.2f TB/s\n", mode, ms * 50, (double)M * N * 4 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
