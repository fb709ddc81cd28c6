// Probe: semantics of __builtin_amdgcn_global_load_lds (16 B) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gvoid;
__global__ void k(const uint4* src, uint4* dst, int n16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave copies 2 KiB-chunks: chunk c -> LDS offset c*1024 (+lane*16 implicit), source permuted: lane reads src[c*64 + (lane ^ 1)]
    for (int c = wave; c < n16 / 64; c += blockDim.x / 64) {
        const uint4* g = src + c * 64 + (lane ^ 1);
        __builtin_amdgcn_global_load_lds((gvoid*)g, (lds_void*)(smem + c * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = ((uint4*)smem)[i];
}
int main() {
    const int n16 = 1024; // 16 KiB
    uint4 *s, *d; hipMalloc(&s, n16 * 16); hipMalloc(&d, n16 * 16);
    uint4* h = (uint4*)malloc(n16 * 16);
    for (int i = 0; i < n16; ++i) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    hipMemcpy(s, h, n16 * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), n16 * 16, 0, s, d, n16);
    hipMemcpy(h, d, n16 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n16; ++i) { int e = (i & ~63) + ((i & 63) ^ 1); if (h[i].x != (unsigned)e || h[i].w != (unsigned)e * 4) bad++; }
    printf("glds probe: %d mismatches of %d (expect 0: LDS[c*64+lane] = src[c*64+(lane^1)])\n", bad, n16);
    return bad != 0;
}
