// Probe: lane mapping of ds_read_b64_tr_b16 (gfx950) for a K-major LDS tile [k][i] with a 256-B row pitch.
// Hypothesis (cdna_hip_programming.md T10): within each 16-lane group the 16 lanes' 8-byte chunks form a [4][16] block
// (lane j supplies row j>>2, columns (j&3)*4..+3) and lane j receives column j of it: 4 consecutive k for one i.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ unsigned short lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)((i / 128) * 1000 + (i % 128));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const int row = g * 8 + (j >> 2), col = 32 + (j & 3) * 4;          // block: k rows g*8..g*8+3, columns 32..47
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + row * 128 + col));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const int want = ((l >> 4) * 8 + e) * 1000 + 32 + (l & 15);
        if (h[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %u want %d\n", l, e, h[l * 4 + e], want); ++bad; }
    }
    printf("lane 0: %u %u %u %u   lane 17: %u %u %u %u   mismatches %d\n", h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71], bad);
    return 0;
}
