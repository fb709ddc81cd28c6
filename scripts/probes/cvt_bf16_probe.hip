#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_hw(float a, float b) {
    f32x2 v = {a, b};
    bf16x2v h = __builtin_convertvector(v, bf16x2v);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__global__ void k_cmp(unsigned long long* bad, unsigned* first) {
    const unsigned long long n = 1ull << 32;
    unsigned long long cnt = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)u;
        const unsigned ex = (bits >> 23) & 0xFF;
        if (ex == 0xFF && (bits & 0x7FFFFF)) continue;   // NaN
        const float f = __uint_as_float(bits);
        const unsigned hw = pk_hw(f, f) & 0xFFFF;
        const unsigned sw = f2bf(f);
        if (hw != sw) { ++cnt; atomicMin(first, bits); }
    }
    if (cnt) atomicAdd(bad, cnt);
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4);
    hipLaunchKernelGGL(k_cmp, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long hb; unsigned hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("mismatches (non-NaN): %llu first bits 0x%08x\n", hb, hf);
    return 0;
}
