// Probe (round 6): cycles of one Philox4x32-10 evaluation per wave on gfx950, one wave per SIMD, with the two multiplies of a round
// as v_mul_hi_u32 + v_mul_lo_u32 (what hipcc emits for csrc/common.h) or as ONE v_mad_u64_u32 each.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/philox_probe scripts/probes/philox_probe.hip && scripts/probes/philox_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__device__ __forceinline__ void philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0, lo0, hi1, lo1;
        if (MODE == 0) {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        } else {
            unsigned long long p0, p1;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "s"(0xD2511F53u) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "s"(0xCD9E8D57u) : "vcc");
            hi0 = (unsigned)(p0 >> 32); lo0 = (unsigned)p0; hi1 = (unsigned)(p1 >> 32); lo1 = (unsigned)p1;
        }
        unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* cyc, int n) {
    unsigned acc = 0, r[4];
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        philox<MODE>(threadIdx.x + i, acc, 7u, 0u, 11u, 13u, r);
        acc ^= r[0] ^ r[1] ^ r[2] ^ r[3];
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int n = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, n);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, n);
            hipDeviceSynchronize();
        }
        long long h[256]; unsigned o[4];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("mode %d (%s): %.1f clock64 ticks per Philox4x32-10 evaluation per wave (one wave per SIMD), check %08x\n", mode,
               mode ? "v_mad_u64_u32" : "v_mul_hi_u32 + v_mul_lo_u32", s / 256 / n, o[0]);
    }
    return 0;
}
