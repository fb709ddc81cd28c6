// Probe: latency of handing a value from one workgroup to another INSIDE a launch on gfx950, by cache-policy
// bits on the store and on the polling load, for two workgroups on the same XCD and on different XCDs.
// Ping-pong: A stores i -> B polls until it reads i, stores i to the return slot -> A polls.  Reported: one-way us.
// Also prices the 3-leg chain used by k_lstm_seq_fwd_persist (data store, vmcnt(0), counter atomic; poll; data load).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ST(bits) asm volatile("global_store_dword %0, %1, off " bits "\n" :: "v"(p), "v"(v) : "memory")
#define LD(bits) asm volatile("global_load_dword %0, %1, off " bits "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory")
template <int M> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (M == 0) ST(""); else if (M == 1) ST("sc0"); else if (M == 2) ST("sc1"); else if (M == 3) ST("sc0 sc1"); else ST("nt sc0 sc1");
}
template <int M> __device__ __forceinline__ unsigned ld(const unsigned* p) {
    unsigned v;
    if (M == 0) LD(""); else if (M == 1) LD("sc0"); else if (M == 2) LD("sc1"); else if (M == 3) LD("sc0 sc1"); else LD("nt sc0 sc1");
    return v;
}
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

template <int SM, int LM> __global__ void pingpong(unsigned* slots, int a, int b, int rounds, long long* out, unsigned* xcc, int* fail) {
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) xcc[wg] = xcc_id();
    if (threadIdx.x != 0 || (wg != a && wg != b)) return;
    unsigned* fwd = slots; unsigned* back = slots + 64;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; ++i) {
        if (wg == a) {
            st<SM>(fwd, i);
            int spin = 0; while (ld<LM>(back) != (unsigned)i) if (++spin > 200000) { *fail = 1; return; }
        } else {
            int spin = 0; while (ld<LM>(fwd) != (unsigned)i) if (++spin > 200000) { *fail = 1; return; }
            st<SM>(back, i);
        }
    }
    if (wg == a) out[0] = wall_clock64() - t0;
}
// chain: producer: store 'data' (SM), waitcnt, atomic add on counter; consumer: poll counter (atomic load), then load data (LM)
template <int SM, int LM> __global__ void chain(unsigned* slots, int a, int b, int rounds, long long* out, int* fail) {
    const int wg = blockIdx.x;
    if (threadIdx.x != 0 || (wg != a && wg != b)) return;
    unsigned* dataF = slots + 128, *ctrF = slots + 192, *dataB = slots + 256, *ctrB = slots + 320;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; ++i) {
        unsigned *sd = (wg == a) ? dataF : dataB, *sc = (wg == a) ? ctrF : ctrB;
        unsigned *rd = (wg == a) ? dataB : dataF, *rc = (wg == a) ? ctrB : ctrF;
        if (wg == a) {
            st<SM>(sd, i); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spin = 0;
        while (__hip_atomic_load(rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) if (++spin > 200000) { *fail = 1; return; }
        if (ld<LM>(rd) != (unsigned)i) { *fail = 2; return; }
        if (wg == b) {
            st<SM>(sd, i); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (wg == a) out[0] = wall_clock64() - t0;
}
static unsigned* slots; static long long* out; static unsigned* xcc; static int* fail;
template <int SM, int LM> void run(const char* name, int a, int b) {
    const int rounds = 2000;
    hipMemset(slots, 0, 4096); hipMemset(fail, 0, 4); hipMemset(out, 0, 8);
    hipLaunchKernelGGL((pingpong<SM, LM>), dim3(16), dim3(64), 0, 0, slots, a, b, rounds, out, xcc, fail);
    hipDeviceSynchronize();
    long long t; int f; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    hipMemset(slots, 0, 4096); hipMemset(fail, 0, 4); hipMemset(out, 0, 8);
    hipLaunchKernelGGL((chain<SM, LM>), dim3(16), dim3(64), 0, 0, slots, a, b, rounds, out, fail);
    hipDeviceSynchronize();
    long long t2; int f2; hipMemcpy(&t2, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&f2, fail, 4, hipMemcpyDeviceToHost);
    printf("  %-28s flag one-way %6.3f us %s | store+ack+atomic / poll+load one-way %6.3f us %s\n", name,
           f ? -1.0 : t * 0.01 / (2.0 * rounds), f ? "(NOT SEEN)" : "", f2 ? -1.0 : t2 * 0.01 / (2.0 * rounds), f2 == 1 ? "(ctr NOT SEEN)" : f2 == 2 ? "(STALE DATA)" : "");
}
int main() {
    hipMalloc(&slots, 4096); hipMalloc(&out, 8); hipMalloc(&xcc, 64); hipMalloc(&fail, 4);
    unsigned hx[16];
    for (int pair = 0; pair < 2; ++pair) {
        const int a = 0, b = pair ? 1 : 8;
        printf("workgroups %d and %d:\n", a, b);
        run<0, 0>("st plain / ld plain", a, b);
        run<0, 1>("st plain / ld sc0", a, b);
        run<0, 2>("st plain / ld sc1", a, b);
        run<1, 1>("st sc0 / ld sc0", a, b);
        run<2, 2>("st sc1 / ld sc1", a, b);
        run<3, 3>("st sc0sc1 / ld sc0sc1", a, b);
        run<2, 1>("st sc1 / ld sc0", a, b);
        run<0, 3>("st plain / ld sc0sc1", a, b);
        hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost);
    }
    printf("xcc id by workgroup:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
    return 0;
}
