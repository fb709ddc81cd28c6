// TEST INFRASTRUCTURE ONLY (not part of the product library): a kernel that keeps `wgs` compute units busy for `usec`
// microseconds -- one workgroup per CU (its LDS request leaves no room for a second one or for a persistent-recurrence
// workgroup), spinning on the 100-MHz wall clock.  tests/test_gpu_dp_contention.py issues it where the data-parallel step
// would issue its all-reduces: an emulation, on ONE GPU, of RCCL's channel kernels taking CUs away from the persistent
// recurrences (which need all their workgroups resident at once).
#include <hip/hip_runtime.h>
extern __shared__ char occupy_lds[];
__global__ __launch_bounds__(256) void k_occupy(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (sink && threadIdx.x == 0 && occupy_lds[0] == 77) sink[0] = 1;      // (keeps the LDS allocation alive)
}
extern "C" int e2t_test_occupy(int wgs, int usec, int lds_bytes, int* sink, void* stream) {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2; attr = true; }
    hipLaunchKernelGGL(k_occupy, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (long long)usec * 100, sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
