// Test helper (tests/test_training_gpu.py::test_cooperative_decoder_fails_safe_when_starved): holds `keep` of the 32 CUs of EVERY XCD for
// `ms` milliseconds -- one 64-thread workgroup with 120 KB of LDS per CU -- so that a kernel which needs 16 co-resident workgroups on
// one XCD (tuber_decoder_coop_fwd, ~70 KB of LDS per workgroup) finds only 32 - keep free CUs there, whichever XCD the hardware picks
// (workgroup i of a launch goes to XCD (i + start) % 8; the start is not the same for every queue).
#include <hip/hip_runtime.h>

__global__ void xcd_blocker_kernel(unsigned long long ticks, int keep, unsigned* started) {
    extern __shared__ char lds[];
    if ((int)(blockIdx.x >> 3) >= keep) return;
    lds[threadIdx.x] = 1;
    if (threadIdx.x == 0) atomicAdd(started, 1u);
    const unsigned long long t0 = wall_clock64();            // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[threadIdx.x ^ 1] == 77) started[1] = 1;          // keeps the LDS allocation observable
}

extern "C" int xcd_blocker_launch(int keep, float ms, unsigned* started, hipStream_t stream) {
    const int lds = 120 * 1024;
    hipError_t e = hipFuncSetAttribute((const void*)xcd_blocker_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(xcd_blocker_kernel, dim3(256), dim3(64), lds, stream, (unsigned long long)(ms * 1e5f), keep, started);
    return (int)hipGetLastError();
}
