// CU hog for tools/probes/power_limit.py (is the board's power limit what holds the clock of the matrix kernels?).
// hog_kernel: `nblocks` workgroups of 64 threads that each claim 160 KB of LDS - so a CU holds exactly one and has no LDS left for any
// block of the kernel under test - and sleep (s_sleep: no issue slots, next to no power) until `ticks` of the constant 100 MHz counter
// have passed.  The kernel under test, launched on another stream while the hog is resident, can only use the remaining CUs: a
// "half grid" without touching its launch code.  Every hog block records (XCC id, SE, CU) from HW_ID so the placement can be checked.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void hog_kernel(unsigned long long ticks, unsigned* where) {
    extern __shared__ char hog_lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[2 * blockIdx.x] = hw;
        where[2 * blockIdx.x + 1] = xcc;
        hog_lds[0] = (char)hw;                              // (the allocation must be used)
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

extern "C" int probe_hog(int nblocks, unsigned long long ticks, unsigned* where, hipStream_t st) {
    const int lds = 160 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(hog_kernel, dim3(nblocks), dim3(64), lds, st, ticks, where);
    return (int)hipGetLastError();
}
