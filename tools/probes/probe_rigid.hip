// Instrumented copy of rigid_update_kernel (csrc/geometry.hip: the same rigid_update_row from csrc/rigid_dev.h) for
// tools/probes/cosched_repro.py: the kernel ITSELF records what it read, beside its outputs, so that a launch whose
// output differs under >= 3 processes on one GPU is caught with its inputs in hand without any extra clone kernel in the loop.
//   dbg row (40 words per residue): [ upd 6 | cur_q 4 | delta_q 4 | cur_t 3 | cur_R 9 | fixed 1 | init_q 4 | init_t 3 ] as read by the
//   thread (34 floats), then HW_ID, MODE at entry, MODE at exit, shader-clock ticks entry -> exit (low 32 bits), XCC_ID, trap status
//   bits: where the wave ran, whether its floating-point mode register changed under it, and whether it was gone for a context switch.
//     hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iabx_amd/csrc -Iinclude -shared tools/probes/probe_rigid.hip \
//           -o tools/probes/bin/libprobe_rigid.so
#include <hip/hip_runtime.h>
#include "rigid_dev.h"

__global__ __launch_bounds__(256) void probe_rigid_update_kernel(const float* __restrict__ upd, const int* __restrict__ fixed,
                                                                 const float* __restrict__ init_q, const float* __restrict__ init_t,
                                                                 float* __restrict__ cur_q, float* __restrict__ cur_t,
                                                                 float* __restrict__ cur_R, float* __restrict__ delta_q, int n, float pscale,
                                                                 unsigned* __restrict__ dbg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned mode0 = __builtin_amdgcn_s_getreg(1 | (0 << 6) | ((32 - 1) << 11));        // hwreg(HW_REG_MODE, 0, 32)
    float u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = upd[(long long)i * 6 + k];
    unsigned* d = dbg + (long long)i * 40;
    // the inputs, as THIS thread reads them (the same addresses rigid_update_row reads; __restrict__ in/out rows are read before written)
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = __float_as_uint(u[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) { d[6 + k] = __float_as_uint(cur_q[i * 4 + k]); d[10 + k] = __float_as_uint(delta_q[i * 4 + k]); }
#pragma unroll
    for (int k = 0; k < 3; ++k) d[14 + k] = __float_as_uint(cur_t[i * 3 + k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) d[17 + k] = __float_as_uint(cur_R[(long long)i * 9 + k]);
    d[26] = (unsigned)fixed[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[27 + k] = __float_as_uint(init_q[i * 4 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[31 + k] = __float_as_uint(init_t[i * 3 + k]);
    rigid_update_row(i, u, fixed, init_q, init_t, cur_q, cur_t, cur_R, delta_q, pscale);
    const unsigned mode1 = __builtin_amdgcn_s_getreg(1 | (0 << 6) | ((32 - 1) << 11));
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));         // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((32 - 1) << 11));         // HW_REG_XCC_ID
    const unsigned trapsts = __builtin_amdgcn_s_getreg(3 | (0 << 6) | ((32 - 1) << 11));      // HW_REG_TRAPSTS
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    d[34] = hwid; d[35] = mode0; d[36] = mode1; d[37] = (unsigned)(t1 - t0); d[38] = xcc; d[39] = trapsts;
}

extern "C" int probe_rigid_update(const float* upd6, const int* fixed_mask, const float* init_q, const float* init_t, float* cur_q,
                                  float* cur_t, float* cur_R, float* delta_q, int n, float pscale, unsigned* dbg, hipStream_t st) {
    hipLaunchKernelGGL(probe_rigid_update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, upd6, fixed_mask, init_q, init_t, cur_q, cur_t,
                       cur_R, delta_q, n, pscale, dbg);
    return (int)hipGetLastError();
}
