// Shared device/host helpers for libabx_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ABX_OK 0
#define ABX_ERR_ARG (-1)

// thread-local last error text (abx_last_error_string)
void abx_set_error(const char* msg);
int abx_check_launch(const char* what);
// Raise a kernel's dynamic-LDS limit once per (device, kernel): the attribute belongs to the device the calling thread has
// current, so a process that drives several GPUs configures each of them (a per-thread flag would skip the second device).
// Returns 0 or the hipError_t; never called inside a graph capture after the first eager launch on a device.
int abx_ensure_dynamic_lds(const void* kernel, int bytes, const char* what);

#define ABX_REQUIRE(cond, msg)            \
    do {                                  \
        if (!(cond)) {                    \
            abx_set_error(msg);           \
            return ABX_ERR_ARG;           \
        }                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> three packed bf16 pairs (lo half = first element) with a + b = p0 + p1 + p2 exactly: round-to-nearest-even
// pieces (v_cvt_pk_bf16_f32), the subtractions are exact.  The operand image of the split-bf16 GEMM kernels (gemm3.hip).
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    f32x2 x = {a, b};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 x0 = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
    f32x2 r = x - x0;
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 x1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    f32x2 r2 = r - x1;
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// Split-f16 weight GEMMs (AbxGemm.b_f16): an activation pair scaled by 2^ABX_F16_A_EXP -> two packed f16 pairs with
// x' = p0 + p1 * 2^-11 (+ <= 2^-23 |x'|: 23 significant bits): round-to-nearest pieces (v_cvt_pk_f16_f32), exact subtraction, the remainder scaled by
// 2^11 so that it never becomes a float16 subnormal before |x'| < 2^-22.  |x'| >= 65520 gives p0 = inf, p1 = -inf: NaN downstream.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define ABX_F16_A_EXP (-4)
__device__ __forceinline__ void split2h(float a, float b, unsigned& p0, unsigned& p1) {
    f32x2 x = {a, b};
    x *= (f32x2){0.0625f, 0.0625f};
    const f16x2 h0 = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(h0, f32x2)) * (f32x2){2048.0f, 2048.0f};
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// one product term of the split GEMMs on 8-element fragments held as raw 16 bytes
template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Gates: 1 / (1 + e^-x) as v_exp_f32 + v_rcp_f32 (1 ulp each: relative error < 3e-7) - 4 VALU instructions instead of the ~18 of
// expf + IEEE division; the pair stack evaluates ~47 G gates per step (glu, final gate, attention gates), beside the MFMAs
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Diagnostics (AbxGemm.clock_probe / AbxTriAttn.clock_probe): shader-clock and constant-100-MHz ticks a workgroup was resident for
struct ClockProbe {
    unsigned long long c0, r0;
    unsigned long long* acc;
    __device__ __forceinline__ explicit ClockProbe(unsigned long long* a) : c0(0), r0(0), acc(a) {
        if (acc) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ __forceinline__ void finish() const {
        if (acc && threadIdx.x == 0) {
            atomicAdd(acc, (unsigned long long)__builtin_amdgcn_s_memtime() - c0);
            atomicAdd(acc + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - r0);
        }
    }
};

#define ABX_NEG_MAX (-3.4028234663852886e38f)  // torch.finfo(float32).min
