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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Split-f16 weight GEMMs (AbxGemm.b_f16): an activation pair scaled by 2^ABX_F16_A_EXP -> two packed f16 pairs with
// x' = p0 + p1 * 2^-11 (+ <= 2^-23 |x'|: 23 significant bits): round-to-nearest pieces (v_cvt_pk_f16_f32), exact subtraction, the remainder scaled by
// 2^11 so that it never becomes a float16 subnormal before |x'| < 2^-22.  |x'| >= 65520 gives p0 = inf, p1 = -inf: NaN downstream.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define ABX_F16_A_EXP (-4)
// (raw: the caller has applied the scale)
__device__ __forceinline__ void split2h_raw(float a, float b, unsigned& p0, unsigned& p1) {
    const f32x2 x = {a, b};
    const f16x2 h0 = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(h0, f32x2)) * (f32x2){2048.0f, 2048.0f};
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// The same pieces with the remainder by the mixed-precision FMA (f16 source x f32 constant + f32 addend -> f16 result): (x - p0) 2^11 =
// x 2^11 - 2^11 p0 is exact in fp32 (x - p0 is, and the scale is a power of two), so one rounding to f16 gives what convert - subtract -
// scale - convert gives, in 5 instructions per pair instead of 7.  For the GEMM main loop only (two pairs per MFMA group); `m2048` is
// -2048.0f in a VGPR of the caller (an SGPR operand costs the main kernels a spill, a literal is not encodable in VOP3P).
__device__ __forceinline__ void split2h_mix(float a, float b, float m2048, unsigned& p0, unsigned& p1) {
    const f32x2 x = (f32x2){a, b} * (f32x2){0.0625f, 0.0625f};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
    const f32x2 xs = (f32x2){a, b} * (f32x2){128.0f, 128.0f};
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(p0), "v"(m2048), "v"(xs[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p1) : "v"(p0), "v"(m2048), "v"(xs[1]));
}
__device__ __forceinline__ void split2h(float a, float b, unsigned& p0, unsigned& p1) {
    split2h_raw(a * 0.0625f, b * 0.0625f, p0, p1);
}
// The partner operand of a split-f16 product when it is an ACTIVATION too (keys / values of the triangle attention): scaled by
// 2^4, three packed f16 planes p0 = f16(x'), p1 = f16(x' - p0), p2 = p0 * 2^-11 - what abx_split_weights_f16 writes for weights,
// with a fixed scale.  |x| < 4095 (beyond: NaN downstream); 23 significant bits from |x| = 2^-6, 2^-29 absolute below.
__device__ __forceinline__ void split2w(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const f32x2 x = (f32x2){a, b} * (f32x2){16.0f, 16.0f};
    const f16x2 h0 = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h0, f32x2);
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    p2 = __builtin_bit_cast(unsigned, h0 * (f16x2){(_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f});
}
// Product terms of one fp32 product, smallest first.  A: the two-piece operand a0 + a1 2^-11; B: planes p0, p1 (in memory) and p2 = p0 2^-11 (derived in registers):
// a1 p2 + a0 p1 + a0 p0 (every f16 x f16 product is exact in fp32; dropped: a1 p1 2^-11 <= 2^-22 |a b|).
struct SplitTerms { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {2, 1, 0}; };
// third term's operand of the plane side: p2 = p0 * 2^-11, derived in registers (one rounding of an exact value: the same bits a
// stored plane would hold) - 4 v_pk_mul_f16 per fragment instead of a third plane in memory, in the DMA stream and in the LDS reads
__device__ __forceinline__ u32x4 f16x8_lo(u32x4 v) {
    const f16x8 h = __builtin_bit_cast(f16x8, v);
    const _Float16 k = (_Float16)4.8828125e-4f;
    return __builtin_bit_cast(u32x4, h * (f16x8){k, k, k, k, k, k, k, k});
}
// the plane side of an activation for C_split outputs: 16 x as p0 = f16(x'), p1 = f16(x' - p0)
__device__ __forceinline__ void split2b(float a, float b, unsigned& p0, unsigned& p1) {
    const f32x2 x = (f32x2){a, b} * (f32x2){16.0f, 16.0f};
    const f16x2 h0 = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h0, f32x2);
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// one product term of the split GEMMs on 8-element fragments held as raw 16 bytes
__device__ __forceinline__ f32x16 mfma_split(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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

// ReLU that keeps a NaN (fmaxf / v_max_f32 return the other operand): an accumulator that an out-of-range split-f16 operand has
// turned into NaN must reach the output and the range probe, not become a silent 0
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

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

// AbxGemm.c_split_tile: the GEMM rows of a plane-output projection are pair positions in (8 i x 16 k) block order, so that the 32 rows
// of a wave tile are two i of ONE k-tile: their plane bytes ([k-tile][plane][i][16]) are 64 contiguous bytes per channel and store
// instruction (row order m = i Lp + k gave two 32-byte pieces a k-tile apart: twice the HBM write bytes on the counter).
__device__ __forceinline__ void pair_tile_decode(int m, int Lp, int& pi, int& pj) {
    const int KT = (Lp + 15) >> 4;
    const int blk = m >> 7, r = m & 127;
    const int ib = blk / KT, kt = blk - ib * KT;
    pi = ib * 8 + (r >> 4);
    pj = kt * 16 + (r & 15);
}

#define ABX_NEG_MAX (-3.4028234663852886e38f)  // torch.finfo(float32).min
