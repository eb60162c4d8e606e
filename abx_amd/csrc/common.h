// Shared device/host helpers for libabx_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ABX_OK 0
#define ABX_ERR_ARG (-1)

// thread-local last error text (abx_last_error_string)
void abx_set_error(const char* msg);
int abx_check_launch(const char* what);

#define ABX_REQUIRE(cond, msg)            \
    do {                                  \
        if (!(cond)) {                    \
            abx_set_error(msg);           \
            return ABX_ERR_ARG;           \
        }                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

#define ABX_NEG_MAX (-3.4028234663852886e38f)  // torch.finfo(float32).min
