// MFMA issue-rate probe: cycles per instruction for the bf16 shapes (one wave per SIMD, 4 independent accumulators).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WHICH>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    bf16x8 a8, b8; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x4 c[4] = {}; f32x16 d[4] = {};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (WHICH == 0) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[u], 0, 0, 0);
            else if (WHICH == 1) c[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[u], 0, 0, 0);
            else d[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, d[u], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int u = 0; u < 4; ++u) { s += c[u][0] + d[u][0]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* o; long long* c; hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 20000;
    const char* names[3] = {"16x16x32_bf16", "16x16x16_bf16_1k", "32x32x16_bf16"};
    for (int w = 0; w < 3; ++w) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (w == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, o, c, iters);
            else if (w == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, o, c, iters);
            else hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, o, c, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            const double flop = (w == 2 ? 32768.0 : (w == 0 ? 16384.0 : 8192.0)) * 4.0 * iters * 4 * 1024;   // 4 waves/block (1 per SIMD)
            printf("%-18s %8.3f ms  %8.1f TFLOP/s  s_memtime ticks per mfma %.2f\n", names[w], ms, flop / ms / 1e9, (double)cy / (4.0 * iters));
        }
    }
    return 0;
}
