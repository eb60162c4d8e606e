// Trivial kernels for tools/probes/cosched_repro.py: launched through ctypes from a separately built shared object, exactly like
// libabx_hip's entry points, but with nothing of the product in them.
//   axpy:   y[i] = a * x[i] + y[i]                       (1 load pair, 1 store per thread)
//   chain:  y[i] = f(x[i], y[i]) with ~60 dependent fp32 ops, sqrtf and a division (the instruction mix of rigid_update)
#include <hip/hip_runtime.h>

__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a * x[i] + y[i];
}

__global__ void chain_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float q[4] = {y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]};
    const float u[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    for (int r = 0; r < 4; ++r) {
        const float a = q[0] - (q[1] * u[0] + q[2] * u[1] + q[3] * u[2]);
        const float b = q[1] + (q[0] * u[0] + q[2] * u[2] - q[3] * u[1]);
        const float c = q[2] + (q[0] * u[1] - q[1] * u[2] + q[3] * u[0]);
        const float d = q[3] + (q[0] * u[2] + q[1] * u[1] - q[2] * u[0]);
        const float nrm = sqrtf(fmaxf(a * a + b * b + c * c + d * d, 1e-12f));
        q[0] = a / nrm; q[1] = b / nrm; q[2] = c / nrm; q[3] = d / nrm;
    }
    for (int k = 0; k < 4; ++k) y[4 * i + k] = q[k];
}

extern "C" int probe_axpy(const float* x, float* y, float a, int n, hipStream_t st) {
    hipLaunchKernelGGL(axpy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, y, a, n);
    return (int)hipGetLastError();
}
extern "C" int probe_chain(const float* x, float* y, int n, hipStream_t st) {
    hipLaunchKernelGGL(chain_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, y, n);
    return (int)hipGetLastError();
}
