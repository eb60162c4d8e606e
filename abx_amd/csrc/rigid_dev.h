// Per-residue frame update of the IPA loop (score_network.py:138-146), shared by rigid_update_kernel (geometry.hip) and the fused
// IPA layer tail (gemm3.hip): device-inline, one thread per residue.
#pragma once
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void quat_to_rot_d(const T* q, T* R) {
    const T a = q[0], b = q[1], c = q[2], d = q[3];
    R[0] = a * a + b * b - c * c - d * d; R[1] = 2 * (b * c - a * d);           R[2] = 2 * (b * d + a * c);
    R[3] = 2 * (b * c + a * d);           R[4] = a * a - b * b + c * c - d * d; R[5] = 2 * (c * d - a * b);
    R[6] = 2 * (b * d - a * c);           R[7] = 2 * (c * d + a * b);           R[8] = a * a - b * b - c * c + d * d;
}

// normalize(q + q (x) (0,v)), eps 1e-12 (quat_affine.py:77-85, utils.py:12-14)
__device__ __forceinline__ void quat_precompose_vec_d(float* q, const float* v) {
    const float a = q[0], b = q[1], c = q[2], d = q[3];
    const float n0 = a + (-b * v[0] - c * v[1] - d * v[2]);
    const float n1 = b + (a * v[0] + c * v[2] - d * v[1]);
    const float n2 = c + (a * v[1] - b * v[2] + d * v[0]);
    const float n3 = d + (a * v[2] + b * v[1] - c * v[0]);
    const float nr = sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3 + 1e-12f);
    q[0] = n0 / nr; q[1] = n1 / nr; q[2] = n2 / nr; q[3] = n3 / nr;
}

// residue i: quaternion / translation update u[6] = (rotation vector part, translation in the OLD frame), fixed residues keep their
// initial frame (quat_affine.py:77-85 pre_compose; score_network.py:140-146)
__device__ __forceinline__ void rigid_update_row(int i, const float* u, const int* __restrict__ fixed, const float* __restrict__ init_q,
                                                 const float* __restrict__ init_t, float* __restrict__ cur_q, float* __restrict__ cur_t,
                                                 float* __restrict__ cur_R, float* __restrict__ delta_q, float pscale) {
    float q[4], dq[4], t[3], R[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k] = cur_q[i * 4 + k]; dq[k] = delta_q[i * 4 + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = cur_t[i * 3 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = cur_R[(long long)i * 9 + k];
    quat_precompose_vec_d(dq, u);
    quat_precompose_vec_d(q, u);
    // translation update with the OLD rotation (score_network.py:140): t + R u_t
    float tn[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tn[r] = t[r] + (R[r * 3] * u[3] + R[r * 3 + 1] * u[4] + R[r * 3 + 2] * u[5]);
    const float dm = (float)(1 - fixed[i]);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = dm * q[k] + (1.f - dm) * init_q[i * 4 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tn[k] = dm * tn[k] + (1.f - dm) * (init_t[i * 3 + k] / pscale);
    quat_to_rot_d<float>(q, R);
#pragma unroll
    for (int k = 0; k < 4; ++k) { cur_q[i * 4 + k] = q[k]; delta_q[i * 4 + k] = dq[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) cur_t[i * 3 + k] = tn[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) cur_R[(long long)i * 9 + k] = R[k];
}

}  // namespace
