// Rigid-frame algebra, score functions, torsion -> frames -> atoms, self-conditioning distogram and small head tails.
// Reference sites: abx/model/quat_affine.py:53-150,234-238 ; abx/model/r3.py:9-59 ; abx/model/score_network.py:100-194 ;
// diffuser/full_diffuser.py:131-142 ; diffuser/so3_diffuser.py:189-205,264-297 ; diffuser/r3_diffuser.py:45-46,150-164 ;
// abx/model/sidechain.py:64-72 ; abx/model/atom.py:9-76 ; abx/model/head.py:162-199 ; abx/model/abx.py:17-26 ;
// abx/model/common_modules.py:62-83,107-120 ; abx/model/utils.py:158-171.
// All of these are per-residue, latency-class kernels (B*L threads); they exist so that a diffusion step never leaves
// the device and is capturable in a hipGraph.
#include "common.h"
#include "abx_hip.h"
#include "rigid_dev.h"

namespace {

template <typename T>
__device__ __forceinline__ void quat_mul_d(const T* p, const T* q, T* o) {
    o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
    o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
    o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
    o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}

// quat_affine.py:113-131 (fp32: the score network works on float32 quaternions)
__device__ __forceinline__ void quat_to_rotvec_f(const float* qi, float* v) {
    float q[4] = {qi[0], qi[1], qi[2], qi[3]};
    if (q[0] < 0.f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const float nr = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float half = atan2f(nr, q[0]);
    const float ang = 2 * half;
    const float s = (fabsf(ang) < 1e-6f) ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
    v[0] = q[1] / s; v[1] = q[2] / s; v[2] = q[3] / s;
}

__global__ __launch_bounds__(256) void frames_init_kernel(const void* __restrict__ rigids, int is_f64, float* __restrict__ init_q,
                                                          float* __restrict__ init_t, float* __restrict__ cur_q,
                                                          float* __restrict__ cur_t, float* __restrict__ cur_R,
                                                          float* __restrict__ delta_q, int n, float pscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
        r[k] = is_f64 ? (float)reinterpret_cast<const double*>(rigids)[(long long)i * 7 + k]
                      : reinterpret_cast<const float*>(rigids)[(long long)i * 7 + k];
    float R[9];
    quat_to_rot_d<float>(r, R);
#pragma unroll
    for (int k = 0; k < 4; ++k) { init_q[i * 4 + k] = r[k]; cur_q[i * 4 + k] = r[k]; delta_q[i * 4 + k] = k == 0 ? 1.f : 0.f; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { init_t[i * 3 + k] = r[4 + k]; cur_t[i * 3 + k] = r[4 + k] / pscale; }
#pragma unroll
    for (int k = 0; k < 9; ++k) cur_R[(long long)i * 9 + k] = R[k];
}

__global__ __launch_bounds__(256) void rigid_update_kernel(const float* __restrict__ upd, const int* __restrict__ fixed,
                                                           const float* __restrict__ init_q, const float* __restrict__ init_t,
                                                           float* __restrict__ cur_q, float* __restrict__ cur_t,
                                                           float* __restrict__ cur_R, float* __restrict__ delta_q, int n,
                                                           float pscale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = upd[(long long)i * 6 + k];
    rigid_update_row(i, u, fixed, init_q, init_t, cur_q, cur_t, cur_R, delta_q, pscale);
}

// sigma(t) and its row index in the discretised table (so3_diffuser.py:189-205), in T = double (loop) or float (warm-up)
template <typename T>
__device__ __forceinline__ int sigma_index(T t, float e_max, float e_min, const float* __restrict__ dsig, int ns) {
    T thr;
    if (sizeof(T) == 4) {
        const float sgf = logf((float)t * e_max + (1.f - (float)t) * e_min);
        thr = (T)(sgf + 1e-5f);
    } else {
        thr = (T)(log((double)t * (double)e_max + (1.0 - (double)t) * (double)e_min) + 1e-5);
    }
    int cnt = 0;
    for (int k = 0; k < ns; ++k) cnt += ((T)dsig[k] <= thr) ? 1 : 0;
    return cnt - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void scores_kernel(const AbxScoreArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = a.B * a.L;
    if (i >= n) return;
    const int b = i / a.L;
    float iq[4], dq[4], qf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { iq[k] = a.init_q[i * 4 + k]; dq[k] = a.delta_q[i * 4 + k]; }
    quat_mul_d<float>(iq, dq, qf);
    const float dm = (float)(1 - a.fixed_mask[i]);
#pragma unroll
    for (int k = 0; k < 4; ++k) qf[k] = dm * qf[k] + (1.f - dm) * iq[k];
    // rot score: rotvec(q0^-1 (x) q_t), q0 = prediction qf, q_t = init (full_diffuser.py:135-142)
    const float nr = sqrtf(qf[0] * qf[0] + qf[1] * qf[1] + qf[2] * qf[2] + qf[3] * qf[3]);
    const float inv[4] = {qf[0] / nr, -qf[1] / nr, -qf[2] / nr, -qf[3] / nr};
    float q0t[4], v[3];
    quat_mul_d<float>(inv, iq, q0t);
    quat_to_rotvec_f(q0t, v);
    const float omega = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-6f;
    const T t = (T)a.t[b];
    const int row = min(max(sigma_index<T>(t, a.exp_max_sigma, a.exp_min_sigma, a.discrete_sigma, a.num_sigma), 0), a.num_sigma - 1);
    // torch.bucketize(omega, discrete_omega[:-1]): number of boundaries strictly below omega (binary search)
    int lo = 0, hi = a.num_omega - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.discrete_omega[mid] < omega) lo = mid + 1; else hi = mid;
    }
    const float sn = a.score_norms[(long long)row * a.num_omega + lo];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.rot_score[(long long)i * 3 + k] = sn * v[k] / (omega + 1e-6f);
    // trans score (r3_diffuser.py:158-164, scale=True)
    const float pscale = a.position_scale;
    const T mb = t * (T)a.min_b + ((T)0.5 * (t * t)) * (T)a.bdiff;
    const T e = (T)exp((double)((T)(-0.5) * mb));
    const T cv = (T)1 - (T)exp((double)(-mb));
    T ef = e, cvf = cv;
    if (sizeof(T) == 4) {
        ef = (T)expf(-0.5f * (float)mb);
        cvf = (T)(1.f - expf(-(float)mb));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ct = a.cur_t[i * 3 + k] * pscale;
        const float xt = a.init_t[i * 3 + k] * a.coord_scale;
        const float x0 = ct * a.coord_scale;
        const T sc = -((T)xt - ef * (T)x0) / cvf;
        if (sizeof(T) == 4) reinterpret_cast<float*>(a.trans_score)[(long long)i * 3 + k] = (float)sc;
        else reinterpret_cast<double*>(a.trans_score)[(long long)i * 3 + k] = (double)sc;
        a.rigids[(long long)i * 7 + 4 + k] = ct;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.rigids[(long long)i * 7 + k] = qf[k];
}

__global__ __launch_bounds__(256) void torsion_finalize_kernel(const float* __restrict__ un, const float* __restrict__ gt,
                                                               const int* __restrict__ fixed, float* __restrict__ ang, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over n*7 (residue, torsion)
    if (i >= n * 7) return;
    const int r = i / 7;
    const float s = un[i * 2], c = un[i * 2 + 1];
    const float nr = sqrtf(s * s + c * c + 1e-12f);
    const bool fx = fixed[r] != 0;
    ang[i * 2] = fx ? gt[i * 2] : s / nr;
    ang[i * 2 + 1] = fx ? gt[i * 2 + 1] : c / nr;
}

__device__ __forceinline__ void compose(const float* Ra, const float* ta, const float* Rb, const float* tb, float* Ro, float* to) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int m = 0; m < 3; ++m) Ro[r * 3 + m] = Ra[r * 3] * Rb[m] + Ra[r * 3 + 1] * Rb[3 + m] + Ra[r * 3 + 2] * Rb[6 + m];
        to[r] = (Ra[r * 3] * tb[0] + Ra[r * 3 + 1] * tb[1] + Ra[r * 3 + 2] * tb[2]) + ta[r];
    }
}

__global__ __launch_bounds__(128) void seq_head_atoms_kernel(const float* __restrict__ logits, const int* __restrict__ fixed,
                                                             const long long* __restrict__ seq_t, const float* __restrict__ rigids,
                                                             const float* __restrict__ angles,
                                                             const long long* __restrict__ a37to14,
                                                             const float* __restrict__ dframes, const int* __restrict__ gidx,
                                                             const float* __restrict__ lit, long long* __restrict__ seq_0,
                                                             float* __restrict__ atom14, float* __restrict__ atom37, int n) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    // argmax over 20 logits, first maximum wins (torch.max on softmax probabilities, head.py:166-167)
    int best = 0;
    float bv = logits[(long long)i * 20];
#pragma unroll
    for (int k = 1; k < 20; ++k) {
        const float v = logits[(long long)i * 20 + k];
        if (v > bv) { bv = v; best = k; }
    }
    const long long fx = fixed[i];
    const long long aa = (long long)best * (1 - fx) + seq_t[i] * fx;
    seq_0[i] = aa;
    // backbone frame from the final quaternion
    float q[4], Rb[9], tb[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = rigids[(long long)i * 7 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tb[k] = rigids[(long long)i * 7 + 4 + k];
    quat_to_rot_d<float>(q, Rb);
    // 8 rigid groups: default frame * Rx(torsion); chi2..4 chained; then backbone (atom.py:14-56)
    float fR[8][9], ft[8][3];
    const float* df = dframes + aa * 8 * 16;
#pragma unroll
    for (int gI = 0; gI < 8; ++gI) {
        const float sn = gI == 0 ? 0.f : angles[((long long)i * 7 + gI - 1) * 2];
        const float cs = gI == 0 ? 1.f : angles[((long long)i * 7 + gI - 1) * 2 + 1];
        const float* m = df + gI * 16;
        // R_default * [[1,0,0],[0,c,-s],[0,s,c]]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float d0 = m[r * 4], d1 = m[r * 4 + 1], d2 = m[r * 4 + 2];
            fR[gI][r * 3 + 0] = d0 * 1.f + d1 * 0.f + d2 * 0.f;
            fR[gI][r * 3 + 1] = d0 * 0.f + d1 * cs + d2 * sn;
            fR[gI][r * 3 + 2] = d0 * 0.f + d1 * (-sn) + d2 * cs;
            ft[gI][r] = m[r * 4 + 3];
        }
    }
    float R5[9], t5[3], R6[9], t6[3], R7[9], t7[3];
    compose(fR[4], ft[4], fR[5], ft[5], R5, t5);
    compose(R5, t5, fR[6], ft[6], R6, t6);
    compose(R6, t6, fR[7], ft[7], R7, t7);
#pragma unroll
    for (int k = 0; k < 9; ++k) { fR[5][k] = R5[k]; fR[6][k] = R6[k]; fR[7][k] = R7[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { ft[5][k] = t5[k]; ft[6][k] = t6[k]; ft[7][k] = t7[k]; }
    float gR[8][9], gt[8][3];
#pragma unroll
    for (int gI = 0; gI < 8; ++gI) compose(Rb, tb, fR[gI], ft[gI], gR[gI], gt[gI]);
    float pos[14][3];
#pragma unroll
    for (int aI = 0; aI < 14; ++aI) {
        const int gg = gidx[aa * 14 + aI];
        const float* lp = lit + (aa * 14 + aI) * 3;
        float R[9], t[3];
        // dynamic group select without runtime-indexed private arrays
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v = gR[0][k];
#pragma unroll
            for (int s = 1; s < 8; ++s) v = gg == s ? gR[s][k] : v;
            R[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = gt[0][k];
#pragma unroll
            for (int s = 1; s < 8; ++s) v = gg == s ? gt[s][k] : v;
            t[k] = v;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            pos[aI][r] = t[r] + (R[r * 3] * lp[0] + R[r * 3 + 1] * lp[1] + R[r * 3 + 2] * lp[2]);
            atom14[((long long)i * 14 + aI) * 3 + r] = pos[aI][r];
        }
    }
    for (int k = 0; k < 37; ++k) {
        const int src = (int)a37to14[(long long)i * 37 + k];
#pragma unroll
        for (int r = 0; r < 3; ++r) atom37[((long long)i * 37 + k) * 3 + r] = atom14[((long long)i * 14 + src) * 3 + r];
    }
}

__global__ __launch_bounds__(256) void prev_pos_kernel(const float* __restrict__ atom37, const float* __restrict__ sq_breaks,
                                                       int nb, long long* __restrict__ out, int B, int L) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * L * L) return;
    const int j = (int)(idx % L);
    const long long bi = idx / L;
    const int b = (int)(bi / L);
    const long long bj = (long long)b * L + j;
    const float* pi = atom37 + bi * 111;
    const float* pj = atom37 + bj * 111;
    float cbi[3], cbj[3];
    {
        const float bx = pi[3] - pi[0], by = pi[4] - pi[1], bz = pi[5] - pi[2];
        const float cx = pi[6] - pi[3], cy = pi[7] - pi[4], cz = pi[8] - pi[5];
        const float ax = by * cz - bz * cy, ay = bz * cx - bx * cz, az = bx * cy - by * cx;
        cbi[0] = -0.58273431f * ax + 0.56802827f * bx - 0.54067466f * cx + pi[3];
        cbi[1] = -0.58273431f * ay + 0.56802827f * by - 0.54067466f * cy + pi[4];
        cbi[2] = -0.58273431f * az + 0.56802827f * bz - 0.54067466f * cz + pi[5];
    }
    {
        const float bx = pj[3] - pj[0], by = pj[4] - pj[1], bz = pj[5] - pj[2];
        const float cx = pj[6] - pj[3], cy = pj[7] - pj[4], cz = pj[8] - pj[5];
        const float ax = by * cz - bz * cy, ay = bz * cx - bx * cz, az = bx * cy - by * cx;
        cbj[0] = -0.58273431f * ax + 0.56802827f * bx - 0.54067466f * cx + pj[3];
        cbj[1] = -0.58273431f * ay + 0.56802827f * by - 0.54067466f * cy + pj[4];
        cbj[2] = -0.58273431f * az + 0.56802827f * bz - 0.54067466f * cz + pj[5];
    }
    const float dx = cbi[0] - cbj[0], dy = cbi[1] - cbj[1], dz = cbi[2] - cbj[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    int bin = 0;
    for (int k = 0; k < nb; ++k) bin += d2 > sq_breaks[k] ? 1 : 0;
    out[idx] = bin;
}

__global__ __launch_bounds__(256) void plddt_kernel(const float* __restrict__ logits, float* __restrict__ out, int n, int bins) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = logits + (long long)i * bins;
    float mx = p[0];
    for (int k = 1; k < bins; ++k) mx = fmaxf(mx, p[k]);
    float sm = 0.f, ex = 0.f;
    const float w = 1.0f / (float)bins;
    for (int k = 0; k < bins; ++k) {
        const float e = expf(p[k] - mx);
        sm += e;
        ex += e * (0.5f * w + (float)k * w);
    }
    out[i] = ex / sm * 100.f;
}

}  // namespace

extern "C" int abx_frames_init(const void* rigids_t, int is_f64, float* init_q, float* init_t, float* cur_q, float* cur_t,
                               float* cur_R, float* delta_q, int n, float position_scale, hipStream_t st) {
    ABX_REQUIRE(rigids_t && init_q && init_t && cur_q && cur_t && cur_R && delta_q && n > 0, "abx_frames_init: bad args");
    hipLaunchKernelGGL(frames_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rigids_t, is_f64, init_q, init_t, cur_q, cur_t,
                       cur_R, delta_q, n, position_scale);
    return abx_check_launch("abx_frames_init");
}

extern "C" int abx_rigid_update(const float* upd6, const int* fixed_mask, const float* init_q, const float* init_t, float* cur_q,
                                float* cur_t, float* cur_R, float* delta_q, int n, float position_scale, hipStream_t st) {
    ABX_REQUIRE(upd6 && fixed_mask && init_q && init_t && cur_q && cur_t && cur_R && delta_q && n > 0, "abx_rigid_update: bad args");
    hipLaunchKernelGGL(rigid_update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, upd6, fixed_mask, init_q, init_t, cur_q, cur_t,
                       cur_R, delta_q, n, position_scale);
    return abx_check_launch("abx_rigid_update");
}

extern "C" int abx_scores(const AbxScoreArgs* ap, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_scores: null");
    const AbxScoreArgs a = *ap;
    ABX_REQUIRE(a.init_q && a.init_t && a.delta_q && a.cur_t && a.fixed_mask && a.t && a.score_norms && a.discrete_sigma &&
                    a.discrete_omega && a.rot_score && a.trans_score && a.rigids, "abx_scores: null operand");
    ABX_REQUIRE(a.B > 0 && a.L > 0 && a.num_sigma > 0 && a.num_omega > 1, "abx_scores: bad sizes");
    const int n = a.B * a.L;
    if (a.t_is_f32) hipLaunchKernelGGL((scores_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((scores_kernel<double>), dim3((n + 255) / 256), dim3(256), 0, st, a);
    return abx_check_launch("abx_scores");
}

extern "C" int abx_torsion_finalize(const float* unnorm, const float* gt_sincos, const int* fixed_mask, float* angles, int n,
                                    hipStream_t st) {
    ABX_REQUIRE(unnorm && gt_sincos && fixed_mask && angles && n > 0, "abx_torsion_finalize: bad args");
    hipLaunchKernelGGL(torsion_finalize_kernel, dim3((n * 7 + 255) / 256), dim3(256), 0, st, unnorm, gt_sincos, fixed_mask, angles, n);
    return abx_check_launch("abx_torsion_finalize");
}

extern "C" int abx_seq_head_atoms(const float* logits, const int* fixed_mask, const long long* seq_t, const float* rigids,
                                  const float* angles, const long long* atom37_to_atom14, const float* default_frames,
                                  const int* group_idx, const float* lit_pos, long long* seq_0, float* atom14, float* atom37, int n,
                                  hipStream_t st) {
    ABX_REQUIRE(logits && fixed_mask && seq_t && rigids && angles && atom37_to_atom14 && default_frames && group_idx && lit_pos &&
                    seq_0 && atom14 && atom37 && n > 0, "abx_seq_head_atoms: bad args");
    hipLaunchKernelGGL(seq_head_atoms_kernel, dim3((n + 127) / 128), dim3(128), 0, st, logits, fixed_mask, seq_t, rigids, angles,
                       atom37_to_atom14, default_frames, group_idx, lit_pos, seq_0, atom14, atom37, n);
    return abx_check_launch("abx_seq_head_atoms");
}

extern "C" int abx_prev_pos(const float* atom37, const float* sq_breaks, int num_breaks, long long* prev_pos, int B, int L,
                            hipStream_t st) {
    ABX_REQUIRE(atom37 && sq_breaks && prev_pos && B > 0 && L > 0 && num_breaks > 0, "abx_prev_pos: bad args");
    const long long total = (long long)B * L * L;
    hipLaunchKernelGGL(prev_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, atom37, sq_breaks, num_breaks,
                       prev_pos, B, L);
    return abx_check_launch("abx_prev_pos");
}

extern "C" int abx_plddt(const float* logits, float* out, int n, int bins, hipStream_t st) {
    ABX_REQUIRE(logits && out && n > 0 && bins > 0, "abx_plddt: bad args");
    hipLaunchKernelGGL(plddt_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logits, out, n, bins);
    return abx_check_launch("abx_plddt");
}
