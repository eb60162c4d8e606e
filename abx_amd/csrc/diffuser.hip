// SE(3) x sequence reverse step and the IGSO(3) tables.
// Reference sites: diffuser/so3_diffuser.py:15-49 (igso3_expansion), :52-69 (density), :72-112 (score), :153-166 (table build),
// :207-216 (diffusion_coef), :328-361 (reverse) ; diffuser/r3_diffuser.py:27-40,110-148 ; diffuser/discrete_diffuser.py:53-67,130-190 ;
// diffuser/full_diffuser.py:12-26,174-227.
//
// Precision follows the reference's *effective* dtypes inside the sampling loop: t, every schedule
// scalar, the SO(3)/R^3 updates and the carried rigids are float64; scores, noise and logits are float32; the 0-dim fp32
// constants (exp(max_sigma), exp(min_sigma), min_b, max_b-min_b, coordinate scale, dt, sqrt(dt)) enter as fp32 values.
#include "common.h"
#include "abx_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// IGSO(3): one thread per (sigma_i, omega_j); terms in fp32 exactly as the reference forms them, accumulated in fp64.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void igso3_kernel(const float* __restrict__ sigma, const float* __restrict__ omega, int ns,
                                                    int no, int Lt, float* __restrict__ pdf, float* __restrict__ score_norms) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)ns * no) return;
    const int i = (int)(idx / no), j = (int)(idx % no);
    const float eps = sigma[i], om = omega[j];
    const float e2 = eps * eps;
    const float lo = sinf(om / 2.f);
    const float dlo = 0.5f * cosf(om / 2.f);
    const float lo2 = lo * lo;
    double acc = 0.0, dacc = 0.0;
    for (int l = 0; l < Lt; ++l) {
        const float fl = (float)l;
        const float w = (float)(2 * l + 1) * expf((float)(-(long long)l * (l + 1)) * e2 / 2.f);
        const float arg = om * (fl + 0.5f);
        const float hi = sinf(arg);
        const float dhi = (fl + 0.5f) * cosf(arg);
        acc += (double)(w * hi / lo);
        dacc += (double)(w * (lo * dhi - hi * dlo) / lo2);
    }
    const float ex = (float)acc;
    pdf[idx] = ex * (1.f - cosf(om)) / 3.14159274101257324f;
    score_norms[idx] = (float)dacc / (ex + 1e-4f);
}

// cdf = cumsum(pdf) / num_omega * pi, sequential fp32 scan per sigma row (so3_diffuser.py:160-161)
__global__ void igso3_cdf_kernel(const float* __restrict__ pdf, int ns, int no, float* __restrict__ cdf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    float run = 0.f;
    for (int j = 0; j < no; ++j) {
        run += pdf[(long long)i * no + j];
        cdf[(long long)i * no + j] = run / (float)no * 3.14159274101257324f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (device noise when none is injected)
// ------------------------------------------------------------------------------------------------------------------
// Counter layout: (sample id, residue, step, stream << 24 | draw).  The first three words identify WHOSE noise this is and are
// never touched by next(); successive draws only advance the low 24 bits of the fourth word, so no two (sample, residue, step,
// stream) tuples can ever share a block (a sample's noise does not depend on the batch it runs in or on its neighbours).
struct Philox {
    uint32_t c[4], k[2];
    __device__ Philox(unsigned long long seed, uint32_t sid, uint32_t res, uint32_t step, uint32_t stream) {
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        c[0] = sid; c[1] = res; c[2] = step; c[3] = stream << 24;
    }
    __device__ void next(uint32_t* out) {
        uint32_t x[4] = {c[0], c[1], c[2], c[3]}, kk[2] = {k[0], k[1]};
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * x[0], p1 = (unsigned long long)0xCD9E8D57u * x[2];
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x[1] ^ kk[0], y1 = (uint32_t)p1;
            const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x[3] ^ kk[1], y3 = (uint32_t)p0;
            x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3;
            kk[0] += 0x9E3779B9u; kk[1] += 0xBB67AE85u;
        }
        out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
        c[3] = (c[3] & 0xff000000u) | ((c[3] + 1u) & 0x00ffffffu);
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

// Poisson(lam) count as a pure function of one uniform u in (0, 1): the smallest k with u <= cdf(k), the pmf terms built by the
// fp32 recurrence p_k = p_{k-1} * (lam / k) from p_0 = (float)exp(-(double)lam) (a correctly rounded start: the same bits on the
// device and in the oracle's restatement, oracle/abx_oracle.py::poisson_icdf).  The walk stops when the fp32 cdf no longer grows
// past the mode (a u above the saturated sum would otherwise run to the cap and throw the token to 0 / 19) and at k = 64.
__device__ __forceinline__ int poisson_icdf(float lam, float u) {
    float pk = (float)exp(-(double)lam), cdf = pk;
    int kk = 0;
    while (u > cdf && kk < 64) {
        ++kk;
        pk *= lam / (float)kk;
        const float nc = cdf + pk;
        if (nc == cdf && (float)kk > lam) break;
        cdf = nc;
    }
    return kk;
}

template <typename T>
__device__ __forceinline__ void rotvec_to_quat(const T* v, T* q) {
    const T ang = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const T half = ang * (T)0.5;
    const T s = (fabs((double)ang) < 1e-6) ? (T)0.5 - (ang * ang) / (T)48 : (T)sin((double)half) / ang;
    q[0] = (T)cos((double)half); q[1] = v[0] * s; q[2] = v[1] * s; q[3] = v[2] * s;
}
__device__ __forceinline__ void rotvec_to_quat_f(const float* v, float* q) {
    const float ang = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float half = ang * 0.5f;
    const float s = (fabsf(ang) < 1e-6f) ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
    q[0] = cosf(half); q[1] = v[0] * s; q[2] = v[1] * s; q[3] = v[2] * s;
}
__device__ __forceinline__ void quat_to_rotvec_dd(const double* qi, double* v) {
    double q[4] = {qi[0], qi[1], qi[2], qi[3]};
    if (q[0] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double nr = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double half = atan2(nr, q[0]);
    const double ang = 2 * half;
    const double s = (fabs(ang) < 1e-6) ? 0.5 - (ang * ang) / 48 : sin(half) / ang;
    v[0] = q[1] / s; v[1] = q[2] / s; v[2] = q[3] / s;
}
__device__ __forceinline__ void quat_to_rotvec_ff(const float* qi, float* v) {
    float q[4] = {qi[0], qi[1], qi[2], qi[3]};
    if (q[0] < 0.f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const float nr = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float half = atan2f(nr, q[0]);
    const float ang = 2 * half;
    const float s = (fabsf(ang) < 1e-6f) ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
    v[0] = q[1] / s; v[1] = q[2] / s; v[2] = q[3] / s;
}

// one workgroup per sample (the R^3 update is centred over ALL residues of the sample, r3_diffuser.py:141-146)
__global__ __launch_bounds__(256) void reverse_step_kernel(const AbxReverseArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sh[];     // [L][3] un-centred x_{t-1} + [3*4] partials
    const int b = blockIdx.x, tid = threadIdx.x, L = a.L;
    const int step = a.step_dev ? *a.step_dev : a.step;
    double* x1s = sh;
    double* part = sh + (size_t)L * 3;                              // [4 waves][3]
    const double t = a.t[b];
    const float dtf = a.dt_dev ? *a.dt_dev : a.dt;                  // fp32 0-dim tensor value
    const double dt = (double)dtf;
    const double sqdt = (double)sqrtf(dtf);
    // so3 diffusion coefficient: sqrt(2 (e^max - e^min) sigma / e^sigma)
    const double sig = log(t * (double)a.exp_max_sigma + (1.0 - t) * (double)a.exp_min_sigma);
    const double g_so3 = sqrt((double)(2.f * (a.exp_max_sigma - a.exp_min_sigma)) * sig / exp(sig));
    // r3: b_t, g = sqrt(b_t)
    const double bt = (double)a.min_b + t * (double)a.bdiff;
    const double g_r3 = sqrt(bt);
    const double cs = (double)a.coord_scale;
    // token transition matrix q_t0 (closed form of V exp(lambda t) V^T for the uniform-rate generator), fp32
    const float tf = (float)t;
    const float ee = expf(-(20.f * a.rate_const) * tf);
    float q_same = ee + (1.f - ee) / 20.f, q_diff = (1.f - ee) / 20.f;
    if (q_same < 1e-8f) q_same = 0.f;
    if (q_diff < 1e-8f) q_diff = 0.f;

    double lsum[3] = {0.0, 0.0, 0.0};
    for (int l = tid; l < L; l += 256) {
        const long long i = (long long)b * L + l;
        // ---------------- inputs
        double q[4], tr[3], rot_t[3];
        if (a.rigid_is_f64) {
            const double* r = reinterpret_cast<const double*>(a.rigid_in) + i * 7;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = r[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) tr[k] = r[4 + k];
            quat_to_rotvec_dd(q, rot_t);
        } else {
            const float* r = reinterpret_cast<const float*>(a.rigid_in) + i * 7;
            float qf[4] = {r[0], r[1], r[2], r[3]}, rv[3];
            quat_to_rotvec_ff(qf, rv);
#pragma unroll
            for (int k = 0; k < 3; ++k) { rot_t[k] = (double)rv[k]; tr[k] = (double)r[4 + k]; }
        }
        // ---------------- noise
        float zr[3], zt[3];
        if (a.z_rot) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { zr[k] = a.z_rot[i * 3 + k]; zt[k] = a.z_trans[i * 3 + k]; }
        } else {
            const long long sid = a.sample_ids ? a.sample_ids[b] : b;
            Philox ph(a.seed, (uint32_t)sid, (uint32_t)l, (uint32_t)step, 0u);
            uint32_t r4[4], r4b[4];
            ph.next(r4); ph.next(r4b);
            const float r0 = sqrtf(-2.f * logf(u01(r4[0]))), th0 = 6.28318530717958647692f * u01(r4[1]);
            const float r1 = sqrtf(-2.f * logf(u01(r4[2]))), th1 = 6.28318530717958647692f * u01(r4[3]);
            const float r2 = sqrtf(-2.f * logf(u01(r4b[0]))), th2 = 6.28318530717958647692f * u01(r4b[1]);
            zr[0] = r0 * cosf(th0); zr[1] = r0 * sinf(th0); zr[2] = r1 * cosf(th1);
            zt[0] = r1 * sinf(th1); zt[1] = r2 * cosf(th2); zt[2] = r2 * sinf(th2);
        }
        // ---------------- SO(3): geodesic random walk
        double perturb[3], pq[4], qt[4], q1[4], rot1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float z = a.noise_scale * zr[k];
            perturb[k] = ((g_so3 * g_so3) * (double)a.rot_score[i * 3 + k]) * dt + (g_so3 * sqdt) * (double)z;
        }
        rotvec_to_quat<double>(perturb, pq);
        if (a.rigid_is_f64) {
            rotvec_to_quat<double>(rot_t, qt);
        } else {
            float rv[3] = {(float)rot_t[0], (float)rot_t[1], (float)rot_t[2]}, qf[4];
            rotvec_to_quat_f(rv, qf);
#pragma unroll
            for (int k = 0; k < 4; ++k) qt[k] = (double)qf[k];
        }
        q1[0] = qt[0] * pq[0] - qt[1] * pq[1] - qt[2] * pq[2] - qt[3] * pq[3];
        q1[1] = qt[0] * pq[1] + qt[1] * pq[0] + qt[2] * pq[3] - qt[3] * pq[2];
        q1[2] = qt[0] * pq[2] - qt[1] * pq[3] + qt[2] * pq[0] + qt[3] * pq[1];
        q1[3] = qt[0] * pq[3] + qt[1] * pq[2] - qt[2] * pq[1] + qt[3] * pq[0];
        quat_to_rotvec_dd(q1, rot1);
        // ---------------- R^3 VP-SDE step (noise scaled by dt, r3_diffuser.py:137)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double x = a.rigid_is_f64 ? tr[k] * cs : (double)((float)tr[k] * a.coord_scale);
            const double f = (-0.5 * bt) * x;
            const double sc = a.ts_is_f32 ? (double)reinterpret_cast<const float*>(a.trans_score)[i * 3 + k]
                                          : reinterpret_cast<const double*>(a.trans_score)[i * 3 + k];
            const float z = a.noise_scale * zt[k];
            const double pt = (f - (g_r3 * g_r3) * sc) * dt + (g_r3 * dt) * (double)z;
            const double x1 = x - pt;
            x1s[(size_t)l * 3 + k] = x1;
            lsum[k] += x1;
        }
        // ---------------- tokens: tau-leaping with Poisson jump counts (discrete_diffuser.py:150-188)
        const float* lg = a.logits + i * 20;
        long long xt = a.seq_in[i];
        xt = xt < 0 ? 0 : (xt > 19 ? 19 : xt);
        float p[20], mx = lg[0];
#pragma unroll
        for (int s = 1; s < 20; ++s) mx = fmaxf(mx, lg[s]);
        float sm = 0.f;
#pragma unroll
        for (int s = 0; s < 20; ++s) { p[s] = expf(lg[s] - mx); sm += p[s]; }
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            const float ps = p[s] / sm;
            const float den = ((long long)s == xt ? q_same : q_diff) + 1e-9f;
            const float r = ps / den;
            p[s] = r;
        }
        float overall = 0.f;
        Philox ph2(a.seed ^ 0x5851F42D4C957F2Dull, (uint32_t)(a.sample_ids ? a.sample_ids[b] : b), (uint32_t)l, (uint32_t)step, 1u);
#pragma unroll 1
        for (int s2 = 0; s2 < 20; ++s2) {
            // inner[s2] = sum_s ratio[s] * q_t0[s][s2] = q_diff * sum_s ratio[s] + (q_same - q_diff) * ratio[s2]
            float inner = 0.f;
#pragma unroll
            for (int s = 0; s < 20; ++s) inner += p[s] * (s == s2 ? q_same : q_diff);
            float rate = ((long long)s2 == xt) ? 0.f : a.rate_const * inner;
            const float lam = rate * dtf;
            if (a.rates_out) a.rates_out[i * 20 + s2] = lam;
            float jumps;
            if (a.jumps) {
                jumps = a.jumps[i * 20 + s2];
            } else {
                float u;
                if (a.u_jumps) {
                    u = a.u_jumps[i * 20 + s2];
                } else {
                    uint32_t r4[4];
                    ph2.next(r4);
                    u = u01(r4[0]);
                }
                jumps = (float)poisson_icdf(lam, u);
            }
            if (a.jumps_out) a.jumps_out[i * 20 + s2] = jumps;
            overall += jumps * (float)((long long)s2 - xt);
        }
        float xp = (float)xt + overall;
        xp = fminf(fmaxf(xp, 0.f), 19.f);
        const int xnew = (int)xp;
        const long long m = a.diffuse_mask[i];
        a.seq_out[i] = m * (long long)xnew + (1 - m) * a.seq_in[i];
        // rotation merge + quaternion now; translation after the centring reduction
        double rm[3], qo[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) rm[k] = (double)m * rot1[k] + (double)(1 - m) * rot_t[k];
        rotvec_to_quat<double>(rm, qo);
#pragma unroll
        for (int k = 0; k < 4; ++k) a.rigid_out[i * 7 + k] = qo[k];
    }
    // ---------------- centre of mass over all residues of the sample
#pragma unroll
    for (int k = 0; k < 3; ++k) lsum[k] = wave_sum_d(lsum[k]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) part[(tid >> 6) * 3 + k] = lsum[k];
    }
    __syncthreads();
    double com[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) com[k] = (part[k] + part[3 + k] + part[6 + k] + part[9 + k]) / (double)(float)L;
    for (int l = tid; l < L; l += 256) {
        const long long i = (long long)b * L + l;
        const long long m = a.diffuse_mask[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double x1 = x1s[(size_t)l * 3 + k];
            if (a.center) x1 -= com[k];
            x1 = x1 / cs;
            const double told = a.rigid_is_f64 ? reinterpret_cast<const double*>(a.rigid_in)[i * 7 + 4 + k]
                                               : (double)reinterpret_cast<const float*>(a.rigid_in)[i * 7 + 4 + k];
            a.rigid_out[i * 7 + 4 + k] = (double)m * x1 + (double)(1 - m) * told;
        }
    }
}

}  // namespace

extern "C" int abx_igso3_tables(const float* sigma, const float* omega, int num_sigma, int num_omega, int L_terms, float* pdf,
                                float* cdf, float* score_norms, hipStream_t st) {
    ABX_REQUIRE(sigma && omega && pdf && cdf && score_norms && num_sigma > 0 && num_omega > 0 && L_terms > 0,
                "abx_igso3_tables: bad args");
    const long long n = (long long)num_sigma * num_omega;
    hipLaunchKernelGGL(igso3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sigma, omega, num_sigma, num_omega, L_terms,
                       pdf, score_norms);
    int rc = abx_check_launch("abx_igso3_tables");
    if (rc) return rc;
    hipLaunchKernelGGL(igso3_cdf_kernel, dim3((num_sigma + 63) / 64), dim3(64), 0, st, pdf, num_sigma, num_omega, cdf);
    return abx_check_launch("abx_igso3_tables(cdf)");
}

extern "C" int abx_reverse_step(const AbxReverseArgs* ap, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_reverse_step: null");
    const AbxReverseArgs a = *ap;
    ABX_REQUIRE(a.rigid_in && a.seq_in && a.rot_score && a.trans_score && a.logits && a.diffuse_mask && a.t && a.rigid_out &&
                    a.seq_out, "abx_reverse_step: null operand");
    ABX_REQUIRE((a.z_rot == nullptr) == (a.z_trans == nullptr), "abx_reverse_step: z_rot and z_trans go together");
    ABX_REQUIRE(a.B > 0 && a.L > 0, "abx_reverse_step: bad sizes");
    const size_t lds = ((size_t)a.L * 3 + 12) * sizeof(double);
    ABX_REQUIRE(lds <= 64 * 1024, "abx_reverse_step: L too large");
    hipLaunchKernelGGL(reverse_step_kernel, dim3(a.B), dim3(256), lds, st, a);
    return abx_check_launch("abx_reverse_step");
}
