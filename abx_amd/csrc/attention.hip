// Pair-biased softmax attention kernels of the Seqformer block.
//
// abx_tri_attn_fwd — triangle attention (reference abx/model/seqformer.py:506-550, Attention.forward :272-312) as a
// flash-style fused kernel: the (B, L, 4, L, L) logits tensor (268 MB / sample at L = 256) is never materialised.
// One workgroup per (b, row s, head h): K [L][48] and V [L][48] of that row are staged ONCE in LDS (padded strides 50 / 52
// floats, conflict-free for the MFMA operand reads below); each of the 4 waves walks query tiles of 32 rows with an
// online softmax over 64-key tiles.  Both contractions run on v_mfma_f32_16x16x4_f32 (exact fp32):
//     S^T[key][q]  = sum_d K[key][d] * Q[q][d]        ("swapped" QK^T: a lane owns 4 keys of ONE query column, so the
//                                                      softmax row reductions are in-lane + two cross-group shuffles)
//     O^T[d][q]   += sum_key V[key][d] * P[key][q]    (P fragments feed the B operand straight from registers)
// The orientation (starting / ending node) is expressed only through strides; no transposed copy of the pair stack.
//
// abx_seq_attn_fwd — sequence attention with 32-head pair bias (seqformer.py:314-356, split_first=False :278-281);
// 0.8 % of the step, one thread per query with K/V of the (b, h) pair in LDS.
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int TD = 48;        // head dim of triangle attention
constexpr int LDK = 50;       // K row stride in LDS (floats): (key*50 + d) mod 32 distinct over 16 keys x 2 d
constexpr int LDV = 52;       // V row stride: 4*52 mod 32 == 16 -> lane groups g land on disjoint bank halves

__global__ __launch_bounds__(256) void tri_attn_kernel(const AbxTriAttn a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L;
    float* Ks = smem;
    float* Vs = smem + (((size_t)L * LDK + 3) & ~(size_t)3);      // keep V rows 16-byte aligned
    const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;

    const long long base = (long long)b * a.sb + (long long)s * a.ss + (long long)h * TD;
    // ---- stage K, V of this (b, s, h) in LDS --------------------------------------------------------------
    for (int idx = tid; idx < L * (TD / 4); idx += 256) {
        const int key = idx / (TD / 4), c4 = idx % (TD / 4);
        const long long off = base + (long long)key * a.sl + c4 * 4;
        const f32x4 kv = *reinterpret_cast<const f32x4*>(a.k + off);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(a.v + off);
        float* kd = Ks + key * LDK + c4 * 4;
        kd[0] = kv[0]; kd[1] = kv[1]; kd[2] = kv[2]; kd[3] = kv[3];
        *reinterpret_cast<f32x4*>(Vs + key * LDV + c4 * 4) = vv;
    }
    __syncthreads();

    const float* biasb = a.bias ? a.bias + (long long)b * a.bias_sb + (long long)h * a.bias_sh : nullptr;
    const float* km = a.keymask ? a.keymask + (long long)b * a.km_sb : nullptr;
    const int nqt = (L + 31) / 32, nkt = (L + 63) / 64;

    for (int qt = wave; qt < nqt; qt += 4) {
        // ---- Q fragments (B operand of the swapped product): lane holds Q[q][kd*4 + g], pre-scaled
        float qf[2][12];
        int qrow[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            qrow[u] = qt * 32 + u * 16 + lq;
            const bool ok = qrow[u] < L;
            const float* qp = a.q + base + (long long)(ok ? qrow[u] : 0) * a.sl + g;
#pragma unroll
            for (int kd = 0; kd < 12; ++kd) qf[u][kd] = ok ? qp[kd * 4] * a.scale : 0.f;
        }
        float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
        f32x4 o[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int d = 0; d < 3; ++d) o[u][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nkt; ++kt) {
            f32x4 sc[2][4];
            // ---- S^T tiles: 4 sub-blocks of 16 keys
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int kb = kt * 64 + sub * 16;
                const int krow = min(kb + lq, L - 1);
                const float* kp = Ks + krow * LDK + g;
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kd = 0; kd < 12; ++kd) {
                    const float kf = kp[kd * 4];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[0][kd], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[1][kd], c1, 0, 0, 0);
                }
                // this lane: keys kb + g*4 + r (r = 0..3), query column lq of sub-tile u
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb + g * 4 + r;
                    const bool kin = key < L;
                    const bool kok = kin && (!km || km[key] != 0.f);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float v = u == 0 ? c0[r] : c1[r];
                        if (biasb && kin && qrow[u] < L)
                            v += biasb[(long long)qrow[u] * a.bias_sq + (long long)key * a.bias_sk];
                        v = kin ? (kok ? v : ABX_NEG_MAX) : -INFINITY;
                        if (u == 0) c0[r] = v; else c1[r] = v;
                    }
                }
                sc[0][sub] = c0;
                sc[1][sub] = c1;
            }
            // ---- online softmax update per query column
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float mx = -INFINITY;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[u][sub][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[u], mx);
                const float alpha = expf(m_run[u] - m_new);     // m_run = -inf on the first tile -> 0
                float rs = 0.f;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = expf(sc[u][sub][r] - m_new);
                        sc[u][sub][r] = p;
                        rs += p;
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run[u] = l_run[u] * alpha + rs;
                m_run[u] = m_new;
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[u][d][r] *= alpha;
            }
            // ---- O^T += V^T P : MFMA step (sub, r) contracts keys {kb + g*4 + r : g = 0..3}
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int kb = kt * 64 + sub * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = min(kb + g * 4 + r, L - 1);    // p == 0 for keys >= L
                    const float* vp = Vs + key * LDV + lq;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const float vf = vp[d * 16];
                        o[0][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, sc[0][sub][r], o[0][d], 0, 0, 0);
                        o[1][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, sc[1][sub][r], o[1][d], 0, 0, 0);
                    }
                }
            }
        }
        // ---- normalise, gate, store.  O^T layout: column = query lq, rows d = dblk*16 + g*4 + r
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (qrow[u] >= L) continue;
            const float inv = 1.0f / l_run[u];
            const long long go = base + (long long)qrow[u] * a.sl;
            float* op = a.out + (long long)b * a.ob + (long long)s * a.os + (long long)qrow[u] * a.ol + h * TD;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int dd = d * 16 + g * 4;
                f32x4 v = o[u][d];
                if (a.gate) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + go + dd);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * inv * (1.0f / (1.0f + expf(-gv[r])));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= inv;
                }
                *reinterpret_cast<f32x4*>(op + dd) = v;
            }
        }
    }
}

// ---- sequence attention: block = (256 queries, h, b); K/V of (b,h) in LDS; one thread per query ----------------
template <int D>
__global__ __launch_bounds__(256) void seq_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                       const float* __restrict__ keymask, const float* __restrict__ gate,
                                                       float* __restrict__ out, int L, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;             // [L][D]
    float* Vs = smem + L * D;     // [L][D]
    const int h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const long long ld = (long long)H * 3 * D;
    const float* rowb = qkv + (long long)b * L * ld + (long long)h * 3 * D;
    for (int idx = threadIdx.x; idx < L * D; idx += 256) {
        const int key = idx / D, d = idx % D;
        Ks[idx] = rowb[(long long)key * ld + D + d];
        Vs[idx] = rowb[(long long)key * ld + 2 * D + d];
    }
    __syncthreads();
    if (qi >= L) return;
    float q[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q[d] = rowb[(long long)qi * ld + d] * scale;
        acc[d] = 0.f;
    }
    const float* bp = bias + (((long long)b * H + h) * L + qi) * L;
    const float* km = keymask ? keymask + (long long)b * L : nullptr;
    float m = -INFINITY, l = 0.f;
    for (int k = 0; k < L; ++k) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[k * D + d], s);
        s += bp[k];
        if (km && km[k] == 0.f) s = ABX_NEG_MAX;
        const float mn = fmaxf(m, s);
        const float al = expf(m - mn), p = expf(s - mn);
        l = l * al + p;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * al + p * Vs[k * D + d];
        m = mn;
    }
    const float inv = 1.0f / l;
    const long long o = ((long long)b * L + qi) * H * D + (long long)h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float v = acc[d] * inv;
        if (gate) v *= 1.0f / (1.0f + expf(-gate[o + d]));
        out[o + d] = v;
    }
}

}  // namespace

extern "C" int abx_tri_attn_fwd(const AbxTriAttn* ap, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_tri_attn_fwd: null descriptor");
    const AbxTriAttn a = *ap;
    ABX_REQUIRE(a.q && a.k && a.v && a.out, "abx_tri_attn_fwd: null operand");
    ABX_REQUIRE(a.D == TD, "abx_tri_attn_fwd: head dim must be 48");
    ABX_REQUIRE(a.B > 0 && a.S > 0 && a.L > 0 && a.H > 0, "abx_tri_attn_fwd: empty problem");
    ABX_REQUIRE(a.S <= 65535 && a.B <= 65535, "abx_tri_attn_fwd: grid too large");
    ABX_REQUIRE((a.sb % 4 == 0) && (a.ss % 4 == 0) && (a.sl % 4 == 0) && (a.ob % 4 == 0) && (a.os % 4 == 0) && (a.ol % 4 == 0),
                "abx_tri_attn_fwd: strides must be multiples of 4 floats");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    ABX_REQUIRE(al16(a.q) && al16(a.k) && al16(a.v) && al16(a.out) && (!a.gate || al16(a.gate)),
                "abx_tri_attn_fwd: pointers must be 16-byte aligned");
    const size_t lds = ((((size_t)a.L * LDK + 3) & ~(size_t)3) + (size_t)a.L * LDV) * sizeof(float);
    ABX_REQUIRE(lds <= 160 * 1024, "abx_tri_attn_fwd: L too large for the single-stage K/V LDS layout (L <= 401)");
    static thread_local size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_attn_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) { abx_set_error("abx_tri_attn_fwd: hipFuncSetAttribute failed"); return (int)e; }
        configured = 160 * 1024;
    }
    hipLaunchKernelGGL(tri_attn_kernel, dim3(a.H, a.S, a.B), dim3(256), lds, st, a);
    return abx_check_launch("abx_tri_attn_fwd");
}

extern "C" int abx_seq_attn_fwd(const float* qkv, const float* bias, const float* keymask, const float* gate, float* out,
                                int B, int L, int H, int D, float scale, hipStream_t st) {
    ABX_REQUIRE(qkv && bias && out && B > 0 && L > 0 && H > 0, "abx_seq_attn_fwd: bad args");
    ABX_REQUIRE(D == 17, "abx_seq_attn_fwd: head dim must be 17 (544 / 32)");
    const size_t lds = (size_t)2 * L * D * sizeof(float);
    ABX_REQUIRE(lds <= 64 * 1024, "abx_seq_attn_fwd: L too large");
    hipLaunchKernelGGL((seq_attn_kernel<17>), dim3((L + 255) / 256, H, B), dim3(256), lds, st, qkv, bias, keymask, gate, out, L,
                       H, scale);
    return abx_check_launch("abx_seq_attn_fwd");
}
