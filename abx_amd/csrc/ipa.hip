// Invariant Point Attention core (reference abx/model/folding.py:47-132), the kernel BASELINE's north_star names:
// coalesced streaming of the (B, L, L, 128) pair slab, LDS-resident logits, wavefront-shuffle softmax reductions.
//
//  abx_ipa_pack : one thread per (residue, head): local points -> global frame (r3.rigids_apply, r3.py:9-16) and repack
//                 the fused projection row into per-head contiguous records so the attention kernel reads them coalesced:
//                    Q[b][i][h][28] = [ q_scalar*w_s (16) | q_point_global (4x3) ]
//                    K[b][h][j][28] = [ k_scalar (16)     | k_point_global (4x3) ]
//                    V[b][h][j][40] = [ v_scalar (16)     | v_point_global (8x3) ]
//  abx_ipa_attn : one workgroup per (b, group of IQ=4 query residues).
//     phase A  logits[iq][h][j] = q.k + pw[h] * sum|q_pt - k_pt|^2 + bias2d[b,i,j,h]  (direct (q-k)^2 as the reference,
//              not the expanded form: no cancellation at |x| ~ 10), mask fill finfo.min, into LDS laid out [iq][j][13]
//     softmax  one wave per (iq, h) row: shuffle max / sum
//     phase B1 scalar + point outputs: thread per (h, 4 channels of the 40) x 4 key groups, 16-byte loads, LDS reduce
//     phase B2 attention over the pair slab: lane -> 4 channels (16-byte loads, 512 B coalesced per j), 16 key groups with
//              >= 4 loads in flight each, 12 heads x 4 channels of accumulators, fixed-order reduction (shuffle + LDS):
//              streams z[b,i,j,0:128] exactly once per query residue -> HBM-bound (33.6 MB / sample / layer at L=256)
//     tail     points back to the local frame (r3.invert_rigids, r3.py:54-59), norms sqrt(sum^2 + 1e-8), concat
//              [scalar 192 | points '(r n)' 288 | norms 96 | pair 1536] = 2112 floats per residue.
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int H = 12, SQK = 16, SV = 16, PQK = 4, PV = 8, CZ = 128;
constexpr int QREC = SQK + 3 * PQK;          // 28
constexpr int VREC = SV + 3 * PV;            // 40
constexpr int NPROJ = H * SQK + H * (SQK + SV) + 3 * H * PQK + 3 * H * (PQK + PV);   // 1152
constexpr int OFF_KV = H * SQK;              // 192
constexpr int OFF_QP = OFF_KV + H * (SQK + SV);      // 576
constexpr int OFF_KVP = OFF_QP + 3 * H * PQK;        // 720
constexpr int NFEAT = H * SV + 3 * H * PV + H * PV + H * CZ;   // 2112
constexpr int IQ = 4;
constexpr int LDH = 13;                      // logits row stride over heads: odd -> lane<->j accesses are conflict-free

__global__ __launch_bounds__(256) void ipa_pack_kernel(const float* __restrict__ proj, const float* __restrict__ rots,
                                                       const float* __restrict__ trans, float* __restrict__ qpack,
                                                       float* __restrict__ kpack, float* __restrict__ vpack, int B, int L,
                                                       float w_s) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * L * H) return;
    const int h = (int)(idx % H);
    const long long row = idx / H;              // b*L + l
    const int b = (int)(row / L), l = (int)(row % L);
    const float* p = proj + row * NPROJ;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = rots[row * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = trans[row * 3 + i];
    float* q = qpack + (row * H + h) * QREC;
    float* k = kpack + (((long long)b * H + h) * L + l) * QREC;
    float* v = vpack + (((long long)b * H + h) * L + l) * VREC;
#pragma unroll
    for (int c = 0; c < SQK; ++c) {
        q[c] = p[h * SQK + c] * w_s;
        k[c] = p[OFF_KV + h * (SQK + SV) + c];
        v[c] = p[OFF_KV + h * (SQK + SV) + SQK + c];
    }
    // points: channel layout '(r n)': r*N + n, n = h*P + pt
#pragma unroll
    for (int pt = 0; pt < PQK; ++pt) {
        float x[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) x[r] = p[OFF_QP + r * (H * PQK) + h * PQK + pt];
#pragma unroll
        for (int r = 0; r < 3; ++r) q[SQK + pt * 3 + r] = t[r] + (R[r * 3 + 0] * x[0] + R[r * 3 + 1] * x[1] + R[r * 3 + 2] * x[2]);
    }
#pragma unroll
    for (int pt = 0; pt < PQK + PV; ++pt) {
        float x[3], y[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) x[r] = p[OFF_KVP + r * (H * (PQK + PV)) + h * (PQK + PV) + pt];
#pragma unroll
        for (int r = 0; r < 3; ++r) y[r] = t[r] + (R[r * 3 + 0] * x[0] + R[r * 3 + 1] * x[1] + R[r * 3 + 2] * x[2]);
        if (pt < PQK) {
#pragma unroll
            for (int r = 0; r < 3; ++r) k[SQK + pt * 3 + r] = y[r];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) v[SV + (pt - PQK) * 3 + r] = y[r];
        }
    }
}

constexpr int IPA_THREADS = 512;                 // 8 waves: enough loads in flight to stream the pair slab
constexpr int NJG2 = IPA_THREADS / 32;           // j-groups of phase B2 (32 lanes x float4 = 128 channels)
constexpr int NJG1 = 4;                          // j-groups of phase B1 (120 (h, c4) items x 4)

__global__ __launch_bounds__(IPA_THREADS) void ipa_attn_kernel(const float* __restrict__ qpack, const float* __restrict__ kpack,
                                                               const float* __restrict__ vpack, const float* __restrict__ bias2d,
                                                               const float* __restrict__ z, const float* __restrict__ mask,
                                                               const float* __restrict__ rots, const float* __restrict__ trans,
                                                               const float* __restrict__ pw, float* __restrict__ feat, int B, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lg = smem;                                           // [IQ][L][LDH] logits -> attention weights
    float* qs = smem + (((size_t)IQ * L * LDH + 3) & ~(size_t)3);   // [IQ][H][QREC]
    float* opt = qs + IQ * H * QREC;                            // [IQ][H][VREC] scalar+point outputs (global frame)
    float* red = opt + IQ * H * VREC;                           // [8 waves][H][CZ] partial sums (16-byte aligned)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * IQ;
    const int niq = min(IQ, L - i0);

    for (int idx = tid; idx < IQ * H * QREC; idx += IPA_THREADS) {
        const int iq = idx / (H * QREC);
        qs[idx] = iq < niq ? qpack[((long long)b * L + i0) * H * QREC + idx] : 0.f;
    }
    __syncthreads();

    // ---- phase A: logits -------------------------------------------------------------------------------------
    // K records of HB heads are fetched together (HB*7 16-byte loads in flight per lane) before any arithmetic
    constexpr int HB = 3;
    for (int j = tid; j < L; j += IPA_THREADS) {
        const float mj = mask[(long long)b * L + j];
        float mi[IQ];
#pragma unroll
        for (int iq = 0; iq < IQ; ++iq) mi[iq] = iq < niq ? mask[(long long)b * L + i0 + iq] : 0.f;
#pragma unroll 1
        for (int h0 = 0; h0 < H; h0 += HB) {
            f32x4 kq[HB][QREC / 4];
            float bz[HB][IQ];
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                const float* kr = kpack + (((long long)b * H + h0 + hh) * L + j) * QREC;
#pragma unroll
                for (int c4 = 0; c4 < QREC / 4; ++c4) kq[hh][c4] = *reinterpret_cast<const f32x4*>(kr + c4 * 4);
#pragma unroll
                for (int iq = 0; iq < IQ; ++iq)
                    bz[hh][iq] = iq < niq ? bias2d[(((long long)b * L + i0 + iq) * L + j) * H + h0 + hh] : 0.f;
            }
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                const int h = h0 + hh;
                const float pwh = pw[h];
#pragma unroll
                for (int iq = 0; iq < IQ; ++iq) {
                    const float* qr = qs + (iq * H + h) * QREC;
                    float sacc = 0.f, d2 = 0.f;
#pragma unroll
                    for (int c = 0; c < SQK; ++c) sacc = fmaf(qr[c], kq[hh][c >> 2][c & 3], sacc);
#pragma unroll
                    for (int c = SQK; c < QREC; ++c) {
                        const float d = qr[c] - kq[hh][c >> 2][c & 3];
                        d2 = fmaf(d, d, d2);
                    }
                    float v = sacc + pwh * d2;
                    if (iq < niq) {
                        v += bz[hh][iq];
                        if (mi[iq] * mj == 0.f) v = ABX_NEG_MAX;
                    }
                    lg[((size_t)iq * L + j) * LDH + h] = v;
                }
            }
        }
    }
    __syncthreads();
    // ---- softmax over j for each (iq, h): 48 rows over 8 waves ------------------------------------------------
    for (int row = wave; row < IQ * H; row += IPA_THREADS / 64) {
        const int iq = row / H, h = row % H;
        float* r = lg + (size_t)iq * L * LDH + h;
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) mx = fmaxf(mx, r[(size_t)j * LDH]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float e = expf(r[(size_t)j * LDH] - mx);
            r[(size_t)j * LDH] = e;
            sm += e;
        }
        sm = wave_sum(sm);
        const float inv = 1.0f / sm;
        for (int j = lane; j < L; j += 64) r[(size_t)j * LDH] *= inv;
    }
    __syncthreads();
    // ---- phase B1: scalar + point outputs.  item = (h, c4): 120 items x NJG1 j-groups, 16-byte loads of V -----------
    {
        const int item = tid % (H * VREC / 4), jg = tid / (H * VREC / 4);
        const int h = item / (VREC / 4), c4 = item % (VREC / 4);
        f32x4 acc[IQ];
#pragma unroll
        for (int iq = 0; iq < IQ; ++iq) acc[iq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (jg < NJG1) {
            const float* vb = vpack + ((long long)b * H + h) * L * VREC + c4 * 4;
            constexpr int UB = 8;                       // loads in flight per lane
            for (int jb = jg; jb < L; jb += NJG1 * UB) {
                f32x4 vv[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = min(jb + u * NJG1, L - 1);
                    vv[u] = *reinterpret_cast<const f32x4*>(vb + (long long)j * VREC);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = jb + u * NJG1;
                    if (j < L) {
#pragma unroll
                        for (int iq = 0; iq < IQ; ++iq) {
                            const float w = lg[((size_t)iq * L + j) * LDH + h];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[iq][c] = fmaf(w, vv[u][c], acc[iq][c]);
                        }
                    }
                }
            }
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq)
                *reinterpret_cast<f32x4*>(red + ((size_t)(jg * IQ + iq) * H + h) * VREC + c4 * 4) = acc[iq];
        }
        __syncthreads();
        for (int idx = tid; idx < IQ * H * VREC; idx += IPA_THREADS) {
            float sacc = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < NJG1; ++g2) sacc += red[(size_t)g2 * IQ * H * VREC + idx];
            opt[idx] = sacc;
        }
        __syncthreads();
    }
    // ---- phase B2: attention over the pair slab: lane -> 4 channels (16-byte loads), 16 j-groups, 48 accumulators ------
    {
        const int c4 = tid & 31, jg = tid >> 5;
        for (int iq = 0; iq < niq; ++iq) {
            const float* zr = z + (((long long)b * L + i0 + iq) * L) * CZ + c4 * 4;
            const float* ar = lg + (size_t)iq * L * LDH;
            f32x4 acc[H];
#pragma unroll
            for (int hh = 0; hh < H; ++hh) acc[hh] = (f32x4){0.f, 0.f, 0.f, 0.f};
            constexpr int UZ = 8;                       // 16-byte slab loads in flight per lane (8 KB per wave)
            for (int jb = jg; jb < L; jb += NJG2 * UZ) {
                f32x4 zv[UZ];
#pragma unroll
                for (int u = 0; u < UZ; ++u) {
                    const int j = min(jb + u * NJG2, L - 1);
                    zv[u] = *reinterpret_cast<const f32x4*>(zr + (long long)j * CZ);
                }
#pragma unroll
                for (int u = 0; u < UZ; ++u) {
                    const int j = jb + u * NJG2;
                    if (j < L) {
                        const float* a12 = ar + (size_t)j * LDH;
#pragma unroll
                        for (int hh = 0; hh < H; ++hh) {
                            const float w = a12[hh];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[hh][c] = fmaf(w, zv[u][c], acc[hh][c]);
                        }
                    }
                }
            }
            // the two j-groups of a wave (lanes l, l+32) first, then the 8 waves through LDS, in a fixed order
#pragma unroll
            for (int hh = 0; hh < H; ++hh)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[hh][c] += __shfl_xor(acc[hh][c], 32, 64);
            if (lane < 32) {
#pragma unroll
                for (int hh = 0; hh < H; ++hh)
                    *reinterpret_cast<f32x4*>(red + ((size_t)wave * H + hh) * CZ + c4 * 4) = acc[hh];
            }
            __syncthreads();
            float* fo = feat + ((long long)b * L + i0 + iq) * NFEAT + (H * SV + 4 * H * PV);
            for (int idx = tid; idx < H * CZ; idx += IPA_THREADS) {
                float sacc = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < IPA_THREADS / 64; ++w8) sacc += red[(size_t)w8 * H * CZ + idx];
                fo[idx] = sacc;
            }
            __syncthreads();
        }
    }
    // ---- tail: scalar copy, points to the local frame, norms ----------------------------------------------------------
    for (int idx = tid; idx < IQ * H * SV; idx += IPA_THREADS) {
        const int iq = idx / (H * SV), r = idx % (H * SV);
        if (iq < niq) feat[((long long)b * L + i0 + iq) * NFEAT + r] = opt[(iq * H + r / SV) * VREC + (r % SV)];
    }
    for (int idx = tid; idx < IQ * H * PV; idx += IPA_THREADS) {
        const int iq = idx / (H * PV), n = idx % (H * PV);
        if (iq >= niq) continue;
        const int h = n / PV, pt = n % PV;
        const long long row = (long long)b * L + i0 + iq;
        const float* R = rots + row * 9;
        const float* t = trans + row * 3;
        const float* g = opt + (iq * H + h) * VREC + SV + pt * 3;
        // invert_rigids: R^T, -R^T t ; apply: R^T g + (-R^T t)   (same association as the reference)
        float loc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float it = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
            loc[r] = it + (R[0 * 3 + r] * g[0] + R[1 * 3 + r] * g[1] + R[2 * 3 + r] * g[2]);
        }
        float* fo = feat + row * NFEAT;
#pragma unroll
        for (int r = 0; r < 3; ++r) fo[H * SV + r * (H * PV) + n] = loc[r];
        fo[H * SV + 3 * H * PV + n] = sqrtf(loc[0] * loc[0] + loc[1] * loc[1] + loc[2] * loc[2] + 1e-8f);
    }
}

}  // namespace

extern "C" int abx_ipa_pack(const float* proj, const float* rots, const float* trans, float* qpack, float* kpack,
                            float* vpack, int B, int L, float scalar_weight, hipStream_t st) {
    ABX_REQUIRE(proj && rots && trans && qpack && kpack && vpack && B > 0 && L > 0, "abx_ipa_pack: bad args");
    const long long n = (long long)B * L * H;
    hipLaunchKernelGGL(ipa_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, proj, rots, trans, qpack, kpack,
                       vpack, B, L, scalar_weight);
    return abx_check_launch("abx_ipa_pack");
}

extern "C" int abx_ipa_attn(const float* qpack, const float* kpack, const float* vpack, const float* bias2d, const float* z,
                            const float* mask, const float* rots, const float* trans, const float* point_weights, float* feat,
                            int B, int L, hipStream_t st) {
    ABX_REQUIRE(qpack && kpack && vpack && bias2d && z && mask && rots && trans && point_weights && feat, "abx_ipa_attn: null");
    ABX_REQUIRE(B > 0 && L > 0 && B <= 65535, "abx_ipa_attn: bad sizes");
    const size_t lds = ((((size_t)IQ * L * LDH + 3) & ~(size_t)3) + IQ * H * QREC + IQ * H * VREC + (size_t)(IPA_THREADS / 64) * H * CZ) * sizeof(float);
    ABX_REQUIRE(lds <= 160 * 1024, "abx_ipa_attn: L too large for LDS-resident logits");
    static thread_local bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) { abx_set_error("abx_ipa_attn: hipFuncSetAttribute failed"); return (int)e; }
        configured = true;
    }
    hipLaunchKernelGGL(ipa_attn_kernel, dim3((L + IQ - 1) / IQ, B), dim3(IPA_THREADS), lds, st, qpack, kpack, vpack, bias2d, z, mask,
                       rots, trans, point_weights, feat, B, L);
    return abx_check_launch("abx_ipa_attn");
}
