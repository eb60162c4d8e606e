// Invariant Point Attention core (reference abx/model/folding.py:47-132), the kernel BASELINE's north_star names:
// coalesced streaming of the (B, L, L, 128) pair slab, LDS-resident logits, wavefront-shuffle softmax reductions.
//
//  abx_ipa_pack : one thread per (residue, head): local points -> global frame (r3.rigids_apply, r3.py:9-16) and repack
//                 the fused projection row into per-head contiguous records so the attention kernel reads them coalesced:
//                    Q[b][i/12][h][(i%12)/2][28][i%2] = [ q_scalar*w_s (16) | q_point_global (4x3) ]  (query pairs interleaved:
//                                     the weights kernel reads them as wave-uniform float2 for packed FMAs; rows padded to 12)
//                    K[b][h][j][28] = [ k_scalar (16)     | k_point_global (4x3) ]
//                    V[b][h][j][40] = [ v_scalar (16)     | v_point_global (8x3) ]
//  abx_ipa_attn : two kernels (abx_ipa_weights + abx_ipa_pair).
//     ipa_weights_kernel  one workgroup per (b, 12 query residues, 4 heads), two per CU, samples pinned to XCDs: logits (direct
//              (q-k)^2 point distances as the reference, not the expanded form: no cancellation at |x| ~ 10; Q through the
//              scalar cache, packed FMAs over query pairs) + pair bias, mask fill finfo.min, LDS-resident [iq][h][j];
//              wave-shuffle softmax; the normalised weights go to HBM as attn[b][i][head group][j][4]; scalar + point outputs
//              (16-byte loads of V, key groups folded through LDS); points back to the local frame (r3.invert_rigids,
//              r3.py:54-59), norms sqrt(sum^2 + 1e-8)
//     ipa_pair_kernel     attention over the pair slab, ONE WAVE per (b, i) row: streams z[b,i,j,0:128] exactly once (16 keys
//              in flight per wave, their weights parked in a wave-private LDS strip), no block barrier, no cross-lane
//              reduction -> the HBM-bound part (6.3 GB per layer at B = 100, L = 352; + 0.6 GB weights)
//     feature row [scalar 192 | points '(r n)' 288 | norms 96 | pair 1536] = 2112 floats per residue.
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

constexpr int IPA_THREADS = 512;
constexpr int IQ = 12;                           // query residues per workgroup of the weights kernel
constexpr int HG = 4, NHG = H / HG;              // heads per workgroup, head groups

__global__ __launch_bounds__(256) void ipa_pack_kernel(const float* __restrict__ proj, const float* __restrict__ rots,
                                                       const float* __restrict__ trans, float* __restrict__ qpack,
                                                       float* __restrict__ kpack, float* __restrict__ vpack, int B, int L,
                                                       float w_s) {
    const int NIB = (L + IQ - 1) / IQ, LQ = NIB * IQ;           // query rows are padded to whole blocks of IQ
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * LQ * H) return;
    const int h = (int)(idx % H);
    const long long prow = idx / H;             // b*LQ + l
    const int b = (int)(prow / LQ), l = (int)(prow % LQ);
    // Q record of (b, l, h): element c of query l sits at [b][l / IQ][h][(l % IQ) / 2][c][l % 2]
    float* q = qpack + ((((long long)b * NIB + l / IQ) * H + h) * (IQ / 2) + (l % IQ) / 2) * (QREC * 2) + (l & 1);
    if (l >= L) {                               // pad rows of the last block: zeros (their logits are never used)
#pragma unroll
        for (int c = 0; c < QREC; ++c) q[2 * c] = 0.f;
        return;
    }
    const long long row = (long long)b * L + l;
    const float* p = proj + row * NPROJ;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = rots[row * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = trans[row * 3 + i];
    float* k = kpack + (((long long)b * H + h) * L + l) * QREC;
    float* v = vpack + (((long long)b * H + h) * L + l) * VREC;
#pragma unroll
    for (int c = 0; c < SQK; ++c) {
        q[2 * c] = p[h * SQK + c] * w_s;
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
        for (int r = 0; r < 3; ++r) q[2 * (SQK + pt * 3 + r)] = t[r] + (R[r * 3 + 0] * x[0] + R[r * 3 + 1] * x[1] + R[r * 3 + 2] * x[2]);
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

// LDS row stride of the logits [IQ * HG][LR]: a multiple of 32 plus 12, so the B1 reads of one wave (2 heads x 12 keys) fall
// into distinct banks
__host__ __device__ inline int ipa_row_stride(int L) { return ((L + 31) / 32) * 32 + 12; }
constexpr int RED_I = IQ * 4 + 4;                       // floats per (key group, item) slot of the B1 fold: 48 + 4 pad
constexpr int RED_G = (HG * VREC / 4) * RED_I + 4;      // floats per key group: 40 slots + 4 pad
// floats of the logits region [IQ * HG][LR]; the fold of phase B1 reuses it as [6 key groups][40 items][IQ][4]
__host__ __device__ inline size_t ipa_logits_floats(int L) {
    const size_t lgf = (size_t)IQ * HG * ipa_row_stride(L), redf = (size_t)6 * RED_G;
    return lgf > redf ? lgf : redf;
}

#ifdef IPA_STAMP
// probe build (tools/probes/kb_ipa_stamps.py): shader-clock stamps of thread 0 at the phase boundaries of 1024 workgroups in the middle
// of the grid
__device__ unsigned long long ipa_stamp_buf[1024 * 8];
#define ISTAMP(k) { if (threadIdx.x == 0 && blockIdx.x >= 4096 && blockIdx.x < 5120) ipa_stamp_buf[(blockIdx.x - 4096) * 8 + (k)] = __builtin_amdgcn_s_memtime(); }
#else
#define ISTAMP(k) {}
#endif
// ---- kernel 1: attention weights + scalar / point outputs ----------------------------------------------------------------
// One workgroup per (b, IQ = 12 query residues, group of 4 heads); two workgroups per CU (78 KB of LDS each at L = 352) so that
// the load-latency-bound phases of one overlap the arithmetic of the other.  12 queries share every K / V record fetched.
//   phase A  wave per (64 keys, head): q.k + pw[h] * sum|q_pt - k_pt|^2 for the 12 queries -> LDS [iq][h][j]
//   bias     thread per (iq, j): + bias2d[b,i,j, 4 heads] (one 16-byte load), mask fill finfo.min
//   softmax  one wave per query (its 4 head rows together), shuffle max / sum; the normalised weights go to LDS and to
//            attn[b][i][head group][j][4] (16 contiguous bytes per lane): the slab kernel streams them
//   phase B1 scalar + point outputs: lane = (h, 4 of the 40 channels) x 12 key groups inside ONE wave, 16-byte loads of V
//            (software-pipelined, first round requested before the softmax), key groups folded through LDS in a fixed order
//   tail     points back to the local frame (r3.invert_rigids, r3.py:54-59), norms sqrt(sum^2 + 1e-8)
__global__ __launch_bounds__(IPA_THREADS, 4) void ipa_weights_kernel(const float* __restrict__ qpack, const float* __restrict__ kpack,
                                                                     const float* __restrict__ vpack, const float* __restrict__ bias2d,
                                                                     const float* __restrict__ mask, const float* __restrict__ rots,
                                                                     const float* __restrict__ trans, const float* __restrict__ pw,
                                                                     float* __restrict__ attn, float* __restrict__ feat, int B, int L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LR = ipa_row_stride(L);
    float* lg = smem;                                           // [IQ * HG][LR] logits -> attention weights
    float* opt = smem + ipa_logits_floats(L);                   // [IQ][HG][VREC] scalar + point outputs (global frame)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD x (= blockIdx & 7: workgroups are dealt round-robin to the 8 XCDs) owns the samples b = x, x + 8, ...: the
    // K / V packs of a sample and the bias lines shared by its three head groups are fetched into ONE L2 instead of eight
    // The samples beyond the last full round of 8 (B % 8 of them: 5 at the 13 samples one of 8 GPUs holds under strong scaling) are dealt to
    // the XCDs workgroup by workgroup instead - whole samples would leave XCDs 5 .. 7 with half the work of XCDs 0 .. 4 (1.31 x the per-sample
    // time of a full batch at 12 samples, 1.21 x at 13: profiles/r06f_bench_L352_b12.json); their K / V packs (1.1 MB per sample) then
    // live in several L2s, which is cheap next to an idle XCD.
    const int NIB = (L + IQ - 1) / IQ, per_b = NIB * NHG;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int full_slots = (B / 8) * per_b;                     // slots of an XCD that belong to whole samples
    int b, wb;
    if (slot < full_slots) {
        b = (slot / per_b) * 8 + xcd;
        wb = slot % per_b;
    } else {
        const int t = (slot - full_slots) * 8 + xcd;            // workgroup t of the B % 8 remaining samples
        if (t >= (B % 8) * per_b) return;
        b = (B / 8) * 8 + t / per_b;
        wb = t % per_b;
    }
    const int hg = wb % NHG, h0 = hg * HG;
    const int iblk = wb / NHG;
    const int i0 = iblk * IQ;
    const int niq = min(IQ, L - i0);

    ISTAMP(0)
    // ---- phase A: scalar + point logits.  A wave takes 64 keys of ONE head per task; the Q records of the block are
    // wave-uniform and come through the scalar cache as (query pair, channel) float2, so every product is one packed FMA
    // over two queries with the key channel broadcast: no LDS traffic, 22 VALU instructions per (key, head, query pair)
    {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float* __restrict__ qb = qpack + (((long long)b * NIB + iblk) * H + h0) * (IQ / 2) * (QREC * 2);
        const int wpk = (L + 63) >> 6;
        for (int task = wave; task < HG * wpk; task += IPA_THREADS / 64) {
            const int hl = __builtin_amdgcn_readfirstlane(task / wpk);
            const int j = (task - hl * wpk) * 64 + lane;
            const float* kr = kpack + (((long long)b * H + h0 + hl) * L + min(j, L - 1)) * QREC;
            f32x4 kq[QREC / 4];
#pragma unroll
            for (int c4 = 0; c4 < QREC / 4; ++c4) kq[c4] = *reinterpret_cast<const f32x4*>(kr + c4 * 4);
            const float pwh = pw[h0 + hl];
            const f32x2* __restrict__ qh = reinterpret_cast<const f32x2*>(qb + (long long)hl * (IQ / 2) * (QREC * 2));
#pragma unroll
            for (int ip = 0; ip < IQ / 2; ++ip) {
                f32x2 sacc = {0.f, 0.f}, d2 = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < SQK; ++c) {
                    const float kc = kq[c >> 2][c & 3];
                    sacc = __builtin_elementwise_fma(qh[ip * QREC + c], (f32x2){kc, kc}, sacc);
                }
#pragma unroll
                for (int c = SQK; c < QREC; ++c) {
                    const float kc = kq[c >> 2][c & 3];
                    const f32x2 d = qh[ip * QREC + c] - (f32x2){kc, kc};
                    d2 = __builtin_elementwise_fma(d, d, d2);
                }
                if (j < L) {
                    lg[(size_t)((2 * ip) * HG + hl) * LR + j] = sacc[0] + pwh * d2[0];
                    lg[(size_t)((2 * ip + 1) * HG + hl) * LR + j] = sacc[1] + pwh * d2[1];
                }
            }
        }
    }
    ISTAMP(1)
    __syncthreads();
    ISTAMP(2)
    // ---- pair bias and mask: thread per (iq, j), the 4 heads of the group in one 16-byte load
    for (int idx = tid; idx < niq * L; idx += IPA_THREADS) {
        const int iq = idx / L, j = idx - iq * L;
        const long long row = (long long)b * L + i0 + iq;
        const f32x4 bz = *reinterpret_cast<const f32x4*>(bias2d + (row * L + j) * H + h0);
        const bool masked = mask[row] * mask[(long long)b * L + j] == 0.f;
        float* r = lg + (size_t)iq * HG * LR + j;
#pragma unroll
        for (int c = 0; c < HG; ++c) r[(size_t)c * LR] = masked ? ABX_NEG_MAX : r[(size_t)c * LR] + bz[c];
    }
    // ---- phase B1 set-up: item = (h, c4): 40 items, 5 per wave; the 12 key groups of an item are lanes il, il + 5, ... of its
    // wave (lanes 60..63 idle).  The first V records are requested now: their latency hides behind the softmax
    constexpr int NJG = 12, IPW = 5, UB = 4;
    const int il = lane % IPW, jg = lane / IPW;
    const int item = wave * IPW + il;                       // 0..39
    const int hl1 = item / (VREC / 4), c41 = item % (VREC / 4);
    const float* vb = vpack + ((long long)b * H + h0 + hl1) * L * VREC + c41 * 4;
    f32x4 vnext[UB];
    auto load_round = [&](int jb) {
#pragma unroll
        for (int u = 0; u < UB; ++u) vnext[u] = *reinterpret_cast<const f32x4*>(vb + (long long)min(jb + u * NJG, L - 1) * VREC);
    };
    if (jg < NJG) load_round(jg);
    __syncthreads();
    ISTAMP(3)
    // ---- softmax over j, one wave per query: its 4 head rows together (interleaved reductions); the normalised weights go
    // back to LDS (phase B1) and to HBM as [i][head group][j][4], 16 contiguous bytes per lane
    for (int iq = wave; iq < niq; iq += IPA_THREADS / 64) {
        float* r = lg + (size_t)iq * HG * LR;
        float mx[HG], sm[HG], inv[HG];
#pragma unroll
        for (int c = 0; c < HG; ++c) { mx[c] = -INFINITY; sm[c] = 0.f; }
        for (int j = lane; j < L; j += 64)
#pragma unroll
            for (int c = 0; c < HG; ++c) mx[c] = fmaxf(mx[c], r[(size_t)c * LR + j]);
#pragma unroll
        for (int c = 0; c < HG; ++c) mx[c] = wave_max(mx[c]);
        for (int j = lane; j < L; j += 64)
#pragma unroll
            for (int c = 0; c < HG; ++c) {
                const float e = expf(r[(size_t)c * LR + j] - mx[c]);
                r[(size_t)c * LR + j] = e;
                sm[c] += e;
            }
#pragma unroll
        for (int c = 0; c < HG; ++c) inv[c] = 1.0f / wave_sum(sm[c]);
        float* o = attn + ((((long long)b * L + i0 + iq) * NHG + hg) * L) * HG;
        for (int j = lane; j < L; j += 64) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < HG; ++c) {
                v[c] = r[(size_t)c * LR + j] * inv[c];
                r[(size_t)c * LR + j] = v[c];
            }
            *reinterpret_cast<f32x4*>(o + (long long)j * HG) = v;
        }
    }
    __syncthreads();
    ISTAMP(4)
    // ---- phase B1: scalar + point outputs; the V records of the next 4 keys are in flight while the current 4 are consumed
    {
        f32x4 acc[IQ];
#pragma unroll
        for (int iq = 0; iq < IQ; ++iq) acc[iq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (jg < NJG) {
            const float* wr = lg + (size_t)hl1 * LR;
            for (int jb = jg; jb < L; jb += NJG * UB) {
                f32x4 vv[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) vv[u] = vnext[u];
                if (jb + NJG * UB < L) load_round(jb + NJG * UB);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = jb + u * NJG;
                    if (j < L) {
#pragma unroll
                        for (int iq = 0; iq < IQ; ++iq) {
                            const float w = wr[(size_t)iq * HG * LR + j];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[iq][c] = fmaf(w, vv[u][c], acc[iq][c]);
                        }
                    }
                }
            }
        }
        // fold the key groups through the (now free) logits region in three rounds, (g, g + 6), then (g, g + 3), then
        // (g0 + g1) + g2: fixed order, no long shuffle chains (their live ranges spilled)
        ISTAMP(5)
        float* red = lg;                                    // [6 key groups][40 items][IQ][4] (+ padding)
        auto slot_of = [&](int g) { return red + (size_t)g * RED_G + (size_t)item * RED_I; };     // padded strides: distinct banks
        __syncthreads();                                    // every wave has consumed its weights
        if (jg >= 6 && jg < NJG) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) *reinterpret_cast<f32x4*>(slot_of(jg - 6) + iq * 4) = acc[iq];
        }
        __syncthreads();
        if (jg < 6) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) {
                const f32x4 o4 = *reinterpret_cast<const f32x4*>(slot_of(jg) + iq * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[iq][c] += o4[c];
            }
        }
        __syncthreads();
        if (jg >= 3 && jg < 6) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) *reinterpret_cast<f32x4*>(slot_of(jg - 3) + iq * 4) = acc[iq];
        }
        __syncthreads();
        if (jg < 3) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) {
                const f32x4 o4 = *reinterpret_cast<const f32x4*>(slot_of(jg) + iq * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[iq][c] += o4[c];
            }
        }
        __syncthreads();
        if (jg == 1 || jg == 2) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) *reinterpret_cast<f32x4*>(slot_of(jg - 1) + iq * 4) = acc[iq];
        }
        __syncthreads();
        if (jg == 0) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) {
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot_of(0) + iq * 4), x2 = *reinterpret_cast<const f32x4*>(slot_of(1) + iq * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[iq][c] = (acc[iq][c] + x1[c]) + x2[c];
            }
        }
        if (jg == 0) {
#pragma unroll
            for (int iq = 0; iq < IQ; ++iq) *reinterpret_cast<f32x4*>(opt + (size_t)(iq * HG + hl1) * VREC + c41 * 4) = acc[iq];
        }
        __syncthreads();
    }
    ISTAMP(6)
    // ---- tail: scalar copy, points to the local frame, norms ----------------------------------------------------------
    for (int idx = tid; idx < niq * HG * SV; idx += IPA_THREADS) {
        const int iq = idx / (HG * SV), r = idx % (HG * SV);
        feat[((long long)b * L + i0 + iq) * NFEAT + h0 * SV + r] = opt[(iq * HG + r / SV) * VREC + (r % SV)];
    }
    for (int idx = tid; idx < niq * HG * PV; idx += IPA_THREADS) {
        const int iq = idx / (HG * PV), nl = idx % (HG * PV);
        const int hl = nl / PV, pt = nl % PV, n = h0 * PV + nl;
        const long long row = (long long)b * L + i0 + iq;
        const float* R = rots + row * 9;
        const float* t = trans + row * 3;
        const float* g = opt + (iq * HG + hl) * VREC + SV + pt * 3;
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
    ISTAMP(7)
}

// ---- kernel 2: attention over the pair slab, out[b,i,h,:] = sum_j attn[b,i,j,h] * z[b,i,j,:] --------------------------------
// The HBM stream of the IPA layer: z[b,i,:,:] (L x 512 bytes) is read exactly once.  ONE WAVE per (b, i) row: a lane owns 2 of
// the 128 channels (8-byte loads, 512 contiguous bytes per wave and key), 16 keys in flight; the 16 x 12 weights of those keys
// are fetched by the same wave as 3 coalesced dwords per lane, together with the slab loads (one latency, not a dependent
// scalar-cache miss per key; layout [head group][key][4]), parked in a wave-private LDS strip and read back as broadcasts; 24 accumulators per lane, keys
// summed in order: no block barrier, no cross-lane reduction.
constexpr int PAIR_WAVES = 4;                    // rows per workgroup
#ifndef ABX_PAIR_UZ
#define ABX_PAIR_UZ 8
#endif
constexpr int UZ = ABX_PAIR_UZ;                  // keys per step (two steps in flight)

__global__ __launch_bounds__(PAIR_WAVES * 64) void ipa_pair_kernel(const float* __restrict__ attn, const float* __restrict__ z,
                                                                   float* __restrict__ feat, long long rows, int L) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float wl[PAIR_WAVES][UZ * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * PAIR_WAVES + wave;      // b * L + i
    if (row >= rows) return;
    const float* __restrict__ wr = attn + row * L * H;
    const float* __restrict__ zr = z + row * L * CZ + lane * 2;
    float* ws = wl[wave];
    const int nw = L * HG;                                       // weights of this row per head group
    f32x2 acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = (f32x2){0.f, 0.f};
    // software pipeline, two register sets in ping-pong: the slab rows and weights of step s + 1 are requested before step s is summed,
    // so a wave always has a step in flight (at a dozen samples per GPU a CU holds ~18 of these waves: nothing else hides a step's HBM
    // round trip)
    f32x2 za[UZ], zb[UZ];
    float wa[NHG], wb[NHG];
    auto request = [&](int j0, f32x2 (&zd)[UZ], float (&wd)[NHG]) {
#pragma unroll
        for (int u = 0; u < UZ; ++u) zd[u] = *reinterpret_cast<const f32x2*>(zr + (long long)min(j0 + u, L - 1) * CZ);
#pragma unroll
        for (int q = 0; q < NHG; ++q) wd[q] = wr[(long long)q * nw + min(j0 * HG + (lane & (UZ * HG - 1)), nw - 1)];   // head group q: UZ keys x 4 heads
    };
    auto sum_step = [&](int j0, const f32x2 (&zd)[UZ], const float (&wd)[NHG]) {
        // the previous step's broadcast reads of this strip have retired (same wave, in-order LDS queue)
#pragma unroll
        for (int q = 0; q < NHG; ++q) ws[q * (UZ * HG) + (lane & (UZ * HG - 1))] = wd[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < UZ; ++u) {
            if (j0 + u < L) {
                f32x4 w4[H / 4];
#pragma unroll
                for (int q = 0; q < NHG; ++q) w4[q] = *reinterpret_cast<const f32x4*>(ws + q * (UZ * HG) + u * HG);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float wh = w4[h >> 2][h & 3];
                    acc[h][0] = fmaf(wh, zd[u][0], acc[h][0]);
                    acc[h][1] = fmaf(wh, zd[u][1], acc[h][1]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    request(0, za, wa);
    for (int j0 = 0; j0 < L; j0 += 2 * UZ) {
        if (j0 + UZ < L) request(j0 + UZ, zb, wb);
        sum_step(j0, za, wa);
        if (j0 + UZ < L) {
            if (j0 + 2 * UZ < L) request(j0 + 2 * UZ, za, wa);
            sum_step(j0 + UZ, zb, wb);
        }
    }
    float* fo = feat + row * NFEAT + (H * SV + 4 * H * PV) + lane * 2;
#pragma unroll
    for (int h = 0; h < H; ++h) *reinterpret_cast<f32x2*>(fo + h * CZ) = acc[h];
}

}  // namespace

extern "C" int abx_ipa_pack(const float* proj, const float* rots, const float* trans, float* qpack, float* kpack,
                            float* vpack, int B, int L, float scalar_weight, hipStream_t st) {
    ABX_REQUIRE(proj && rots && trans && qpack && kpack && vpack && B > 0 && L > 0, "abx_ipa_pack: bad args");
    const long long n = (long long)B * ((L + IQ - 1) / IQ) * IQ * H;
    hipLaunchKernelGGL(ipa_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, proj, rots, trans, qpack, kpack,
                       vpack, B, L, scalar_weight);
    return abx_check_launch("abx_ipa_pack");
}

extern "C" long long abx_ipa_qpack_bytes(int B, int L) {
    return (long long)B * ((L + IQ - 1) / IQ) * IQ * H * QREC * sizeof(float);
}

extern "C" long long abx_ipa_attn_workspace_bytes(int B, int L) {
    return (long long)B * L * L * H * sizeof(float);
}

extern "C" int abx_ipa_weights(const float* qpack, const float* kpack, const float* vpack, const float* bias2d, const float* mask,
                               const float* rots, const float* trans, const float* point_weights, float* attn_ws, float* feat, int B,
                               int L, hipStream_t st) {
    ABX_REQUIRE(qpack && kpack && vpack && bias2d && mask && rots && trans && point_weights && attn_ws && feat, "abx_ipa_weights: null");
    ABX_REQUIRE(B > 0 && L > 0 && B <= 65535 && (long long)B * L < (1ll << 31), "abx_ipa_weights: bad sizes");
    const size_t lds = (ipa_logits_floats(L) + IQ * HG * VREC) * sizeof(float);
    ABX_REQUIRE(lds <= 160 * 1024, "abx_ipa_weights: L too large for LDS-resident logits");
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(ipa_weights_kernel), 160 * 1024, "abx_ipa_weights")) return rc;
    const long long per_b = (long long)((L + IQ - 1) / IQ) * NHG;
    const long long nwg = 8 * ((B / 8) * per_b + ((B % 8) * per_b + 7) / 8);        // (full rounds of 8 samples, then the rest workgroup by workgroup)
    ABX_REQUIRE(nwg < (1LL << 31), "abx_ipa_weights: grid too large");
    hipLaunchKernelGGL(ipa_weights_kernel, dim3((unsigned)nwg), dim3(IPA_THREADS), lds, st, qpack, kpack, vpack, bias2d, mask,
                       rots, trans, point_weights, attn_ws, feat, B, L);
    return abx_check_launch("abx_ipa_weights");
}

#ifdef IPA_STAMP
extern "C" int abx_ipa_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ipa_stamp_buf), sizeof(unsigned long long) * 1024 * 8);
}
#endif

extern "C" int abx_ipa_pair(const float* attn_ws, const float* z, float* feat, int B, int L, hipStream_t st) {
    ABX_REQUIRE(attn_ws && z && feat && B > 0 && L > 0 && (long long)B * L < (1ll << 31), "abx_ipa_pair: bad args");
    const long long rows = (long long)B * L;
    hipLaunchKernelGGL(ipa_pair_kernel, dim3((unsigned)((rows + PAIR_WAVES - 1) / PAIR_WAVES)), dim3(PAIR_WAVES * 64), 0, st, attn_ws, z,
                       feat, rows, L);
    return abx_check_launch("abx_ipa_pair");
}

extern "C" int abx_ipa_attn(const float* qpack, const float* kpack, const float* vpack, const float* bias2d, const float* z,
                            const float* mask, const float* rots, const float* trans, const float* point_weights, float* attn_ws,
                            float* feat, int B, int L, hipStream_t st) {
    ABX_REQUIRE(z != nullptr, "abx_ipa_attn: null");
    const int rc = abx_ipa_weights(qpack, kpack, vpack, bias2d, mask, rots, trans, point_weights, attn_ws, feat, B, L, st);
    if (rc) return rc;
    return abx_ipa_pair(attn_ws, z, feat, B, L, st);
}
