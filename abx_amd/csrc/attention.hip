// Pair-biased softmax attention kernels of the Seqformer block.
//
// abx_tri_attn_fwd — triangle attention (reference abx/model/seqformer.py:506-550, Attention.forward :272-312) as a
// flash-style fused kernel: the (B, L, 4, L, L) logits tensor (268 MB / sample at L = 256) is never materialised.
// One workgroup per (b, row s, head h): K [L][48] and V [L][48] of that row are staged ONCE in LDS (row stride 52 floats;
// K rows permuted so a lane's 12 QK^T operands are three 16-byte reads); each of the 12 waves (3 per SIMD, so one wave's softmax VALU / LDS
// latency hides under the others' MFMAs) walks query tiles of 16 rows with a base-2 online softmax over 64-key tiles; the pair bias of
// a tile is fetched before its QK^T MFMAs.  Both contractions run on v_mfma_f32_16x16x4_f32 (exact fp32):
//     S^T[key][q]  = sum_d K[key][d] * Q[q][d]        ("swapped" QK^T: a lane owns 4 keys of ONE query column, so the
//                                                      softmax row reductions are in-lane + two cross-group shuffles)
//     O^T[d][q]   += sum_key V[key][d] * P[key][q]    (P fragments feed the B operand straight from registers)
// The orientation (starting / ending node) is expressed only through strides; no transposed copy of the pair stack.
//
// abx_seq_attn_fwd — sequence attention with 32-head pair bias (seqformer.py:314-356, split_first=False :278-281):
// lanes are keys so that the bias rows are read coalesced, K/V of the (b, h) pair in LDS.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int TD = 48;        // head dim of triangle attention
constexpr int LDK = 52;       // K rows are stored permuted [key][g][kd] (d = kd*4 + g): a lane's 12 operands are 3 x 16-byte reads
constexpr int LDV = 52;       // V row stride: 4*52 mod 32 == 16 -> lane groups g land on disjoint bank halves

constexpr int TRI_THREADS = 768;      // 12 waves = 3 per SIMD: softmax VALU / LDS latency of one wave hide under the others' MFMAs
constexpr float LOG2E = 1.4426950408889634f;

// KC: keys per staged chunk (== L: the whole row once, the round-1 form; otherwise a multiple of 64).  A row longer than the LDS holds (L > 389) is
// walked in key chunks: the 12 waves take 12 query tiles at a time through ALL chunks (online softmax across chunks, state in
// registers), re-staging the chunks for every group of 12 query tiles - the exact kernel is the fallback of the split-f16 one, the
// re-reads are the price of having no length limit (seqformer.py:272-312 has none).
__global__ __launch_bounds__(TRI_THREADS) void tri_attn_kernel(const AbxTriAttn a, const int KC) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L;
    float* Ks = smem;
    float* Vs = smem + (((size_t)KC * LDK + 3) & ~(size_t)3);     // keep V rows 16-byte aligned
    float* Ms = Vs + (size_t)KC * LDV;                            // [KC + 1]: key classes of the chunk's slots, [KC] = any key masked
    const int h = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int nchunk = (L + KC - 1) / KC;

    const long long base = (long long)b * a.sb + (long long)s * a.ss + (long long)h * TD;
    const float* km = a.keymask ? a.keymask + (long long)b * a.km_sb : nullptr;
    // does the sample mask any key?  (decides the clamp-free path, as a whole-row property like before)
    // (through a slot of the dynamic LDS: __syncthreads_or would add static LDS, and the kernel asks for all 160 KB dynamically)
    float* flag = Ms + ((KC + 63) / 64) * 64;
    if (tid == 0) *flag = 0.f;
    __syncthreads();
    if (km) {
        bool any_masked = false;
        for (int key = tid; key < L; key += TRI_THREADS) any_masked |= km[key] == 0.f;
        if (any_masked) *flag = 1.f;                             // benign race: every writer stores the same value
    }
    __syncthreads();
    const bool has_mask = *flag != 0.f;
    // ---- stage K, V and the key classes of chunk c: 0 (valid), 1 (masked: the reference REPLACES the logit by finfo.min), 2 (beyond L)
    auto stage = [&](int c) {
        const int c0 = c * KC, nk = min(KC, L - c0);
        for (int idx = tid; idx < nk * (TD / 4); idx += TRI_THREADS) {
            const int key = idx / (TD / 4), c4 = idx % (TD / 4);
            const long long off = base + (long long)(c0 + key) * a.sl + c4 * 4;
            const f32x4 kv = *reinterpret_cast<const f32x4*>(a.k + off);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(a.v + off);
            float* kd = Ks + key * LDK + c4;                     // element d = c4*4 + j goes to [g = j][kd = c4]
            kd[0] = kv[0]; kd[12] = kv[1]; kd[24] = kv[2]; kd[36] = kv[3];
            *reinterpret_cast<f32x4*>(Vs + key * LDV + c4 * 4) = vv;
        }
        const int npad = ((nk + 63) / 64) * 64;
        for (int key = tid; key < npad; key += TRI_THREADS)
            Ms[key] = key < nk ? ((!km || km[c0 + key] != 0.f) ? 0.f : 1.f) : 2.f;
    };
    if (nchunk == 1) {
        stage(0);
        __syncthreads();
    }

    const float* biasb = a.bias ? a.bias + (long long)b * a.bias_sb + (long long)h * a.bias_sh : nullptr;
    const int nqt = (L + 15) / 16;
    const bool bias_vec = biasb && a.bias_sk == 1 && (a.bias_sq % 4 == 0) && (L % 4 == 0) &&
                          ((reinterpret_cast<uintptr_t>(biasb) & 15) == 0);
    const float qscale = a.scale * LOG2E;                       // softmax evaluated in base 2: exp(x) = exp2(x log2 e)
    const float bias_l2 = a.bias_log2 ? 0.0078125f : LOG2E;     // (a bias that arrives as ABX_TRI_BIAS_LOG2 x the pair bias: back to base-2 logits)

    for (int qg = 0; qg < nqt; qg += TRI_THREADS / 64) {
        const int qt = qg + wave;
        const bool has_tile = qt < nqt;                          // (a wave without a tile still joins the chunk barriers)
        // ---- Q fragment (B operand of the swapped product): lane holds Q[q][kd*4 + g], pre-scaled
        const int qrow = qt * 16 + lq;
        const bool qok = has_tile && qrow < L;
        float qf[12];
        {
            const float* qp = a.q + base + (long long)(qok ? qrow : 0) * a.sl + g;
#pragma unroll
            for (int kd = 0; kd < 12; ++kd) qf[kd] = qok ? qp[kd * 4] * qscale : 0.f;
        }
        const float* brow = biasb ? biasb + (long long)(qok ? qrow : 0) * a.bias_sq : nullptr;
        float m_run = -INFINITY, l_run = 0.f;
        f32x4 o[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // one 64-key tile of the staged chunk (first key c0, nk keys).  FAST: the tile lies fully inside the chunk and no key is masked
        // -> no clamps, no mask arithmetic, LDS addresses are a per-tile base plus compile-time offsets.
        auto tile = [&](const int c0, const int nk, const int kt, auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const int k0 = kt * 64;                               // chunk-local
            // ---- bias of this tile: issued first so the loads fly under the QK^T MFMAs
            float bz[4][4];
            if (bias_vec && (FAST || k0 + 64 <= nk)) {           // 4 consecutive keys per lane -> one 16-B load (c0 % 64 == 0)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(brow + c0 + k0 + sub * 16 + g * 4);
                    bz[sub][0] = t4[0]; bz[sub][1] = t4[1]; bz[sub][2] = t4[2]; bz[sub][3] = t4[3];
                }
            } else {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = FAST ? k0 + sub * 16 + g * 4 + r : min(k0 + sub * 16 + g * 4 + r, nk - 1);
                        bz[sub][r] = brow ? brow[(long long)(c0 + key) * a.bias_sk] : 0.f;
                    }
            }
            f32x4 sc[4];
            // ---- S^T tiles: 4 sub-blocks of 16 keys
            const float* kbase = Ks + (k0 + lq) * LDK + g * 12;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const f32x4* kp = reinterpret_cast<const f32x4*>(
                    FAST ? kbase + sub * 16 * LDK : Ks + min(k0 + sub * 16 + lq, nk - 1) * LDK + g * 12);
                const f32x4 ka = kp[0], kb4 = kp[1], kc = kp[2];
                f32x4 c0a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kd = 0; kd < 4; ++kd) c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kd], qf[kd], c0a, 0, 0, 0);
#pragma unroll
                for (int kd = 0; kd < 4; ++kd) c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(kb4[kd], qf[4 + kd], c0a, 0, 0, 0);
#pragma unroll
                for (int kd = 0; kd < 4; ++kd) c0a = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[kd], qf[8 + kd], c0a, 0, 0, 0);
                sc[sub] = c0a;
            }
            // this lane: keys k0 + sub*16 + g*4 + r (r = 0..3) of query column lq
            float mx = -INFINITY;
            if (FAST) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = fmaf(bz[sub][r], bias_l2, sc[sub][r]);
                        sc[sub][r] = v;
                        mx = fmaxf(mx, v);
                    }
            } else {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + k0 + sub * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = fmaf(bz[sub][r], bias_l2, sc[sub][r]);
                        v = mk[r] == 0.f ? v : (mk[r] == 1.f ? ABX_NEG_MAX : -INFINITY);
                        sc[sub][r] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            }
            // ---- online softmax update of this query column
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // m_run = -inf on the first tile -> 0
            float rs = 0.f;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sc[sub][r] - m_new);   // arguments <= 0: raw v_exp_f32
                    sc[sub][r] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {                        // the running maximum rarely moves after the first tiles
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
            }
            // ---- O^T += V^T P : MFMA step (sub, r) contracts keys {k0 + sub*16 + g*4 + r : g = 0..3}
            const float* vbase = Vs + (k0 + g * 4) * LDV + lq;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* vp = FAST ? vbase + (sub * 16 + r) * LDV
                                           : Vs + min(k0 + sub * 16 + g * 4 + r, nk - 1) * LDV + lq;   // p == 0 for keys beyond the row
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[d * 16], sc[sub][r], o[d], 0, 0, 0);
                }
            }
        };
        for (int c = 0; c < nchunk; ++c) {
            const int c0 = c * KC, nk = min(KC, L - c0);
            if (nchunk > 1) {
                __syncthreads();                                // every wave has left the chunk staged before
                stage(c);
                __syncthreads();
            }
            if (has_tile) {
                const int nkt = (nk + 63) / 64, nfast = has_mask ? 0 : nk / 64;
                for (int kt = 0; kt < nfast; ++kt) tile(c0, nk, kt, std::true_type{});
                for (int kt = nfast; kt < nkt; ++kt) tile(c0, nk, kt, std::false_type{});
            }
        }
        // ---- normalise, gate, store.  O^T layout: column = query lq, rows d = dblk*16 + g*4 + r
        if (qok) {
            const float inv = 1.0f / l_run;
            const long long go = base + (long long)qrow * a.sl;
            float* op = a.out + (long long)b * a.ob + (long long)s * a.os + (long long)qrow * a.ol + h * TD;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int dd = d * 16 + g * 4;
                f32x4 v = o[d];
                if (a.gate) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + go + dd);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * inv * sigmoidf_(gv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= inv;
                }
                *reinterpret_cast<f32x4*>(op + dd) = v;
            }
        }
    }
}

// ---- triangle attention on the float16 matrix cores (split-f16, see gemm3.hip for the arithmetic) ----------------------------
// Same (b, row s, head h) decomposition, online softmax and swapped S^T product as tri_attn_kernel, with every fp32 product evaluated
// from 3 exact f16 products (v_mfma_f32_16x16x32_f16, fp32 accumulate): keys / values are the plane side (16 k, 16 v as p0, p1,
// p2 = p0 2^-11), queries and softmax weights the two-piece side (q / 16, P / 16 as a0, a1), so the products need no rescale.
//   K, V of the row are split ONCE, while they are staged, in chunks of 192 / 128 keys (p0, p1; p2 = p0 2^-11 is derived at the
//   fragment: two planes per operand leave room for 192-key chunks, i.e. two chunks and two barriers at L = 352), DOUBLE BUFFERED: the global loads of chunk
//   c + 1 are issued before the wave computes on chunk c and are split + written afterwards, ONE barrier per chunk.
//     K planes [2][key][48 d] f16 (96-byte rows)   -> A operand of S^T = K Q^T          (lane: key, 8 consecutive d; ds_read_b128)
//     V planes [2][key][48 d] f16 (same image)     -> A operand of O^T += V^T P         (lane: d, 8 keys) through the transposing
//                                                     LDS read ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 d] block
//                                                     and lane n receives its column n; no transposed staging writes
//   Q is split per query tile in registers; P per key tile in registers: the S^T accumulators of two 16-key sub-blocks (lane:
//   query lq, keys 4g + r) are exactly the 8 k-values lane (m = lq, k-group g) feeds to the A operand of one PV MFMA.
//   O^T[d][q] accumulators: column = query like the S^T tiles, so the running (max, sum) and the rescale are lane-local; the
//   four-lane reductions of a query column use v_permlane16/32_swap (VALU), never the LDS queue.
//   The pair bias is read in rows of bias_sq floats (bias_sk == 1: rows padded to a multiple of 4 -> 16-byte loads for any L).
//   Grid: 1-D, ordered so that the workgroups resident on one XCD walk the rows of ONE (b, h) pair: its (L, L) bias (495 KB at
//   L = 352) stays in that XCD's L2 instead of being re-fetched from HBM by every row.
constexpr int RST = 96;                  // bytes per key row of a plane: 32 * odd -> 16-byte fragment reads of 16 consecutive rows hit 64 banks

// two-piece split of a pre-scaled pair without the 2^11 lift of the remainder (tri_attn8_kernel: operands scaled so that it stays a
// normal float16 where it matters): x = p0 + p1 (+ <= 2^-24 |x|, or 2^-25 absolute below |x| = 2^-2)
__device__ __forceinline__ void split2h_ns(float a, float b, unsigned& p0, unsigned& p1) {
    // p0 = (f16(a), f16(b)); p1 = (f16(a - p0.lo), f16(b - p0.hi)): the differences are exact in fp32, so the mixed-precision FMA
    // (f16 source, f32 addend, f16 result: one rounding) gives the bits of convert - subtract - convert in 3 instructions instead of 5
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(p1) : "v"(p0), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p1) : "v"(p0), "v"(b));
}

// common.h split2b (planes p0 = f16(16 x), p1 = f16(16 x - p0)) on the mixed-precision FMA: 16 x - p0 is exact in fp32, so one
// rounding to f16 gives the bits of scale - convert - convert back - subtract - convert, in 4 instructions per pair instead of 6
// (the producer wave of tri_attn8_kernel is what the chunk barrier waits for).  c16 = 16.0f in a VGPR (no literal in VOP3P).
__device__ __forceinline__ void split2b_mix(float a, float b, float c16, unsigned& p0, unsigned& p1) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(p0) : "v"(a), "v"(c16));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(p0) : "v"(b), "v"(c16));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(p1) : "v"(a), "v"(c16), "v"(p0));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p1) : "v"(b), "v"(c16), "v"(p0));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// single-instruction max / min (the compiler otherwise canonicalises MFMA results with an extra v_max before fmaxf / fminf)
__device__ __forceinline__ float vmax(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float vmin(float x, float y) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__device__ __forceinline__ float vmax3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
// Reductions over the four lanes {l, l^16, l^32, l^48} (one query column of the swapped S^T tiles) on the VALU path:
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves in a few cycles, where __shfl_xor goes through the LDS
// queue (ds_bpermute) behind the K / V fragment reads of all 12 waves - four serial LDS round trips per key tile were the
// largest single cost of this kernel.
__device__ __forceinline__ float quad_max(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = vmax(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// KC4: keys per chunk; DB: double-buffered chunks (next chunk's loads ride under this chunk's compute, one barrier per chunk) or
// a single buffer (load -> split -> write between two barriers at every chunk start, but fewer, larger chunks)
template <int MAXQ, int KC4, bool DB, int NTH, bool PROD>
__global__ __launch_bounds__(NTH) void tri_attn4_kernel(const AbxTriAttn a) {
    constexpr int PLN = KC4 * RST;           // bytes per plane
    constexpr int BUF4 = 4 * PLN;            // K planes + V planes (p0, p1 each; p2 = p0 2^-11 is derived at the fragment) of one chunk
    constexpr int NIT = (KC4 * (TD / 4) + NTH - 1) / NTH;     // staging items per thread and chunk
    // PROD: the last wave is a PRODUCER - it alone stages the next chunk (loads, splits, LDS writes) while the other NCW waves
    // compute.  Vector-memory results return in issue order, so a consumer that also carried staging loads (HBM latency) waited for
    // them at its next bias use (L2 latency); with L = 352 the 22 query tiles are exactly 2 slots of 11 consumer waves.
    constexpr int NCW = NTH / 64 - (PROD ? 1 : 0);
    static_assert(!PROD || DB, "producer wave: double-buffered chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    float* Msb = reinterpret_cast<float*>(lds + (DB ? 2 : 1) * BUF4);       // [2][KC4] key-mask clamps of the chunks in flight
    const int L = a.L;
    // ---- (b, h, s) of this workgroup: XCD x (blockIdx & 7) owns the (b, h) pairs x, x + 8, ... and walks their rows in order
    // Long rows: the query tiles of a row are dealt to q_parts workgroups (each stages all keys of the row; neighbours in the grid, so
    // the row's K / V come from the L2 the second time): a wave never carries more than MAXQ tiles of online-softmax state.
    // When B * H is not a multiple of 8 (one sample: 4 pairs - half of the XCDs would idle) the rows of every pair are dealt to
    // row_groups = 2 virtual pairs (even / odd rows), which always gives a multiple of 8 with H = 4.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = a.row_groups, rows_g = (a.S + G - 1) / G;
    const int per_vp = rows_g * a.q_parts, sp = slot % per_vp;
    const int vp = (slot / per_vp) * 8 + xcd;
    const int bh = vp / G, s = (sp / a.q_parts) * G + vp % G, part = sp % a.q_parts;
    if (bh >= a.B * a.H || s >= a.S) return;
    const ClockProbe probe(a.clock_probe);
    const int b = bh / a.H, h = bh % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const long long base = (long long)b * a.sb + (long long)s * a.ss + (long long)h * TD;
    const float* km = a.keymask ? a.keymask + (long long)b * a.km_sb : nullptr;
    const float* biasb = a.bias ? a.bias + (long long)b * a.bias_sb + (long long)h * a.bias_sh : nullptr;
    const int nqt_row = (L + 15) / 16, tpp = (nqt_row + a.q_parts - 1) / a.q_parts;   // query tiles of the row / of one part
    const int qt0 = part * tpp, nqt = min(nqt_row, qt0 + tpp);
    const bool bias_vec = biasb && a.bias_sk == 1 && (a.bias_sq % 4 == 0) && ((reinterpret_cast<uintptr_t>(biasb) & 15) == 0);
    const int bias_row = (int)a.bias_sq;                        // readable floats per bias row when bias_sk == 1
    // split-f16 scales: keys / values are staged as 16 x, queries and softmax weights enter as x / 16, so the products need no rescale
    const float qscale = a.scale * LOG2E * 0.0625f;
    const float bias_l2 = a.bias_log2 ? 0.0078125f : LOG2E;     // (see tri_attn_kernel)
    // wave-uniform: does this sample mask any key?  (12 waves x 64 lanes cover L <= 768 in one pass of the ballot loop)
    bool any_masked = false;
    if (km) {
        for (int j = lane; j < L; j += 64) any_masked |= km[j] == 0.f;
        any_masked = __any(any_masked);
    }

    float m_run[MAXQ], l_run[MAXQ];
    f32x4 o[MAXQ][3];
#pragma unroll
    for (int sl = 0; sl < MAXQ; ++sl) {
        m_run[sl] = -INFINITY;
        l_run[sl] = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) o[sl][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // staging share of this thread: 2 (key, 4-channel) items of a chunk: 128 keys x 12 float4 = 1536 items over 768 threads.
    // Item `it` of the NEXT chunk is loaded before and written after the wave's query-tile slot `it` of the current chunk (the
    // other buffer was last read before the previous barrier), so only one item is live in registers at a time.
    f32x4 kreg, vreg;
    auto stage_load = [&](int c0, int idx) {
        const int kk = idx / (TD / 4), c4 = idx % (TD / 4);
        kreg = (f32x4){0.f, 0.f, 0.f, 0.f};
        vreg = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kk < KC4 && c0 + kk < L) {
            const long long off = base + (long long)(c0 + kk) * a.sl + c4 * 4;
            // read once per launch: non-temporal, so that the K / V stream does not push the Q rows (re-read at every chunk) and the
            // pair's bias out of the L2
            kreg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.k + off));
            vreg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.v + off));
        }
    };
    auto stage_mask = [&](int c0, int buf, int t) {
        Msb[buf * KC4 + t] = (c0 + t < L) ? ((!km || km[c0 + t] != 0.f) ? INFINITY : ABX_NEG_MAX) : -INFINITY;
    };
    auto stage_write = [&](int c0, int buf, int idx) {
        char* Kp = lds + buf * BUF4;
        char* Vp = Kp + 2 * PLN;
        const int kk = idx / (TD / 4), c4 = idx % (TD / 4);
        if (kk >= KC4) return;
        unsigned a0, a1, b0, b1;
        split2b(kreg[0], kreg[1], a0, a1);
        split2b(kreg[2], kreg[3], b0, b1);
        char* kd = Kp + kk * RST + c4 * 8;
        *reinterpret_cast<u32x2*>(kd) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(kd + PLN) = u32x2{a1, b1};
        split2b(vreg[0], vreg[1], a0, a1);
        split2b(vreg[2], vreg[3], b0, b1);
        char* vd = Vp + kk * RST + c4 * 8;
        *reinterpret_cast<u32x2*>(vd) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(vd + PLN) = u32x2{a1, b1};
    };
    // (key-mask clamps are applied as logit = min(logit, clamp): +inf valid, finfo.min masked (the reference REPLACES the logit by
    // finfo.min: every finite logit is >= finfo.min, so min() is that replacement), -inf beyond L)

    if (DB) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { stage_load(0, tid + it * NTH); stage_write(0, 0, tid + it * NTH); }
        if (tid < KC4) stage_mask(0, 0, tid);
        __syncthreads();
    }

    const int nchunk = (L + KC4 - 1) / KC4;
    // the two on-demand global streams of a consumer wave: the raw Q rows of a query tile, the bias of a key tile.  (Requesting them
    // one slot / one tile ahead was measured: no gain with 8 waves of 256 registers, spills with 12 waves - tools/probes/kb_tri.py)
    f32x4 qraw[4], bz[4];
    auto q_load = [&](int qt_) {
        const int qrow_ = qt_ * 16 + lq;
        const bool ok_ = qrow_ < L;
        const float* qp = a.q + base + (long long)(ok_ ? qrow_ : 0) * a.sl;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const bool live = ok_ && (hh == 0 || g < 2);            // d 32..47 only: lane groups 2, 3 of the second step are 0
            qraw[2 * hh] = live ? *reinterpret_cast<const f32x4*>(qp + hh * 32 + g * 8) : (f32x4){0.f, 0.f, 0.f, 0.f};
            qraw[2 * hh + 1] = live ? *reinterpret_cast<const f32x4*>(qp + hh * 32 + g * 8 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // bias of the tile (query tile qt_, chunk at key c0_, keys k0_ + sub*16 + 4g + r of nkeys_ in the chunk)
    auto bias_load = [&](int qt_, int c0_, int nkeys_, int k0_) {
        const int qrow_ = qt_ * 16 + lq;
        const float* brow = biasb ? biasb + (long long)(qrow_ < L ? qrow_ : 0) * a.bias_sq + (long long)c0_ * a.bias_sk : nullptr;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int kq = k0_ + sub * 16 + g * 4;                  // first of this lane's 4 keys (chunk-relative)
            if (bias_vec && c0_ + kq + 4 <= bias_row) {
                bz[sub] = *reinterpret_cast<const f32x4*>(brow + kq);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = min(kq + r, nkeys_ - 1);
                    bz[sub][r] = brow ? brow[(long long)key * a.bias_sk] : 0.f;
                }
            }
        }
    };
    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = ch * KC4, buf = DB ? (ch & 1) : 0;
        const int nkeys = min(KC4, L - c0);
        const int nkt = (nkeys + 63) / 64;
        const bool more = DB && ch + 1 < nchunk;
        if (!DB) {
            __syncthreads();                                    // the previous chunk has been consumed
#pragma unroll
            for (int it = 0; it < NIT; ++it) { stage_load(c0, tid + it * NTH); stage_write(c0, 0, tid + it * NTH); }
            if (tid < KC4) stage_mask(c0, 0, tid);
            __syncthreads();
        }
        const char* Kp = lds + buf * BUF4;
        const char* Vp = Kp + 2 * PLN;
        const float* Ms = Msb + buf * KC4;
        const bool masked_chunk = any_masked;                   // (per sample: a masked key anywhere -> clamp every tile)

        if (PROD && wave == NCW) {
            // ---- producer wave: the whole next chunk, 4 (key, 4-channel) items per lane in flight
            if (more) {
                constexpr int NPI = KC4 * (TD / 4) / 64;            // items per lane
                f32x4 kr[4], vr[4];
                for (int j0 = 0; j0 < NPI; j0 += 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { stage_load(c0 + KC4, lane + (j0 + j) * 64); kr[j] = kreg; vr[j] = vreg; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { kreg = kr[j]; vreg = vr[j]; stage_write(c0 + KC4, buf ^ 1, lane + (j0 + j) * 64); }
                }
                for (int t = lane; t < KC4; t += 64) stage_mask(c0 + KC4, buf ^ 1, t);
            }
        } else
#pragma unroll
        for (int sl = 0; sl < MAXQ; ++sl) {
            const int qt = qt0 + wave + sl * NCW;
            // (no producer wave) staging item sl of the next chunk rides along with slot sl (slots >= NIT carry none; a wave without a
            // query tile in this slot still does its share)
            if (!PROD && sl < NIT && more) stage_load(c0 + KC4, tid + sl * NTH);
            if (qt < nqt) {
            // ---- Q fragments (B operand of the swapped product), pre-scaled, split: lane holds Q[q][dbase + 8g .. +7]
            f16x8 qf[2][2];
            q_load(qt);
            {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float x[8];
                    const f32x4 lo = qraw[2 * hh], hi = qraw[2 * hh + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = lo[e] * qscale; x[4 + e] = hi[e] * qscale; }
                    unsigned q0[4], q1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split2h_raw(x[2 * e], x[2 * e + 1], q0[e], q1[e]);
                    qf[hh][0] = __builtin_bit_cast(f16x8, u32x4{q0[0], q0[1], q0[2], q0[3]});
                    qf[hh][1] = __builtin_bit_cast(f16x8, u32x4{q1[0], q1[1], q1[2], q1[3]});
                }
            }
            float mr = m_run[sl], lr = l_run[sl];
            f32x4 oo[3] = {o[sl][0], o[sl][1], o[sl][2]};
            using T = SplitTerms;                         // A: the two-piece operand (Q, P), B: the plane operand (K, V)

            for (int kt = 0; kt < nkt; ++kt) {
                const int k0 = kt * 64;
                // ---- bias of this tile (keys k0 + sub*16 + 4g + r): issued first, consumed after the QK^T MFMAs
                bias_load(qt, c0, nkeys, k0);
                f32x4 sc[4];
                // ---- S^T tiles: 4 sub-blocks of 16 keys x 2 d-steps x 6 products
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const char* kr = Kp + (k0 + sub * 16 + lq) * RST + g * 16;
                    f16x8 ka[2][3];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        ka[0][p] = *reinterpret_cast<const f16x8*>(kr + p * PLN);
                        // d 32..47: lane groups 0, 1 read d 32 + 8g; groups 2, 3 re-read a valid address and are multiplied by Q = 0
                        ka[1][p] = *reinterpret_cast<const f16x8*>(kr + p * PLN + 64 - (g >> 1) * 32);
                    }
                    ka[0][2] = __builtin_bit_cast(f16x8, f16x8_lo(__builtin_bit_cast(u32x4, ka[0][0])));
                    ka[1][2] = __builtin_bit_cast(f16x8, f16x8_lo(__builtin_bit_cast(u32x4, ka[1][0])));
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int term = 0; term < T::N; ++term)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[hh][T::B[term]], qf[hh][T::A[term]], c, 0, 0, 0);
                    sc[sub] = c;
                }
                // ---- bias, mask, online softmax (base 2) of this lane's query column
                float mx;
                // logits (base 2) = S^T + bias * log2(e); the key clamps only where the tile has masked / padded keys
                const bool clamp = masked_chunk || k0 + 64 > nkeys;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[sub][r] = fmaf(bz[sub][r], bias_l2, sc[sub][r]);
                }
                if (clamp) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + k0 + sub * 16 + g * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[sub][r] = vmin(sc[sub][r], mk[r]);
                    }
                }
                {   // tree maximum of the lane's 16 logits, then over the 4 lanes of the query column
                    const float m0 = vmax3(sc[0][0], sc[0][1], sc[0][2]), m1 = vmax3(sc[0][3], sc[1][0], sc[1][1]);
                    const float m2 = vmax3(sc[1][2], sc[1][3], sc[2][0]), m3 = vmax3(sc[2][1], sc[2][2], sc[2][3]);
                    const float m4 = vmax3(sc[3][0], sc[3][1], sc[3][2]);
                    mx = vmax(vmax3(m0, m1, m2), vmax3(m3, m4, sc[3][3]));
                }
                mx = quad_max(mx);
                const float m_new = vmax(mr, mx);
                const float alpha = __builtin_amdgcn_exp2f(mr - m_new);
                const float m_sh = m_new + 4.0f;
                float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sc[sub][r] - m_sh);     // P / 16
                        sc[sub][r] = p;
                        rs4[r] += p;
                    }
                const float rs = quad_sum((rs4[0] + rs4[1]) + (rs4[2] + rs4[3]));
                lr = lr * alpha + rs;
                mr = m_new;
                if (!__all(alpha == 1.0f)) {                        // O^T columns are the queries: lane-local rescale
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oo[d][r] *= alpha;
                }
                // ---- O^T += V^T P: one MFMA step contracts the 32 keys of two sub-blocks; A = V^T (transposing reads of the row-major
                // V planes: lane d receives 8 keys of its column), B = P (registers)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f16x8 pa[2];
                    unsigned p0[4], p1[4];
                    split2h_raw(sc[2 * m][0], sc[2 * m][1], p0[0], p1[0]);
                    split2h_raw(sc[2 * m][2], sc[2 * m][3], p0[1], p1[1]);
                    split2h_raw(sc[2 * m + 1][0], sc[2 * m + 1][1], p0[2], p1[2]);
                    split2h_raw(sc[2 * m + 1][2], sc[2 * m + 1][3], p0[3], p1[3]);
                    pa[0] = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p0[2], p0[3]});
                    pa[1] = __builtin_bit_cast(f16x8, u32x4{p1[0], p1[1], p1[2], p1[3]});
                    // lane i of a 16-lane group supplies row (4g + i/4) of the sub-block, 4 channels (i%4)*4.. of the 16-channel block
                    const char* vr = Vp + (k0 + m * 32 + 4 * g + (lq >> 2)) * RST + (lq & 3) * 8;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        f16x8 vb[3];
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const s16x4 lo = lds_tr16(vr + p * PLN + d * 32);
                            const s16x4 hi = lds_tr16(vr + p * PLN + d * 32 + 16 * RST);
                            vb[p] = __builtin_bit_cast(f16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                        }
                        vb[2] = __builtin_bit_cast(f16x8, f16x8_lo(__builtin_bit_cast(u32x4, vb[0])));
#pragma unroll
                        for (int term = 0; term < T::N; ++term)
                            oo[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[T::B[term]], pa[T::A[term]], oo[d], 0, 0, 0);
                    }
                }
            }
            m_run[sl] = mr;
            l_run[sl] = lr;
            o[sl][0] = oo[0]; o[sl][1] = oo[1]; o[sl][2] = oo[2];
            }
            if (!PROD && sl < NIT && more) {
                stage_write(c0 + KC4, buf ^ 1, tid + sl * NTH);
                if (sl == 0 && tid < KC4) stage_mask(c0 + KC4, buf ^ 1, tid);
            }
        }
        if (DB) __syncthreads();
    }
    if (a.range_flag) {                                         // range safety, see tri_attn8_kernel
        float z = 0.f;
#pragma unroll
        for (int sl = 0; sl < MAXQ; ++sl) {
            const int qt = qt0 + wave + sl * NCW;
            const bool live = wave < NCW && qt < nqt && qt * 16 + lq < L;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) z = fmaf(live ? o[sl][d][r] : 0.f, 0.f, z);
            z = fmaf(live ? l_run[sl] : 0.f, 0.f, z);
        }
        if (__any(z != z) && lane == 0) atomicOr(a.range_flag, a.range_tag);
    }
    // ---- normalise, gate, store.  O^T layout: column = query lq, rows d = dblk*16 + g*4 + r
#pragma unroll
    for (int sl = 0; sl < MAXQ; ++sl) {
        const int qt = qt0 + wave + sl * NCW;
        const int qrow = qt * 16 + lq;
        if (wave >= NCW || qt >= nqt || qrow >= L) continue;
        const float inv = 0.0625f / l_run[sl];                  // l_run accumulated P / 16
        const long long go = base + (long long)qrow * a.sl;
        float* op = a.out + (long long)b * a.ob + (long long)s * a.os + (long long)qrow * a.ol + h * TD;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int dd = d * 16 + g * 4;
            f32x4 v = o[sl][d];
            if (a.gate) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + go + dd);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * inv * sigmoidf_(gv[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= inv;
            }
            *reinterpret_cast<f32x4*>(op + dd) = v;
        }
    }
    probe.finish();
}

// ---- triangle attention, paired query tiles (split-f16) ----------------------------------------------------------------------
// Same (b, row, head) decomposition and LDS image as tri_attn4_kernel (K / V of a row staged in key chunks as two f16 planes each,
// swapped S^T = K Q^T, transposing V reads, online base-2 softmax per query column).  What is different:
//   * PERSISTENT workgroups: the grid is one workgroup per CU, workgroup j of XCD x walks the slots j, j + 32, ... of that XCD's
//     (b, h, row) list (so the 32 CUs of an XCD still work on neighbouring rows of ONE (b, h) pair: its bias stays in that L2), and
//     the PRODUCER wave streams key chunks across row boundaries: while the 11 computing waves work on the last chunk of a row it
//     stages the first chunk of the next one.  One workgroup owns a CU (99 / 147 KB of LDS), so in the one-row-per-workgroup form
//     nothing ran under a row's prologue (K / V of chunk 0 from HBM, first barrier) and under the dispatch of the next workgroup:
//     a third of the kernel.
//   * a computing wave walks its TWO query tiles (2w, 2w + 1 of the row's <= 22) together over the key tiles: every K / V fragment
//     read serves two S^T / O^T tiles (half the LDS fragment traffic per product) and the matrix pipe sees two independent
//     accumulator chains (a wave whose second tile lies beyond the row computes it on the clamped last query row and drops it);
//   * the Q fragments of both tiles are loaded and split ONCE per row;
//   * the pair bias is the INITIAL VALUE of the S^T accumulators: the loads of the next key tile's bias go into the registers the
//     softmax weights of this tile have just left, under the PV matrix work;
//   * operand scales chosen so that the second piece of a query / softmax weight needs no 2^11 lift: the three product terms use the
//     stored planes p0, p1 as they are (no p2 = p0 2^-11 derivation);
//   * the 48-wide head is 32 + 16 channels: the second k-step carries BOTH plane terms of channels 32..47 (lane groups 0, 1 read
//     p0, groups 2, 3 read p1 of the same 16 channels, the query piece a0 is held twice): 5 matrix instructions per 16-key
//     sub-block instead of 6, 3 fragment reads instead of 4;
//   * a last key tile with <= 32 keys runs half a tile.
// The kernel is instruction-issue bound (one paired tile step: 76 MFMA + ~300 VALU + 36 LDS per wave, three waves per SIMD).
// Rows with more than 22 query tiles are dealt to q_parts slots.
// BVEC: the bias rows are key-contiguous, 16-byte aligned and padded to a multiple of 4 floats (what model/forward.py passes): one
// 16-byte load per sub-block; otherwise (any strides, or no bias) the generic element loads
template <int KC4, int NTH, bool BVEC>
__global__ __launch_bounds__(NTH) void tri_attn8_kernel(const AbxTriAttn a) {
    constexpr int PLN = KC4 * RST;           // bytes per plane
    constexpr int BUF4 = 4 * PLN;            // K planes + V planes (p0, p1) of one chunk
    constexpr int NCW = NTH / 64 - 1;        // computing waves; the last wave is the producer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    float* Msb = reinterpret_cast<float*>(lds + 2 * BUF4);       // [2][KC4] key-mask clamps of the chunks in flight
    const int L = a.L;
    const ClockProbe probe(a.clock_probe);
#ifdef TRI8_STAMP
    // diagnostic build: shader-clock stamps of wave 0 (slots 0..15) and of the producer wave (16..31) while the workgroup is on its
    // FOURTH row (steady state), one record per workgroup
    unsigned long long* st_buf = a.clock_probe ? a.clock_probe + 16 + (size_t)blockIdx.x * 32 : nullptr;
    int st_n = 0, st_row = 0;
#define STAMP() { if (st_buf && st_row == 3 && (threadIdx.x & 63) == 0 && st_n < 16) st_buf[(threadIdx.x >= NCW * 64 ? 16 : 0) + st_n] = __builtin_amdgcn_s_memtime(); if (st_row == 3) ++st_n; }
#else
#define STAMP() {}
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g = lane >> 4;
    // ---- the slots of this workgroup: XCD x = blockIdx & 7 owns the virtual pairs x, x + 8, ...; a slot is (pair, row, part)
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, NW = gridDim.x >> 3;
    const int G = a.row_groups, rows_g = (a.S + G - 1) / G;
    const int per_vp = rows_g * a.q_parts;
    const int nvp = a.B * a.H * G;
    const long long nslots = (long long)((nvp - xcd + 7) / 8) * per_vp;          // slots of this XCD
    struct Row { long long base; const float* km; const float* biasb; int b, h, s, part; };
    auto decode = [&](long long slot, Row& r) __attribute__((always_inline)) -> bool {
        const int sp = (int)(slot % per_vp), vp = (int)(slot / per_vp) * 8 + xcd;
        const int bh = vp / G;
        r.s = (sp / a.q_parts) * G + vp % G;
        r.part = sp % a.q_parts;
        if (r.s >= a.S) return false;                                              // (ragged last row group)
        r.b = bh / a.H;
        r.h = bh % a.H;
        // (wave-uniform, but the divisions above run on the vector unit: back to scalar registers - the producer wave holds two rows)
        r.b = __builtin_amdgcn_readfirstlane(r.b);
        r.h = __builtin_amdgcn_readfirstlane(r.h);
        r.s = __builtin_amdgcn_readfirstlane(r.s);
        r.part = __builtin_amdgcn_readfirstlane(r.part);
        r.base = (long long)r.b * a.sb + (long long)r.s * a.ss + (long long)r.h * TD;
        r.km = a.keymask ? a.keymask + (long long)r.b * a.km_sb : nullptr;
        r.biasb = a.bias ? a.bias + (long long)r.b * a.bias_sb + (long long)r.h * a.bias_sh : nullptr;
        return true;
    };
    auto next_slot = [&](long long slot, Row& r) __attribute__((always_inline)) -> long long {       // first valid slot >= slot, or -1
        for (; slot < nslots; slot += NW)
            if (decode(slot, r)) return slot;
        return -1;
    };
    const int nchunk = (L + KC4 - 1) / KC4;
    const int nqt_row = (L + 15) / 16, tpp = (nqt_row + a.q_parts - 1) / a.q_parts;

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;

    if (wave == NCW) {
        // ================= producer wave: K / V of the chunk after the one being computed, across row boundaries ==============
        // (the last-dispatched wave of the workgroup loses every VALU / LDS issue arbitration against the older computing waves of
        // its SIMD, and it is the one the chunk barrier waits for: static priority)
        __builtin_amdgcn_s_setprio(3);
        // per-lane constants of the staging: item j of a round is (key kk0[j] + 32 rd, 4-channel group c4) with idx = lane + 64 j =
        // 12 kk0 + c4 (a round of 6 items advances every key by 384 / 12 = 32): LDS byte offset inside a plane, byte offset inside a row
        constexpr int NPI = KC4 * (TD / 4) / 64;                // (key, 4-channel) items per lane: 36 (192-key chunks) / 24
        constexpr int NPF = 6, NRD = NPI / NPF;                 // rounds of 6 items (12 loads of 16 bytes per lane)
        static_assert(NPI % NPF == 0 && NRD % 2 == 0 && (NPF * 64) % (TD / 4) == 0, "rounds");
        constexpr int KRD = NPF * 64 / (TD / 4);                // keys a round advances: 32
        const unsigned sl4 = (unsigned)(a.sl * 4);             // row stride in bytes (< 2^24, rows of one (b, s) slab < 4 GB: the dispatch)
        float c16 = 16.0f;                                      // (VOP3P takes no literal; a VGPR: see split2b_mix)
        asm volatile("" : "+v"(c16));
        auto stage = [&](const Row& r, int c0, int buf) __attribute__((always_inline)) {
            char* Kp = lds + buf * BUF4;
            char* Vp = Kp + 2 * PLN;
            if (a.kv_planes) {
                // Round 6: k | v arrive as the operand images the projection wrote (AbxGemm.c_planes_from: per key and head [p0: 48 f16 | p1: 48
                // f16] of 16 x value, at the byte address of the fp32 head slice).  A plane of the chunk is a dense [KC4][96 bytes] array in
                // LDS, so staging is a straight DMA: instruction i of a plane fills bytes [1024 i, 1024 i + 1024) - lane l the 16 bytes at
                // o = 1024 i + 16 l = (key o / 96, byte o % 96); three instructions cover 32 keys exactly, so a lane has three (key, byte)
                // constants.  No registers, no split, no LDS writes by this wave: 4 planes x PLN / 1024 instructions per chunk, one wait.
                constexpr int NI = PLN / 1024;
                static_assert(PLN % 3072 == 0, "a plane of the chunk is a whole number of 32-key DMA rounds");
                int ln = lane;
                asm volatile("" : "+v"(ln));                     // (per chunk, behind an opaque lane id: see the staging constants below)
                const char* kb = reinterpret_cast<const char*>(a.k + r.base);
                const char* vb = reinterpret_cast<const char*>(a.v + r.base);
                int kt[3];
                unsigned cb[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int o = t * 1024 + ln * 16;
                    kt[t] = o / RST;
                    cb[t] = (unsigned)(o - kt[t] * RST);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const unsigned off = __umul24((unsigned)min(c0 + 32 * (i / 3) + kt[i % 3], L - 1), sl4) + cb[i % 3];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + off + p * RST),
                                                         (__attribute__((address_space(3))) void*)(Kp + p * PLN + i * 1024), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + off + p * RST),
                                                         (__attribute__((address_space(3))) void*)(Vp + p * PLN + i * 1024), 16, 0, 0);
                    }
                }
                for (int t = lane; t < KC4; t += 64)
                    Msb[buf * KC4 + t] = (c0 + t < L) ? ((!r.km || r.km[c0 + t] != 0.f) ? INFINITY : ABX_NEG_MAX) : -INFINITY;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the images have landed before this wave joins the chunk barrier)
                return;
            }
            // (recomputed per chunk behind an opaque lane id: as invariants of the whole kernel they are hoisted above the wave-role
            // branch and cost the computing waves 18 registers - spills)
            int ln = lane;
            asm volatile("" : "+v"(ln));
        int kk0[NPF];
        unsigned ldo[NPF], gco[NPF];
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            const int idx = ln + j * 64;
            kk0[j] = idx / (TD / 4);
            ldo[j] = (unsigned)(kk0[j] * RST + (idx % (TD / 4)) * 8);
            gco[j] = (unsigned)((idx % (TD / 4)) * 16);
        }
            // wave-uniform slab bases + 32-bit per-lane offsets (scalar-base loads: no 64-bit vector address arithmetic per item)
            const char* kb = reinterpret_cast<const char*>(a.k + r.base);
            const char* vb = reinterpret_cast<const char*>(a.v + r.base);
            // software pipeline over the rounds: the loads of round rd + 1 are in flight while round rd is split and written
            f32x4 kr[2][NPF], vr[2][NPF];
            auto issue = [&](int rd, auto set_) __attribute__((always_inline)) {
                constexpr int set = decltype(set_)::value;
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    // UNCONDITIONAL loads (a conditional one makes the number of loads in flight unknown to the compiler: it then waits
                    // for vmcnt(0) before the previous round's values, and the pipeline is gone).  Keys beyond L re-read the last key:
                    // their logits are clamped to -inf (Msb), their softmax weights are exactly 0 against a finite V row
                    const unsigned off = __umul24((unsigned)min(c0 + rd * KRD + kk0[j], L - 1), sl4) + gco[j];
                    // read once per launch: non-temporal, so that the K / V stream does not push the pair's bias out of the L2
#ifdef TRI8_TEMPORAL                                                    // (probe: plain loads)
                    kr[set][j] = *reinterpret_cast<const f32x4*>(kb + off);
                    vr[set][j] = *reinterpret_cast<const f32x4*>(vb + off);
#else
                    kr[set][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kb + off));
                    vr[set][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vb + off));
#endif
                }
            };
            auto convert = [&](int rd, auto set_) __attribute__((always_inline)) {
                constexpr int set = decltype(set_)::value;
                const unsigned rdo = (unsigned)(rd * KRD * RST);
#ifdef TRI8_LOADS_ONLY     // (probe build: every K / V load still issues and is waited for, but nothing is split or written except one word per
                           //  round - the kernel's time with the HBM stream and without the producer's VALU / LDS work: what DMA staging could return)
                unsigned x = 0;
#pragma unroll
                for (int j = 0; j < NPF; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x ^= __builtin_bit_cast(unsigned, kr[set][j][e]) ^ __builtin_bit_cast(unsigned, vr[set][j][e]);
                *reinterpret_cast<unsigned*>(Kp + rdo + ldo[0]) = x;
                return;
#endif
#pragma unroll
                for (int j = 0; j < NPF; ++j) {
                    unsigned a0, a1, b0, b1;
                    split2b_mix(kr[set][j][0], kr[set][j][1], c16, a0, a1);
                    split2b_mix(kr[set][j][2], kr[set][j][3], c16, b0, b1);
                    char* kd = Kp + rdo + ldo[j];
                    *reinterpret_cast<u32x2*>(kd) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(kd + PLN) = u32x2{a1, b1};
                    split2b_mix(vr[set][j][0], vr[set][j][1], c16, a0, a1);
                    split2b_mix(vr[set][j][2], vr[set][j][3], c16, b0, b1);
                    char* vd = Vp + rdo + ldo[j];
                    *reinterpret_cast<u32x2*>(vd) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(vd + PLN) = u32x2{a1, b1};
                }
            };
#ifndef TRI8_NO_PRODUCER   // (probe build, tools/probes/build_noprod.sh: no K / V staging at all - wrong results, the kernel's time without
                           //  any producer work: the ceiling of what staging by DMA from pre-split planes could return)
            // (a real loop over round pairs; scheduling barriers: the compiler would hoist every round's loads to the top and spill)
            issue(0, I0{});
#pragma unroll 1
            for (int rd = 0; rd < NRD - 2; rd += 2) {           // (every issue unconditional: see above)
                issue(rd + 1, I1{});
                __builtin_amdgcn_sched_barrier(0);
                convert(rd, I0{});
                __builtin_amdgcn_sched_barrier(0);
                issue(rd + 2, I0{});
                __builtin_amdgcn_sched_barrier(0);
                convert(rd + 1, I1{});
                __builtin_amdgcn_sched_barrier(0);
            }
            issue(NRD - 1, I1{});
            __builtin_amdgcn_sched_barrier(0);
            convert(NRD - 2, I0{});
            __builtin_amdgcn_sched_barrier(0);
            convert(NRD - 1, I1{});
            __builtin_amdgcn_sched_barrier(0);
#endif
            // key clamps, applied as logit = min(logit, clamp): +inf valid, finfo.min masked (the reference REPLACES the logit by
            // finfo.min: every finite logit is >= finfo.min), -inf beyond L
            for (int t = lane; t < KC4; t += 64)
                Msb[buf * KC4 + t] = (c0 + t < L) ? ((!r.km || r.km[c0 + t] != 0.f) ? INFINITY : ABX_NEG_MAX) : -INFINITY;
        };
        Row cur, nxt;
        long long slot = next_slot(wg, cur);
        int cc = 0;                                             // chunks staged so far: chunk cc goes to buffer cc & 1
        if (slot >= 0) stage(cur, 0, 0);
        __syncthreads();
        while (slot >= 0) {
            const long long slot2 = next_slot(slot + NW, nxt);
            for (int ch = 0; ch < nchunk; ++ch) {
                ++cc;
                STAMP()
                if (ch + 1 < nchunk) stage(cur, (ch + 1) * KC4, cc & 1);
                else if (slot2 >= 0) stage(nxt, 0, cc & 1);
                STAMP()
                __syncthreads();
            }
            STAMP()
#ifdef TRI8_STAMP
            ++st_row;
#endif
            slot = slot2;
            cur = nxt;
        }
        probe.finish();
        return;
    }

    // ================= computing waves ======================================================================================
    // scales (powers of two: exact): keys / values are staged as 16 x (planes p0, p1 = f16(x'), f16(x' - p0)); queries enter as
    // q scale log2(e) 2^3, so the S^T accumulators hold 2^7 x the base-2 logits; softmax weights as P 2^8.  With these scales the
    // second piece of a query / weight, f16(x' - f16(x')), needs no 2^11 lift (a float16 subnormal only below |x'| = 2^-2, where its
    // 2^-25 absolute error is far below the rounding of the neighbouring products), so the three product terms a1 p0 + a0 p1 + a0 p0
    // use the stored planes as they are
    const float qscale = a.scale * LOG2E * 8.0f;
    constexpr float SCL = 128.0f, ISCL = 1.0f / 128.0f, PEXP = 8.0f;
    const int koff0 = lq * RST + g * 16;                                  // K fragment, channels 8g .. (plane p: + p PLN)
    const int koffc = lq * RST + (g >> 1) * PLN + 64 + (g & 1) * 16;      // channels 32 + 8(g&1) .. of plane g >> 1
    const int voff = (4 * g + (lq >> 2)) * RST + (lq & 3) * 8;            // transposing V reads (see tri_attn4_kernel)

    Row cur;
    long long slot = next_slot(wg, cur);
    int cc = 0;                                                 // chunks consumed so far
    bool any_masked = false;
    int b_masked = -1;
    __syncthreads();                                            // the first chunk of the first row is staged
    while (slot >= 0) {
        const int qt0 = cur.part * tpp, nqt = min(nqt_row, qt0 + tpp);
        const int qtA = qt0 + 2 * wave;
        const bool has_tile = qtA < nqt;                        // (a wave without tiles still joins the barriers)
        if (wave == 0) STAMP()
        if (cur.b != b_masked) {                                // wave-uniform: does this sample mask any key?  (once per sample: the
            any_masked = false;                                 // rows of a workgroup walk one (b, h) pair before the next)
            if (cur.km) {
                for (int j = lane; j < L; j += 64) any_masked |= cur.km[j] == 0.f;
                any_masked = __any(any_masked);
            }
            b_masked = cur.b;
        }
        // ---- Q fragments (B operand of the swapped product), pre-scaled, split once per row.
        // qf[X][0], [1]: pieces a0, a1 of channels 8g .. 8g+7 (first k-step); qf[X][2]: a0 of channels 32 + 8(g&1) .. (second k-step:
        // held by groups g and g + 2, against p0 | p1); qf[X][3]: a1 of the same channels in groups 0, 1, zero in groups 2, 3
        f16x8 qf[2][4];
        unsigned boff[2];                                        // byte offset of (query row, key 4g) inside the pair's bias (< 4 GB)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int qrow = min((qtA + X) * 16 + lq, L - 1);     // beyond the row: the last query row again (never stored)
            const float* qp = a.q + cur.base + (long long)qrow * a.sl;
            boff[X] = (unsigned)(((long long)qrow * a.bias_sq + g * 4) * 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp + g * 8);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qp + g * 8 + 4);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(qp + 32 + (g & 1) * 8);
            const f32x4 q3 = *reinterpret_cast<const f32x4*>(qp + 32 + (g & 1) * 8 + 4);
            unsigned p0[4], p1[4];
            split2h_ns(q0[0] * qscale, q0[1] * qscale, p0[0], p1[0]);
            split2h_ns(q0[2] * qscale, q0[3] * qscale, p0[1], p1[1]);
            split2h_ns(q1[0] * qscale, q1[1] * qscale, p0[2], p1[2]);
            split2h_ns(q1[2] * qscale, q1[3] * qscale, p0[3], p1[3]);
            qf[X][0] = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p0[2], p0[3]});
            qf[X][1] = __builtin_bit_cast(f16x8, u32x4{p1[0], p1[1], p1[2], p1[3]});
            split2h_ns(q2[0] * qscale, q2[1] * qscale, p0[0], p1[0]);
            split2h_ns(q2[2] * qscale, q2[3] * qscale, p0[1], p1[1]);
            split2h_ns(q3[0] * qscale, q3[1] * qscale, p0[2], p1[2]);
            split2h_ns(q3[2] * qscale, q3[3] * qscale, p0[3], p1[3]);
            qf[X][2] = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p0[2], p0[3]});
            qf[X][3] = g < 2 ? __builtin_bit_cast(f16x8, u32x4{p1[0], p1[1], p1[2], p1[3]}) : __builtin_bit_cast(f16x8, u32x4{0u, 0u, 0u, 0u});
        }
        // running maximum (in accumulator units) and the running sum of the softmax weights: the sums come off the MATRIX pipe as a
        // 49th output channel (A operand = ones: every row of ls is the column sum of the weight pieces the PV product uses) - 8 MFMA
        // issue slots per tile pair instead of the 48 VALU ones of the add tree + the four-lane reduction; the kernel is issue bound
        float mr[2] = {-INFINITY, -INFINITY};
        f32x4 ls[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        f32x4 oo[2][3];
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int d = 0; d < 3; ++d) oo[X][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // logits of the key tile in flight: bias -> 2^7 (bias log2 e + S^T) -> softmax weights P 2^8 -> (the next tile's bias)
        f32x4 sc[2][4];
        __amdgpu_buffer_rsrc_t brsrc;
        {
            const unsigned long long bu = reinterpret_cast<unsigned long long>(cur.biasb);
            // (readfirstlane returns int: through unsigned, or a low half with bit 31 set sign-extends into the high half)
            const unsigned long long bu_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bu >> 32)) << 32) |
                                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bu);
            brsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(bu_u), 0,
                                                      __builtin_amdgcn_readfirstlane((int)(L * a.bias_sq * 4)), 0x00020000);
        }
        // bias of sub-blocks [S0, S1) of the key tile that starts at key k_abs of the row, into sc
        auto bias_issue = [&](int k_abs, auto s0_, auto s1_) __attribute__((always_inline)) {
            constexpr int S0 = decltype(s0_)::value, S1 = decltype(s1_)::value;
            if (BVEC) {
                // buffer loads: the pair's (L, Lp) bias as the resource (wave-uniform descriptor), (query row, key) in the per-lane 32-bit
                // offset, the sub-block as immediate: no 64-bit vector address arithmetic.  Keys beyond the padded row read the next row's
                // floats (beyond the matrix: the range check returns 0): replaced by the -inf clamp
                // (the tile's first key goes into the per-lane offset, not the scalar one: only the former is range-checked)
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    const unsigned vo = boff[X] + (unsigned)k_abs * 4u;
#pragma unroll
                    for (int sub = S0; sub < S1; ++sub)
                        sc[X][sub] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, vo + sub * 64, 0, 0));
                }
            } else {
#pragma unroll
                for (int X = 0; X < 2; ++X)
#pragma unroll
                    for (int sub = S0; sub < S1; ++sub) {
                        const int kq = k_abs + sub * 16 + g * 4;
                        const float* brow = cur.biasb ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(cur.biasb) + boff[X]) - g * 4 : nullptr;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sc[X][sub][r] = brow ? brow[(long long)min(kq + r, L - 1) * a.bias_sk] : 0.f;
                    }
            }
        };
        if (has_tile) bias_issue(0, I0{}, I4{});
        if (wave == 0) STAMP()

        for (int ch = 0; ch < nchunk; ++ch, ++cc) {
            const int c0 = ch * KC4, buf = cc & 1;
            const int nkeys = min(KC4, L - c0);
            const int nkt = (nkeys + 63) / 64;
            const bool more = ch + 1 < nchunk;
            const char* Kp = lds + buf * BUF4;
            const char* Vp = Kp + 2 * PLN;
            const float* Ms = Msb + buf * KC4;
            if (has_tile)
            for (int kt = 0; kt < nkt; ++kt) {
                const int k0 = kt * 64;
                const bool full = nkeys - k0 > 32;              // otherwise: sub-blocks 0, 1 and the first PV step only
                // next key tile of this row (its bias is requested under the PV work below): first key, or -1
                const int k_next = kt + 1 < nkt ? c0 + k0 + 64 : (more ? c0 + KC4 : -1);
                if ((BVEC || cur.biasb) && !a.bias_log2) {      // (bias_log2: the projection applied the factor, AbxGemm.alpha)
#pragma unroll
                    for (int X = 0; X < 2; ++X)
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sc[X][sub][r] *= LOG2E * SCL;
                }
                // ---- S^T tiles (accumulators start from the bias): terms a1 p0, a0 p1, a0 p0, smallest first
                auto qk = [&](auto sub_) __attribute__((always_inline)) {
                    constexpr int sub = decltype(sub_)::value;
                    const char* kr = Kp + (k0 + sub * 16) * RST;
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(kr + koff0);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(kr + koff0 + PLN);
                    const f16x8 rc = *reinterpret_cast<const f16x8*>(kr + koffc);
#pragma unroll
                    for (int X = 0; X < 2; ++X) sc[X][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0, qf[X][1], sc[X][sub], 0, 0, 0);
#pragma unroll
                    for (int X = 0; X < 2; ++X) sc[X][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rc, qf[X][3], sc[X][sub], 0, 0, 0);
#pragma unroll
                    for (int X = 0; X < 2; ++X) sc[X][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r1, qf[X][0], sc[X][sub], 0, 0, 0);
#pragma unroll
                    for (int X = 0; X < 2; ++X) sc[X][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rc, qf[X][2], sc[X][sub], 0, 0, 0);
#pragma unroll
                    for (int X = 0; X < 2; ++X) sc[X][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r0, qf[X][0], sc[X][sub], 0, 0, 0);
                };
                qk(I0{});
                qk(I1{});
                if (full) {
                    qk(I2{});
                    qk(I3{});
                } else {
#pragma unroll
                    for (int X = 0; X < 2; ++X) {
                        sc[X][2] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        sc[X][3] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                }
                // ---- key clamps (masked / padded keys), online softmax (base 2) of this lane's query columns
                if (any_masked || k0 + 64 > nkeys) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + k0 + sub * 16 + g * 4);
#pragma unroll
                        for (int X = 0; X < 2; ++X)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sc[X][sub][r] = vmin(sc[X][sub][r], mk[r]);
                    }
                }
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    const float m0 = vmax3(sc[X][0][0], sc[X][0][1], sc[X][0][2]), m1 = vmax3(sc[X][0][3], sc[X][1][0], sc[X][1][1]);
                    const float m2 = vmax3(sc[X][1][2], sc[X][1][3], sc[X][2][0]), m3 = vmax3(sc[X][2][1], sc[X][2][2], sc[X][2][3]);
                    const float m4 = vmax3(sc[X][3][0], sc[X][3][1], sc[X][3][2]);
                    const float mx = quad_max(vmax(vmax3(m0, m1, m2), vmax3(m3, m4, sc[X][3][3])));
                    const float m_new = vmax(mr[X], mx);
                    const float alpha = __builtin_amdgcn_exp2f((mr[X] - m_new) * ISCL);
                    const float m_sh = PEXP - m_new * ISCL;
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[X][sub][r] = __builtin_amdgcn_exp2f(fmaf(sc[X][sub][r], ISCL, m_sh));     // P 2^8
                    mr[X] = m_new;
                    if (!__all(alpha == 1.0f)) {                    // O^T columns are the queries: lane-local rescale
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int r = 0; r < 4; ++r) oo[X][d][r] *= alpha;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ls[X][r] *= alpha;
                    }
                }
                // ---- O^T += V^T P: one step contracts the 32 keys of two sub-blocks; the bias of the next tile goes into the
                // registers the weights leave
                auto pv = [&](auto m_) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_)::value;
                    f16x8 pa[2][2];
#pragma unroll
                    for (int X = 0; X < 2; ++X) {
                        unsigned p0[4], p1[4];
                        split2h_ns(sc[X][2 * m][0], sc[X][2 * m][1], p0[0], p1[0]);
                        split2h_ns(sc[X][2 * m][2], sc[X][2 * m][3], p0[1], p1[1]);
                        split2h_ns(sc[X][2 * m + 1][0], sc[X][2 * m + 1][1], p0[2], p1[2]);
                        split2h_ns(sc[X][2 * m + 1][2], sc[X][2 * m + 1][3], p0[3], p1[3]);
                        pa[X][0] = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p0[2], p0[3]});
                        pa[X][1] = __builtin_bit_cast(f16x8, u32x4{p1[0], p1[1], p1[2], p1[3]});
                    }
                    if (k_next >= 0) bias_issue(k_next, std::integral_constant<int, 2 * m>{}, std::integral_constant<int, 2 * m + 2>{});
                    {
                        const _Float16 one = (_Float16)1.0f;
                        const f16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
                        for (int X = 0; X < 2; ++X) ls[X] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pa[X][1], ls[X], 0, 0, 0);
#pragma unroll
                        for (int X = 0; X < 2; ++X) ls[X] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pa[X][0], ls[X], 0, 0, 0);
                    }
                    const char* vr = Vp + (k0 + m * 32) * RST + voff;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        f16x8 vb[2];
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const s16x4 lo = lds_tr16(vr + p * PLN + d * 32);
                            const s16x4 hi = lds_tr16(vr + p * PLN + d * 32 + 16 * RST);
                            vb[p] = __builtin_bit_cast(f16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                        }
#pragma unroll
                        for (int X = 0; X < 2; ++X) oo[X][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0], pa[X][1], oo[X][d], 0, 0, 0);
#pragma unroll
                        for (int X = 0; X < 2; ++X) oo[X][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[1], pa[X][0], oo[X][d], 0, 0, 0);
#pragma unroll
                        for (int X = 0; X < 2; ++X) oo[X][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[0], pa[X][0], oo[X][d], 0, 0, 0);
                    }
                };
                pv(I0{});
                if (full) {
                    pv(I1{});
                } else if (k_next >= 0) {
                    bias_issue(k_next, I2{}, I4{});             // (a half tile is the last of its row: not reached)
                }
            }
            if (wave == 0) STAMP()
            __syncthreads();
            if (wave == 0) STAMP()
        }
        bool bad = false;           // range safety (AbxTriAttn.range_flag): a key / value / query / gate beyond the split ranges, or not finite
        // ---- normalise, gate, store.  O^T layout: column = query lq, rows d = dblk*16 + g*4 + r
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int qrow = (qtA + X) * 16 + lq;
            if (qtA + X >= nqt || qrow >= L) continue;
            const float inv = 0.0625f / ls[X][0];               // O^T accumulated (16 v) (P 2^8), ls = sum P 2^8 (every row)
            const long long go = cur.base + (long long)qrow * a.sl;
            float* op = a.out + (long long)cur.b * a.ob + (long long)cur.s * a.os + (long long)qrow * a.ol + cur.h * TD;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int dd = d * 16 + g * 4;
                f32x4 v = oo[X][d];
                if (a.gate) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gate + go + dd);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * inv * sigmoidf_(gv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= inv;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) bad |= __builtin_amdgcn_classf(v[r], 0x207);
                *reinterpret_cast<f32x4*>(op + dd) = v;
            }
        }
        if (a.range_flag && __any(bad) && lane == 0) atomicOr(a.range_flag, a.range_tag);
        if (wave == 0) STAMP()
#ifdef TRI8_STAMP
        ++st_row;
#endif
        slot = next_slot(slot + NW, cur);
    }
#undef STAMP
    probe.finish();
}

// ---- sequence attention with pair bias (seqformer.py:314-356): 32 heads x 17 channels, bias (b, h, q, k) ---------------------------
// The bias stream (L^2 floats per (b, h): 1.6 GB per launch at B = 100, L = 352) is the only real traffic, so the kernel is laid
// out around reading it coalesced: a wave takes 4 queries at a time and its LANES ARE KEYS (key = lane + 64 m), each bias row is
// read as contiguous 256-byte segments.  K and V of the (b, h) pair sit in LDS as [key][20] (17 channels + zero pad: 16-byte
// reads); the 4 x 17 query values are wave-uniform.
//   logits  lane: 17 FMAs per (query, key) against its K row (read once for the 4 queries), + bias, key mask
//   softmax per query over lanes (wave max / sum, base 2)
//   PV      the weights go through a wave-private LDS strip and the lanes regroup as (query, 4 channels, key quarter): one
//           16-byte V read + one weight read per 4 FMAs; the 17th channel as (query, key sixteenth); shuffle folds, fixed order
// NT threads share the K / V of one (b, h) pair: 16 waves (4 per SIMD, hiding the LDS / shuffle latencies) while the weight strips fit
template <int NK, int NT>
__global__ __launch_bounds__(NT) void seq_attn2_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                        const float* __restrict__ keymask, const float* __restrict__ gate,
                                                        float* __restrict__ out, int L, int H, float scale, int qblk) {
    constexpr int D = 17, DP = 20, QW = 4, LP = NK * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [L][DP]
    float* Vs = smem + (size_t)L * DP;      // [L][DP]
    float* pw = Vs + (size_t)L * DP;        // [waves][QW][LP]
    const int h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ld = (long long)H * 3 * D;
    const float* rowb = qkv + (long long)b * L * ld + (long long)h * 3 * D;
    for (int idx = tid; idx < L * DP; idx += NT) {
        const int key = idx / DP, d = idx - key * DP;
        Ks[idx] = d < D ? rowb[(long long)key * ld + D + d] : 0.f;
        Vs[idx] = d < D ? rowb[(long long)key * ld + 2 * D + d] : 0.f;
    }
    __syncthreads();
    const float qs = scale * LOG2E;
    const float* km = keymask ? keymask + (long long)b * L : nullptr;
    // per-lane key clamps: +inf valid, finfo.min masked (the reference REPLACES the logit: every finite logit is >= finfo.min),
    // -inf beyond L
    float clampv[NK];
#pragma unroll
    for (int m = 0; m < NK; ++m) {
        const int k = lane + 64 * m;
        clampv[m] = k < L ? ((!km || km[k] != 0.f) ? INFINITY : ABX_NEG_MAX) : -INFINITY;
    }
    float* pww = pw + (size_t)wave * QW * LP;
    const int q_end = min((int)(blockIdx.x + 1) * qblk, L);
    for (int qb = blockIdx.x * qblk + wave * QW; qb < q_end; qb += (NT / 64) * QW) {
        // ---- bias rows of the 4 queries (coalesced), issued first
        float bz[QW][NK];
#pragma unroll
        for (int qq = 0; qq < QW; ++qq) {
            const float* bp = bias + (((long long)b * H + h) * L + min(qb + qq, L - 1)) * L;
#pragma unroll
            for (int m = 0; m < NK; ++m) bz[qq][m] = bp[min(lane + 64 * m, L - 1)];
        }
        // ---- logits (base 2)
        float sc[QW][NK];
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            const float* kr = Ks + (size_t)min(lane + 64 * m, L - 1) * DP;
            f32x4 kv[DP / 4];
#pragma unroll
            for (int c = 0; c < DP / 4; ++c) kv[c] = *reinterpret_cast<const f32x4*>(kr + c * 4);
#pragma unroll
            for (int qq = 0; qq < QW; ++qq) {
                const float* qp = rowb + (long long)min(qb + qq, L - 1) * ld;          // wave-uniform
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) acc = fmaf(qp[d] * qs, kv[d >> 2][d & 3], acc);
                sc[qq][m] = fminf(fmaf(bz[qq][m], LOG2E, acc), clampv[m]);
            }
        }
        // ---- softmax over the keys (lanes x NK), the 4 queries interleaved
        float inv[QW];
#pragma unroll
        for (int qq = 0; qq < QW; ++qq) {
            float mx = sc[qq][0];
#pragma unroll
            for (int m = 1; m < NK; ++m) mx = fmaxf(mx, sc[qq][m]);
            mx = wave_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < NK; ++m) {
                const float pv = __builtin_amdgcn_exp2f(sc[qq][m] - mx);
                sc[qq][m] = pv;
                sm += pv;
            }
            inv[qq] = 1.0f / wave_sum(sm);
        }
#pragma unroll
        for (int qq = 0; qq < QW; ++qq)
#pragma unroll
            for (int m = 0; m < NK; ++m) pww[qq * LP + lane + 64 * m] = sc[qq][m];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- PV, channels 0..15: lane = (query, channel quad, key quarter)
        const int qq = lane >> 4, qi = qb + qq;
        const float myinv = qq == 0 ? inv[0] : (qq == 1 ? inv[1] : (qq == 2 ? inv[2] : inv[3]));
        const float* pr = pww + qq * LP;
        const long long orow = ((long long)b * L + min(qi, L - 1)) * H * D + (long long)h * D;
        {
            const int dg = (lane >> 2) & 3, kg = lane & 3;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k = kg; k < L; k += 4) {
                const float w = pr[k];
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(Vs + (size_t)k * DP + dg * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, v4[c], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] += __shfl_xor(acc[c], 1, 64);
                acc[c] += __shfl_xor(acc[c], 2, 64);
            }
            if (kg == 0 && qi < q_end) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = acc[c] * myinv;
                    if (gate) v *= sigmoidf_(gate[orow + dg * 4 + c]);
                    out[orow + dg * 4 + c] = v;
                }
            }
        }
        // ---- channel 16: lane = (query, key sixteenth)
        {
            const int kg = lane & 15;
            float acc = 0.f;
#pragma unroll 4
            for (int k = kg; k < L; k += 16) acc = fmaf(pr[k], Vs[(size_t)k * DP + 16], acc);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (kg == 0 && qi < q_end) {
                float v = acc * myinv;
                if (gate) v *= sigmoidf_(gate[orow + 16]);
                out[orow + 16] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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
    ABX_REQUIRE(!a.kv_planes || !a.exact, "abx_tri_attn_fwd: kv_planes goes with the split-f16 kernel (exact = 0)");
    if (!a.exact) {
        // split-f16 kernel: K / V staged in (double-buffered) key chunks, a wave keeps the online-softmax state of up to MAXQ query tiles
        // A workgroup carries 2 query-tile slots per wave: 24 tiles with 12 computing waves, 22 with 11 + the producer wave.  Rows with
        // more tiles are dealt to q_parts workgroups of (almost) equal share; the producer variant whenever the share fits its 22.
        // (Every query's arithmetic is the same in all variants: results are bit-identical.)  tune bit 0: never the producer wave.
        AbxTriAttn aa = a;
        const int nqt = (a.L + 15) / 16, nw = TRI_THREADS / 64;
        // tri_attn8_kernel: 11 computing waves x 2 query tiles walked together (its producer addresses a (b, s) slab of K / V with 32-bit
        // byte offsets and a 24-bit row stride; any other layout takes tri_attn4)
        const bool paired = !(a.tune & 4) && a.sl * 4 < (1LL << 24) && (long long)a.L * a.sl * 4 < (1LL << 32);
        ABX_REQUIRE(!a.kv_planes || paired, "abx_tri_attn_fwd: kv_planes (k | v as operand images) is served by tri_attn8_kernel only (tune bit 2 off, "
                                            "key stride below 2^24 bytes)");
        aa.q_parts = paired ? (nqt + 2 * (nw - 1) - 1) / (2 * (nw - 1)) : (nqt + 2 * nw - 1) / (2 * nw);
        const int tpp = (nqt + aa.q_parts - 1) / aa.q_parts;
        const bool prod = tpp <= 2 * (nw - 1) && !(a.tune & 1);
        aa.row_groups = ((long long)a.B * a.H) % 8 == 0 ? 1 : 2;
        const long long nvp8 = ((long long)a.B * a.H * aa.row_groups + 7) / 8 * 8, rows_g = (a.S + aa.row_groups - 1) / aa.row_groups;
        ABX_REQUIRE(nvp8 * rows_g * aa.q_parts < (1LL << 31), "abx_tri_attn_fwd: grid too large");
        // tri_attn4: one workgroup per (b, h, row, part) slot; tri_attn8: persistent, one workgroup per CU (8 XCDs x CUs / 8), each
        // walking the slots of its XCD with a stride of the workgroups per XCD
        static int n_cu = 0;
        if (paired && n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
                abx_set_error("abx_tri_attn_fwd: hipGetDeviceProperties failed");
                return ABX_ERR_ARG;
            }
            n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
        }
        const long long slots_xcd = nvp8 / 8 * rows_g * aa.q_parts;
        const unsigned wg_xcd = paired ? (unsigned)std::min<long long>(n_cu / 8, slots_xcd) : 0;
        const dim3 grid(paired ? 8 * wg_xcd : (unsigned)(nvp8 * rows_g * aa.q_parts)), block(TRI_THREADS);
        auto launch = [&](auto kern, size_t lds4) -> int {
            if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds4, "abx_tri_attn_fwd")) return rc;
            hipLaunchKernelGGL(kern, grid, block, lds4, st, aa);
            return abx_check_launch("abx_tri_attn_fwd");
        };
        auto lds_of = [](int kc, bool db) { return (size_t)(db ? 2 : 1) * (4 * kc * RST + kc * sizeof(float)); };
        if (paired) {
            const float* bb = a.bias;
            const bool bvec = bb && a.bias_sk == 1 && a.bias_sq % 4 == 0 && a.bias_sq >= 4 && a.bias_sb % 4 == 0 && a.bias_sh % 4 == 0 && al16(bb);
            // key chunks of 192 unless 128 gives the same number of chunks (then: less LDS to fill before the first tile, a less ragged
            // last chunk; L = 231: 2.52 vs 2.79 ms at 20 samples, L = 352: 21.2 vs 19.2 ms at 100).  tune bit 1 flips the choice.
            const bool kc128 = (((a.L + 127) / 128 == (a.L + 191) / 192) != ((a.tune & 2) != 0));
            if (kc128)
                return bvec ? launch(&tri_attn8_kernel<128, TRI_THREADS, true>, lds_of(128, true))
                            : launch(&tri_attn8_kernel<128, TRI_THREADS, false>, lds_of(128, true));
            return bvec ? launch(&tri_attn8_kernel<192, TRI_THREADS, true>, lds_of(192, true))
                        : launch(&tri_attn8_kernel<192, TRI_THREADS, false>, lds_of(192, true));
        }
        if (prod && (a.tune & 2)) return launch(&tri_attn4_kernel<2, 128, true, TRI_THREADS, true>, lds_of(128, true));   // benchmarking
        if (prod) return launch(&tri_attn4_kernel<2, 192, true, TRI_THREADS, true>, lds_of(192, true));
        return launch(&tri_attn4_kernel<2, 128, true, TRI_THREADS, false>, lds_of(128, true));
    }
    // exact fp32 kernel: the whole row's K / V in LDS when it fits (L <= 389), key chunks of 320 otherwise (no length limit)
    auto lds_of_exact = [](int kc) { return ((((size_t)kc * LDK + 3) & ~(size_t)3) + (size_t)kc * LDV + (size_t)((kc + 63) / 64) * 64 + 4) * sizeof(float); };
    const int kc = lds_of_exact(a.L) <= 160 * 1024 ? a.L : 320;      // (chunks start at multiples of 64: aligned bias loads)
    const size_t lds = lds_of_exact(kc);
    ABX_REQUIRE(lds <= 160 * 1024, "abx_tri_attn_fwd: internal: exact-kernel LDS layout");
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(tri_attn_kernel), 160 * 1024, "abx_tri_attn_fwd")) return rc;
    hipLaunchKernelGGL(tri_attn_kernel, dim3(a.H, a.S, a.B), dim3(TRI_THREADS), lds, st, a, kc);
    return abx_check_launch("abx_tri_attn_fwd");
}

extern "C" int abx_seq_attn_fwd(const float* qkv, const float* bias, const float* keymask, const float* gate, float* out,
                                int B, int L, int H, int D, float scale, hipStream_t st) {
    ABX_REQUIRE(qkv && bias && out && B > 0 && L > 0 && H > 0, "abx_seq_attn_fwd: bad args");
    ABX_REQUIRE(D == 17, "abx_seq_attn_fwd: head dim must be 17 (544 / 32)");
    ABX_REQUIRE(H <= 65535 && B <= 65535, "abx_seq_attn_fwd: grid too large");
    const int nk = (L + 63) / 64;
    ABX_REQUIRE(nk <= 12, "abx_seq_attn_fwd: L too large (L <= 716)");
    const int nkt = nk <= 2 ? 2 : (nk <= 4 ? 4 : (nk <= 6 ? 6 : (nk <= 8 ? 8 : 12)));       // keys per lane of the instantiation
    const int nt = nkt <= 6 ? 1024 : (nkt <= 8 ? 512 : 256);
    // one workgroup per (b, h) (every wave passes over 4 queries at a time); long complexes split the queries
    const int nb = (L + 511) / 512;
    const int qblk = (((L + nb - 1) / nb) + 3) / 4 * 4;
    const size_t lds = ((size_t)2 * L * 20 + (size_t)(nt / 64) * 4 * nkt * 64) * sizeof(float);
    ABX_REQUIRE(lds <= 160 * 1024, "abx_seq_attn_fwd: L too large for the LDS-resident K / V (L <= 716)");
    const dim3 grid((L + qblk - 1) / qblk, H, B), block(nt);
    auto launch = [&](auto kern, int) -> int {
        if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, "abx_seq_attn_fwd")) return rc;
        hipLaunchKernelGGL(kern, grid, block, lds, st, qkv, bias, keymask, gate, out, L, H, scale, qblk);
        return abx_check_launch("abx_seq_attn_fwd");
    };
    switch (nkt) {
        case 2: return launch(&seq_attn2_kernel<2, 1024>, 0);
        case 4: return launch(&seq_attn2_kernel<4, 1024>, 1);
        case 6: return launch(&seq_attn2_kernel<6, 1024>, 2);
        case 8: return launch(&seq_attn2_kernel<8, 512>, 3);
        default: return launch(&seq_attn2_kernel<12, 256>, 4);
    }
}
