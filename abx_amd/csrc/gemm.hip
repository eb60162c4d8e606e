// Generic batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain)
// with a fused LayerNorm and a fused epilogue.  One kernel serves every dense contraction of the score network
// (reference sites: every abx.model.common_modules.Linear on the hot path, the TriangleMultiplication einsum
// seqformer.py:490-493, the transitions seqformer.py:358-376).
//
//   acc[b][m][n] = sum_k A'[b][m][k] * B[b][k][n]                      A' = relu?(A)
//   LayerNorm over k (when ln_stats != NULL) is applied ALGEBRAICALLY in the epilogue: the caller passes B already scaled by
//   gamma (B[k][n] = gamma[k] W[n][k]), csum[n] = sum_k B[k][n] and bias[n] = sum_k beta[k] W[n][k] + b[n]; then
//       LN(A) W^T + b  ==  rstd[m] * (acc - mean[m] * csum[n]) + bias[n]
//   so the operand loads are plain 16-byte loads that stay in flight under the MFMAs.
//   epi(v) = ((ln(v) + bias[n]) * alpha) -> act -> * rowscale[m] -> * (sigmoid?)(gate[m][n]) -> + resid[m][n]
//
// Layout: A is k-contiguous (sAk==1) or m-contiguous (sAm==1); B is n-contiguous (sBn==1, the packed weight layout
// Wt[K][N]) or k-contiguous (sBk==1).  C is n-contiguous, or stored transposed (c_transposed: element (m,n) at
// C + b*sCb + n*sCm + m): the epilogue stages 32x32 sub-tiles through LDS so the transposed store is made of full 128-byte
// row segments — this is how the pair stack turns channel-major for the triangle-multiplication contraction without a
// separate transpose pass.
//
// Tiling: 256 threads = 4 waves; block tile BMxBNx16 staged through double-buffered LDS as [k][m] / [k][n] (+4 pad) so the
// MFMA operand reads (lane -> m, lane>>5 -> k) are conflict-free; tile t+1 is prefetched into registers while tile t is on
// the matrix cores; one barrier per k-tile.  Interior blocks run a guard-free path; edge blocks a fully predicated one.
// Block ids are remapped so that the N-tiles of one M-panel run on the same XCD (private L2) back to back.
#include <stdlib.h>
#include "common.h"
#include "abx_hip.h"
#include "gemm_epilogue.h"

int abx_gemm3_dispatch(const AbxGemm& g, hipStream_t st, int* rc);     // gemm3.hip

namespace {

template <int BMN, int BK, bool KC, bool EDGE>
struct TileLoader {
    static constexpr int NVEC = BMN * BK / 4;            // float4 slots in the tile
    static constexpr int NV = (NVEC + 255) / 256;        // per thread
    static constexpr int LD = BMN + 4;
    f32x4 v[NV];

    // base already offset by batch.  s_mn / s_k: element strides of the (m|n) and k dims.
    __device__ __forceinline__ void load(const float* __restrict__ base, long long s_mn, long long s_k, int mn0, int k0,
                                         int MN, int K, bool vec_ok) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (NVEC % 256 == 0 || idx < NVEC) {
                if (KC) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
                    const int mn = mn0 + row, k = k0 + kq * 4;
                    const float* p = base + (long long)mn * s_mn + k;
                    if (!EDGE) {
                        r = *reinterpret_cast<const f32x4*>(p);
                    } else if (mn < MN && k < K) {
                        if (vec_ok && k + 3 < K) {
                            r = *reinterpret_cast<const f32x4*>(p);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (k + c < K) r[c] = p[c];
                        }
                    }
                } else {
                    const int krow = idx / (BMN / 4), mq = idx % (BMN / 4);
                    const int k = k0 + krow, mn = mn0 + mq * 4;
                    const float* p = base + (long long)k * s_k + mn;
                    if (!EDGE) {
                        r = *reinterpret_cast<const f32x4*>(p);
                    } else if (k < K && mn < MN) {
                        if (vec_ok && mn + 3 < MN) {
                            r = *reinterpret_cast<const f32x4*>(p);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (mn + c < MN) r[c] = p[c];
                        }
                    }
                }
            }
            v[i] = r;
        }
    }

    // LayerNorm statistics of the A rows, accumulated from the operand registers while they wait for their LDS slot
    // (inline-LN mode).  KC: slot i = one row, sums of (x - shift[i]); RC: slots 0..3 = the four rows of this lane's float4.
    __device__ __forceinline__ void stats_accum(float* __restrict__ s, float* __restrict__ q, const float* __restrict__ shift,
                                                int k0, int K) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (NVEC % 256 == 0 || idx < NVEC) {
                if (KC) {
                    const int kq = idx % (BK / 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (!EDGE || k0 + kq * 4 + c < K) {
                            const float d = v[i][c] - shift[i];
                            s[i] += d;
                            q[i] = fmaf(d, d, q[i]);
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s[c] += v[i][c];
                        q[c] = fmaf(v[i][c], v[i][c], q[c]);
                    }
                }
            }
        }
    }

    // shift (KC, inline LayerNorm): the row's first element is subtracted from the operand so that the folded LayerNorm
    // rstd * (acc - (mean - shift) * csum) has no mean >> sigma cancellation
    __device__ __forceinline__ void store(float* __restrict__ lds, bool relu, const float* __restrict__ shift = nullptr) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (NVEC % 256 == 0 || idx < NVEC) {
                f32x4 r = v[i];
                if (KC && shift) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] -= shift[i];
                }
                if (relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = fmaxf(r[c], 0.f);
                }
                if (KC) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) lds[(kq * 4 + c) * LD + row] = r[c];
                } else {
                    const int krow = idx / (BMN / 4), mq = idx % (BMN / 4);
                    *reinterpret_cast<f32x4*>(&lds[krow * LD + mq * 4]) = r;
                }
            }
        }
    }
};

template <int BM, int BN, int WM, int WN, int BK, bool AKC, bool BNC, bool EDGE, bool TS>
__device__ __forceinline__ void gemm_block(const AbxGemm& g, float* smem, int mt, int nt, int b) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const float* Ab = g.A + (long long)b * g.sAb;
    const float* Bb = g.B + (long long)b * g.sBb;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<BM, BK, AKC, EDGE> la;
    TileLoader<BN, BK, !BNC, EDGE> lb;   // B n-contiguous == row-contiguous loader; B k-contiguous == KC loader
    const bool a_vec = g.a_vec_ok != 0, b_vec = g.b_vec_ok != 0, relu = g.a_relu != 0;
    const long long a_smn = g.sAm, a_sk = g.sAk, b_smn = g.sBn, b_sk = g.sBk;
    const int nk = (g.K + BK - 1) / BK;

    // inline LayerNorm statistics: computed from the A operand stream itself (no separate statistics pass over HBM)
    const bool ln_inline = g.ln_csum != nullptr && g.ln_stats == nullptr;
    float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f}, lshift[4] = {0.f, 0.f, 0.f, 0.f};

    la.load(Ab, a_smn, a_sk, m0, 0, g.M, g.K, a_vec);
    lb.load(Bb, b_smn, b_sk, n0, 0, g.N, g.K, b_vec);
    if (ln_inline) {
        if (AKC) {      // shift = first element of the row (held by the kq == 0 lane of the quad): kills the E[x^2]-mean^2 cancellation
#pragma unroll
            for (int i = 0; i < TileLoader<BM, BK, AKC, EDGE>::NV; ++i) lshift[i] = __shfl(la.v[i][0], lane & ~3, 64);
        }
        la.stats_accum(ls, lq, lshift, 0, g.K);
    }
    const float* shp = (ln_inline && AKC) ? lshift : nullptr;
    la.store(As, relu, shp);
    lb.store(Bs, false);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            la.load(Ab, a_smn, a_sk, m0, (t + 1) * BK, g.M, g.K, a_vec);
            lb.load(Bb, b_smn, b_sk, n0, (t + 1) * BK, g.N, g.K, b_vec);
        }
        const float* as = As + cur * BK * LDA + wm * WM + (lane & 31);
        const float* bs = Bs + cur * BK * LDB + wn * WN + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kr = kk * 2 + (lane >> 5);
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[kr * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = bs[kr * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) {
            if (ln_inline) la.stats_accum(ls, lq, lshift, (t + 1) * BK, g.K);
            la.store(As + (cur ^ 1) * BK * LDA, relu, shp);
            lb.store(Bs + (cur ^ 1) * BK * LDB, false);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // LDS is free now: stage the LayerNorm row statistics of this M-panel, then the per-wave store scratch behind them
    float* st_lds = smem;                                   // [BM][2]
    const float* gstats = g.ln_stats ? g.ln_stats + 2 * (long long)b * g.sSb : nullptr;
    const bool stats = gstats != nullptr || ln_inline;
    if (gstats) {
        for (int idx = threadIdx.x; idx < 2 * BM; idx += 256) {
            const int m = m0 + (idx >> 1);
            st_lds[idx] = (!EDGE || m < g.M) ? gstats[2 * (long long)m + (idx & 1)] : 0.f;
        }
        __syncthreads();
    } else if (ln_inline) {
        const float invK = 1.0f / (float)g.K;
        if (AKC) {
            constexpr int NVA = TileLoader<BM, BK, AKC, EDGE>::NV;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                float sm = ls[i], sq = lq[i];
                sm += __shfl_xor(sm, 1, 64); sq += __shfl_xor(sq, 1, 64);
                sm += __shfl_xor(sm, 2, 64); sq += __shfl_xor(sq, 2, 64);
                const int idx = threadIdx.x + i * 256;
                if ((idx & 3) == 0 && idx < BM * BK / 4) {
                    const int row = idx / (BK / 4);
                    const float dm = sm * invK;
                    st_lds[2 * row] = dm;                       // the operand was shifted: only (mean - shift) remains
                    st_lds[2 * row + 1] = 1.0f / sqrtf(fmaxf(sq * invK - dm * dm, 0.f) + g.ln_eps);
                }
            }
            __syncthreads();
        } else {
            // m-contiguous A: this lane saw rows mq*4..+3 for the k-rows (tid/32 + 8 i); combine the 8 k-groups through LDS
            float* red = smem + 2 * BM;                      // [8][BM][2]
            const int mq = threadIdx.x % (BM / 4), kg = threadIdx.x / (BM / 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[(kg * BM + mq * 4 + c) * 2] = ls[c];
                red[(kg * BM + mq * 4 + c) * 2 + 1] = lq[c];
            }
            __syncthreads();
            for (int r = threadIdx.x; r < BM; r += 256) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int kg2 = 0; kg2 < 256 / (BM / 4); ++kg2) {
                    sm += red[(kg2 * BM + r) * 2];
                    sq += red[(kg2 * BM + r) * 2 + 1];
                }
                const float mean = sm * invK;
                st_lds[2 * r] = mean;
                st_lds[2 * r + 1] = 1.0f / sqrtf(fmaxf(sq * invK - mean * mean, 0.f) + g.ln_eps);
            }
            __syncthreads();
        }
    }
    gemm_epilogue<BM, BN, WM, WN, EDGE, TS, false, false>(g, st_lds, smem + 2 * BM, acc, m0, n0, b, stats);
}

template <int BM, int BN, int WM, int WN, int BK, bool AKC, bool BNC, bool TS, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(const AbxGemm g) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int OPER = 2 * BK * LDA + 2 * BK * LDB;
    constexpr int SCR = 4 * 32 * ((TS ? WM : WN) + 4);                         // per-wave store scratch (4 waves)
    constexpr int RED = AKC ? 0 : (256 / (BM / 4)) * BM * 2;                     // inline-LN reduction of m-contiguous A
    constexpr int EPI = 2 * BM + (SCR > RED ? SCR : RED);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntn = (g.N + BN - 1) / BN;
    // XCD-aware remap (blocks are dispatched round-robin over the 8 XCDs): give each XCD a contiguous range of tiles so the
    // N-tiles that share an A panel hit the same private L2.  Bijective for any grid size.
    int wgid = blockIdx.x;
    if (!(g.tune & 1)) {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = wgid / ntn, nt = wgid % ntn;
    const int b = blockIdx.z;
    const bool interior = g.fast_ok && (mt + 1) * BM <= g.M && (nt + 1) * BN <= g.N && (g.K % BK) == 0;
    if (interior) gemm_block<BM, BN, WM, WN, BK, AKC, BNC, false, TS>(g, smem, mt, nt, b);
    else gemm_block<BM, BN, WM, WN, BK, AKC, BNC, true, TS>(g, smem, mt, nt, b);
}

// ---- LayerNorm statistics -------------------------------------------------------------------------------
// k-contiguous rows: one wave per row, two-pass (mean, then centred variance), rstd = 1/sqrt(var + eps).
__global__ __launch_bounds__(256) void row_stats_kc(const float* __restrict__ x, long long s_row, int rows, int K,
                                                    float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * s_row;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += p[k];
    const float mean = wave_sum(s) / (float)K;
    float q = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float d = p[k] - mean;
        q += d * d;
    }
    const float var = wave_sum(q) / (float)K;
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = 1.0f / sqrtf(var + eps);
    }
}
// m-contiguous (channel-major [K][rows] per batch): one thread per row.
__global__ __launch_bounds__(256) void row_stats_mc(const float* __restrict__ x, long long s_b, long long s_k, int rows,
                                                    int K, float eps, float* __restrict__ stats) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (r >= rows) return;
    const float* p = x + (long long)b * s_b + r;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += p[(long long)k * s_k];
    const float mean = s / (float)K;
    float q = 0.f;
    for (int k = 0; k < K; ++k) {
        const float d = p[(long long)k * s_k] - mean;
        q += d * d;
    }
    const long long o = 2 * ((long long)b * rows + r);
    stats[o] = mean;
    stats[o + 1] = 1.0f / sqrtf(q / (float)K + eps);
}

// Materialised LayerNorm (k-contiguous rows), out may alias x.  Optional residual: out = res + LN(x).
__global__ __launch_bounds__(256) void layernorm_kc(const float* __restrict__ x, long long s_row, int rows, int K,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, float* __restrict__ out, long long s_out,
                                                    const float* __restrict__ res, long long s_res) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * s_row;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += p[k];
    const float mean = wave_sum(s) / (float)K;
    float q = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float d = p[k] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + eps);
    float* o = out + row * s_out;
    for (int k = lane; k < K; k += 64) {
        float v = (p[k] - mean) * rstd * gamma[k] + beta[k];
        if (res) v += res[row * s_res + k];
        o[k] = v;
    }
}

// K == 128 (the IPA pair representation, B*L*L rows): 32 lanes x float4 per row, two rows per wave, 16-byte accesses.
__global__ __launch_bounds__(256) void layernorm128_kernel(const float* __restrict__ x, long long s_row, long long rows,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ out, long long s_out,
                                                           const float* __restrict__ res, long long s_res) {
    const int l32 = threadIdx.x & 31;
    const long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x + row * s_row)[l32];
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / 128.0f;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float d = v[c] - mean;
        q = fmaf(d, d, q);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / 128.0f + eps);
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[l32], be = reinterpret_cast<const f32x4*>(beta)[l32];
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = (v[c] - mean) * rstd * ga[c] + be[c];
    if (res) {
        const f32x4 rv = reinterpret_cast<const f32x4*>(res + row * s_res)[l32];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] += rv[c];
    }
    reinterpret_cast<f32x4*>(out + row * s_out)[l32] = r;
}

template <int BM, int BN, int WM, int WN, int BK = 16, int MINW = 3>
int launch_cfg(const AbxGemm& g, hipStream_t st) {
    const long long mt = ((long long)g.M + BM - 1) / BM, ntn = ((long long)g.N + BN - 1) / BN;
    dim3 grid((unsigned)(mt * ntn), 1, (unsigned)g.batch), block(256);
    const bool akc = g.sAk == 1;
    const bool bnc = g.sBn == 1;
    if (g.c_transposed) {       // transposed store: weight GEMMs only (A k-contiguous activations, B packed weights)
        if (!(akc && bnc)) { abx_set_error("abx_gemm: transposed store needs k-contiguous A and n-contiguous B"); return ABX_ERR_ARG; }
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, true, true, true, MINW>), grid, block, 0, st, g);
    } else if (akc && bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, true, true, false, MINW>), grid, block, 0, st, g);
    else if (akc && !bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, true, false, false, MINW>), grid, block, 0, st, g);
    else if (!akc && bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, false, true, false, MINW>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, false, false, false, MINW>), grid, block, 0, st, g);
    return abx_check_launch("abx_gemm");
}

// tuning variants of the main weight-GEMM configuration (selected by AbxGemm.tune bits 1..3; 0 = default)
template <int BK, int MINW>
int launch_main_variant(const AbxGemm& g, hipStream_t st) {
    const long long mt = ((long long)g.M + 127) / 128, ntn = ((long long)g.N + 127) / 128;
    dim3 grid((unsigned)(mt * ntn), 1, (unsigned)g.batch), block(256);
    hipLaunchKernelGGL((gemm_kernel<128, 128, 64, 64, BK, true, true, false, MINW>), grid, block, 0, st, g);
    return abx_check_launch("abx_gemm");
}

}  // namespace

// The mode table of include/abx_hip.h ("MODES"): which fused forms of the descriptor may be combined.
extern "C" int abx_gemm_check_modes(const AbxGemm* gp) {
    ABX_REQUIRE(gp != nullptr, "abx_gemm: null descriptor");
    const AbxGemm& g = *gp;
    enum { GLU, MLP, DUAL, C_SPLIT, OUT_LN, A_SPLIT, PAIR, EXACT, NMODE };
    static const char* const names[NMODE] = {"glu", "mlp", "dual (A2 / B2_split)", "c_split", "out_ln", "a_split", "pair-row maps", "exact = 1"};
    const bool on[NMODE] = {g.glu != 0, g.mlp != 0, g.A2 != nullptr, g.C_split != nullptr, g.out_ln_w != nullptr, g.A_split != nullptr,
                            g.pair_Lp > 0 || g.a_pair != 0 || g.c_pair != 0 || g.a_pair_transpose > 0, g.exact == 1};
    // 1: the pair may be combined
    static const unsigned char ok[NMODE][NMODE] = {
        /* glu     */ {1, 0, 0, 1, 0, 0, 1, 0},
        /* mlp     */ {0, 1, 0, 0, 0, 0, 0, 0},
        /* dual    */ {0, 0, 1, 0, 0, 0, 1, 0},
        /* c_split */ {1, 0, 0, 1, 0, 0, 1, 0},
        /* out_ln  */ {0, 0, 0, 0, 1, 0, 0, 0},
        /* a_split */ {0, 0, 0, 0, 0, 1, 0, 0},
        /* pair    */ {1, 0, 1, 1, 0, 0, 1, 0},
        /* exact   */ {0, 0, 0, 0, 0, 0, 0, 1},
    };
    static thread_local char msg[160];
    for (int i = 0; i < NMODE; ++i)
        for (int j = i + 1; j < NMODE; ++j)
            if (on[i] && on[j] && !ok[i][j]) {
                snprintf(msg, sizeof(msg), "abx_gemm: modes '%s' and '%s' are mutually exclusive (include/abx_hip.h, MODES)", names[i], names[j]);
                abx_set_error(msg);
                return ABX_ERR_ARG;
            }
    ABX_REQUIRE(!on[GLU] || (g.c_transposed && g.N % 128 == 0 && !g.gate), "abx_gemm: glu needs a transposed store, N % 128 == 0, no gate");
    ABX_REQUIRE(!on[C_SPLIT] || g.c_transposed, "abx_gemm: c_split needs the transposed store (c_transposed)");
    ABX_REQUIRE(!on[MLP] || (g.B2_split && g.ln_csum && !g.ln_stats && g.N2 > 0 && g.N2 <= 192 && !g.rowscale && !g.c_transposed &&
                             (g.mlp == 2 ? (g.act == 2 && g.gate != nullptr) : (g.act == 1 && !g.gate))),
                "abx_gemm: mlp needs B2_split, a folded LayerNorm (ln_csum, no ln_stats), 0 < N2 <= 192, no rowscale / transposed store; mlp = 1: "
                "act = 1, no gate; mlp = 2 (gated tail): act = 2 and the gate operand");
    ABX_REQUIRE(!on[DUAL] || (g.B2_split && g.ln2_csum && !g.c_transposed), "abx_gemm: dual needs B2_split, ln2_csum and a plain store");
    ABX_REQUIRE(!on[OUT_LN] || (g.out_ln_b && g.N <= 128 && !g.c_transposed), "abx_gemm: out_ln needs out_ln_b, N <= 128 and a plain store");
    ABX_REQUIRE(!on[A_SPLIT] || (!g.ln_csum && !g.a_relu), "abx_gemm: a_split (plane x plane contraction) takes no LayerNorm and no relu-on-load");
    return ABX_OK;
}

// validation of a descriptor and the vector-access flags the kernels read (shared by abx_gemm and abx_gemm_side)
static int gemm_prepare(const AbxGemm* gp, AbxGemm& g) {
    if (int rc = abx_gemm_check_modes(gp)) return rc;
    g = *gp;
    ABX_REQUIRE((g.A || g.A_split) && (g.B || g.B_split) && (g.C || g.C_split), "abx_gemm: null operand");
    ABX_REQUIRE(!g.glu || (g.c_transposed && g.N % 128 == 0 && !g.gate), "abx_gemm: glu needs a transposed store, N % 128 == 0, no gate");
    const long long tile_rows = ((long long)(g.pair_L + 7) / 8 * 8) * ((long long)(g.pair_Lp + 15) / 16 * 16);
    ABX_REQUIRE(!g.c_split_tile || (g.C_split && g.a_pair && g.pair_Lp > 0 && g.M == tile_rows && !g.gate && !g.resid),
                "abx_gemm: c_split_tile needs C_split, a_pair, pair_L / pair_Lp, M == ceil8(pair_L) * ceil16(pair_Lp), no gate / resid");
    ABX_REQUIRE(!g.C_split || (g.c_transposed && g.c_split_L > 0 && g.c_split_L % 4 == 0 && (g.c_split_tile || g.M % g.c_split_L == 0)),
                "abx_gemm: C_split needs the transposed store of a (padded) pair tensor (M = rows * c_split_L, c_split_L % 4 == 0)");
    ABX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, "abx_gemm: empty problem");
    ABX_REQUIRE(g.pair_Lp == 0 || (g.pair_L > 0 && g.pair_Lp >= g.pair_L && g.pair_Lp % 4 == 0 &&
                                   (g.c_split_tile || g.M == (long long)g.pair_L * g.pair_Lp) && (!g.C_split || g.c_split_L == g.pair_Lp)),
                "abx_gemm: padded pair rows need M == pair_L * pair_Lp, pair_Lp % 4 == 0 (and c_split_L == pair_Lp)");
    ABX_REQUIRE((!g.a_pair && !g.c_pair) || g.pair_Lp > 0, "abx_gemm: a_pair / c_pair need pair_L / pair_Lp");
    ABX_REQUIRE(!g.A || g.sAk == 1 || g.sAm == 1, "abx_gemm: A must be k- or m-contiguous");
    ABX_REQUIRE(!g.B || g.sBn == 1 || g.sBk == 1, "abx_gemm: B must be n- or k-contiguous");
    ABX_REQUIRE(!g.ln_stats || g.ln_csum, "abx_gemm: LayerNorm needs the column sums of the gamma-scaled weights");
    ABX_REQUIRE(!(g.ln_csum && g.a_relu), "abx_gemm: LayerNorm and relu-on-load are exclusive");
    if (g.ln_csum && !g.ln_stats && g.ln_eps <= 0.f) g.ln_eps = 1e-5f;
    if (g.out_ln_w && g.out_ln_eps <= 0.f) g.out_ln_eps = 1e-5f;
    const bool akc = g.sAk == 1;
    // 16-byte vector loads need aligned bases and strides that are multiples of 4 elements
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.a_vec_ok = al16(g.A) && (g.sAb % 4 == 0) && (akc ? (g.sAm % 4 == 0) : (g.sAk % 4 == 0));
    g.b_vec_ok = al16(g.B) && (g.sBb % 4 == 0) && (g.sBn == 1 ? (g.sBk % 4 == 0) : (g.sBn % 4 == 0));
    g.fast_ok = g.a_vec_ok && g.b_vec_ok;
    g.c_vec_ok = g.C_split ? ((reinterpret_cast<uintptr_t>(g.C_split) & 7) == 0 && g.sCb % 4 == 0 && g.sCm % 4 == 0 && g.sCp % 4 == 0)
                           : (al16(g.C) && (g.sCb % 4 == 0) && (g.sCm % 4 == 0));
    g.g_vec_ok = g.gate && al16(g.gate) && (g.sGb % 4 == 0) && (g.sGm % 4 == 0);
    g.r_vec_ok = g.resid && al16(g.resid) && (g.sRb % 4 == 0) && (g.sRm % 4 == 0);
    g.rs_vec_ok = g.rowscale && al16(g.rowscale) && (g.sRSb % 4 == 0);
    ABX_REQUIRE(g.c_planes_from == 0 || (g.c_planes_from > 0 && g.c_planes_from % 4 == 0 && g.c_planes_group > 0 && g.c_planes_group % 4 == 0 &&
                                         g.N % 4 == 0 && g.c_planes_from < g.N && (g.N - g.c_planes_from) % g.c_planes_group == 0 && g.C && g.c_vec_ok &&
                                         !g.c_transposed && !g.C_split && !g.glu && !g.gate && !g.resid && !g.mlp && !g.A2 && !g.out_ln_w && !g.c_pair &&
                                         g.act == 0 && g.exact != 1),
                "abx_gemm: c_planes_from needs a plain, 16-byte aligned fp32 store without gate / resid / activation on the split-f16 path, from, "
                "group and N multiples of 4, (N - from) % group == 0");
    return ABX_OK;
}

int abx_gemm3_side_dispatch(const AbxGemm& g, const AbxGemm& s2, hipStream_t st, int* rc);
int abx_gemm_as_dispatch(const AbxGemm& g, const AbxGemm* side, hipStream_t st, int* rc);      // gemm_as.hip

// 1 when a (main, side) pair of M rows takes the A-stationary kernel, i.e. may ask for plane output (AbxGemm.c_planes_from): the size rule of
// abx_gemm_as_dispatch (the walk pays from 1 024 blocks of 64 rows on).  Callers that switch the k | v format on it stay bit-invariant under
// chunking: the attention kernel computes the same pieces from fp32 k | v as the projection writes as planes.
extern "C" int abx_gemm_planes_ok(long long M) {
    static const bool off = getenv("ABX_NO_GEMM_AS") != nullptr || getenv("ABX_NO_GEMM_SIDE") != nullptr;
    return (!off && (M + 63) / 64 >= 1024) ? 1 : 0;
}

extern "C" int abx_gemm_side(const AbxGemm* main_gemm, const AbxGemm* side_gemm, hipStream_t st) {
    ABX_REQUIRE(main_gemm && side_gemm, "abx_gemm_side: null descriptor");
    AbxGemm g, s2;
    if (int rc = gemm_prepare(main_gemm, g)) return rc;
    if (int rc = gemm_prepare(side_gemm, s2)) return rc;
    int rc = 0;
    static const bool off = getenv("ABX_NO_GEMM_SIDE") != nullptr;          // (A / B measurements: always the two launches)
    if (!off && !abx_gemm_as_dispatch(g, &s2, st, &rc)) return rc;      // A-stationary walk: the side rides in the ragged last N-tile
    // (plane output exists in the A-stationary kernel only: a caller asks for it when the launch qualifies - abx_gemm_planes_ok)
    ABX_REQUIRE(g.c_planes_from == 0, "abx_gemm_side: c_planes_from is served by the A-stationary kernel only (>= 65 536 rows, K = 192, N % 128 == 64, "
                                      "planes in groups of 48 from a multiple of 32, a side projection): ask abx_gemm_planes_ok first");
    if (!off && !abx_gemm3_side_dispatch(g, s2, st, &rc)) return rc;
    // not a (128 x 128 plain, skinny transposed) split-f16 pair: the two launches
    if (int r1 = abx_gemm(main_gemm, st)) return r1;
    return abx_gemm(side_gemm, st);
}

extern "C" int abx_gemm(const AbxGemm* gp, hipStream_t st) {
    AbxGemm g;
    if (int rc = gemm_prepare(gp, g)) return rc;
    ABX_REQUIRE(g.c_planes_from == 0, "abx_gemm: c_planes_from is served through abx_gemm_side by the A-stationary kernel only (abx_gemm_planes_ok)");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    (void)al16;
    const bool akc = g.sAk == 1;
    if (g.exact != 1) {
        int rc = 0;
        if (!abx_gemm_as_dispatch(g, nullptr, st, &rc)) return rc;
        if (!abx_gemm3_dispatch(g, st, &rc)) return rc;
    }
    g.range_flag = nullptr;                 // exact fp32 products from here on: no operand range, a non-finite result is the input's
    ABX_REQUIRE(g.A && g.B, "abx_gemm: split-f16 operands given but the problem does not qualify for the split kernels "
                            "(K % 16, alignment, size) and no fp32 operands were passed for the exact kernel");
    ABX_REQUIRE(g.a_pair_transpose <= 0 && g.pair_Lp == 0, "abx_gemm: pair-row remapping is served by the split-f16 kernels only");
    ABX_REQUIRE(g.batch <= 65535, "abx_gemm: batch > 65535 (exact fp32 kernels)");
    ABX_REQUIRE(!g.glu, "abx_gemm: glu is served by the split-f16 kernels only (large problems, K % 16 == 0)");
    ABX_REQUIRE(!g.A2, "abx_gemm: the dual GEMM is served by the split-f16 kernels only");
    ABX_REQUIRE(!g.out_ln_w, "abx_gemm: out_ln is served by the split-f16 kernels only");
    ABX_REQUIRE(!g.mlp, "abx_gemm: the fused transition (mlp) is served by the split-f16 kernels only (exact = 2, K % 16 == 0)");
    const long long mt128 = ((long long)g.M + 127) / 128;
    if (g.N <= 32) return launch_cfg<128, 32, 32, 32>(g, st);
    if (g.N <= 64) return launch_cfg<128, 64, 32, 64>(g, st);
    if (mt128 * (((long long)g.N + 127) / 128) * g.batch < 512) return launch_cfg<64, 64, 32, 32>(g, st);
    if (g.tune >> 1) {
        ABX_REQUIRE(!g.c_transposed && akc && g.sBn == 1, "abx_gemm: tuning variants exist for the plain weight GEMM only");
        switch (g.tune >> 1) {
            case 1: return launch_main_variant<16, 2>(g, st);
            case 2: return launch_main_variant<16, 4>(g, st);
            case 3: return launch_cfg<128, 128, 64, 64, 16, 3>(g, st);
            default: break;
        }
    }
    // 128x192 tiles (96 accumulator registers per lane) when they waste no more columns than 128x128 tiles do: N = 192 (every
    // residual-stream update of the pair stack), 768 (q|k|v|gate, transition hidden), 544 ...
    const long long pad128 = ((g.N + 127) / 128) * 128, pad192 = ((g.N + 191) / 192) * 192;
    if (pad192 <= pad128 && !(g.tune >> 1)) return launch_cfg<128, 192, 64, 96, 16, 2>(g, st);
    return launch_cfg<128, 128, 64, 64>(g, st);
}

extern "C" int abx_row_stats(const float* x, long long s_b, long long s_row, long long s_k, int batch, int rows, int K,
                             float eps, float* stats, hipStream_t st) {
    ABX_REQUIRE(x && stats && rows > 0 && K > 0 && batch > 0, "abx_row_stats: bad args");
    if (s_k == 1) {
        ABX_REQUIRE(batch == 1 || s_b == (long long)rows * s_row, "abx_row_stats: k-contiguous rows must be dense over batch");
        const long long total = (long long)batch * rows;
        hipLaunchKernelGGL(row_stats_kc, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, x, s_row, (int)total, K, eps, stats);
    } else {
        ABX_REQUIRE(s_row == 1, "abx_row_stats: rows must be contiguous when k is strided");
        hipLaunchKernelGGL(row_stats_mc, dim3((unsigned)((rows + 255) / 256), (unsigned)batch), dim3(256), 0, st, x, s_b, s_k,
                           rows, K, eps, stats);
    }
    return abx_check_launch("abx_row_stats");
}

extern "C" int abx_layernorm(const float* x, long long s_row, long long rows, int K, const float* gamma, const float* beta,
                             float eps, float* out, long long s_out, const float* res, long long s_res, hipStream_t st) {
    ABX_REQUIRE(x && out && gamma && beta && rows > 0 && K > 0, "abx_layernorm: bad args");
    auto al16v = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (K == 128 && s_row % 4 == 0 && s_out % 4 == 0 && al16v(x) && al16v(out) && al16v(gamma) && al16v(beta) &&
        (!res || (al16v(res) && s_res % 4 == 0))) {
        hipLaunchKernelGGL(layernorm128_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, x, s_row, rows, gamma, beta, eps,
                           out, s_out, res, s_res);
        return abx_check_launch("abx_layernorm");
    }
    hipLaunchKernelGGL(layernorm_kc, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, s_row, (int)rows, K, gamma, beta,
                       eps, out, s_out, res, s_res);
    return abx_check_launch("abx_layernorm");
}
