// Generic batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain)
// with a LayerNorm-on-load prologue and a fused epilogue.  One kernel serves every dense contraction of the
// score network (reference sites: every abx.model.common_modules.Linear on the hot path, the
// TriangleMultiplication einsum seqformer.py:490-493, the transitions seqformer.py:358-376).
//
//   C[b][m][n] = epi( sum_k A'[b][m][k] * B[b][k][n] )
//   A' = relu?( LN?(A) )           LN uses per-row (mean, rstd) from abx_row_stats and gamma/beta over k
//   epi(v)  = ((v + bias[n]) * alpha) -> act -> * rowscale[m] -> * (sigmoid?)(gate[m][n]) -> + resid[m][n]
//
// Layout: A is either k-contiguous (sAk==1) or m-contiguous (sAm==1); B is n-contiguous (sBn==1, the packed
// weight layout Wt[K][N]) or k-contiguous (sBk==1).  C is n-contiguous, or stored transposed (c_transposed:
// element (m,n) at C + b*sCb + n*sCm + m) which is how the pair stack is turned channel-major for the
// triangle-multiplication contraction without a separate transpose pass.
//
// Tiling: 256 threads = 4 waves; block tile BMxBNx16 staged through double-buffered LDS as [k][m] / [k][n]
// (+4 pad) so the MFMA operand reads (lane -> m, lane>>5 -> k) are conflict-free ds_read_b32; global->register
// prefetch of tile t+1 overlaps the MFMAs of tile t; one barrier per k-tile.
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int BK = 16;

template <int BMN, bool KC>
struct TileLoader {
    static constexpr int NVEC = BMN * BK / 4;            // float4 slots in the tile
    static constexpr int NV = (NVEC + 255) / 256;        // per thread
    static constexpr int LD = BMN + 4;
    f32x4 v[NV];

    // base already offset by batch.  s_mn / s_k: element strides of the (m|n) and k dims.
    __device__ __forceinline__ void load(const float* __restrict__ base, long long s_mn, long long s_k, int mn0,
                                         int k0, int MN, int K, bool vec_ok, const float* __restrict__ stats,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         bool relu) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (idx < NVEC) {
                if (KC) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
                    const int mn = mn0 + row, k = k0 + kq * 4;
                    if (mn < MN && k < K) {
                        const float* p = base + (long long)mn * s_mn + k;
                        if (vec_ok && k + 3 < K) {
                            r = *reinterpret_cast<const f32x4*>(p);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (k + c < K) r[c] = p[c];
                        }
                        if (stats) {
                            const float mean = stats[2 * (long long)mn], rstd = stats[2 * (long long)mn + 1];
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (k + c < K) r[c] = (r[c] - mean) * rstd * gamma[k + c] + beta[k + c];
                        }
                        if (relu) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) r[c] = fmaxf(r[c], 0.f);
                        }
                    }
                } else {
                    const int krow = idx / (BMN / 4), mq = idx % (BMN / 4);
                    const int k = k0 + krow, mn = mn0 + mq * 4;
                    if (k < K && mn < MN) {
                        const float* p = base + (long long)k * s_k + mn;
                        if (vec_ok && mn + 3 < MN) {
                            r = *reinterpret_cast<const f32x4*>(p);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (mn + c < MN) r[c] = p[c];
                        }
                        if (stats) {
                            const float ga = gamma[k], be = beta[k];
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (mn + c < MN)
                                    r[c] = (r[c] - stats[2 * (long long)(mn + c)]) * stats[2 * (long long)(mn + c) + 1] * ga + be;
                        }
                        if (relu) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) r[c] = fmaxf(r[c], 0.f);
                        }
                    }
                }
            }
            v[i] = r;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (idx < NVEC) {
                if (KC) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) lds[(kq * 4 + c) * LD + row] = v[i][c];
                } else {
                    const int krow = idx / (BMN / 4), mq = idx % (BMN / 4);
                    *reinterpret_cast<f32x4*>(&lds[krow * LD + mq * 4]) = v[i];
                }
            }
        }
    }
};

template <int BM, int BN, int WM, int WN, bool AKC, bool BNC>
__global__ __launch_bounds__(256) void gemm_kernel(const AbxGemm g) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int ntn = (g.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
    const int b = blockIdx.z;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const float* Ab = g.A + (long long)b * g.sAb;
    const float* Bb = g.B + (long long)b * g.sBb;
    const float* stats = g.ln_stats ? g.ln_stats + 2 * (long long)b * g.sSb : nullptr;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<BM, AKC> la;
    TileLoader<BN, !BNC> lb;   // B n-contiguous == "row-contiguous" loader; B k-contiguous == KC loader
    const bool a_vec = g.a_vec_ok != 0, b_vec = g.b_vec_ok != 0;
    const long long a_smn = g.sAm, a_sk = g.sAk, b_smn = g.sBn, b_sk = g.sBk;
    const int nk = (g.K + BK - 1) / BK;

    la.load(Ab, a_smn, a_sk, m0, 0, g.M, g.K, a_vec, stats, g.ln_gamma, g.ln_beta, g.a_relu != 0);
    lb.load(Bb, b_smn, b_sk, n0, 0, g.N, g.K, b_vec, nullptr, nullptr, nullptr, false);
    la.store(As);
    lb.store(Bs);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            la.load(Ab, a_smn, a_sk, m0, (t + 1) * BK, g.M, g.K, a_vec, stats, g.ln_gamma, g.ln_beta, g.a_relu != 0);
            lb.load(Bb, b_smn, b_sk, n0, (t + 1) * BK, g.N, g.K, b_vec, nullptr, nullptr, nullptr, false);
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
            la.store(As + (cur ^ 1) * BK * LDA);
            lb.store(Bs + (cur ^ 1) * BK * LDB);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = g.C + (long long)b * g.sCb;
    const float* rs = g.rowscale ? g.rowscale + (long long)b * g.sRSb : nullptr;
    const float* gt = g.gate ? g.gate + (long long)b * g.sGb : nullptr;
    const float* rd = g.resid ? g.resid + (long long)b * g.sRb : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + (lane & 31);
            const bool nok = n < g.N;
            const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int mbase = m0 + wm * WM + i * 32 + 8 * rq + 4 * (lane >> 5);
                f32x4 out;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int m = mbase + c;
                    float v = (acc[i][j][rq * 4 + c] + bias) * g.alpha;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = 1.0f / (1.0f + expf(-v));
                    if (nok && m < g.M) {
                        if (rs) v *= rs[m];
                        if (gt) {
                            const float gv = gt[(long long)m * g.sGm + n];
                            v *= g.gate_sigmoid ? 1.0f / (1.0f + expf(-gv)) : gv;
                        }
                        if (rd) v += rd[(long long)m * g.sRm + n];
                        if (!g.c_transposed) Cb[(long long)m * g.sCm + n] = v;
                    }
                    out[c] = v;
                }
                if (g.c_transposed && nok) {
                    float* p = Cb + (long long)n * g.sCm + mbase;
                    if (g.c_vec_ok && mbase + 3 < g.M) {
                        *reinterpret_cast<f32x4*>(p) = out;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (mbase + c < g.M) p[c] = out[c];
                    }
                }
            }
        }
    }
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

template <int BM, int BN, int WM, int WN>
int launch_cfg(const AbxGemm& g, hipStream_t st) {
    const long long mt = ((long long)g.M + BM - 1) / BM, ntn = ((long long)g.N + BN - 1) / BN;
    dim3 grid((unsigned)(mt * ntn), 1, (unsigned)g.batch), block(256);
    const bool akc = g.sAk == 1 && !(g.sAm == 1 && g.force_a_mcontig);
    const bool bnc = g.sBn == 1;
    if (akc && bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, g);
    else if (akc && !bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, g);
    else if (!akc && bnc) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, g);
    return abx_check_launch("abx_gemm");
}

}  // namespace

extern "C" int abx_gemm(const AbxGemm* gp, hipStream_t st) {
    ABX_REQUIRE(gp != nullptr, "abx_gemm: null descriptor");
    AbxGemm g = *gp;
    ABX_REQUIRE(g.A && g.B && g.C, "abx_gemm: null operand");
    ABX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, "abx_gemm: empty problem");
    ABX_REQUIRE(g.batch <= 65535, "abx_gemm: batch > 65535");
    ABX_REQUIRE(g.sAk == 1 || g.sAm == 1, "abx_gemm: A must be k- or m-contiguous");
    ABX_REQUIRE(g.sBn == 1 || g.sBk == 1, "abx_gemm: B must be n- or k-contiguous");
    ABX_REQUIRE(!g.ln_stats || (g.ln_gamma && g.ln_beta), "abx_gemm: LN needs gamma/beta");
    const bool akc = g.sAk == 1 && !(g.sAm == 1 && g.force_a_mcontig);
    // 16-byte vector loads need aligned bases and strides that are multiples of 4 elements
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    g.a_vec_ok = al16(g.A) && (g.sAb % 4 == 0) && (akc ? (g.sAm % 4 == 0) : (g.sAk % 4 == 0));
    g.b_vec_ok = al16(g.B) && (g.sBb % 4 == 0) && (g.sBn == 1 ? (g.sBk % 4 == 0) : (g.sBn % 4 == 0));
    g.c_vec_ok = al16(g.C) && (g.sCb % 4 == 0) && (g.sCm % 4 == 0);
    const long long mt128 = ((long long)g.M + 127) / 128;
    if (g.N <= 32) return launch_cfg<128, 32, 32, 32>(g, st);
    if (g.N <= 64) return launch_cfg<128, 64, 32, 64>(g, st);
    if (mt128 * (((long long)g.N + 127) / 128) * g.batch < 512) return launch_cfg<64, 64, 32, 32>(g, st);
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
    hipLaunchKernelGGL(layernorm_kc, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, s_row, (int)rows, K, gamma, beta,
                       eps, out, s_out, res, s_res);
    return abx_check_launch("abx_layernorm");
}
