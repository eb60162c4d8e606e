// Fused GEMM epilogue shared by the exact-fp32 (gemm.hip) and the split-f16 (gemm3.hip) kernels.
// The accumulators use the C/D layout of the 32x32 MFMAs (dtype independent on gfx950):
//   col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
// st_lds: [BM][2] (mean - shift, rstd) of this M-panel when `stats`; scratch: per-wave store staging (4 waves).
#pragma once
#include "common.h"
#include "abx_hip.h"

// PROBE: the range probe (AbxGemm.range_flag) is compiled in.  Off for the exact fp32 kernels (no operand range) and for the 128 x 128
// split tiles at four blocks per CU, whose 128 registers have no room for it: their outputs (q | k | v | gate, gated projections) are
// operands of kernels that carry the probe (triangle attention, contraction), so a NaN row made there is reported one kernel later.
template <int BM, int BN, int WM, int WN, bool EDGE, bool TS, bool OLN = false, bool PROBE = true, int TGO = 0>
__device__ __forceinline__ void gemm_epilogue(const AbxGemm& g, const float* __restrict__ st_lds, float* __restrict__ scratch,
                                              f32x16 (&acc)[WM / 32][WN / 32], int m0, int n0, int b, bool stats,
                                              f32x16 (*acc2)[WM / 32][WN / 32] = nullptr, const float* __restrict__ st2_lds = nullptr,
                                              const f32x16 (*gatev)[WM / 32][WN / 32] = nullptr) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float* Cb = g.C + (long long)b * g.sCb;
    const float* rs = g.rowscale ? g.rowscale + (long long)b * g.sRSb : nullptr;
    const float* gt = g.gate ? g.gate + (long long)b * g.sGb : nullptr;
    const float* rd = g.resid ? g.resid + (long long)b * g.sRb : nullptr;
    const bool c_vec = g.c_vec_ok != 0, g_vec = g.g_vec_ok != 0, r_vec = g.r_vec_ok != 0, gsig = g.gate_sigmoid != 0;
    // range safety (AbxGemm.range_flag): an operand beyond the split-f16 ranges has become inf - inf = NaN in the accumulators of its
    // row / column, and NaN survives everything this epilogue does (relu_keep_nan, gates, residual, LayerNorm), so the probe looks at
    // the values on their way to memory: v_cmp_class per stored value, OR-ed as lane masks in SGPRs (no vector registers)
    bool bad = false;
    float bias[TN], csum[TN], bias2[TN], csum2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        const bool nok = !EDGE || n < g.N;
        bias[j] = (g.bias && nok) ? g.bias[n] : 0.f;
        csum[j] = (stats && nok) ? g.ln_csum[n] : 0.f;
        bias2[j] = (acc2 && g.bias2 && nok) ? g.bias2[n] : 0.f;
        csum2[j] = (acc2 && nok) ? g.ln2_csum[n] : 0.f;
    }
    // per-element part that needs no global operand: folded LayerNorm, bias, alpha, activation
    auto epi1 = [&](float v, int ml, int j) -> float {
        if (stats) v = st_lds[2 * ml + 1] * (v - st_lds[2 * ml] * csum[j]);
        v = (v + bias[j]) * g.alpha;
        if (g.act == 1) v = relu_keep_nan(v);
        else if (g.act == 2) v = sigmoidf_(v);
        return v;
    };
    // 4 consecutive elements along the contiguous output dimension: gate / residual reads and the store are 16-byte accesses
    // when the operand allows it (cnt < 4: ragged tail)
    auto load4 = [&](const float* p, bool vec, int cnt) -> f32x4 {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (vec && cnt == 4) r = *reinterpret_cast<const f32x4*>(p);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < cnt) r[c] = p[c];
        }
        return r;
    };
    auto epi2_store = [&](f32x4 v, long long off_c, long long off_g, long long off_r, int cnt, int m_idx, int n_idx) {
        if (gt) {
            f32x4 gv = load4(gt + off_g, g_vec, cnt);
            if (gsig) {
#pragma unroll
                for (int c = 0; c < 4; ++c) gv[c] = sigmoidf_(gv[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= gv[c];
        }
        if (rd) {
            const f32x4 rv = load4(rd + off_r, r_vec, cnt);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += rv[c];
        }
        if constexpr (PROBE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bad |= __builtin_amdgcn_classf(v[c], 0x207) && c < cnt;
        }
        if (g.C_split) {
            // output as the pre-split f16 operand image of the next contraction, two planes per channel: channels below c_split_nA are
            // its A side (pieces of v 2^-4: a0, a1 = (v' - a0) 2^11), the others its B side (planes of v 2^4: p0, p1 = v' - p0) -
            // common.h split2h / split2b
            unsigned a0, a1, b0, b1;
            if (n_idx < g.c_split_nA) {
                split2h(v[0], v[1], a0, a1);
                split2h(v[2], v[3], b0, b1);
            } else {
                split2b(v[0], v[1], a0, a1);
                split2b(v[2], v[3], b0, b1);
            }
            // m = i * L + k: the planes are the k-tiled operand image [k/16][plane][i][16] of the following contraction
            int ii, kk;
            if (g.c_split_tile) {
                pair_tile_decode(m_idx, g.c_split_L, ii, kk);
                if (ii >= g.pair_L || kk >= g.c_split_L) return;    // padding rows of the (8 i x 16 k) blocks: nothing to store
            } else {
                ii = m_idx / g.c_split_L;
                kk = m_idx - ii * g.c_split_L;
            }
            unsigned short* cs = g.C_split + (long long)b * g.sCb + (long long)n_idx * g.sCm + (kk >> 4) * g.sCk + ii * 16 + (kk & 15);
            if (c_vec && cnt == 4) {
                *reinterpret_cast<u32x2*>(cs) = u32x2{a0, b0};
                *reinterpret_cast<u32x2*>(cs + g.sCp) = u32x2{a1, b1};
            } else {
                const unsigned pa[2] = {a0, a1}, pb[2] = {b0, b1};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    unsigned short* o = cs + p * g.sCp;
                    if (cnt > 0) o[0] = (unsigned short)(pa[p] & 0xffffu);
                    if (cnt > 1) o[1] = (unsigned short)(pa[p] >> 16);
                    if (cnt > 2) o[2] = (unsigned short)(pb[p] & 0xffffu);
                    if (cnt > 3) o[3] = (unsigned short)(pb[p] >> 16);
                }
            }
        }
#ifdef ABX_EPI_NT_STORE   // (probe build: non-temporal stores of the plain fp32 rows)
        else if (c_vec && cnt == 4) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Cb + off_c));
#else
        else if (c_vec && cnt == 4) *reinterpret_cast<f32x4*>(Cb + off_c) = v;
#endif
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < cnt) Cb[off_c + c] = v[c];
        }
    };
    // LayerNorm over the output columns (out_ln_w; its own kernel instantiation: the extra live ranges must not cost the plain
    // kernels registers): the wave holds whole rows (WAVES_N == 1, one n-tile), so the row statistics
    // are a 32-lane butterfly over the accumulators; two passes (mean, then centred squares) like torch's LayerNorm
    constexpr bool pre = OLN;
    if constexpr (OLN) {
        static_assert(!TS && WAVES_N == 1, "out_ln: plain store, whole rows per wave");
        {
            float gam[TN], bet[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 32 + (lane & 31);
                const bool nok = n < g.N;
                gam[j] = nok ? g.out_ln_w[n] : 0.f;
                bet[j] = nok ? g.out_ln_b[n] : 0.f;
            }
            const float invn = 1.0f / (float)g.N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * WM + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    float sm = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bool nok = n0 + j * 32 + (lane & 31) < g.N;
                        float v = epi1(acc[i][j][r], ml, j);
                        if (acc2) {
                            const float gv = st2_lds[2 * ml + 1] * ((*acc2)[i][j][r] - st2_lds[2 * ml] * csum2[j]) + bias2[j];
                            v *= sigmoidf_(gv);
                        }
                        acc[i][j][r] = nok ? v : 0.f;
                        sm += acc[i][j][r];
                    }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o, 64);
                    const float mean = sm * invn;
                    float sq = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bool nok = n0 + j * 32 + (lane & 31) < g.N;
                        const float d = nok ? acc[i][j][r] - mean : 0.f;
                        sq = fmaf(d, d, sq);
                    }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) sq += __shfl_xor(sq, o, 64);
                    const float rstd = 1.0f / sqrtf(sq * invn + g.out_ln_eps);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] = (acc[i][j][r] - mean) * rstd * gam[j] + bet[j];
                }
            }
        }
    }
    if constexpr (!TS) {
        // plain store: each 32-row band goes through the wave's row-major LDS scratch [32][GW + 4] in column groups of GW <= 96
        // and leaves as float4 along n
        constexpr int TG = TGO > 0 ? TGO : (TN > 3 ? (TN % 3 == 0 ? 3 : 2) : TN);     // sub-tiles per column group (TGO: the caller's choice)
        static_assert(TN % TG == 0, "column groups");
        constexpr int GW = TG * 32, LW = GW + 4, C4 = GW / 4, NQ = 32 * C4 / 64;
        float* wsc = scratch + wave * (32 * LW);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int jg = 0; jg < TN; jg += TG) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int mloc = 8 * rq + 4 * (lane >> 5) + c;
                        const int ml = wm * WM + i * 32 + mloc;
#pragma unroll
                        for (int j = 0; j < TG; ++j) {
                            float v = pre ? acc[i][jg + j][rq * 4 + c] : epi1(acc[i][jg + j][rq * 4 + c], ml, jg + j);
                            if (acc2 && !pre) {      // dual GEMM: times sigmoid(LN-folded gate accumulator + bias2)
                                const float gv = st2_lds[2 * ml + 1] * ((*acc2)[i][jg + j][rq * 4 + c] - st2_lds[2 * ml] * csum2[jg + j]) + bias2[jg + j];
                                v *= sigmoidf_(gv);
                            }
                            // (one-walk dual GEMM: the gate values sigmoid(...) themselves, kept in registers since the gate walk)
                            if (gatev && !pre) v *= (*gatev)[i][jg + j][rq * 4 + c];
                            wsc[mloc * LW + j * 32 + (lane & 31)] = v;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll 6
                for (int q = 0; q < NQ; ++q) {          // 6 rows of gate / residual loads in flight per lane, not NQ
                    const int f = lane + 64 * q;
                    const int row = f / C4, c4 = f % C4;
                    const int m = m0 + wm * WM + i * 32 + row;
                    const int n = n0 + wn * WN + jg * 32 + c4 * 4;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&wsc[row * LW + c4 * 4]);
                    int cnt = 4;
                    if (EDGE) {
                        if (m >= g.M || n >= g.N) continue;
                        cnt = min(4, g.N - n);
                    }
                    if (rs) {
                        const float s = rs[m];
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] *= s;
                    }
                    long long mo = m;                       // row of C / gate / resid
                    if (g.c_pair) {                         // padded pair position (i, j) -> unpadded pair row; pad columns dropped
                        const int pi = m / g.pair_Lp, pj = m - pi * g.pair_Lp;
                        if (pj >= g.pair_L) continue;
                        mo = (long long)pi * g.pair_L + pj;
                    }
                    epi2_store(v, mo * g.sCm + n, mo * g.sGm + n, mo * g.sRm + n, cnt, m, n);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    } else {
        // transposed store (C, gate and resid are all addressed [n][m], m contiguous): each 32-column sub-tile goes through the
        // wave's LDS scratch [32 n][WM + 4] (the 4 accumulator rows of a register quad are contiguous there) and leaves as
        // float4 along m: WM * 4 contiguous bytes per output row
        constexpr int LWT = WM + 4, M4 = WM / 4, NQ = 32 * M4 / 64;
        float* wsc = scratch + wave * (32 * LWT);
        const bool rs_vec = g.rs_vec_ok != 0;
        // glu: the tile's sub-tiles come in (value, gate) pairs (2jo, 2jo + 1) of the SAME output channels (the weights are
        // packed that way): out = value * sigmoid(gate), N / 2 output columns - the gated projections of the triangle
        // multiplication (seqformer.py:480-485) without a round trip of the gates through HBM
        const bool glu = g.glu != 0;
        const int jstep = glu ? 2 : 1;
        const int Nout = glu ? g.N / 2 : g.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (glu && (j & 1)) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int mloc = i * 32 + 8 * rq + 4 * (lane >> 5);
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = epi1(acc[i][j][rq * 4 + c], wm * WM + mloc + c, j);
                    if (glu) {
                        if constexpr (TN % 2 == 0) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float gv = epi1(acc[i][(j | 1) < TN ? (j | 1) : j][rq * 4 + c], wm * WM + mloc + c, (j | 1) < TN ? (j | 1) : j);
                                v[c] *= sigmoidf_(gv);
                            }
                        }
                    }
                    *reinterpret_cast<f32x4*>(&wsc[(lane & 31) * LWT + mloc]) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int f = lane + 64 * q;
                const int nl = f / M4, m4 = f % M4;
                const int n = (n0 + wn * WN + j * 32) / jstep + nl;
                const int m = m0 + wm * WM + m4 * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(&wsc[nl * LWT + m4 * 4]);
                int cnt = 4;
                if (EDGE) {
                    if (m >= g.M || n >= Nout) continue;
                    cnt = min(4, g.M - m);
                }
                if (rs) {
                    long long rsi = m;
                    if (g.c_split_tile) {                       // the row scale stays in (i, k) order: [i * Lp + k]
                        int pi, pj;
                        pair_tile_decode(m, g.pair_Lp, pi, pj);
                        if (pi >= g.pair_L || pj >= g.pair_Lp) continue;
                        rsi = (long long)pi * g.pair_Lp + pj;
                    }
                    const f32x4 s = load4(rs + rsi, rs_vec, cnt);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] *= s[c];
                }
                epi2_store(v, (long long)n * g.sCm + m, (long long)n * g.sGm + m, (long long)n * g.sRm + m, cnt, m, n);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if constexpr (PROBE) {
        if (g.range_flag && __any(bad) && lane == 0) atomicOr(g.range_flag, g.range_tag);
    }
}
