// fp32 GEMM on the float16 matrix cores by operand splitting ("split-f16"): the large contractions of the pair stack.
//
// An fp32 product is evaluated from three exact partial products on v_mfma_f32_32x32x16_f16 with fp32 accumulation
//     x y ~ a1 p2 + a0 p1 + a0 p0                              (smallest terms first)
//     A side (two pieces):   x' = x 2^-4,  a0 = f16(x'),  a1 = f16((x' - a0) 2^11)
//     B side (two planes):   y' = y 2^e,   p0 = f16(y'),  p1 = f16(y' - p0);  p2 = f16(p0 2^-11) is derived in registers
// Each f16 x f16 product is exact in fp32.  A piece pair holds 23 significant bits and the dropped term a1 p1 2^-11 is <= 2^-22 |x y|,
// mean zero: measured against fp64 (tests/test_gpu_kernels.py) the error is that of the exact v_mfma_f32_32x32x2_f32 kernel of
// gemm.hip, while the matrix cores run 16/3 = 5.3x faster than their fp32 rate.  The power-of-two scales keep the pieces normal
// float16 numbers (range contract and what happens beyond it: include/abx_hip.h, "Split-f16 operands").  The
// exact kernel remains selectable (AbxGemm.exact).  Rounds 1-2 used three bf16 pieces per operand and six products.
//
// Data movement is all asynchronous global->LDS DMA (global_load_lds_dwordx4): no operand passes through VGPRs on its way
// to LDS (no staging registers, no ds_write pass, no vector address arithmetic); the k-loop waits with an explicit
// s_waitcnt + raw s_barrier and is double buffered.
//   A  fp32, k-contiguous rows (AMODE 0): 2 stages [BM][16] fp32 (64-byte rows, 16-byte slots XOR-swizzled through
//      the SOURCE address).  Every wave reads its own rows as fp32 fragments and splits them in registers right in front of
//      the MFMAs; the LayerNorm statistics (inline mode), the mean shift and relu-on-load are applied to those registers.
//      Optional pair transposition of the rows (a_pair_transpose): the DMA source address is per lane, so the incoming
//      TriangleMultiplication reads z[k][i] rows in (i,k) order for free.
//   A  fp32, row-contiguous / channel-major (AMODE 1): 2 stages [16 k][BM] fp32, fragments by 4-byte LDS reads.
//   A  pre-split pieces (AMODE 2: planes 0, 1) and B always pre-split planes, k-TILED in memory: [K/16][2][rows][16] so that the
//      32 bytes a row contributes to a k-tile sit next to the neighbouring rows' (full 128-byte lines per DMA instead of a
//      quarter line per row, which the 32 KB L1 cannot keep until the next k-tile): weights from abx_split_weights_f16, or
//      activations written by a producer GEMM with C_split (the TriangleMultiplication einsum seqformer.py:490-493 takes
//      both operands this way).  LDS image [plane][row][32 B], the two 16-byte halves swapped on odd row-octets:
//      conflict-free ds_read_b128 fragment reads (lane -> row, lane >> 5 -> k half).
// Requirements (abx_gemm falls back to the exact kernel otherwise): K % 16 == 0, 16-byte aligned operand rows.
// Epilogue: gemm_epilogue.h (shared with gemm.hip).
#include "common.h"
#include "abx_hip.h"
#include "gemm_epilogue.h"
#include "rigid_dev.h"

namespace {

constexpr int BK = 16;
#ifndef ABX_SPLIT_MIN_TILES
#define ABX_SPLIT_MIN_TILES 256     // smaller problems (sequence track, IPA) stay on the exact fp32 kernels
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// 16 bytes per lane, global -> LDS at (wave-uniform) dst + lane * 16
__device__ __forceinline__ void glds16(const void* src, char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst_wave_base, 16, 0, 0);
}

// A per-lane 32-bit byte offset as the compiler must see it AT the DMA instruction: defined in the same basic block, so that the
// address is (scalar base) + zext(32-bit VGPR) and the load takes the scalar-base form `global_load_lds v, s[a:a+1]`.  Hoisted out
// of the k loop the zero-extension becomes a 64-bit VGPR pair and every DMA instruction a v_lshl_add_u64 first (4 VALU instructions
// and 4 VGPRs per k-step of the 128 x 128 tile).
__device__ __forceinline__ unsigned vgpr32(unsigned o) {
    asm volatile("" : "+v"(o));
    return o;
}

template <int N> __device__ __forceinline__ void wait_vm_and_barrier() {
    // own DMA writes for the next tile have landed (the newest N stay in flight), own LDS reads retired, then the block barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// byte offset of (plane, row, 16-byte half) in a [2][ROWS][16] 16-bit tile image
template <int ROWS>
__device__ __forceinline__ int plane_off(int plane, int row, int half) {
    return plane * (ROWS * 32) + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
}

// DMA source byte offsets (from the operand's batch base) of one wave for a [2][ROWS][16] 16-bit plane image stage: chunk c (1 KB of LDS) = wave * NL + i.
// Surplus chunks (image not a multiple of 4 KB) and rows past the matrix re-read a valid row; their LDS bytes are never used
// for valid outputs.
template <int ROWS, int NL, int NPL = 2>
__device__ __forceinline__ void plane_sources(long long s_plane, long long s_row, int r0, int R, unsigned (&off)[NL]) {
    // (the k-tile stride is applied by the caller: one k-tile = 16 consecutive k of every row)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int o = (wave * NL + i) * 1024 + lane * 16;                 // byte offset in the stage
        if (o >= NPL * ROWS * 32) o -= NPL * ROWS * 32;             // surplus chunk: duplicate of the image start
        const int plane = o / (ROWS * 32), rem = o % (ROWS * 32);
        const int row = rem >> 5, hp = (rem >> 4) & 1;
        const int half = hp ^ ((row >> 3) & 1);
        const int gr = min(r0 + row, R - 1);
        off[i] = (unsigned)((plane * s_plane + (long long)gr * s_row + 8 * half) * 2);
    }
}

// Main loop of one output tile: acc += A' B for the operands of `g`; for the inline LayerNorm also the per-row partial sums
// (ls, lq over this lane's k half, shifted by lshift).  Ends with all DMA drained and a block barrier (the LDS is free again).
// SWAP: the MFMA operands trade places, acc[i][j] holds the TRANSPOSED 32 x 32 tile (rows = the tile's B rows / output columns, lane =
// A row): the fused transition consumes it as the A operand of its second GEMM straight from the registers.
// STATS: accumulate the inline LayerNorm partial sums and set lshift (the per-row shift of the operand, part of the statistics'
// meaning); a caller that walks the same A rows again passes STATS = false and the lshift of its first walk.
// AMODE 3: the A operand is resident in LDS (a_lds: BM fp32 rows of a_lds_stride bytes, k-contiguous; written by the caller before
// the call): no A stage, no A DMA - the fused IPA tail chains its GEMMs this way.
template <int BM, int BN, int WM, int WN, int AMODE, bool SWAP = false, bool STATS = true, int RING = 2, int JGO = 0>
__device__ __forceinline__ void gemm3_mainloop(const AbxGemm& g, float* smem, int mt, int nt, int b, f32x16 (&acc)[WM / 32][WN / 32],
                                               float (&ls)[WM / 32], float (&lq)[WM / 32], float (&lshift)[WM / 32],
                                               const char* a_lds = nullptr, int a_lds_stride = 0) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int A_IMG = AMODE == 3 ? 0 : (AMODE == 2 ? 2 * BM * 32 : BM * 64);   // bytes per A stage (planes: the two pieces a0, a1)
    constexpr int NLA = AMODE == 3 ? 1 : (A_IMG + 4095) / 4096;                 // DMA instructions per wave per A tile
    constexpr int A_STAGE = RING == 2 ? A_IMG : (A_IMG + 4095) / 4096 * 4096;
    constexpr int B_IMG = 2 * BN * 32;
    constexpr int NLB = (B_IMG + 4095) / 4096;
    // with counted waits every wave must issue the same number of DMA instructions (stage padded to 4 KB multiples); the
    // 2-stage protocol waits for everything, so the surplus chunks are simply skipped and the stage is the bare image
    constexpr int B_STAGE = RING == 2 ? B_IMG : (B_IMG + 4095) / 4096 * 4096;   // (counted waits: the surplus chunks of a padded stage re-read the image start)
    // RING stages per operand: tile t + RING - 1 is fetched (DMA) while tile t is consumed.  RING = 2 (tile t + 1 waited for at the
    // end of step t) for the pair-stack GEMMs: with K <= 192 the fixed per-tile latencies (dispatch, first DMA, epilogue) weigh more
    // than pipeline depth, and the smaller LDS footprint buys a third / fourth resident block per CU (40 KB with 128x128 tiles, 52 KB
    // with 128x192) that hides them.  RING = 3 where ONE block owns the CU anyway (the IPA tail: 180 k-steps per block and nothing
    // else to hide a DMA round trip): two tiles in flight, counted waits (vmcnt: results return in issue order, so every wave must
    // issue the same NI instructions per tile - a stage is then padded to 4 KB and the surplus chunk loads rows nobody reads).
    constexpr int NI = (AMODE == 3 ? 0 : NLA) + NLB;
    constexpr int INFLIGHT = (RING - 2) * NI;                  // DMA instructions of this wave allowed to be outstanding at a wait

    char* As = reinterpret_cast<char*>(smem);                 // RING stages
    char* Bs = As + RING * A_STAGE;                           // RING stages
    const int m0 = mt * BM, n0 = nt * BN;
    // (the wave index as a scalar: the LDS destinations of the DMA instructions - m0 - are then SALU arithmetic, not a VALU
    // computation + v_readfirstlane per instruction)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int h = lane >> 5;

    // ---- DMA sources: a block-uniform base pointer (advanced per k-tile) plus per-lane 32-bit byte offsets, so the loads
    // use the scalar-base addressing form and cost no vector address arithmetic in the loop
    unsigned offsA[NLA], offsB[NLB];
    // The weight side first: its sources need two shifts, and with two stages its first k-tile is requested here, before the
    // integer divisions of the A-row maps below - a block's first DMA round trip is exposed (nothing of this block can run
    // before it lands), so it starts as early as the block knows where to read
    const char* baseB = reinterpret_cast<const char*>(
        g.B_split + ((g.batch_inner > 0 && g.A_split) ? (long long)(b / g.batch_inner) * g.sB3b + (long long)(b % g.batch_inner) * g.sB3i
                                                      : (long long)b * g.sB3b));
    plane_sources<BN, NLB>(g.sB3p, g.sB3n, n0, g.N, offsB);
    const long long b_step = g.sB3k * 2;
    const int nk = g.K / BK;
    auto issue_b = [&](int tile) {
        char* dst = Bs + (tile % RING) * B_STAGE + wave * NLB * 1024;
        const char* src = baseB + tile * b_step;
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            if (RING > 2 || B_IMG % 4096 == 0 || (wave * NLB + i) * 1024 < B_IMG) glds16(src + vgpr32(offsB[i]), dst + i * 1024);
    };
    if constexpr (RING == 2) issue_b(0);
    const char* baseA = nullptr;
    long long a_step = 0;                                      // bytes per k-tile
    if constexpr (AMODE == 3) {
        offsA[0] = 0;
    } else if constexpr (AMODE == 0) {
        // offsets are relative to the tile's first row (to the batch base when the rows are pair-transposed)
        const bool remap = g.a_pair_transpose > 0 || g.a_pair != 0;
        const long long row0 = remap ? 0 : m0;
        baseA = reinterpret_cast<const char*>(g.A + (long long)b * g.sAb + row0 * g.sAm);
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int r = (wave * NLA + i) * 16 + (lane >> 2), p = lane & 3;
            const int kq = p ^ ((r >> 2) & 3);                 // physical 16-byte slot p of row r holds logical k-quad kq
            const int gri = min(m0 + r, g.M - 1);                    // (32-bit: M < 2^31; 64-bit divisions cost ~80 instructions)
            long long gr = gri;
            if (g.a_pair) {
                // padded pair position (i, j) of the GEMM -> row of the unpadded pair tensor (pad columns re-read column L-1;
                // their outputs are zeroed by the row scale / never stored)
                int pi, pj;
                if (g.c_split_tile) {
                    pair_tile_decode(gri, g.pair_Lp, pi, pj);
                    pi = min(pi, g.pair_L - 1);
                    pj = min(pj, g.pair_L - 1);
                } else {
                    pi = gri / g.pair_Lp;
                    pj = min(gri - pi * g.pair_Lp, g.pair_L - 1);
                }
                gr = g.a_pair_transpose > 0 ? (long long)pj * g.pair_L + pi : (long long)pi * g.pair_L + pj;
            } else if (g.a_pair_transpose > 0) {
                const int qi = gri / g.a_pair_transpose;
                gr = (long long)(gri - qi * g.a_pair_transpose) * g.a_pair_transpose + qi;
            }
            offsA[i] = (unsigned)(((gr - row0) * g.sAm + kq * 4) * 4);
        }
        a_step = BK * 4;
    } else if constexpr (AMODE == 1) {
        baseA = reinterpret_cast<const char*>(g.A + (long long)b * g.sAb + m0);
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int kl = (wave * NLA + i) * 2 + (lane >> 5), mq = lane & 31;     // LDS [k][BM]: 2 k-rows per instruction
            const int gm = min(m0 + mq * 4, g.M - 4) - m0;                         // (M % 4 == 0: checked by the dispatch)
            offsA[i] = (unsigned)(((long long)kl * g.sAk + gm) * 4);
        }
        a_step = (long long)BK * g.sAk * 4;
    } else {
        baseA = reinterpret_cast<const char*>(g.A_split + (g.batch_inner > 0 ? (long long)(b / g.batch_inner) * g.sA3b + (long long)(b % g.batch_inner) * g.sA3i
                                                                                  : (long long)b * g.sA3b));
        plane_sources<BM, NLA>(g.sA3p, g.sA3m, m0, g.M, offsA);
        a_step = g.sA3k * 2;
    }

    auto issue_a = [&](int tile) {          // tile index clamped by the caller
        if constexpr (AMODE == 3) return;
        char* dst = As + (tile % RING) * A_STAGE + wave * NLA * 1024;
        const char* src = baseA + tile * a_step;
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            if (RING > 2 || A_IMG % 4096 == 0 || (wave * NLA + i) * 1024 < A_IMG) glds16(src + vgpr32(offsA[i]), dst + i * 1024);
    };

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float m2048 = -2048.0f;
    asm volatile("" : "+v"(m2048));                           // (a VGPR constant of split2h_mix; not rematerialised into an SGPR)
    const bool relu = g.a_relu != 0;
    const bool ln_inline = g.ln_csum != nullptr && g.ln_stats == nullptr;
    f32x2 ls2[TM], lq2[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        ls[i] = lq[i] = 0.f;
        if constexpr (STATS) lshift[i] = 0.f;
        ls2[i] = lq2[i] = (f32x2){0.f, 0.f};
    }

    // prologue
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) {
        issue_a(min(r, nk - 1));
        if constexpr (RING != 2) issue_b(min(r, nk - 1));
    }
    wait_vm_and_barrier<INFLIGHT>();

    // per-lane LDS fragment offsets
    int offA[TM][(AMODE == 0 || AMODE == 3) ? 2 : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + (lane & 31);
        if constexpr (AMODE == 3) {
            offA[i][0] = r * a_lds_stride + (2 * h) * 16;
            offA[i][1] = r * a_lds_stride + (2 * h + 1) * 16;
        } else if constexpr (AMODE == 0) {
            const int x = (r >> 2) & 3;
            offA[i][0] = r * 64 + (((2 * h) ^ x) << 4);
            offA[i][1] = r * 64 + (((2 * h + 1) ^ x) << 4);
        } else if constexpr (AMODE == 1) {
            offA[i][0] = (8 * h) * (BM * 4) + r * 4;
        } else {
            offA[i][0] = plane_off<BM>(0, r, h);
        }
    }
    int offB[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = plane_off<BN>(0, wn * WN + j * 32 + (lane & 31), h);

    // a wave whose 32 rows all lie beyond M (the last row tile of a ragged M: e.g. the 4th wave of the third 128-row tile of the
    // L = 352 contraction) takes part in the DMA and the barriers but skips its B-fragment reads and MFMAs
    const bool rows_live = m0 + wm * WM < g.M;
    if (!rows_live) {
        for (int t = 0; t < nk; ++t) {
            issue_b(min(t + RING - 1, nk - 1));
            issue_a(min(t + RING - 1, nk - 1));
            wait_vm_and_barrier<INFLIGHT>();
        }
    } else
    for (int t = 0; t < nk; ++t) {
        // next tiles
        issue_b(min(t + RING - 1, nk - 1));
        issue_a(min(t + RING - 1, nk - 1));
        const char* as = AMODE == 3 ? a_lds + t * 64 : As + (t % RING) * A_STAGE;
        const char* bs = Bs + (t % RING) * B_STAGE;
        u32x4 a[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (AMODE == 2) {
#pragma unroll
                for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const u32x4*>(as + offA[i][0] + p * (BM * 32));
            } else {
                float x[8];
                if constexpr (AMODE == 0 || AMODE == 3) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(as + offA[i][0]);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(as + offA[i][1]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { x[c] = lo[c]; x[4 + c] = hi[c]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const float*>(as + offA[i][0] + e * (BM * 4));
                }
                if (ln_inline) {
                    // statistics of the row from this lane's 8 k (the other k half sits in lane ^ 32), shifted by the row's
                    // first element so that E[x^2] - mean^2 does not cancel when |mean| >> sigma.  Packed math: two elements
                    // per VALU instruction (even / odd partial sums, folded after the loop)
                    if (STATS && t == 0) lshift[i] = __shfl(x[0], lane & 31, 64);
                    const f32x2 sh2 = {lshift[i], lshift[i]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 xp = {x[2 * e], x[2 * e + 1]};
                        xp -= sh2;
                        if constexpr (STATS) {
                            ls2[i] += xp;
                            lq2[i] = __builtin_elementwise_fma(xp, xp, lq2[i]);
                        }
                        x[2 * e] = xp[0];
                        x[2 * e + 1] = xp[1];
                    }
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
                }
                unsigned q0[4], q1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    split2h_mix(x[2 * e], x[2 * e + 1], m2048, q0[e], q1[e]);
                }
                a[i][0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
                a[i][1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
            }
        }
        // product terms, smallest first (SplitTerms); the B fragments are fetched per group of JG sub-tiles (register budget);
        // consecutive MFMAs hit different accumulators
        using T = SplitTerms;
        constexpr int JG = JGO > 0 ? JGO : (TN > 3 ? (TN % 3 == 0 ? 3 : 2) : TN);
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += JG) {
            u32x4 bb[JG][3];
#pragma unroll
            for (int j = 0; j < JG; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) bb[j][p] = *reinterpret_cast<const u32x4*>(bs + offB[j0 + j] + p * (BN * 32));
#pragma unroll
            for (int j = 0; j < JG; ++j) bb[j][2] = f16x8_lo(bb[j][0]);
#pragma unroll
            for (int term = 0; term < T::N; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < JG; ++j)
                        acc[i][j0 + j] = SWAP ? mfma_split(bb[j][T::B[term]], a[i][T::A[term]], acc[i][j0 + j])
                                              : mfma_split(a[i][T::A[term]], bb[j][T::B[term]], acc[i][j0 + j]);
        }
        wait_vm_and_barrier<INFLIGHT>();
    }
    if constexpr (AMODE != 2) {
        // the planes hold w * 2^b_exp, the activations went in as x * 2^ABX_F16_A_EXP: exact power-of-two rescale
        // (plane x plane: the images were written as x 2^-4 and y 2^4, nothing to undo)
        const float cs = __builtin_ldexpf(1.0f, -ABX_F16_A_EXP - g.b_exp);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= cs;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        ls[i] = ls2[i][0] + ls2[i][1];
        lq[i] = lq2[i][0] + lq2[i][1];
    }
    // drain the (redundant) tail DMA before the epilogue reuses the LDS
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

// LayerNorm row statistics of this M-panel -> st_lds [BM][2] (mean - shift, rstd); returns whether the GEMM is LN-folded
template <int BM, int BN, int WM, int WN, bool EDGE>
__device__ __forceinline__ bool gemm3_row_stats(const AbxGemm& g, float* st_lds, int mt, int b, const float (&ls)[WM / 32],
                                                const float (&lq)[WM / 32]) {
    constexpr int TM = WM / 32, WAVES_N = BN / WN;
    const int m0 = mt * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, h = lane >> 5;
    const bool ln_inline = g.ln_csum != nullptr && g.ln_stats == nullptr;
    const float* gstats = g.ln_stats ? g.ln_stats + 2 * (long long)b * g.sSb : nullptr;
    if (gstats) {
        for (int idx = threadIdx.x; idx < 2 * BM; idx += 256) {
            const int m = m0 + (idx >> 1);
            st_lds[idx] = (!EDGE || m < g.M) ? gstats[2 * (long long)m + (idx & 1)] : 0.f;
        }
    } else if (ln_inline) {
        const float invK = 1.0f / (float)g.K;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float sm = ls[i] + __shfl_xor(ls[i], 32, 64), sq = lq[i] + __shfl_xor(lq[i], 32, 64);
            if (wn == 0 && h == 0) {                            // the wn waves hold identical copies
                const int row = wm * WM + i * 32 + lane;
                const float dm = sm * invK;
                st_lds[2 * row] = dm;                           // the operand was shifted: only (mean - shift) remains
                st_lds[2 * row + 1] = 1.0f / sqrtf(fmaxf(sq * invK - dm * dm, 0.f) + g.ln_eps);
            }
        }
    }
    return gstats != nullptr || ln_inline;
}

template <int BM, int BN, int WM, int WN, int AMODE, bool EDGE, bool TS, bool OLN = false, bool PROBE = true, int RING = 2, int JGO = 0>
__device__ __forceinline__ void gemm3_block(const AbxGemm& g, float* smem, int mt, int nt, int b) {
    constexpr int TM = WM / 32, TN = WN / 32;
    f32x16 acc[TM][TN];
    float ls[TM], lq[TM], lsh[TM];
    gemm3_mainloop<BM, BN, WM, WN, AMODE, false, true, RING, JGO>(g, smem, mt, nt, b, acc, ls, lq, lsh);
    float* st_lds = smem;                                   // [BM][2]
    const bool stats = gemm3_row_stats<BM, BN, WM, WN, EDGE>(g, st_lds, mt, b, ls, lq);
    __syncthreads();
    gemm_epilogue<BM, BN, WM, WN, EDGE, TS, OLN, PROBE>(g, st_lds, smem + 2 * BM, acc, mt * BM, nt * BN, b, stats);
}

// Dual GEMM (the TriangleMultiplication tail, seqformer.py:496-503): out = epi(A' B) * sigmoid(LN(A2) B2 + bias2) (+ resid).
// Two main loops over the same output tile into two accumulator sets: the channel-major product (AMODE 1) against proj_out, then
// the rows of z (k-contiguous, inline LayerNorm) against the final-gate weights; the gate never travels through HBM and z is
// read once for both the gate and the residual.
template <int BM, int BN, int WM, int WN, bool EDGE, int JGO = 0, int TGO = 0>
__device__ __forceinline__ void gemm3_dual_block(const AbxGemm& g, float* smem, int mt, int nt, int b) {
    constexpr int TM = WM / 32, TN = WN / 32;
    f32x16 acc[TM][TN], acc2[TM][TN];
    float ls[TM], lq[TM], ls2[TM], lq2[TM], lsh[TM], lsh2[TM];
    AbxGemm g2 = g;                                          // operand view of the gate GEMM
    g2.A = g.A2; g2.sAb = g.sA2b; g2.sAm = g.sA2m; g2.sAk = 1; g2.K = g.K2;
    g2.A_split = nullptr; g2.a_relu = 0; g2.a_pair_transpose = 0; g2.a_pair = g.pair_Lp > 0 ? 1 : 0;
    g2.B_split = g.B2_split; g2.sB3p = g.sB23p; g2.sB3n = g.sB23n; g2.sB3k = g.sB23k; g2.sB3b = 0;
    g2.ln_csum = g.ln2_csum; g2.ln_stats = nullptr; g2.batch_inner = 0; g2.b_exp = g.b2_exp;
    if (g.sAk == 1) gemm3_mainloop<BM, BN, WM, WN, 0, false, true, 2, JGO>(g, smem, mt, nt, b, acc, ls, lq, lsh);
    else gemm3_mainloop<BM, BN, WM, WN, 1, false, true, 2, JGO>(g, smem, mt, nt, b, acc, ls, lq, lsh);
    gemm3_mainloop<BM, BN, WM, WN, 0, false, true, 2, JGO>(g2, smem, mt, nt, b, acc2, ls2, lq2, lsh2);
    float* st_lds = smem;                                   // [BM][2] + [BM][2]
    const bool stats = gemm3_row_stats<BM, BN, WM, WN, EDGE>(g, st_lds, mt, b, ls, lq);
    gemm3_row_stats<BM, BN, WM, WN, EDGE>(g2, st_lds + 2 * BM, mt, b, ls2, lq2);
    __syncthreads();
    gemm_epilogue<BM, BN, WM, WN, EDGE, false, false, true, TGO>(g, st_lds, smem + 4 * BM, acc, mt * BM, nt * BN, b, stats, &acc2, st_lds + 2 * BM);
}

// Round 6: the same dual GEMM with ONE walk over the rows of z.  The 128 x 96 tiles above give every row tile two blocks (one per column
// half) that each walk the product rows AND the z rows: the A side of both main loops (DMA from the fabric, fragment reads, LayerNorm
// partial sums, f16 splits) is paid twice per row, 9 MFMAs per k-step.  Under the board's power limit that is what the launch costs
// (profiles/r06a_power_limit.txt; the gated attention tail gained 11 % from the same change).  Here a block owns 128 rows and all
// 192 columns:
//   gate walk   LN(z) against the final-gate weights for all 192 columns (128 x 192 tile, 18 MFMAs per k-step), 96 accumulator
//               registers, turned into the gate VALUES sigmoid(.) in place and kept;
//   two passes  over the output columns (0 .. 95, 96 .. 191): the product rows against proj_out into 48 accumulator registers, then the
//               epilogue of that half: folded LayerNorm, bias, x the kept gate values, + z, store.
// z is walked once instead of twice, the product rows twice as before.  Every accumulator receives the products of the two-tile form
// in the same order and the gate is the same expression: bit-identical results.  MEASURED SLOWER (10.97 vs 10.30 ms at 100 samples of
// L = 352, profiles/r06c_kb_dual.txt): 96 + 48 live accumulator registers leave two blocks per CU where the two-tile form runs three;
// kept behind AbxGemm.tune bit 7 as the record of the experiment and a cross-check (test_gemm_dual_walk_variants_bit_identical).
template <bool EDGE>
__device__ __forceinline__ void gemm3_dual1_block(const AbxGemm& g, float* smem, int mt, int b) {
    constexpr int BM = 128, BN = 192, WM = 32, BH = 96, TH = BH / 32, TN = BN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    AbxGemm g2 = g;                                          // operand view of the gate GEMM
    g2.A = g.A2; g2.sAb = g.sA2b; g2.sAm = g.sA2m; g2.sAk = 1; g2.K = g.K2;
    g2.A_split = nullptr; g2.a_relu = 0; g2.a_pair_transpose = 0; g2.a_pair = g.pair_Lp > 0 ? 1 : 0;
    g2.B_split = g.B2_split; g2.sB3p = g.sB23p; g2.sB3n = g.sB23n; g2.sB3k = g.sB23k; g2.sB3b = 0;
    g2.ln_csum = g.ln2_csum; g2.ln_stats = nullptr; g2.batch_inner = 0; g2.b_exp = g.b2_exp;
    float* st_lds = smem;                                   // [BM][2] (product rows) + [BM][2] (z rows)
    f32x16 gatev[1][TN];
    {
        float ls2[1], lq2[1], lsh2[1];
        gemm3_mainloop<BM, BN, WM, BN, 0, false, true, 2>(g2, smem, mt, 0, b, gatev, ls2, lq2, lsh2);
        gemm3_row_stats<BM, BN, WM, BN, EDGE>(g2, st_lds + 2 * BM, mt, b, ls2, lq2);
        __syncthreads();
        const float* st2 = st_lds + 2 * BM;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + (lane & 31);
            const bool nok = !EDGE || n < g.N;
            const float csum2 = nok ? g.ln2_csum[n] : 0.f, bias2 = (g.bias2 && nok) ? g.bias2[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wave * WM + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                // (gemm_epilogue.h, the acc2 path: the same expression, the same rounding)
                const float gv = st2[2 * ml + 1] * (gatev[0][j][r] - st2[2 * ml] * csum2) + bias2;
                gatev[0][j][r] = sigmoidf_(gv);
            }
        }
        __syncthreads();                                    // (the statistics are read: the next main loop's stages overlay them)
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 acc[1][TH];
        float ls[1], lq[1], lsh[1];
        if (g.sAk == 1) gemm3_mainloop<BM, BH, WM, BH, 0, false, true, 2>(g, smem, mt, half, b, acc, ls, lq, lsh);
        else gemm3_mainloop<BM, BH, WM, BH, 1, false, true, 2>(g, smem, mt, half, b, acc, ls, lq, lsh);
        const bool stats = gemm3_row_stats<BM, BH, WM, BH, EDGE>(g, st_lds, mt, b, ls, lq);
        __syncthreads();
        gemm_epilogue<BM, BH, WM, BH, EDGE, false, false, true>(g, st_lds, smem + 4 * BM, acc, mt * BM, half * BH, b, stats, nullptr, nullptr,
                                                                reinterpret_cast<const f32x16 (*)[1][TH]>(&gatev[0][half * TH]));
        if (half == 0) __syncthreads();                     // (the scratch and the statistics are read: the next main loop overlays them)
    }
}

__global__ __launch_bounds__(256, 2) void gemm3_dual1_kernel(const AbxGemm g) {
    constexpr int OPER = (2 * 128 * 64 + 2 * 2 * 192 * 32) / 4;
    constexpr int EPI = 4 * 128 + 4 * 32 * (3 * 32 + 4);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntm = (g.M + 127) / 128;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const ClockProbe probe(g.clock_probe);
    if ((mt + 1) * 128 <= g.M && g.N == 192) gemm3_dual1_block<false>(g, smem, mt, b);
    else gemm3_dual1_block<true>(g, smem, mt, b);
    probe.finish();
}

template <int BM, int BN, int WM, int WN, int AMODE, bool TS, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm3_kernel(const AbxGemm g) {
    constexpr int A_IMG = AMODE == 2 ? 2 * BM * 32 : BM * 64;
    constexpr int A_STAGE = A_IMG;
    constexpr int B_STAGE = 2 * BN * 32;
    constexpr int RING = 2;       // (3 stages at 3 blocks per CU, 4 at 2: 4.08 / 4.67 ms against 3.68 at N = 768, K = 192, 20 samples - DESIGN.md 4c)
    constexpr int OPER = RING * (A_STAGE + B_STAGE) / 4;                           // floats
    constexpr int TNW = WN / 32, TGW = TNW > 3 ? (TNW % 3 == 0 ? 3 : 2) : TNW;      // epilogue column group (gemm_epilogue.h)
    constexpr int SCR = 4 * 32 * ((TS ? WM : TGW * 32) + 4);
    constexpr int EPI = 2 * BM + SCR;
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    // 1-D grid over (batch, m-tile, n-tile).  XCD-aware remap (blocks are placed round-robin over the 8 XCDs): every XCD gets a
    // contiguous range of tiles, so the N-tiles that share an A panel - and all tiles of one batch entry of the
    // triangle-multiplication contraction - meet in one private L2.  Bijective for any grid size.
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const int per_batch = ntn * ntm;
    const ClockProbe probe(g.clock_probe);
    constexpr bool PROBE = !(BM == 128 && BN == 128 && MINW >= 4);      // (no register for it at four blocks per CU: gemm_epilogue.h)
#ifdef ABX_PERSIST
    if (g.tune & 256) {
        // experiment: resident blocks walk the tiles of their XCD's range (slot s takes x0 + s, x0 + s + slots, ...)
        const long long total = (long long)per_batch * g.batch;
        const long long q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
        const long long x0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, xn = q + (xcd < r ? 1 : 0);
        for (long long i = slot; i < xn; i += slots) {
            const long long wgid = x0 + i;
            const int b = (int)(wgid / per_batch);
            const int rem = (int)(wgid - (long long)b * per_batch);
            const int mt = rem / ntn, nt = rem % ntn;
            const bool interior = (mt + 1) * BM <= g.M && (nt + 1) * BN <= g.N;
            if (interior) gemm3_block<BM, BN, WM, WN, AMODE, false, TS, false, PROBE, RING>(g, smem, mt, nt, b);
            else gemm3_block<BM, BN, WM, WN, AMODE, true, TS, false, PROBE, RING>(g, smem, mt, nt, b);
            __syncthreads();
        }
        return;
    }
#endif
    // (32-bit: the dispatch keeps the grid below 2^31 tiles; a 64-bit division is ~80 instructions in front of the block's first DMA)
    unsigned wgid = blockIdx.x;
    if (!(g.tune & 1)) {
        const unsigned nwg = gridDim.x, bid = blockIdx.x;
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int b = (int)(wgid / (unsigned)per_batch);
    const int rem = (int)(wgid - (unsigned)b * (unsigned)per_batch);
    const int mt = rem / ntn, nt = rem % ntn;
    const bool interior = (mt + 1) * BM <= g.M && (nt + 1) * BN <= g.N;
    if (interior) gemm3_block<BM, BN, WM, WN, AMODE, false, TS, false, PROBE, RING>(g, smem, mt, nt, b);
    else gemm3_block<BM, BN, WM, WN, AMODE, true, TS, false, PROBE, RING>(g, smem, mt, nt, b);
    probe.finish();
}

// A 128 x 128 plain-store GEMM with a SIDE GEMM in its grid: the skinny (N <= 32, transposed store) projection of the SAME rows - the
// triangle attention's pair bias next to its q | k | v | gate projection (seqformer.py:520-531).  Side tiles (128 rows x 32 columns,
// the block body of the narrow kernel) are dealt evenly between the main tiles in launch order, so a side tile runs while the main
// tiles of the same rows are resident on its XCD and takes its A panel from that L2 instead of a second 9.5 GB HBM read of z in a
// launch of its own.  Every tile's arithmetic is what the two separate launches do: results are bit-identical.
__global__ __launch_bounds__(256, 4) void gemm3_side_kernel(const AbxGemm g, const AbxGemm s2) {
    constexpr int BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) float smem[35840 / 4];
    const unsigned ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM, per_batch = ntn * ntm;
    const unsigned stm = (s2.M + BM - 1) / BM, ts = stm * (unsigned)s2.batch;
    const unsigned total = gridDim.x, bid = blockIdx.x;
    const unsigned q = total >> 3, r = total & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    // slot w is a side tile when floor((w + 1) ts / total) > floor(w ts / total); it is side tile floor(w ts / total)
    const unsigned sw = (unsigned)(((unsigned long long)wgid * ts) / total), sw1 = (unsigned)(((unsigned long long)(wgid + 1) * ts) / total);
    if (sw1 > sw) {
        const int b = (int)(sw / stm), mt = (int)(sw - (unsigned)b * stm);
        if ((mt + 1) * BM <= s2.M && 32 <= s2.N) gemm3_block<BM, 32, 32, 32, 0, false, true>(s2, smem, mt, 0, b);
        else gemm3_block<BM, 32, 32, 32, 0, true, true>(s2, smem, mt, 0, b);
        return;
    }
    const unsigned w = wgid - sw;
    const int b = (int)(w / per_batch);
    const int rem = (int)(w - (unsigned)b * per_batch);
    const int mt = rem / (int)ntn, nt = rem % (int)ntn;
    if ((mt + 1) * BM <= g.M && (nt + 1) * BN <= g.N) gemm3_block<BM, BN, 32, 128, 0, false, false, false, false>(g, smem, mt, nt, b);
    else gemm3_block<BM, BN, 32, 128, 0, true, false, false, false>(g, smem, mt, nt, b);
}

// Probe (round 6, AbxGemm.tune bit 9): the skinny N <= 32 projections - HBM streams of 9.5 / 6.3 GB at 4.2 - 4.5 TB/s - with a deeper operand ring
// (RING k-tiles of A in flight per block instead of one); bit-identical to gemm3_kernel<128, 32, 32, 32, 0, TS, 6>
template <bool TS, int RING, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm3_narrow_ring_kernel(const AbxGemm g) {
    constexpr int BM = 128, BN = 32;
    constexpr int OPER = RING * (BM * 64 + 4096) / 4;
    constexpr int EPI = 2 * BM + 4 * 32 * ((TS ? 32 : 32) + 4);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntm = (g.M + BM - 1) / BM;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    if ((mt + 1) * BM <= g.M && BN <= g.N) gemm3_block<BM, BN, 32, 32, 0, false, TS, false, true, RING>(g, smem, mt, 0, b);
    else gemm3_block<BM, BN, 32, 32, 0, true, TS, false, true, RING>(g, smem, mt, 0, b);
}

// Linear -> LayerNorm over the output row (out_ln): k-contiguous fp32 A, one n-tile, plain store
template <int BM, int BN, int WM, int WN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm3_oln_kernel(const AbxGemm g) {
    constexpr int OPER = (2 * BM * 64 + 2 * 2 * BN * 32) / 4;
    constexpr int TNW = WN / 32, TGW = TNW > 3 ? (TNW % 3 == 0 ? 3 : 2) : TNW;
    constexpr int EPI = 2 * BM + 4 * 32 * (TGW * 32 + 4);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntm = (g.M + BM - 1) / BM;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const bool interior = (mt + 1) * BM <= g.M && BN <= g.N;
    if (interior) gemm3_block<BM, BN, WM, WN, 0, false, false, true>(g, smem, mt, 0, b);
    else gemm3_block<BM, BN, WM, WN, 0, true, false, true>(g, smem, mt, 0, b);
}

template <int BM, int BN, int WM, int WN, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm3_dual_kernel(const AbxGemm g) {
    constexpr int A_STAGE = BM * 64, B_STAGE = 2 * BN * 32;
    constexpr int OPER = (2 * A_STAGE + 2 * B_STAGE) / 4;
    // four blocks per CU (MINW 4): one B sub-tile in flight and 32-column store groups keep the two accumulator sets inside 128
    // VGPRs and the block inside 40 KB of LDS
    constexpr int JGO = MINW >= 4 ? 1 : 0, TGO = MINW >= 4 ? 1 : 0;
    constexpr int TNW = WN / 32, TGW = TGO > 0 ? TGO : (TNW > 3 ? (TNW % 3 == 0 ? 3 : 2) : TNW);
    constexpr int EPI = 4 * BM + 4 * 32 * (TGW * 32 + 4);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per_batch = ntn * ntm;
    const int b = (int)(wgid / (unsigned)per_batch);
    const int rem = (int)(wgid - (unsigned)b * (unsigned)per_batch);
    const int mt = rem / ntn, nt = rem % ntn;
    const bool interior = (mt + 1) * BM <= g.M && (nt + 1) * BN <= g.N;
    if (interior) gemm3_dual_block<BM, BN, WM, WN, false, JGO, TGO>(g, smem, mt, nt, b);
    else gemm3_dual_block<BM, BN, WM, WN, true, JGO, TGO>(g, smem, mt, nt, b);
}

// Fused two-layer transition (seqformer.py:358-376: LayerNorm -> Linear(4x) -> ReLU -> Linear, + residual) for 128 rows per block:
//     out = relu(LN(A) W1 + b1) W2 + b2 (+ resid)          A: k-contiguous fp32 rows (K = g.K), hidden width g.N, output width g.N2
// The hidden activations never exist in memory.  The hidden dimension is walked in chunks of 128:
//   GEMM 1 of a chunk is the ordinary main loop with the MFMA operands SWAPPED, so a wave's accumulators hold the TRANSPOSED tiles
//   H^T[hidden][row]: lane = row, registers = 16 hidden channels.  Folded LayerNorm (the row statistics are lane-local in this
//   orientation), bias and ReLU are applied to the registers, which are then split into f16 pieces and ARE the A operand of GEMM 2
//   (lane = row, 8 k values per k-step): a 32 x 32 tile feeds two k-steps whose k slots (lane half h, slot i) hold the hidden
//   channel 8 (i >> 2) + 4 h + (i & 3) of a 16-channel k-tile.  The W2 planes are stored with exactly that order inside every
//   k-tile (AbxGemm.B2_split of an mlp descriptor; abx_amd.ops.permute_k16), so its fragments are the usual 16-byte reads.
//   GEMM 2 streams the chunk's 8 k-tiles of W2 (all N2 <= 192 columns) through a double-buffered LDS stage of its own; its first
//   tile is requested before GEMM 1 of the chunk starts.
// A is read from HBM once (the 6 chunk walks of a block hit the L2 / MALL), the hidden never leaves the CU, the output rows are
// written once: 2 x 192 floats of HBM traffic per pair row instead of 2 x 192 + 2 x 768.
constexpr int MLP_NH = 768;                                                  // widest hidden layer of the fused transition (LDS table)
// R1: stages of GEMM 1's operand ring (2: a step waits for the tiles it requested at its own start; 3: requested one step earlier, counted waits)
template <bool EDGE, int R1 = 2>
__device__ __forceinline__ void gemm3_mlp_block(const AbxGemm& g, float* smem, int mt, int b) {
    constexpr int BM = 128, BN = 128, WM = 32, WN = 128, BN2 = 192, TN2 = BN2 / 32;
    constexpr int G1_BYTES = R1 * (BM * 64 + 2 * BN * 32);                   // stages of GEMM 1 (A fp32 + W1 planes)
    constexpr int B2_IMG = 2 * BN2 * 32;                                     // one k-tile of W2: [2][192][16] f16
    constexpr int NL2 = (B2_IMG + 4095) / 4096;
    char* W2s = reinterpret_cast<char*>(smem) + G1_BYTES;                    // 2 stages
    // column sums and biases of the first layer in LDS [2][768] (read per k-step of GEMM 2: as global loads every step waited out an L2
    // round trip - and, in issue order, the weight DMA requested before them)
    float* cst = reinterpret_cast<float*>(W2s + 2 * B2_IMG);
    for (int i = threadIdx.x; i < MLP_NH; i += 256) {
        cst[i] = i < g.N ? g.ln_csum[i] : 0.f;
        cst[MLP_NH + i] = (i < g.N && g.bias) ? g.bias[i] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    f32x16 acc2[1][TN2];
#pragma unroll
    for (int t = 0; t < TN2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][t][r] = 0.f;
    // DMA sources of the W2 k-tiles (all N2 columns)
    unsigned offs2[NL2];
    plane_sources<BN2, NL2>(g.sB23p, g.sB23n, 0, g.N2, offs2);
    const char* base2 = reinterpret_cast<const char*>(g.B2_split);
    const long long step2 = g.sB23k * 2;
    auto issue_w2 = [&](int ktile, int stage) {
        char* dst = W2s + stage * B2_IMG + wave * NL2 * 1024;
        const char* src = base2 + ktile * step2;
#pragma unroll
        for (int i = 0; i < NL2; ++i)
            if (B2_IMG % 4096 == 0 || (wave * NL2 + i) * 1024 < B2_IMG) glds16(src + offs2[i], dst + i * 1024);
    };
    // fragment of n-tile t: + t * 1024 bytes (32 rows of 32 bytes; the half swap of plane_off depends on (row >> 3) & 1 only)
    const int offB2 = plane_off<BN2>(0, lane & 31, h);
    constexpr int TG2 = 2;                                                   // n-tiles per fragment group (register budget)
    const bool rows_live = mt * BM + wave * WM < g.M;

    float ls[1], lq[1], lsh[1] = {0.f};
    float rstd = 0.f, dmean = 0.f;
    const int nchunk = (g.N + BN - 1) / BN;
    for (int c = 0; c < nchunk; ++c) {
        issue_w2(c * (BN / 16), 0);                                          // first W2 k-tile of the chunk: lands under GEMM 1
        f32x16 acc1[1][BN / 32];
        if (c == 0) {
            gemm3_mainloop<BM, BN, WM, WN, 0, true, true, R1>(g, smem, mt, c, b, acc1, ls, lq, lsh);
            // row statistics of this lane's row (the two lane halves hold the two k halves)
            const float invK = 1.0f / (float)g.K;
            const float sm = ls[0] + __shfl_xor(ls[0], 32, 64), sq = lq[0] + __shfl_xor(lq[0], 32, 64);
            dmean = sm * invK;
            rstd = 1.0f / sqrtf(fmaxf(sq * invK - dmean * dmean, 0.f) + g.ln_eps);
        } else {
            gemm3_mainloop<BM, BN, WM, WN, 0, true, false, R1>(g, smem, mt, c, b, acc1, ls, lq, lsh);
        }
        // (the main loop ended with every DMA drained - the W2 tile included - and a block barrier)
        const int hid0 = c * BN;
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) {
            // ---- H^T tile j: folded LayerNorm, bias, ReLU, split -> A fragments of two k-steps (one k-step = 8 registers of the tile),
            // each followed by its k-tile of GEMM 2
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 hf[2];
                {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int hc = hid0 + j * 32 + 8 * (2 * s2 + q) + 4 * h;    // 4 consecutive hidden channels
                        const int hcl = min(hc, MLP_NH - 4);                 // (channels beyond N: zeros in the table, masked below)
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(cst + hcl), bi = *reinterpret_cast<const f32x4*>(cst + MLP_NH + hcl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = rstd * (acc1[0][j][8 * s2 + 4 * q + e] - dmean * cs[e]) + bi[e];
                            x = relu_keep_nan(x);
                            v[4 * q + e] = (hc + e < g.N) ? x : 0.f;
                        }
                    }
                    // the hidden activations enter GEMM 2 as UNLIFTED pieces of h 2^4 (a0 = f16(h'), a1 = f16(h' - a0): what split2b writes
                    // for plane operands), so its three terms a1 p0 + a0 p1 + a0 p0 need no p2 = p0 2^-11 derived from the W2 fragments
                    // (192 v_pk_mul_f16 per chunk walk and 8 registers less).  |h| < 4094 (beyond: NaN -> the range probe -> the exact
                    // kernels); 22 significant bits from |h| = 2^-7, 2^-29 absolute below - post-ReLU hidden values next to an O(1) sum
                    unsigned q0[4], q1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split2b(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
                    hf[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
                    hf[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
                }
                const int kl = 2 * j + s2;                                   // k-tile of the chunk, stage kl & 1
                const int knext = c * (BN / 16) + kl + 1;
                if (kl + 1 < BN / 16 && knext * 16 < g.N) issue_w2(knext, (kl + 1) & 1);
                const char* ws = W2s + (kl & 1) * B2_IMG + offB2;
                if (rows_live && (c * (BN / 16) + kl) * 16 < g.N) {
#pragma unroll
                    for (int t0 = 0; t0 < TN2; t0 += TG2) {
                        u32x4 wb[TG2][2];
#pragma unroll
                        for (int t = 0; t < TG2; ++t)
#pragma unroll
                            for (int p = 0; p < 2; ++p) wb[t][p] = *reinterpret_cast<const u32x4*>(ws + (t0 + t) * 1024 + p * (BN2 * 32));
                        // smallest first: a1 p0, a0 p1, a0 p0
                        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int t = 0; t < TG2; ++t)
                                acc2[0][t0 + t] = mfma_split(hf[TA[term]], wb[t][TB[term]], acc2[0][t0 + t]);
                    }
                }
                wait_vm_and_barrier<0>();
            }
        }
    }
    {
        const float cs2 = __builtin_ldexpf(1.0f, -4 - g.b2_exp);             // (the hidden went in as h 2^4: split2b)
#pragma unroll
        for (int t = 0; t < TN2; ++t) acc2[0][t] *= cs2;
    }
    // ---- epilogue: + bias2 (+ resid), plain store.  A view of the descriptor whose "GEMM" is the second layer.
    AbxGemm g2 = g;
    g2.N = g.N2; g2.bias = g.bias2; g2.ln_csum = nullptr; g2.ln_stats = nullptr; g2.act = 0; g2.alpha = 1.0f;
    __syncthreads();
    gemm_epilogue<BM, BN2, WM, BN2, EDGE, false>(g2, smem, smem + 2 * BM, acc2, mt * BM, 0, b, false);
}

template <int MINB>
__global__ __launch_bounds__(256, MINB) void gemm3_mlp_kernel(const AbxGemm g) {
    constexpr int OPER = (2 * 128 * 64 + 2 * 2 * 128 * 32 + 2 * 2 * 192 * 32) / 4 + 2 * MLP_NH;           // floats (stages + the constants table)
    constexpr int EPI = 2 * 128 + 4 * 32 * (3 * 32 + 4);
    __shared__ __attribute__((aligned(16))) float smem[OPER > EPI ? OPER : EPI];
    const int ntm = (g.M + 127) / 128;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const ClockProbe probe(g.clock_probe);
    if ((mt + 1) * 128 <= g.M && g.N2 == 192) gemm3_mlp_block<false>(g, smem, mt, b);
    else gemm3_mlp_block<true>(g, smem, mt, b);
    probe.finish();
}

// the same with a three-stage ring for GEMM 1 (dynamic LDS: 79.9 KB, still two blocks per CU)
constexpr int MLP3_LDS = 3 * (128 * 64 + 2 * 128 * 32) + 2 * 2 * 192 * 32 + 2 * MLP_NH * 4;
__global__ __launch_bounds__(256, 2) void gemm3_mlp3_kernel(const AbxGemm g) {
    extern __shared__ __attribute__((aligned(16))) float mlp3_smem[];
    const int ntm = (g.M + 127) / 128;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const ClockProbe probe(g.clock_probe);
    if ((mt + 1) * 128 <= g.M && g.N2 == 192) gemm3_mlp_block<false, 3>(g, mlp3_smem, mt, b);
    else gemm3_mlp_block<true, 3>(g, mlp3_smem, mt, b);
    probe.finish();
}

// Gated tail of the TriangleAttention (seqformer.py:300-312: gate = sigmoid(gate_proj(LN z)), out = proj_out(gate * o), z += out) for 128
// rows per block, in the shape of the fused transition above:
//     out = (sigmoid(LN(A) Wg + bg) * G) Wo + bo (+ resid)        A: k-contiguous fp32 rows (the pair rows z, K = g.K), G = g.gate: the
//                                                                attention output o, [M][N] fp32 rows; N = g.N gate channels, N2 outputs
// The gate never exists in memory (9.5 GB written by the q | k | v | gate projection and read back by the attention kernel per block at
// 100 samples of L = 352) and the output projection is no launch of its own.  The gate channels are walked in chunks of 96:
//   GEMM 1 of a chunk = the main loop with swapped MFMA operands: the accumulators hold the transposed tiles gate^T[channel][row], lane =
//   row; folded LayerNorm (lane-local statistics), bias, sigmoid on the registers;
//   the G operand streams as k-tiles [128 rows][16 channels] fp32 through a double-buffered LDS stage of its own (DMA, the A-stage layout of
//   the main loop); a lane reads the 8 channels its 8 gate registers of a k-step stand for (channels 8 q + 4 h + e of the 16-tile: the
//   k order of permute_k16), multiplies, splits (unlifted pieces of v 2^4: split2b) - that IS the A operand of GEMM 2;
//   GEMM 2 streams the W_o planes (k order permuted inside every 16-tile like the fused transition's second layer).
constexpr int GT_NG = 192;                                                   // gate channels (LDS table)
template <bool EDGE>
__device__ __forceinline__ void gemm3_gtail2w_block(const AbxGemm& g, float* smem, int mt, int b) {
    constexpr int BM = 128, BN = 96, WM = 32, WN = 96, BN2 = 192, TN2 = BN2 / 32;
    constexpr int G1_BYTES = 2 * BM * 64 + 2 * 2 * BN * 32;                  // stages of GEMM 1 (A fp32 + gate-weight planes)
    constexpr int B2_IMG = 2 * BN2 * 32;                                     // one k-tile of W_o: [2][192][16] f16
    constexpr int NL2 = (B2_IMG + 4095) / 4096;
    constexpr int G_IMG = BM * 64;                                           // one k-tile of G: fp32 [128][16]
    char* W2s = reinterpret_cast<char*>(smem) + G1_BYTES;                    // 2 stages
    // G comes from HBM (the W_o planes from the L2): THREE stages, requested two k-steps ahead - with two, every k-step of GEMM 2 waited
    // out an HBM round trip that had started less than one step before (11.7 ms per launch at 100 samples of L = 352)
    char* Gs = W2s + 2 * B2_IMG;
    // column sums and biases of the gate projection in LDS [2][192] (see gemm3_mlp_block)
    float* cst = reinterpret_cast<float*>(Gs + 3 * G_IMG);
    for (int i = threadIdx.x; i < GT_NG; i += 256) {
        cst[i] = i < g.N ? g.ln_csum[i] : 0.f;
        cst[GT_NG + i] = (i < g.N && g.bias) ? g.bias[i] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5;
    const int m0 = mt * BM;
    f32x16 acc2[1][TN2];
#pragma unroll
    for (int t = 0; t < TN2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][t][r] = 0.f;
    unsigned offs2[NL2];
    plane_sources<BN2, NL2>(g.sB23p, g.sB23n, 0, g.N2, offs2);
    const char* base2 = reinterpret_cast<const char*>(g.B2_split);
    const long long step2 = g.sB23k * 2;
    // G k-tiles: wave w fetches rows 32 w .. 32 w + 31 (two 1 KB chunks of 16 rows), 16-byte slots XOR-swizzled through the source address
    unsigned offsG[2];
    const char* baseG = reinterpret_cast<const char*>(g.gate + (long long)b * g.sGb + (long long)m0 * g.sGm);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2), p = lane & 3;
        const int kq = p ^ ((r >> 2) & 3);
        const int gri = min(m0 + r, g.M - 1) - m0;
        offsG[i] = (unsigned)(((long long)gri * g.sGm + kq * 4) * 4);
    }
    static_assert(B2_IMG % 4096 == 0, "uniform DMA instruction counts per wave (counted waits)");
    auto issue_w2 = [&](int ktile) {                                         // k-tile kt of W_o -> stage kt & 1
        char* dst = W2s + (ktile & 1) * B2_IMG + wave * NL2 * 1024;
        const char* src = base2 + ktile * step2;
#pragma unroll
        for (int i = 0; i < NL2; ++i) glds16(src + vgpr32(offs2[i]), dst + i * 1024);
    };
    const int nkt2 = g.N / 16;                                               // k-tiles of GEMM 2 (N % 16 == 0)
    auto issue_g = [&](int ktile) {                                          // k-tile kt of G -> stage kt % 3; beyond the end: the last one again
        const int kt = min(ktile, nkt2 - 1);                                 // (uniform instruction counts; its stage is not read any more)
        char* dg = Gs + (ktile % 3) * G_IMG + wave * 2048;
        const char* sg = baseG + kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(sg + vgpr32(offsG[i]), dg + i * 1024);
    };
    const int offB2 = plane_off<BN2>(0, lane & 31, h);
    // this lane's 2 x 4 channels of a G k-tile: logical 16-byte slots h and 2 + h of its row
    const int grow = wave * WM + (lane & 31), gx = (grow >> 2) & 3;
    const int offG0 = grow * 64 + ((h ^ gx) << 4), offG1 = grow * 64 + (((2 + h) ^ gx) << 4);
    constexpr int TG2 = 2;
    const bool rows_live = m0 + wave * WM < g.M;

    float ls[1], lq[1], lsh[1] = {0.f};
    float rstd = 0.f, dmean = 0.f;
    constexpr int KC = BN / 16;                                              // k-tiles of GEMM 2 per chunk (6: even, the stage parity runs on)
    const int nchunk = (g.N + BN - 1) / BN;
    for (int c = 0; c < nchunk; ++c) {
        // first W_o k-tile and first TWO G k-tiles of the chunk: they land under GEMM 1 (whose main loop ends with every DMA drained)
        issue_w2(c * KC);
        issue_g(c * KC);
        issue_g(c * KC + 1);
        f32x16 acc1[1][BN / 32];
        if (c == 0) {
            gemm3_mainloop<BM, BN, WM, WN, 0, true, true>(g, smem, mt, c, b, acc1, ls, lq, lsh);
            const float invK = 1.0f / (float)g.K;
            const float sm = ls[0] + __shfl_xor(ls[0], 32, 64), sq = lq[0] + __shfl_xor(lq[0], 32, 64);
            dmean = sm * invK;
            rstd = 1.0f / sqrtf(fmaxf(sq * invK - dmean * dmean, 0.f) + g.ln_eps);
        } else {
            gemm3_mainloop<BM, BN, WM, WN, 0, true, false>(g, smem, mt, c, b, acc1, ls, lq, lsh);
        }
        const int hid0 = c * BN;
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int kl = 2 * j + s2;                                   // k-tile of the chunk
                const int kg = c * KC + kl;                                  // k-tile of GEMM 2: W_o stage kg & 1, G stage kg % 3
                // requests of this step, W_o first: the wait at its end leaves the two newest instructions - the G tile - in flight
                if (kl + 1 < KC && (kg + 1) * 16 < g.N) {
                    issue_w2(kg + 1);
                    issue_g(kg + 2);
                }
                u32x4 hf[2];
                {
                    const char* gs = Gs + (kg % 3) * G_IMG;
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(gs + offG0), o1 = *reinterpret_cast<const f32x4*>(gs + offG1);
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int hc = hid0 + j * 32 + 8 * (2 * s2 + q) + 4 * h;    // 4 consecutive gate channels
                        const int hcl = min(hc, GT_NG - 4);
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(cst + hcl), bi = *reinterpret_cast<const f32x4*>(cst + GT_NG + hcl);
                        const f32x4 ov = q ? o1 : o0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = rstd * (acc1[0][j][8 * s2 + 4 * q + e] - dmean * cs[e]) + bi[e];
                            v[4 * q + e] = (hc + e < g.N) ? ov[e] * sigmoidf_(x) : 0.f;
                        }
                    }
                    // unlifted pieces of v 2^4 (split2b): |gate * o| < 4094, else NaN -> the range probe -> the exact kernels
                    unsigned q0[4], q1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split2b(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
                    hf[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
                    hf[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
                }
                const char* ws = W2s + (kg & 1) * B2_IMG + offB2;
                if (rows_live && kg * 16 < g.N) {
#pragma unroll
                    for (int t0 = 0; t0 < TN2; t0 += TG2) {
                        u32x4 wb[TG2][2];
#pragma unroll
                        for (int t = 0; t < TG2; ++t)
#pragma unroll
                            for (int p = 0; p < 2; ++p) wb[t][p] = *reinterpret_cast<const u32x4*>(ws + (t0 + t) * 1024 + p * (BN2 * 32));
                        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};      // smallest first: a1 p0, a0 p1, a0 p0
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int t = 0; t < TG2; ++t)
                                acc2[0][t0 + t] = mfma_split(hf[TA[term]], wb[t][TB[term]], acc2[0][t0 + t]);
                    }
                }
                if (kl + 1 < KC && (kg + 1) * 16 < g.N) wait_vm_and_barrier<2>();
                else wait_vm_and_barrier<0>();                               // last k-tile of the chunk: nothing was requested
            }
        }
    }
    {
        const float cs2 = __builtin_ldexpf(1.0f, -4 - g.b2_exp);
#pragma unroll
        for (int t = 0; t < TN2; ++t) acc2[0][t] *= cs2;
    }
    AbxGemm g2 = g;
    g2.N = g.N2; g2.bias = g.bias2; g2.ln_csum = nullptr; g2.ln_stats = nullptr; g2.act = 0; g2.alpha = 1.0f; g2.gate = nullptr;
    __syncthreads();
    gemm_epilogue<BM, BN2, WM, BN2, EDGE, false>(g2, smem, smem + 2 * BM, acc2, mt * BM, 0, b, false);
}

constexpr int GT_LDS = 2 * 128 * 64 + 2 * 2 * 96 * 32 + 2 * 2 * 192 * 32 + 3 * 128 * 64 + 2 * GT_NG * 4;      // 79 360 bytes (two blocks per CU)

// the round-5 form (two walks over the z rows, one per gate chunk of 96): AbxGemm.tune bit 6, kept for A / B runs and as a cross-check
// (test_gemm_gated_tail_walk_variants_bit_identical)
__global__ __launch_bounds__(256, 2) void gemm3_gtail2w_kernel(const AbxGemm g) {
    extern __shared__ __attribute__((aligned(16))) float gt_smem[];
    const int ntm = (g.M + 127) / 128;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const ClockProbe probe(g.clock_probe);
    if ((mt + 1) * 128 <= g.M && g.N2 == 192) gemm3_gtail2w_block<false>(g, gt_smem, mt, b);
    else gemm3_gtail2w_block<true>(g, gt_smem, mt, b);
    probe.finish();
}

// Round 6: ONE walk over the z rows.  The board's power limit holds the clock of every large kernel (profiles/r06a_power_limit.txt), so
// what a launch costs is its energy, and the two-walk form paid the whole A side of GEMM 1 twice: 2 x 98 KB of z rows per block from the
// fabric (the second walk misses the 4 MB L2: 64 resident blocks x 98 KB per XCD), 2 x 12 k-steps of fragment reads, LayerNorm partial
// sums and f16 splits.  Here:
//   GEMM 1   all 192 gate channels in one main loop (128 x 192 tile, swapped operands: gate^T[channel][row], lane = row), 96 accumulator
//            registers - GEMM 2's accumulators do not exist yet;
//   GEMM 2   in two PASSES over the output columns (0 .. 95, then 96 .. 191), 48 accumulator registers each.  Pass A turns the gate
//            accumulators into the operand pieces of sigmoid(gate) * o k-tile by k-tile (the G stream, as before) and KEEPS them (the 96
//            registers the gate accumulators leave); pass B replays them against the other half of W_o.  Each pass streams its half of
//            the W_o planes [2][96][16] per k-tile (the same bytes as one pass over 192 columns), and ends with its own epilogue.
// Every accumulator sees the products of the two-walk form in the same order: bit-identical results.
constexpr int GT1_G1 = 2 * 128 * 64 + 2 * 2 * 192 * 32;                      // stages of GEMM 1: A fp32 + the gate-weight planes, 40 960
constexpr int GT1_W2 = 2 * 96 * 32;                                          // one k-tile of a W_o half: [2][96][16] f16, 6 144
constexpr int GT1_OFF_G = GT1_G1, GT1_OFF_W2 = GT1_OFF_G + 3 * 128 * 64, GT1_OFF_CST = GT1_OFF_W2 + 2 * GT1_W2;
constexpr int GT1_LDS = GT1_OFF_CST + 2 * GT_NG * 4;                         // 79 360 bytes (two blocks per CU)
template <bool EDGE>
__device__ __forceinline__ void gemm3_gtail_block(const AbxGemm& g, float* smem, int mt, int b) {
    constexpr int BM = 128, BN = 192, WM = 32, BH = 96, TH = BH / 32, NKT = BN / 16;
    char* lds = reinterpret_cast<char*>(smem);
    char* Gs = lds + GT1_OFF_G;                                              // 3 stages of G k-tiles (HBM: requested two steps ahead)
    char* W2s = lds + GT1_OFF_W2;                                            // 2 stages; beyond the epilogue's scratch (52 KB from the base)
    float* cst = reinterpret_cast<float*>(lds + GT1_OFF_CST);
    for (int i = threadIdx.x; i < GT_NG; i += 256) {
        cst[i] = i < g.N ? g.ln_csum[i] : 0.f;
        cst[GT_NG + i] = (i < g.N && g.bias) ? g.bias[i] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5;
    const int m0 = mt * BM;
    // W_o half-tiles: 6 chunks of 1 KB per k-tile - waves 0 .. 2 fetch two each, wave 3 none (its counted waits below only ever leave the
    // G pair in flight, which is the newest pair of every wave)
    unsigned offs2[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) plane_sources<BH, 2>(g.sB23p, g.sB23n, half * BH, g.N2, offs2[half]);
    const char* base2 = reinterpret_cast<const char*>(g.B2_split);
    const long long step2 = g.sB23k * 2;
    auto issue_w2 = [&](int half, int ktile) {                               // k-tile kt of the W_o half -> stage kt & 1
        if (wave == 3) return;
        char* dst = W2s + (ktile & 1) * GT1_W2 + wave * 2048;
        const char* src = base2 + ktile * step2;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(src + vgpr32(offs2[half][i]), dst + i * 1024);
    };
    unsigned offsG[2];
    const char* baseG = reinterpret_cast<const char*>(g.gate + (long long)b * g.sGb + (long long)m0 * g.sGm);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2), p = lane & 3;
        const int kq = p ^ ((r >> 2) & 3);
        const int gri = min(m0 + r, g.M - 1) - m0;
        offsG[i] = (unsigned)(((long long)gri * g.sGm + kq * 4) * 4);
    }
    const int nkt2 = g.N / 16;                                               // k-tiles of GEMM 2 (N % 16 == 0; <= NKT)
    auto issue_g = [&](int ktile) {                                          // k-tile kt of G -> stage kt % 3; beyond the end: the last one again
        const int kt = min(ktile, nkt2 - 1);
        char* dg = Gs + (ktile % 3) * (BM * 64) + wave * 2048;
        const char* sg = baseG + kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(sg + vgpr32(offsG[i]), dg + i * 1024);
    };
    const int offB2 = plane_off<BH>(0, lane & 31, h);
    const int grow = wave * WM + (lane & 31), gx = (grow >> 2) & 3;
    const int offG0 = grow * 64 + ((h ^ gx) << 4), offG1 = grow * 64 + (((2 + h) ^ gx) << 4);
    const bool rows_live = m0 + wave * WM < g.M;

    // ---- GEMM 1: gate^T for all gate channels; the first W_o half-tile and the first two G k-tiles land under it
    issue_w2(0, 0);
    issue_g(0);
    issue_g(1);
    float ls[1], lq[1], lsh[1] = {0.f};
    f32x16 acc1[1][BN / 32];
    gemm3_mainloop<BM, BN, WM, BN, 0, true, true>(g, smem, mt, 0, b, acc1, ls, lq, lsh);
    const float invK = 1.0f / (float)g.K;
    const float sm = ls[0] + __shfl_xor(ls[0], 32, 64), sq = lq[0] + __shfl_xor(lq[0], 32, 64);
    const float dmean = sm * invK;
    const float rstd = 1.0f / sqrtf(fmaxf(sq * invK - dmean * dmean, 0.f) + g.ln_eps);

    AbxGemm g2 = g;
    g2.N = g.N2; g2.bias = g.bias2; g2.ln_csum = nullptr; g2.ln_stats = nullptr; g2.act = 0; g2.alpha = 1.0f; g2.gate = nullptr;
    const float cs2 = __builtin_ldexpf(1.0f, -4 - g.b2_exp);
    u32x4 hfa[NKT][2];                                                       // the pieces of sigmoid(gate) * o, k-tile by k-tile (pass A -> pass B)
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};                      // smallest first: a1 p0, a0 p1, a0 p0
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 acc2[1][TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][t][r] = 0.f;
        if (half == 1) issue_w2(1, 0);                                       // (requested after the barrier that ended pass A's last step)
#pragma unroll
        for (int kg = 0; kg < NKT; ++kg) {
            const bool more = (kg + 1) * 16 < g.N && kg + 1 < NKT;
            if (half == 1 && kg == 0) wait_vm_and_barrier<0>();               // pass B's first half-tile has landed
            // requests of this step, W_o first: the wait at its end leaves the two newest instructions - the G tile - in flight
            if (more) {
                issue_w2(half, kg + 1);
                if (half == 0) issue_g(kg + 2);
            }
            if (half == 0) {
                const int j = kg >> 1, s2 = kg & 1;
                const char* gs = Gs + (kg % 3) * (BM * 64);
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(gs + offG0), o1 = *reinterpret_cast<const f32x4*>(gs + offG1);
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int hc = j * 32 + 8 * (2 * s2 + q) + 4 * h;        // 4 consecutive gate channels
                    const int hcl = min(hc, GT_NG - 4);
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(cst + hcl), bi = *reinterpret_cast<const f32x4*>(cst + GT_NG + hcl);
                    const f32x4 ov = q ? o1 : o0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = rstd * (acc1[0][j][8 * s2 + 4 * q + e] - dmean * cs[e]) + bi[e];
                        v[4 * q + e] = (hc + e < g.N) ? ov[e] * sigmoidf_(x) : 0.f;
                    }
                }
                // unlifted pieces of v 2^4 (split2b): |gate * o| < 4094, else NaN -> the range probe -> the exact kernels
                unsigned q0[4], q1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2b(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
                hfa[kg][0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
                hfa[kg][1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
            }
            const char* ws = W2s + (kg & 1) * GT1_W2 + offB2;
            if (rows_live && kg * 16 < g.N) {
                u32x4 wb[TH][2];
#pragma unroll
                for (int t = 0; t < TH; ++t)
#pragma unroll
                    for (int p = 0; p < 2; ++p) wb[t][p] = *reinterpret_cast<const u32x4*>(ws + t * 1024 + p * (BH * 32));
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int t = 0; t < TH; ++t) acc2[0][t] = mfma_split(hfa[kg][TA[term]], wb[t][TB[term]], acc2[0][t]);
            }
            if (more && half == 0) wait_vm_and_barrier<2>();
            else wait_vm_and_barrier<0>();                                   // (pass B, or the last k-tile: nothing but W_o was requested)
        }
#pragma unroll
        for (int t = 0; t < TH; ++t) acc2[0][t] *= cs2;
        // (every DMA has landed and every wave is past its last stage read: the scratch of the epilogue overlays the GEMM 1 stages and
        // the head of the G ring, not the W_o stages - pass B's first half-tile is requested above, after this epilogue)
        gemm_epilogue<BM, BH, WM, BH, EDGE, false>(g2, smem, smem + 2 * BM, acc2, m0, half * BH, b, false);
        if (half == 0) __syncthreads();
    }
}

__global__ __launch_bounds__(256, 2) void gemm3_gtail_kernel(const AbxGemm g) {
    extern __shared__ __attribute__((aligned(16))) float gt_smem[];
    const int ntm = (g.M + 127) / 128;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / (unsigned)ntm), mt = (int)(wgid - (unsigned)b * (unsigned)ntm);
    const ClockProbe probe(g.clock_probe);
    if ((mt + 1) * 128 <= g.M && g.N2 == 192 && g.N == 192) gemm3_gtail_block<false>(g, gt_smem, mt, b);
    else gemm3_gtail_block<true>(g, gt_smem, mt, b);
    probe.finish();
}

// ---- IPA layer tail (score_network.py:126-163 per layer: the single representation after the attention) in ONE kernel:
//     s <- LN1(s + feat W_final + b)                                 attention_module.final_proj + attention_layer_norm
//     s <- LN2(s + relu(relu(s W0 + b0) W2 + b2) W4 + b4)            transition_module.0 / .2 / .4 + transition_layer_norm
// A block owns 32 rows and all 256 channels: 4 waves x (32 rows x 64 columns), the main loop with swapped MFMA operands (lane = row, 16
// columns per 32 x 32 tile in registers), so bias / ReLU / residual are lane-local and the LayerNorm statistics need one 32-lane
// exchange + a 4-wave fold through LDS.  The activations between the GEMMs never leave the CU: each epilogue writes the block's
// 32 x 256 tile to one of two LDS buffers (1040-byte rows: conflict-free 16-byte fragment reads), which IS the A operand of the next
// main loop (AMODE 3); only the weight planes stream (DMA, 24 KB per k-tile).  Six launches per layer become one; the arithmetic of a
// row does not depend on the batch (one instantiation for every size).
constexpr int IT_C = 256;
constexpr int IT_ASTR = IT_C * 4 + 16;                                      // bytes per activation row in LDS
constexpr int IT_RING = 4;
constexpr int IT_OPER = IT_RING * (4096 + 2 * IT_C * 32);                   // main-loop stages: A 3 x 4 KB (padded) + weights 3 x 16 KB
constexpr int IT_LDS = IT_OPER + 2 * 32 * IT_ASTR + 2 * 4 * 32 * 4;

__global__ __launch_bounds__(256, 1) void ipa_tail_kernel(const AbxIpaTail a) {
    constexpr int BM = 32, BN = IT_C, WM = 32, WN = 64, TN = 2;
    extern __shared__ __attribute__((aligned(16))) float it_smem[];
    char* lds = reinterpret_cast<char*>(it_smem);
    char* act[2] = {lds + IT_OPER, lds + IT_OPER + 32 * IT_ASTR};
    float* red = reinterpret_cast<float*>(lds + IT_OPER + 2 * 32 * IT_ASTR);    // [2][4][32]
    const int mt = blockIdx.x, m0 = mt * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, row = lane & 31;
    const int grc = min(m0 + row, a.M - 1);
    // column of accumulator register r of tile j (swapped operands): wave * 64 + j * 32 + 8 (r >> 2) + 4 h + (r & 3)
    const int colb = wave * WN + 4 * h;

    AbxGemm g = {};
    g.M = a.M; g.N = IT_C; g.batch = 1; g.b_f16 = 1;
    g.sB3k = 2 * IT_C * 16; g.sB3p = IT_C * 16; g.sB3n = 16;
    f32x16 acc[1][TN];
    float ls[1], lq[1], lsh[1];
    float x[TN][16];

    // rows of 4 consecutive channels <-> registers 4 q .. 4 q + 3 of tile j
    auto for_groups = [&](auto&& fn) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) fn(j, q, colb + j * 32 + 8 * q);
    };
    // LayerNorm over the 256 channels of this lane's row (values in x): two passes like torch, gamma / beta applied in place
    auto layer_norm = [&](const float* gamma, const float* beta) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += x[j][r];
        sm += __shfl_xor(sm, 32, 64);
        if (h == 0) red[wave * 32 + row] = sm;
        __syncthreads();
        const float mean = (red[row] + red[32 + row] + red[64 + row] + red[96 + row]) * (1.0f / IT_C);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[j][r] -= mean; sq = fmaf(x[j][r], x[j][r], sq); }
        sq += __shfl_xor(sq, 32, 64);
        if (h == 0) red[128 + wave * 32 + row] = sq;
        __syncthreads();
        const float var = (red[128 + row] + red[160 + row] + red[192 + row] + red[224 + row]) * (1.0f / IT_C);
        const float rstd = 1.0f / sqrtf(var + a.ln_eps);
        for_groups([&](int j, int q, int c) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[j][4 * q + e] = x[j][4 * q + e] * rstd * gm[e] + bt[e];
        });
    };
    auto to_lds = [&](char* buf) {
        for_groups([&](int j, int q, int c) {
            *reinterpret_cast<f32x4*>(buf + row * IT_ASTR + c * 4) = (f32x4){x[j][4 * q], x[j][4 * q + 1], x[j][4 * q + 2], x[j][4 * q + 3]};
        });
    };
    auto bias_act = [&](const float* bias, bool relu) {
        for_groups([&](int j, int q, int c) {
            const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[0][j][4 * q + e] + bi[e];
                x[j][4 * q + e] = relu ? relu_keep_nan(v) : v;
            }
        });
    };

    // ---- 1: s + feat W_final + b -> LN1 -> act[0] (and kept in registers: the residual of the transition)
    if (a.partial) {
        // feat W_final comes as n_partial K-slice products of a split-K GEMM launched before (the 132-k-step chain of this block becomes
        // 12 k-steps of 11 x as many blocks): summed here in slice order, then the bias
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        for (int sl = 0; sl < a.n_partial; ++sl) {
            const float* pp = a.partial + (long long)sl * a.s_partial + (long long)grc * IT_C;
            for_groups([&](int j, int q, int c) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][j][4 * q + e] += pv[e];
            });
        }
    } else {
        g.A = a.feat; g.sAm = a.s_feat; g.sAk = 1; g.K = a.K1;
        g.B_split = a.W_final; g.b_exp = a.e_final;
        gemm3_mainloop<BM, BN, WM, WN, 0, true, false, IT_RING>(g, it_smem, mt, 0, 0, acc, ls, lq, lsh);
    }
    bias_act(a.b_final, false);
    for_groups([&](int j, int q, int c) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(a.s + (long long)grc * a.s_s + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[j][4 * q + e] += rv[e];
    });
    layer_norm(a.ln1_w, a.ln1_b);
    float res[TN][16];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) res[j][r] = x[j][r];
    to_lds(act[0]);
    __syncthreads();
    // ---- 2, 3: relu(. W0 + b0) -> act[1], relu(. W2 + b2) -> act[0]
    g.K = IT_C;
    g.B_split = a.W_t0; g.b_exp = a.e_t0;
    gemm3_mainloop<BM, BN, WM, WN, 3, true, false, IT_RING>(g, it_smem, mt, 0, 0, acc, ls, lq, lsh, act[0], IT_ASTR);
    bias_act(a.b_t0, true);
    to_lds(act[1]);
    __syncthreads();
    g.B_split = a.W_t2; g.b_exp = a.e_t2;
    gemm3_mainloop<BM, BN, WM, WN, 3, true, false, IT_RING>(g, it_smem, mt, 0, 0, acc, ls, lq, lsh, act[1], IT_ASTR);
    bias_act(a.b_t2, true);
    to_lds(act[0]);
    __syncthreads();
    // ---- 4: residual + . W4 + b4 -> LN2 -> act[1] -> the rows of s, coalesced
    g.B_split = a.W_t4; g.b_exp = a.e_t4;
    gemm3_mainloop<BM, BN, WM, WN, 3, true, false, IT_RING>(g, it_smem, mt, 0, 0, acc, ls, lq, lsh, act[0], IT_ASTR);
    bias_act(a.b_t4, false);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[j][r] += res[j][r];
    layer_norm(a.ln2_w, a.ln2_b);
    if (a.range_flag) {                                         // range safety (AbxGemm.range_flag): a row that left the split range is NaN by now
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) z = fmaf(x[j][r], 0.f, z);
        if (__any(z != z) && lane == 0) atomicOr(a.range_flag, a.range_tag);
    }
    to_lds(act[1]);
    __syncthreads();
    for (int idx = threadIdx.x; idx < BM * (IT_C / 4); idx += 256) {
        const int r = idx / (IT_C / 4), c4 = idx % (IT_C / 4);
        if (m0 + r < a.M)
            *reinterpret_cast<f32x4*>(a.s + (long long)(m0 + r) * a.s_s + c4 * 4) = *reinterpret_cast<const f32x4*>(act[1] + r * IT_ASTR + c4 * 16);
    }
    // ---- 5 (optional): affine_update (256 -> 6, fp32 FMA chain in channel order) and the frame update of the block's residues
    if (a.W_aff) {
        const int r = threadIdx.x >> 3, o = threadIdx.x & 7;
        if (o < 6) {
            const float* y = reinterpret_cast<const float*>(act[1] + r * IT_ASTR);
            float u = 0.f;
#pragma unroll 8
            for (int c = 0; c < IT_C; ++c) u = fmaf(y[c], a.W_aff[c * 6 + o], u);
            red[r * 8 + o] = u + a.b_aff[o];
        }
        __syncthreads();
        if (threadIdx.x < BM && m0 + threadIdx.x < a.M)
            rigid_update_row(m0 + threadIdx.x, red + threadIdx.x * 8, a.fixed, a.init_q, a.init_t, a.cur_q, a.cur_t, a.cur_R, a.delta_q, a.pscale);
    }
}

// ---- Per-residue heads on the final single representation in ONE kernel: the torsion ResNet (sidechain.py:28-62), the SequenceHead MLP
// and - on the last pass - the PredictedLDDTHead MLP (head.py:143-226).  Thirteen (eighteen) launches of 352 ... 35 200-row GEMMs and
// LayerNorms otherwise, each a latency chain of its own at small batches.
//     t  = relu(s) W_act + b + relu(s0) W_init + b;   t += relu(relu(t) W_r0 + b) W_r1 + b;   t += relu(relu(t) W_r2 + b) W_r3 + b
//     un = relu(t) W_proj + b                                            (14 columns: the unnormalised torsion sin / cos)
//     logits = relu(relu(LN_s(s) W_s1 + b) W_s3 + b) W_s5 + b            (20 columns);  pLDDT logits likewise (50 columns)
// A block owns 32 rows; 4 waves x (32 rows x 32 of the 128 hidden columns), swapped MFMA operands (lane = row), activations in LDS
// between the GEMMs (AMODE 3 main loops as in ipa_tail_kernel), only the weight planes stream.  The 14 / 20 / 50-column projections
// run as 128-column GEMMs on zero-padded planes (negligible work; one code path).  Split-f16 arithmetic throughout.
constexpr int HT_C = 256, HT_H = 128;
constexpr int HT_ASTR = HT_C * 4 + 16, HT_HSTR = HT_H * 4 + 16;             // bytes per activation row in LDS
constexpr int HT_RING = 4;
constexpr int HT_OPER = HT_RING * (2 * HT_H * 32);                           // 4 weight stages of 8 KB
constexpr int HT_LDS = HT_OPER + 2 * 32 * HT_ASTR + 2 * 32 * HT_HSTR;

__global__ __launch_bounds__(256, 1) void heads_tail_kernel(const AbxHeadsTail a) {
    constexpr int BM = 32, BN = HT_H, WM = 32, WN = 32;
    extern __shared__ __attribute__((aligned(16))) float ht_smem[];
    char* lds = reinterpret_cast<char*>(ht_smem);
    char* S = lds + HT_OPER;
    char* S0 = S + 32 * HT_ASTR;                                            // s0, later LN(s)
    char* T1 = S0 + 32 * HT_ASTR;
    char* T2 = T1 + 32 * HT_HSTR;
    const int mt = blockIdx.x, m0 = mt * BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, row = lane & 31;
    const int colb = wave * WN + 4 * h;               // column of accumulator register r (swapped operands): colb + 8 (r >> 2) + (r & 3)
    const bool row_ok = m0 + row < a.M;

    for (int idx = threadIdx.x; idx < 32 * (HT_C / 4); idx += 256) {
        const int r = idx / (HT_C / 4), c4 = idx % (HT_C / 4);
        const long long gr = min(m0 + r, a.M - 1);
        *reinterpret_cast<f32x4*>(S + r * HT_ASTR + c4 * 16) = *reinterpret_cast<const f32x4*>(a.s + gr * a.s_s + c4 * 4);
        *reinterpret_cast<f32x4*>(S0 + r * HT_ASTR + c4 * 16) = *reinterpret_cast<const f32x4*>(a.s0 + gr * a.s_s0 + c4 * 4);
    }
    __syncthreads();

    AbxGemm g = {};
    g.M = a.M; g.N = HT_H; g.batch = 1; g.b_f16 = 1;
    g.sB3k = 2 * HT_H * 16; g.sB3p = HT_H * 16; g.sB3n = 16;
    f32x16 acc[1][1];
    float ls[1], lq[1], lsh[1];
    float x[16], y[16];
    bool bad = false;

    auto run = [&](const unsigned short* W, int e, int K, const char* abuf, int astr, int a_relu) {
        g.K = K; g.B_split = W; g.b_exp = e; g.a_relu = a_relu;
        gemm3_mainloop<BM, BN, WM, WN, 3, true, false, HT_RING>(g, ht_smem, mt, 0, 0, acc, ls, lq, lsh, abuf, astr);
    };
    // v = acc + bias (relu: after the Linear); into `dst` (add: on top of what it holds)
    auto bias_to = [&](const float* bias, bool relu, bool add, float (&dst)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + colb + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[0][0][4 * q + e] + bi[e];
                if (relu) v = relu_keep_nan(v);
                dst[4 * q + e] = add ? dst[4 * q + e] + v : v;
            }
        }
    };
    auto to_lds = [&](char* buf, int stride, const float (&src)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(buf + row * stride + (colb + 8 * q) * 4) = (f32x4){src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]};
    };
    auto store_cols = [&](float* out, int ncol, const float (&src)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = colb + 8 * q + e;
                if (c < ncol) {
                    bad |= !(fabsf(src[4 * q + e]) <= 3.0e38f);
                    if (row_ok) out[(long long)(m0 + row) * ncol + c] = src[4 * q + e];
                }
            }
    };
    // LayerNorm of the block's rows of s -> S0 (two passes like torch): 8 threads per row, 32 channels each
    auto layer_norm_s = [&](const float* gamma, const float* beta) {
        const int r = threadIdx.x >> 3, p = threadIdx.x & 7;
        const float* src = reinterpret_cast<const float*>(S + r * HT_ASTR) + p * 32;
        float* dst = reinterpret_cast<float*>(S0 + r * HT_ASTR) + p * 32;
        float v[32], sm = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) { v[c] = src[c]; sm += v[c]; }
        sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
        const float mean = sm * (1.0f / HT_C);
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) { v[c] -= mean; sq = fmaf(v[c], v[c], sq); }
        sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / HT_C) + a.ln_eps);
#pragma unroll
        for (int c = 0; c < 32; ++c) dst[c] = v[c] * rstd * gamma[p * 32 + c] + beta[p * 32 + c];
    };
    // LayerNorm -> Linear -> ReLU -> Linear -> ReLU -> Linear of one head (S0 holds LN(s) on entry)
    auto head = [&](const unsigned short* W1, int e1, const float* b1, const unsigned short* W3, int e3, const float* b3,
                    const unsigned short* W5, int e5, const float* b5, float* out, int ncol) {
        run(W1, e1, HT_C, S0, HT_ASTR, 0);
        bias_to(b1, true, false, y);
        to_lds(T1, HT_HSTR, y);
        __syncthreads();
        run(W3, e3, HT_H, T1, HT_HSTR, 0);
        bias_to(b3, true, false, y);
        to_lds(T2, HT_HSTR, y);
        __syncthreads();
        run(W5, e5, HT_H, T2, HT_HSTR, 0);
        bias_to(b5, false, false, y);
        store_cols(out, ncol, y);
    };

    // ---- torsion module
    run(a.W_act, a.e_act, HT_C, S, HT_ASTR, 1);
    bias_to(a.b_act, false, false, x);
    run(a.W_init, a.e_init, HT_C, S0, HT_ASTR, 1);
    bias_to(a.b_init, false, true, x);
    to_lds(T1, HT_HSTR, x);
    __syncthreads();
    {
        const unsigned short* Wr[4] = {a.W_r0, a.W_r1, a.W_r2, a.W_r3};
        const int er[4] = {a.e_r0, a.e_r1, a.e_r2, a.e_r3};
        const float* br[4] = {a.b_r0, a.b_r1, a.b_r2, a.b_r3};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            run(Wr[2 * blk], er[2 * blk], HT_H, T1, HT_HSTR, 1);
            bias_to(br[2 * blk], false, false, y);
            to_lds(T2, HT_HSTR, y);
            __syncthreads();
            run(Wr[2 * blk + 1], er[2 * blk + 1], HT_H, T2, HT_HSTR, 1);
            bias_to(br[2 * blk + 1], false, true, x);
            to_lds(T1, HT_HSTR, x);
            __syncthreads();
        }
    }
    run(a.W_proj, a.e_proj, HT_H, T1, HT_HSTR, 1);
    bias_to(a.b_proj, false, false, y);
    store_cols(a.un, 14, y);
    // ---- sequence head, pLDDT head
    layer_norm_s(a.lns_w, a.lns_b);
    __syncthreads();
    head(a.W_s1, a.e_s1, a.b_s1, a.W_s3, a.e_s3, a.b_s3, a.W_s5, a.e_s5, a.b_s5, a.logits, 20);
    if (a.W_p1) {
        layer_norm_s(a.lnp_w, a.lnp_b);          // (every read of S0 by the sequence head ended with its first main loop's barrier)
        __syncthreads();
        head(a.W_p1, a.e_p1, a.b_p1, a.W_p3, a.e_p3, a.b_p3, a.W_p5, a.e_p5, a.b_p5, a.pl, 50);
    }
    if (a.range_flag && __any(bad) && lane == 0) atomicOr(a.range_flag, a.range_tag);
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch3(const AbxGemm& g, hipStream_t st) {
    const long long mt = ((long long)g.M + BM - 1) / BM, ntn = ((long long)g.N + BN - 1) / BN;
    dim3 grid((unsigned)(mt * ntn * g.batch), 1, 1), block(256);
#ifdef ABX_PERSIST
    if ((g.tune & 256) && grid.x > 256u * MINW) grid.x = 256u * MINW;
#endif
    const int amode = g.A_split ? 2 : (g.sAk == 1 ? 0 : 1);
    if (g.c_transposed) {
        if (amode != 0) { abx_set_error("abx_gemm: transposed store needs a k-contiguous fp32 A"); return ABX_ERR_ARG; }
        hipLaunchKernelGGL((gemm3_kernel<BM, BN, WM, WN, 0, true, MINW>), grid, block, 0, st, g);
    } else if (amode == 0) hipLaunchKernelGGL((gemm3_kernel<BM, BN, WM, WN, 0, false, MINW>), grid, block, 0, st, g);
    else if (amode == 1) hipLaunchKernelGGL((gemm3_kernel<BM, BN, WM, WN, 1, false, MINW>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm3_kernel<BM, BN, WM, WN, 2, false, MINW>), grid, block, 0, st, g);
    return abx_check_launch("abx_gemm");
}

// fp32 weights -> k-tiled float16 planes [Kp/16][2][N][16] of w' = w * scale: p0 = f16(w'), p1 = f16(w' - p0) (AbxGemm.b_f16)
__global__ __launch_bounds__(256) void split_weights_f16_kernel(const float* __restrict__ w, long long s_n, long long s_k, int N, int K,
                                                                int Kp, float scale, unsigned short* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * Kp) return;
    const int n = (int)(idx / Kp), k = (int)(idx % Kp);
    const float x = k < K ? w[n * s_n + k * s_k] * scale : 0.f;
    const _Float16 h0 = (_Float16)x;
    const _Float16 h1 = (_Float16)(x - (float)h0);
    const long long o = ((long long)(k >> 4) * 2 * N + n) * 16 + (k & 15);
    out[o] = __builtin_bit_cast(unsigned short, h0);
    out[o + (long long)N * 16] = __builtin_bit_cast(unsigned short, h1);
}

}  // namespace

// Called by abx_gemm (gemm.hip) once the descriptor has been validated and the vector flags filled.
// Returns 1 when the problem is not served by this path (the caller falls back to the exact kernel).
int abx_gemm3_dispatch(const AbxGemm& g, hipStream_t st, int* rc) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // N <= 32 with split weights: the skinny projections of the pair stack (4 / 12 / 32 bias channels from 192- / 128-wide rows):
    // HBM streams of the A operand, served by a 128 x 32 tile of the same DMA pipeline (7 blocks per CU keep the stream fed)
    const bool narrow = g.B_split && g.N <= 32 && !g.A_split && g.sAk == 1 && !g.glu && !g.C_split && !g.A2 && !g.out_ln_w &&
                        g.a_pair_transpose <= 0 && g.pair_Lp == 0;
    if (!g.B_split || g.K % 16 != 0 || (g.N <= 64 && !narrow)) return 1;
    if ((g.A_split != nullptr) == (g.b_f16 != 0) || g.b_exp < -100 || g.b_exp > 100 || g.b2_exp < -100 || g.b2_exp > 100) {
        abx_set_error("abx_gemm: B_split must be float16 weight planes (abx_split_weights_f16, b_f16 = 1, |b_exp| <= 100) with an fp32 A "
                      "and activation images (b_f16 = 0) with A_split");
        *rc = ABX_ERR_ARG;
        return 0;
    }
    const long long mt128 = ((long long)g.M + 127) / 128;
    // exact == 2: the caller fixed the arithmetic class of this op (results must not depend on how many samples share a launch)
    if (g.exact != 2 && mt128 * (((long long)g.N + 127) / 128) * g.batch < ABX_SPLIT_MIN_TILES) return 1;
    if (!al16(g.B_split) || g.sB3n % 8 != 0 || g.sB3p % 8 != 0 || g.sB3b % 8 != 0 || g.sB3k % 8 != 0) return 1;
    if (g.A_split) {
        if (!al16(g.A_split) || g.sA3m % 8 != 0 || g.sA3p % 8 != 0 || g.sA3b % 8 != 0 || g.sA3k % 8 != 0 || g.c_transposed) return 1;
        if (g.batch_inner > 0 && (g.sA3i % 8 != 0 || g.sB3i % 8 != 0)) return 1;
    } else if (g.sAk == 1) {
        if (!al16(g.A) || g.sAm % 4 != 0 || g.sAb % 4 != 0) return 1;
        if (g.a_pair_transpose > 0 && !g.a_pair && (long long)g.a_pair_transpose * g.a_pair_transpose != g.M) return 1;
    } else {
        if (!al16(g.A) || g.sAk % 4 != 0 || g.sAb % 4 != 0 || g.M % 4 != 0 || g.M < 4 || g.a_pair_transpose > 0) return 1;
    }
    // 128x128 tiles run 3 blocks per CU (48 KB LDS, ~150 VGPRs), 128x192 tiles 2: the narrow tile wins whenever it wastes no
    // columns (N = 768: q|k|v|gate, transition hidden); N = 192 / 448 take the wide tile
    // per-lane DMA offsets are 32-bit, relative to the tile's first row (k-contiguous A), to the batch base (pair-transposed
    // rows, plane operands) or to the tile's first column (row-contiguous A)
    if (!g.A_split && g.sAk == 1 && ((g.a_pair_transpose > 0 || g.a_pair) ? (long long)g.M * g.sAm : 128LL * g.sAm) >= (1LL << 30)) return 1;
    if (g.pair_Lp > 0 && g.c_pair && g.c_transposed) return 1;
    if (g.a_pair && (g.A_split || g.sAk != 1)) return 1;
    if (!g.A_split && g.sAk != 1 && 16LL * g.sAk + g.M >= (1LL << 30)) return 1;
    if (((long long)g.M + 127) / 128 * (((long long)g.N + 127) / 128) * g.batch >= (1LL << 31)) return 1;
    if (g.A_split && (long long)(g.K / 16) * g.sA3k >= (1LL << 31)) return 1;
    if ((long long)(g.K / 16) * g.sB3k >= (1LL << 31)) return 1;
    if (g.out_ln_w) {
        if (g.N > 128 || g.c_transposed || g.C_split || g.A2 || g.A_split || g.sAk != 1 || !g.out_ln_b) {
            abx_set_error("abx_gemm: out_ln needs N <= 128, a k-contiguous fp32 A and a plain store (split-f16 path)");
            *rc = ABX_ERR_ARG;
            return 0;
        }
        const long long mt = ((long long)g.M + 127) / 128;
        hipLaunchKernelGGL((gemm3_oln_kernel<128, 128, 32, 128, 4>), dim3((unsigned)(mt * g.batch)), dim3(256), 0, st, g);
        *rc = abx_check_launch("abx_gemm(out_ln)");
        return 0;
    }
    if (g.mlp == 2) {
        // gated attention tail: out = (sigmoid(LN(A) B + bias) * gate) B2 + bias2 (+ resid)
        if (g.A_split || g.sAk != 1 || g.c_transposed || g.C_split || g.glu || g.A2 || g.out_ln_w || !g.B2_split || g.N2 <= 0 || g.N2 > 192 ||
            g.N % 16 != 0 || g.N > GT_NG || g.act != 2 || g.alpha != 1.0f || g.rowscale || !g.gate || !g.ln_csum || g.ln_stats || g.a_relu ||
            g.a_pair_transpose > 0 || g.pair_Lp > 0 || !al16(g.B2_split) || g.sB23n % 8 != 0 || g.sB23p % 8 != 0 || g.sB23k % 8 != 0 ||
            (long long)(g.N / 16) * g.sB23k >= (1LL << 31) || !al16(g.ln_csum) || (g.bias && !al16(g.bias)) || !al16(g.gate) || g.sGm % 4 != 0 ||
            g.sGb % 4 != 0 || 128LL * g.sGm >= (1LL << 30)) {
            abx_set_error("abx_gemm: mlp = 2 (gated tail) needs a k-contiguous fp32 A with folded LayerNorm, act = 2, a 16-byte aligned gate "
                          "operand [M][N], N % 16 == 0, N2 <= 192, plain store, B2_split planes in the permuted k order");
            *rc = ABX_ERR_ARG;
            return 0;
        }
        const long long mt = ((long long)g.M + 127) / 128;
        if (g.tune & 64) {                                                   // tune bit 6: the two-walk form of round 5 (A / B runs, cross-check)
            if (int e = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm3_gtail2w_kernel), GT_LDS, "abx_gemm(gated tail)")) { *rc = e; return 0; }
            hipLaunchKernelGGL(gemm3_gtail2w_kernel, dim3((unsigned)(mt * g.batch)), dim3(256), GT_LDS, st, g);
            *rc = abx_check_launch("abx_gemm(gated tail, two walks)");
            return 0;
        }
        if (int e = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm3_gtail_kernel), GT1_LDS, "abx_gemm(gated tail)")) { *rc = e; return 0; }
        hipLaunchKernelGGL(gemm3_gtail_kernel, dim3((unsigned)(mt * g.batch)), dim3(256), GT1_LDS, st, g);
        *rc = abx_check_launch("abx_gemm(gated tail)");
        return 0;
    }
    if (g.mlp) {
        // fused two-layer transition: hidden = relu(LN(A) B + bias) never stored, out = hidden B2 + bias2 (+ resid)
        if (g.A_split || g.sAk != 1 || g.c_transposed || g.C_split || g.glu || g.A2 || g.out_ln_w || !g.B2_split || g.N2 <= 0 || g.N2 > 192 ||
            g.N % 16 != 0 || g.N > MLP_NH || g.act != 1 || g.alpha != 1.0f || g.rowscale || g.gate || !g.ln_csum || g.ln_stats || g.a_relu ||
            g.a_pair_transpose > 0 || g.pair_Lp > 0 || !al16(g.B2_split) || g.sB23n % 8 != 0 || g.sB23p % 8 != 0 || g.sB23k % 8 != 0 ||
            (long long)(g.N / 16) * g.sB23k >= (1LL << 31) || !al16(g.ln_csum) || (g.bias && !al16(g.bias))) {
            abx_set_error("abx_gemm: mlp needs a k-contiguous fp32 A with folded LayerNorm, relu, N % 16 == 0, N <= 768, N2 <= 192, plain store, "
                          "B2_split planes in the permuted k order");
            *rc = ABX_ERR_ARG;
            return 0;
        }
        const long long mt = ((long long)g.M + 127) / 128;
        if ((g.tune >> 5) & 1) {                                                                                                   // benchmarking
            if (int e = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm3_mlp3_kernel), MLP3_LDS, "abx_gemm(mlp)")) { *rc = e; return 0; }
            hipLaunchKernelGGL(gemm3_mlp3_kernel, dim3((unsigned)(mt * g.batch)), dim3(256), MLP3_LDS, st, g);
        } else if ((g.tune >> 4) & 1) hipLaunchKernelGGL(gemm3_mlp_kernel<1>, dim3((unsigned)(mt * g.batch)), dim3(256), 0, st, g);       // benchmarking
        else hipLaunchKernelGGL(gemm3_mlp_kernel<2>, dim3((unsigned)(mt * g.batch)), dim3(256), 0, st, g);
        *rc = abx_check_launch("abx_gemm(mlp)");
        return 0;
    }
    if (g.A2) {
        // dual GEMM: 128 x 96 tiles (two accumulator sets of 48 registers)
        if (g.A_split || g.c_transposed || g.glu || !g.B2_split || g.K2 % 16 != 0 || !al16(g.A2) || g.sA2m % 4 != 0 || g.sA2b % 4 != 0 ||
            !al16(g.B2_split) || g.sB23n % 8 != 0 || g.sB23p % 8 != 0 || g.sB23k % 8 != 0 || !g.ln2_csum ||
            (g.pair_Lp > 0 ? (long long)g.pair_L * g.pair_L * g.sA2m : 128LL * g.sA2m) >= (1LL << 30) ||
            (long long)(g.K2 / 16) * g.sB23k >= (1LL << 31)) {
            abx_set_error("abx_gemm: dual (A2 / B2_split) operands do not qualify for the split-f16 dual kernel");
            *rc = ABX_ERR_ARG;
            return 0;
        }
        // (128 x 192 tiles - A read and split once per row - need 2 x 96 accumulator registers and spill at 256 VGPRs)
        const long long mt = ((long long)g.M + 127) / 128, ntn = ((long long)g.N + 95) / 96;
        // Three blocks per CU (142 VGPRs).  tune bit 10: the four-block build (one B sub-tile in flight, 32-column store groups, 128 VGPRs):
        // 5 % faster (9.75 vs 10.28 ms at 100 samples), but the accumulator set of the first main loop is parked in scratch across the
        // second and that is + 9 GB of HBM traffic per launch on the counters (43.9 vs 34.8 GB) - measured in profiles/r04l, not the default
        // tune bit 7 (round 6 probe, 96 < N <= 192): one block per row tile, z walked once (gemm3_dual1_kernel) - bit-identical, but its 96 kept
        // gate registers + 48 accumulators leave two blocks per CU instead of three (256 VGPRs, 7 spilled): 10.97 vs 10.30 ms at 100 samples
        // (profiles/r06c_kb_dual.txt); NOT the default
        if (g.N > 96 && g.N <= 192 && (g.tune & 128)) hipLaunchKernelGGL(gemm3_dual1_kernel, dim3((unsigned)(mt * g.batch)), dim3(256), 0, st, g);
        else if (g.tune & 1024) hipLaunchKernelGGL((gemm3_dual_kernel<128, 96, 32, 96, 4>), dim3((unsigned)(mt * ntn * g.batch)), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm3_dual_kernel<128, 96, 32, 96, 3>), dim3((unsigned)(mt * ntn * g.batch)), dim3(256), 0, st, g);
        *rc = abx_check_launch("abx_gemm(dual)");
        return 0;
    }
    if (narrow) {
        const long long mt = ((long long)g.M + 127) / 128;
        dim3 grid((unsigned)(mt * g.batch), 1, 1), block(256);
        if ((g.tune >> 9) & 3) {                                     // probe: deeper ring (1: 3 stages at 4 blocks per CU, 2: 4 stages at 3)
            const int v = (g.tune >> 9) & 3;
            if (g.c_transposed) { if (v == 1) hipLaunchKernelGGL((gemm3_narrow_ring_kernel<true, 3, 4>), grid, block, 0, st, g); else hipLaunchKernelGGL((gemm3_narrow_ring_kernel<true, 4, 3>), grid, block, 0, st, g); }
            else { if (v == 1) hipLaunchKernelGGL((gemm3_narrow_ring_kernel<false, 3, 4>), grid, block, 0, st, g); else hipLaunchKernelGGL((gemm3_narrow_ring_kernel<false, 4, 3>), grid, block, 0, st, g); }
            *rc = abx_check_launch("abx_gemm(narrow, ring)");
            return 0;
        }
        if (g.c_transposed) hipLaunchKernelGGL((gemm3_kernel<128, 32, 32, 32, 0, true, 6>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((gemm3_kernel<128, 32, 32, 32, 0, false, 6>), grid, block, 0, st, g);
        *rc = abx_check_launch("abx_gemm(narrow)");
        return 0;
    }
    const long long pad128 = ((g.N + 127) / 128) * 128, pad192 = ((g.N + 191) / 192) * 192;
    const int force = (g.tune >> 1) & 7;                       // 1: 128x128, 2: 128x192 (benchmarking)
    // (the plane x plane contraction at L = 352 pads to 384 either way: the wide tile measured 10 % faster)
    const bool wide = g.glu ? false : force ? force == 2 : (g.A_split ? pad192 <= pad128 : (pad192 <= pad128 && g.N % 128 != 0));
    // waves are stacked along M (4 x 1): every wave owns 32 rows and the full tile width, so each A row is read from LDS and
    // split into f16 pieces by exactly one wave (the VALU issue slots next to the MFMAs are the scarce resource)
    // Small problems (the per-residue GEMMs of the IPA loop / sequence track at a dozen samples per GPU: 4 224 rows x 256 columns = 66
    // tiles of 128 x 128 on 256 CUs): 64 x 128 tiles, 2 x 2 waves, twice the workgroups.  Every output element still accumulates its k
    // in the same order (same MFMA, same term order, same row statistics), so the tile choice does not change a single bit and
    // results stay independent of how many samples share a launch.
    const long long ntn128 = ((long long)g.N + 127) / 128;
    if (!wide && !g.glu && !force && mt128 * ntn128 * g.batch < 384 && g.M > 64) *rc = launch3<64, 128, 32, 64, 4>(g, st);
    else if (wide) *rc = launch3<128, 192, 32, 192, 3>(g, st);
    else *rc = launch3<128, 128, 32, 128, 4>(g, st);
    return 0;
}

// abx_gemm_side (gemm.hip): both descriptors validated and their vector flags filled.  Returns 1 when the pair is not served by
// the side kernel (the caller then issues two launches).
int abx_gemm3_side_dispatch(const AbxGemm& g, const AbxGemm& s2, hipStream_t st, int* rc) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    auto plain_a = [&](const AbxGemm& d) {
        return d.B_split && d.b_f16 && !d.A_split && d.A && d.sAk == 1 && d.K % 16 == 0 && al16(d.A) && d.sAm % 4 == 0 && d.sAb % 4 == 0 &&
               al16(d.B_split) && d.sB3n % 8 == 0 && d.sB3p % 8 == 0 && d.sB3k % 8 == 0 && d.sB3b % 8 == 0 && d.b_exp >= -100 && d.b_exp <= 100 &&
               !d.glu && !d.C_split && !d.A2 && !d.out_ln_w && !d.mlp && d.a_pair_transpose <= 0 && d.pair_Lp == 0 && d.batch_inner == 0 &&
               128LL * d.sAm < (1LL << 30) && (long long)(d.K / 16) * d.sB3k < (1LL << 31) && d.exact != 1;
    };
    if (!plain_a(g) || !plain_a(s2)) return 1;
    if (g.c_transposed || g.N % 128 != 0 || g.N < 128) return 1;                      // main: the 128 x 128 plain-store kernel
    if (!s2.c_transposed || s2.N > 32) return 1;                                      // side: the 128 x 32 transposed-store tile
    const long long tq = ((long long)g.M + 127) / 128 * (g.N / 128) * g.batch, ts = ((long long)s2.M + 127) / 128 * s2.batch;
    if (tq < 384 || ts <= 0 || ts > tq || tq + ts >= (1LL << 31)) return 1;           // (small problems keep their own tile choice)
    hipLaunchKernelGGL(gemm3_side_kernel, dim3((unsigned)(tq + ts)), dim3(256), 0, st, g, s2);
    *rc = abx_check_launch("abx_gemm_side");
    return 0;
}

// resident workgroups per CU of the main instantiations (diagnostics for tools/kbench.py)
extern "C" int abx_gemm3_occupancy(int which) {
    int n = -1;
    hipError_t e = hipErrorInvalidValue;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm3_kernel<128, 192, 32, 192, 0, false, 3>), 256, 0);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm3_kernel<128, 128, 32, 128, 0, false, 4>), 256, 0);
    else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm3_kernel<128, 128, 32, 128, 0, true, 4>), 256, 0);
    else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm3_kernel<128, 192, 32, 192, 2, false, 3>), 256, 0);
    return e == hipSuccess ? n : -(int)e - 1000;
}

extern "C" int abx_split_weights_f16(const float* w, long long s_n, long long s_k, int N, int K, int scale_exp, unsigned short* out,
                                     hipStream_t st) {
    ABX_REQUIRE(w && out && N > 0 && K > 0 && scale_exp >= -100 && scale_exp <= 100, "abx_split_weights_f16: bad args");
    const int Kp = (K + 15) / 16 * 16;
    const long long total = (long long)N * Kp;
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, s_n, s_k, N, K, Kp,
                       ldexpf(1.0f, scale_exp), out);
    return abx_check_launch("abx_split_weights_f16");
}

extern "C" int abx_ipa_tail(const AbxIpaTail* ap, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_ipa_tail: null descriptor");
    const AbxIpaTail a = *ap;
    auto al16 = [](const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    ABX_REQUIRE(a.M > 0 && a.C == IT_C, "abx_ipa_tail: M > 0, C == 256");
    ABX_REQUIRE(al16(a.s) && a.s_s % 4 == 0 && a.s_s >= IT_C, "abx_ipa_tail: s must be 16-byte aligned rows");
    if (a.partial) {
        ABX_REQUIRE(al16(a.partial) && a.n_partial > 0 && a.s_partial >= (long long)a.M * IT_C && a.s_partial % 4 == 0,
                    "abx_ipa_tail: partial = n_partial K-slice products (M, 256) of feat W_final, s_partial floats apart");
    } else {
        ABX_REQUIRE(a.K1 > 0 && a.K1 % 16 == 0, "abx_ipa_tail: K1 % 16 == 0");
        ABX_REQUIRE(al16(a.feat) && a.s_feat % 4 == 0 && a.s_feat >= a.K1, "abx_ipa_tail: feat must be 16-byte aligned rows");
        ABX_REQUIRE(32LL * a.s_feat < (1LL << 28), "abx_ipa_tail: feature rows too long");
        ABX_REQUIRE(al16(a.W_final), "abx_ipa_tail: weight planes (abx_split_weights_f16)");
    }
    ABX_REQUIRE(al16(a.W_t0) && al16(a.W_t2) && al16(a.W_t4), "abx_ipa_tail: weight planes (abx_split_weights_f16)");
    ABX_REQUIRE(al16(a.b_final) && al16(a.b_t0) && al16(a.b_t2) && al16(a.b_t4) && al16(a.ln1_w) && al16(a.ln1_b) && al16(a.ln2_w) && al16(a.ln2_b),
                "abx_ipa_tail: biases and LayerNorm parameters ([256], 16-byte aligned)");
    for (int e : {a.e_final, a.e_t0, a.e_t2, a.e_t4}) ABX_REQUIRE(e >= -100 && e <= 100, "abx_ipa_tail: weight exponent out of range");
    ABX_REQUIRE(!a.W_aff || (a.b_aff && a.fixed && a.init_q && a.init_t && a.cur_q && a.cur_t && a.cur_R && a.delta_q && a.pscale != 0.f),
                "abx_ipa_tail: the affine / frame update needs all of its operands");
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&ipa_tail_kernel), IT_LDS, "abx_ipa_tail")) return rc;
    hipLaunchKernelGGL(ipa_tail_kernel, dim3((unsigned)((a.M + 31) / 32)), dim3(256), IT_LDS, st, a);
    return abx_check_launch("abx_ipa_tail");
}

extern "C" int abx_heads_tail(const AbxHeadsTail* ap, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_heads_tail: null descriptor");
    const AbxHeadsTail a = *ap;
    auto al16 = [](const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    ABX_REQUIRE(a.M > 0 && al16(a.s) && al16(a.s0) && a.s_s % 4 == 0 && a.s_s0 % 4 == 0 && a.s_s >= HT_C && a.s_s0 >= HT_C,
                "abx_heads_tail: s / s0 are (M, 256) fp32 rows, 16-byte aligned");
    ABX_REQUIRE(a.un && a.logits, "abx_heads_tail: un (M, 14) and logits (M, 20) outputs");
    const unsigned short* Ws[] = {a.W_act, a.W_init, a.W_r0, a.W_r1, a.W_r2, a.W_r3, a.W_proj, a.W_s1, a.W_s3, a.W_s5};
    const float* bs[] = {a.b_act, a.b_init, a.b_r0, a.b_r1, a.b_r2, a.b_r3, a.b_proj, a.b_s1, a.b_s3, a.b_s5};
    for (int i = 0; i < 10; ++i)
        ABX_REQUIRE(al16(Ws[i]) && al16(bs[i]), "abx_heads_tail: weight planes (abx_split_weights_f16, 128 columns) and [128] biases, 16-byte aligned");
    for (int e : {a.e_act, a.e_init, a.e_r0, a.e_r1, a.e_r2, a.e_r3, a.e_proj, a.e_s1, a.e_s3, a.e_s5})
        ABX_REQUIRE(e >= -100 && e <= 100, "abx_heads_tail: weight exponent out of range");
    ABX_REQUIRE(al16(a.lns_w) && al16(a.lns_b), "abx_heads_tail: LayerNorm parameters of the sequence head ([256])");
    if (a.W_p1) {
        ABX_REQUIRE(al16(a.W_p1) && al16(a.W_p3) && al16(a.W_p5) && al16(a.b_p1) && al16(a.b_p3) && al16(a.b_p5) && al16(a.lnp_w) && al16(a.lnp_b) && a.pl,
                    "abx_heads_tail: the pLDDT head needs all of its operands");
        for (int e : {a.e_p1, a.e_p3, a.e_p5}) ABX_REQUIRE(e >= -100 && e <= 100, "abx_heads_tail: weight exponent out of range");
    }
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&heads_tail_kernel), HT_LDS, "abx_heads_tail")) return rc;
    hipLaunchKernelGGL(heads_tail_kernel, dim3((unsigned)((a.M + 31) / 32)), dim3(256), HT_LDS, st, a);
    return abx_check_launch("abx_heads_tail");
}
