// A-STATIONARY, N-WALKING split-f16 GEMM for the LayerNorm -> Linear projections of the pair stack (K = 192 input channels):
//     TriangleAttention   q | k | v projection + its 4-column pair-bias projection        (seqformer.py:520-531)
//     TriangleMultiplication  [left | right] (value, gate) projections -> operand images   (seqformer.py:480-485)
//
// The tile kernels of gemm3.hip give every 128 x 128 output tile its own block: the 128 x 192 fp32 rows of z are fetched, LayerNorm-
// summed and split into f16 pieces once per N-tile (six times for the 768-wide projection), every tile pays a DMA round trip per k-step
// from a cold start, and its 64 KB store burst runs after its last MFMA with nothing of the same block to overlap it: the K-independent
// part of such a launch is as long as its k loops.  Here a block owns 64 ROWS and walks ALL N-tiles:
//   * its 64 x 192 fp32 rows arrive in ONE DMA burst (48 KB); the inline LayerNorm statistics and the split into the two f16 pieces
//     (a0, a1: common.h split2h_mix) are done once, IN PLACE and wave-locally: the fp32 image [32 rows][16] of a (k-tile, row half) and
//     its piece image [2][32][16] are both 2 KB.  The A fragments of the walk are two 16-byte LDS reads per k-step, no VALU;
//   * only the weight planes stream (8 KB per k-step, L2 resident), through a 3-stage ring requested three k-steps ahead with counted
//     vmcnt waits; the fragments of step s + 1 are read from LDS while the MFMAs of step s issue (two register sets);
//   * the epilogue of N-tile n (folded LayerNorm, bias, store) is cut into slices that ride in the k-steps of tile n + 1 on a second
//     accumulator set, registers only (a 4 x 4 DPP transpose per lane quad / v_permlane32_swap instead of an LDS staging buffer): the
//     stores leave under the next tile's matrix work instead of after the block's last MFMA;
//   * the FIRST tile has no epilogue to carry: its k-steps carry the statistics + split of the k-tiles two steps ahead instead, so
//     the walk starts when the first two k-tiles of the burst have landed and proceeds at the pace of the HBM fetch.
// 4 waves = 2 (rows) x 2 (columns), wave tile 32 x 64, with two ROLES: vector-memory results return in issue order, so a wave that
// waits for its weight DMA also waits for every store it issued before - the column-0 waves fetch and split A and never wait on vmcnt
// after the first tile (their stores run free), the column-1 waves stream the weights (and carry the coupling for their own stores).
// 75 KB of LDS, two blocks per CU: one block's prologue and its last, un-overlapped epilogue run under the other block's walk.
// The arithmetic of every output element is that of gemm3_kernel (same pieces, same three product terms per k-step in the same
// order, same statistics code in the same order): results are bit-identical to the tile kernels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "abx_hip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int AS_BM = 64, AS_BN = 128, AS_NK = 12, AS_RING = 3;
constexpr int AS_KT = 4096;                              // bytes of one k-tile of the A image: fp32 [64][16] or pieces [2][64][16]
constexpr int AS_A = AS_NK * AS_KT;                      // 49 152
constexpr int AS_STAGE = 2 * AS_BN * 32;                 // one k-tile of weight planes [2][128][16] f16: 8 192
constexpr int AS_HALF = AS_KT / 2;                       // one (k-tile, 32-row half): fp32 [32][16] (landing) or pieces [2][32][16]
constexpr int AS_OFF_RING = AS_A;
constexpr int AS_OFF_ST = AS_OFF_RING + AS_RING * AS_STAGE;         // [4][64]: mean - shift | rstd | row scale | row valid
constexpr int AS_OFF_CONST = AS_OFF_ST + 4 * 64 * 4;                // [2 tiles][csum 128 | bias 128]: column constants of a tile
constexpr int AS_LDS = AS_OFF_CONST + 2 * 1024;                     // 76 800 = 60 granules of 1 280 (two blocks per CU: <= 81 920)

__device__ __forceinline__ void as_glds16(const void* src, char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned as_vgpr32(unsigned o) {
    asm volatile("" : "+v"(o));
    return o;
}
template <int I> using IC = std::integral_constant<int, I>;
template <int I0, int I1, class F> __device__ __forceinline__ void as_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(IC<I0>{});
        as_static_for<I0 + 1, I1>(f);
    }
}

// byte offset of (plane, row, 16-byte half) in a [2][ROWS][16] 16-bit tile image (the layout of gemm3.hip: conflict-free 16-byte
// fragment reads with lane -> row, lane >> 5 -> k half)
template <int ROWS> __device__ __forceinline__ int as_plane_off(int plane, int row, int half) {
    return plane * (ROWS * 32) + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
}

// ---- the A image ------------------------------------------------------------------------------------------------------------------------
// source row of GEMM row gri (the row maps of gemm3_mainloop: padded pair positions, (8 i x 16 k) block order, pair transposition)
__device__ __forceinline__ long long as_a_row(const AbxGemm& g, int gri) {
    long long gr = gri;
    if (g.a_pair) {
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
    return gr;
}

// The burst of a column-0 wave (wm, 0): its 32 rows x 192 fp32, 24 DMA instructions in k-tile order (two per k-tile: rows 0 - 15 and
// 16 - 31 of the half).  Landing image of a (k-tile, half): [32 rows][4 slots of 16 bytes], slot p of row r holds the k-quad
// p ^ ((r >> 2) & 3) (conflict-free 16-byte reads of a row's two k halves by the 32 x 2 lanes of the splitting wave).
__device__ __forceinline__ void as_issue_a(const AbxGemm& g, char* lds, int m0, int b, int wm) {
    const int lane = threadIdx.x & 63;
    const bool remap = g.a_pair_transpose > 0 || g.a_pair != 0;
    const long long row0 = remap ? 0 : m0;
    const char* baseA = reinterpret_cast<const char*>(g.A + (long long)b * g.sAb + row0 * g.sAm);
    unsigned offA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * i + (lane >> 2), p = lane & 3;
        const int kq = p ^ ((r >> 2) & 3);
        const long long gr = as_a_row(g, min(m0 + wm * 32 + r, g.M - 1));
        offA[i] = (unsigned)(((gr - row0) * g.sAm + kq * 4) * 4);
    }
    char* dst = lds + wm * AS_HALF;
#pragma unroll
    for (int kt = 0; kt < AS_NK; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i) as_glds16(baseA + kt * 64 + as_vgpr32(offA[i]), dst + kt * AS_KT + i * 1024);
}

// statistics + split of ONE (k-tile, row half) by the wave that fetched it: lane = (row, k half) exactly as a fragment lane of the tile
// kernels, k-tiles in order, so the statistics add up in the order gemm3_mainloop adds them (bit-identical mean / rstd).  In place:
// a wave's LDS instructions execute in order, every read of the 2 KB image precedes the first write.
struct AsConv {
    float lshift, m2048;
    f32x2 ls2, lq2;
    int rd0, rd1, wr;
    bool ln, relu;
};
__device__ __forceinline__ void as_conv_init(AsConv& c, const AbxGemm& g) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, x = (r >> 2) & 3;
    c.rd0 = r * 64 + (((2 * h) ^ x) << 4);
    c.rd1 = r * 64 + (((2 * h + 1) ^ x) << 4);
    c.wr = as_plane_off<32>(0, r, h);
    c.lshift = 0.f;
    c.ls2 = c.lq2 = (f32x2){0.f, 0.f};
    c.m2048 = -2048.0f;
    asm volatile("" : "+v"(c.m2048));
    c.ln = g.ln_csum != nullptr;
    c.relu = g.a_relu != 0;
}
template <bool FIRST>
__device__ __forceinline__ void as_conv_ktile(AsConv& c, char* img) {
    const int lane = threadIdx.x & 63;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(img + c.rd0);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(img + c.rd1);
    float xv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] = lo[e]; xv[4 + e] = hi[e]; }
    if (c.ln) {
        if (FIRST) c.lshift = __shfl(xv[0], lane & 31, 64);      // the row's first element
        const f32x2 sh2 = {c.lshift, c.lshift};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x2 xp = {xv[2 * e], xv[2 * e + 1]};
            xp -= sh2;
            c.ls2 += xp;
            c.lq2 = __builtin_elementwise_fma(xp, xp, c.lq2);
            xv[2 * e] = xp[0];
            xv[2 * e + 1] = xp[1];
        }
    }
    if (c.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = fmaxf(xv[e], 0.f);
    }
    unsigned q0[4], q1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2h_mix(xv[2 * e], xv[2 * e + 1], c.m2048, q0[e], q1[e]);
    *reinterpret_cast<u32x4*>(img + c.wr) = u32x4{q0[0], q0[1], q0[2], q0[3]};
    *reinterpret_cast<u32x4*>(img + c.wr + 1024) = u32x4{q1[0], q1[1], q1[2], q1[3]};
}
// (mean - shift, rstd * cs) of the wave's 32 rows -> st.  cs = the accumulator scale 2^-(a_exp + b_exp) of the split-f16 product: the
// epilogues compute (rstd cs) (acc - dmean (csum / cs)) instead of rstd (acc cs - dmean csum) - the same bits (every scaling by the
// power of two is exact and commutes with the roundings), one multiplication less per element
__device__ __forceinline__ void as_conv_finish(const AsConv& c, float* st, int wm, float eps, float cs) {
    const int lane = threadIdx.x & 63;
    const float ls = c.ls2[0] + c.ls2[1], lq = c.lq2[0] + c.lq2[1];
    const float invK = 1.0f / (float)(AS_NK * 16);
    const float sm = ls + __shfl_xor(ls, 32, 64), sq = lq + __shfl_xor(lq, 32, 64);
    if (lane < 32) {
        const float dm = sm * invK;
        st[wm * 32 + lane] = dm;
        st[64 + wm * 32 + lane] = (1.0f / sqrtf(fmaxf(sq * invK - dm * dm, 0.f) + eps)) * cs;
    }
}

// 4 x 4 transpose inside every lane quad: lane q of a quad holds x[0 .. 3] = rows 0 .. 3 of column q and receives o[0 .. 3] = columns
// 0 .. 3 of row q.  Two exchange stages (lane ^ 1, lane ^ 2), each ONE v_cndmask_b32 with a DPP quad_perm source per value (the select
// and the cross-lane move in one instruction; vcc = the lanes that keep their own value): 8 VALU instructions instead of the 24 the
// compiler makes of update_dpp + select (a zeroed destination, the move, the select).  The leading s_nop covers the two wait states
// between the VALU write of x and its DPP read; inside the statement every DPP source is written >= 2 instructions earlier.
__device__ __forceinline__ void as_quad_transpose(const float (&x)[4], float (&o)[4]) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"                                    // even lanes keep rows 0, 2
        "v_cndmask_b32_dpp %4, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"               // a0 = even ? x0 : x1 of lane ^ 1
        "v_cndmask_b32_dpp %6, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"             // a2 = even ? x2 : x3 of lane ^ 1
        "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\ts_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"                                    // odd lanes keep rows 1, 3
        "v_cndmask_b32_dpp %5, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"               // a1 = odd ? x1 : x0 of lane ^ 1
        "v_cndmask_b32_dpp %7, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"             // a3 = odd ? x3 : x2 of lane ^ 1
        "s_mov_b32 vcc_lo, 0x33333333\n\ts_mov_b32 vcc_hi, 0x33333333\n\t"                                    // lanes 0, 1 of a quad
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"               // o0 = low ? a0 : a2 of lane ^ 2
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"               // o1 = low ? a1 : a3 of lane ^ 2
        "s_mov_b32 vcc_lo, 0xcccccccc\n\ts_mov_b32 vcc_hi, 0xcccccccc\n\t"                                    // lanes 2, 3 of a quad
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"               // o2 = high ? a2 : a0 of lane ^ 2
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"                     // o3 = high ? a3 : a1 of lane ^ 2
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])
        : "vcc");
}

// ---- operations a COLUMN-1 wave queues behind the weight DMA of a k-step (for its counted waits) -------------------------------------
// Per tile: two more DMA instructions at k-step 0 (the tile's column constants); NST stores at each of the k-steps [S0, S0 + NSL) of
// every tile but the first (the slices of the previous tile's epilogue).  S0 + NSL <= 10: the steps 10, 11 of a tile carry nothing, so
// the counts at the head of the next tile do not depend on which tile came before.
template <int S0, int NSL, int NST, int NDMA = 4>
struct AsSched {
    static_assert(S0 >= 2 && S0 + NSL <= 10, "slice window");
    static constexpr int after(int kt, bool first) { return (kt == 0 ? 2 : 0) + ((!first && kt >= S0 && kt < S0 + NSL) ? NST : 0); }
    // vector-memory operations that may still be in flight at the top of step kt while the DMA of step kt + 1 must have landed:
    // everything issued after it = the tail of step kt - 2, the DMA of step kt + 2 (four instructions) and the tail of step kt - 1
    static constexpr int allow(int kt, bool first) {
        const int k2 = (kt + AS_NK - 2) % AS_NK, k1 = (kt + AS_NK - 1) % AS_NK;
        return after(k2, first) + NDMA + after(k1, first);
    }
};

enum { AS_EPI_PLAIN = 0, AS_EPI_GLU = 1 };

// the k-step's rendezvous: the weight tile of the NEXT step has landed (own DMA: the newest N vector-memory operations may stay in
// flight - results return in issue order), own LDS reads have returned, block barrier.  The fragment registers of THIS step (read
// from LDS during the previous one) pass through the statement: its MFMAs cannot be scheduled above it, and the reads not below.
template <int N>
__device__ __forceinline__ void as_rendezvous(u32x4 (&fa)[2], u32x4 (&fb)[2][2]) {
    asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)\n\ts_barrier"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])
                 : "n"(N)
                 : "memory");
}
// the column-0 waves: no vector-memory wait (they issue no weight DMA)
__device__ __forceinline__ void as_rendezvous_free(u32x4 (&fa)[2], u32x4 (&fb)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])
                 :
                 : "memory");
}
template <int N> __device__ __forceinline__ void as_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- the block ------------------------------------------------------------------------------------------------------------------------
// EPI_PLAIN: C[m][n] = rstd (acc - dmean csum[n]) + bias[n] (the folded LayerNorm; the dispatch admits alpha == 1, no activation), 16-byte
//            stores of 128-byte row segments.
//            SIDE: the last N-tile of g has only its first 64 columns (N % 128 == 64); the other half of that tile computes the first 32
//            columns of s2 (N <= 32, transposed store (b, n, m)) for the same rows: the pair bias next to q | k | v.
// EPI_GLU:   the wave tile is a (value, gate) pair of 32 output channels: out = epi(value) sigmoid(epi(gate)) rowscale, written as
//            the k-tiled f16 operand image of the following contraction (C_split, c_split_tile row order: gemm_epilogue.h).
// PL (EPI_PLAIN): the columns n >= g.c_planes_from leave as two float16 planes of 16 x value in groups of 48 channels (AbxGemm.c_planes_from:
//            the k | v columns of the q | k | v projection as the operand image the triangle attention stages by DMA).  Every slice issues
//            TWO 8-byte stores per lane - the planes p0 / p1 of 4 channels, or the two halves of the 16 fp32 bytes - so that the counted
//            waits of the streaming waves stay exact whichever kind of column a slice holds.
template <int EPI, bool SIDE, bool EDGE, int ABL = 0, bool PL = false>
__device__ __forceinline__ void as_block(const AbxGemm& g, const AbxGemm& s2, char* lds, int mt, int b) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
    const int m0 = mt * AS_BM;
    float* st = reinterpret_cast<float*>(lds + AS_OFF_ST);       // [4][64]: mean - shift | rstd | row scale | row valid
    // ABL (probe builds, -DABX_AS_ABLATE + tune bits 12 - 13): 1 = no slice stores, 2 = no MFMA, 3 = half of the weight DMA inside the walk
    using Sched = typename std::conditional<EPI == AS_EPI_PLAIN, AsSched<2, 8, (ABL == 1 ? 0 : (PL ? 2 : 1)), (ABL == 3 ? 2 : 4)>, AsSched<6, 2, (ABL == 1 ? 0 : 2), (ABL == 3 ? 2 : 4)>>::type;

    // ---- weight operand: fetched by the column-1 waves, wave (wm, 1) = plane wm of a stage (four 1 KB chunks of 32 rows)
    const int ntiles = (g.N + AS_BN - 1) / AS_BN;
    const bool ragged = (g.N % AS_BN) != 0;                    // the last tile has 64 columns: wave column 1 idles there (or runs the side)
    const char* bbase = reinterpret_cast<const char*>(g.B_split);
    const long long bkstep = g.sB3k * 2;
    // per-lane source offset of chunk 0 (rows 0 .. 31 of the wave's plane); chunk i = + i * chunk_step, tile t = + t * tile_step.  The
    // chunks 2, 3 of the LAST tile are apart: columns beyond N clamped (their products are never stored), or the side operand
    unsigned off0, offU[2];
    {
        const int row = lane >> 1, hp = lane & 1;
        off0 = (unsigned)((wm * g.sB3p + (long long)row * g.sB3n + 8 * (hp ^ ((row >> 3) & 1))) * 2);
#pragma unroll
        for (int i = 2; i < 4; ++i) {
            const int r2 = i * 32 + row;                         // (bit 3 of the row - the half swap - is that of `row`)
            const int half = hp ^ ((row >> 3) & 1);
            if (SIDE) offU[i - 2] = (unsigned)((wm * s2.sB3p + (long long)min(r2 - 64, s2.N - 1) * s2.sB3n + 8 * half) * 2);
            else offU[i - 2] = (unsigned)((wm * g.sB3p + (long long)min((ntiles - 1) * AS_BN + r2, g.N - 1) * g.sB3n + 8 * half) * 2);
        }
    }
    const char* bbaseS = SIDE ? reinterpret_cast<const char*>(s2.B_split) : bbase;
    const long long bkstepS = SIDE ? s2.sB3k * 2 : bkstep;
    const unsigned tile_step = (unsigned)(AS_BN * g.sB3n * 2), chunk_step = (unsigned)(32 * g.sB3n * 2);
    // stage of step (tile, kt): ring slot kt % 3 (12 % 3 == 0)
    auto issue_b = [&](int tile, int kt, int slot, bool half_only = false) __attribute__((always_inline)) {
        char* dst = lds + AS_OFF_RING + slot * AS_STAGE + wm * 4096;
        const int t = min(tile, ntiles - 1);                    // (tiles beyond the walk re-request the last one: uniform instruction counts)
        const char* src = bbase + kt * bkstep + (long long)t * tile_step;
#pragma unroll
        for (int i = 0; i < 2; ++i) as_glds16(src + i * chunk_step + as_vgpr32(off0), dst + i * 1024);
        if (half_only) return;
        if (t == ntiles - 1) {
            const char* srcU = bbaseS + kt * bkstepS;
#pragma unroll
            for (int i = 2; i < 4; ++i) as_glds16(srcU + as_vgpr32(offU[i - 2]), dst + i * 1024);
        } else {
#pragma unroll
            for (int i = 2; i < 4; ++i) as_glds16(src + i * chunk_step + as_vgpr32(off0), dst + i * 1024);
        }
    };
    // column constants of a tile -> LDS [slot = tile & 1][csum 128 | bias 128] by DMA (4 bytes per lane: wave (0, 1) the column sums,
    // wave (1, 1) the biases): no load result for the compiler to wait on inside the walk
    const float* csrc = (wm ? g.bias : g.ln_csum);
    auto issue_consts = [&](int tile) __attribute__((always_inline)) {
        const int t = min(tile, ntiles - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = min(t * AS_BN + i * 64 + lane, g.N - 1);
            char* dst = lds + AS_OFF_CONST + (tile & 1) * 1024 + wm * 512 + i * 256;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(csrc + n), (lds_ptr_t)dst, 4, 0, 0);
        }
    };

    // ---- prologue.  Column 0: the A burst, statistics + split of its first two k-tiles.  Column 1: the first three weight stages
#ifdef ABX_AS_STAMP
    // probe build: shader-clock ticks of wave 0 per phase, one record per block behind clock_probe[16]
    unsigned long long t_wait = 0, t_vm = 0, t_bar = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_issued = 0, as_t_landed = 0;
#endif
    AsConv cv;
    char* a_half = lds + wm * AS_HALF;                          // + kt * AS_KT: the (k-tile, row half) images of this wave row
    if (wn == 0) {
        as_issue_a(g, lds, m0, b, wm);
#ifdef ABX_AS_STAMP
        t_issued = __builtin_amdgcn_s_memtime();
#endif
        as_conv_init(cv, g);
        as_wait_vm<20>();                                       // k-tiles 0, 1 of the burst (its first four instructions) have landed
#ifdef ABX_AS_STAMP
        as_t_landed = __builtin_amdgcn_s_memtime();
#endif
        as_conv_ktile<true>(cv, a_half);
        as_conv_ktile<false>(cv, a_half + AS_KT);
    } else {
        issue_b(0, 0, 0);
        issue_b(0, 1, 1);
        issue_b(0, 2, 2);
        if (EPI == AS_EPI_GLU && wm == 0) {
            // row scale (pair mask) and validity of the block's rows; rows = pair positions in (8 i x 16 k) block order
            const int m = m0 + lane;
            int pi, pj;
            pair_tile_decode(m, g.pair_Lp, pi, pj);
            const bool ok = m < g.M && pi < g.pair_L && pj < g.c_split_L;
            st[128 + lane] = (ok && g.rowscale) ? g.rowscale[(long long)b * g.sRSb + (long long)pi * g.pair_Lp + pj] : (ok ? 1.f : 0.f);
            st[192 + lane] = ok ? 1.f : 0.f;
        }
        as_wait_vm<8>();                                        // stage 0 has landed
    }
    __syncthreads();

    const float cs = __builtin_ldexpf(1.0f, -ABX_F16_A_EXP - g.b_exp), inv_cs = __builtin_ldexpf(1.0f, ABX_F16_A_EXP + g.b_exp);
    // (side operand: its own weight exponent; st holds rstd * cs)
    const float csS = SIDE ? __builtin_ldexpf(1.0f, -ABX_F16_A_EXP - s2.b_exp) : 0.f;
    float* Cb = g.C + (long long)b * g.sCb;
    bool bad = false;

    // ---- fragment addressing
    const int offAf = wm * AS_HALF + as_plane_off<32>(0, lane & 31, h);                     // + kt * AS_KT + p * 1024
    const int offBf = AS_OFF_RING + as_plane_off<AS_BN>(0, wn * 64 + (lane & 31), h);       // + slot * AS_STAGE + p * 4096 + j * 1024
    u32x4 fa[2][2], fb[2][2][2];                                // [set][piece], [set][sub-tile][plane]
    auto read_frags = [&](int set, int kt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) fa[set][p] = *reinterpret_cast<const u32x4*>(lds + kt * AS_KT + offAf + p * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                fb[set][j][p] = *reinterpret_cast<const u32x4*>(lds + offBf + slot * AS_STAGE + p * (AS_BN * 32) + j * 1024);
    };

    f32x16 acc[2], accp[2];
    float csump[2] = {0.f, 0.f}, biasp[2] = {0.f, 0.f};
    int tile_p = 0;                                             // tile the previous accumulators belong to
    // column constants of the previous tile: LDS -> registers (k-step 1 of the tile that carries its epilogue; the final flush)
    auto read_consts = [&]() __attribute__((always_inline)) {
        const float* cc = reinterpret_cast<const float*>(lds + AS_OFF_CONST + (tile_p & 1) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            csump[j] = cc[wn * 64 + j * 32 + (lane & 31)] * inv_cs;      // csum / cs (as_conv_finish)
            biasp[j] = cc[128 + wn * 64 + j * 32 + (lane & 31)];
        }
    };

    // ---- epilogue slices of the PREVIOUS tile (accp, csump, biasp, tile_p): registers only, no LDS round trip
    // row statistics of this lane's 16 rows (rows 8 rq + 4 h + c of the wave's 32): (mean - shift) and rstd
    // (st is written by the column-0 waves during the first tile; two 16-byte LDS reads per slice - 32 registers would hold them for
    // the whole walk, and the kernel has none to spare)
    const float* st_w = st + wm * 32 + 4 * h;
    // PLAIN: slice q = (row quad rq = q >> 1, sub-tile j = q & 1).  A lane holds 4 rows x 1 column; a 4 x 4 transpose inside every
    // lane quad (two DPP quad_perm exchanges) gives it 1 row x 4 consecutive columns: one 16-byte store per lane, 8 lanes = the 128
    // contiguous bytes of a row segment
    auto plain_slice = [&](auto q_) __attribute__((always_inline)) {
        constexpr int q = decltype(q_)::value, rq = q >> 1, j = q & 1;
        float x[4], o[4];
        const f32x4 dm4 = *reinterpret_cast<const f32x4*>(st_w + 8 * rq), rs4 = *reinterpret_cast<const f32x4*>(st_w + 64 + 8 * rq);
        // (the slice's arithmetic must stay in ITS k-step: the compiler would hoist the pure-VALU part of all eight slices to the head
        // of the tile - one long live range per value, spills; a volatile statement is not moved across the step's rendezvous)
        asm volatile("" : "+v"(accp[j][4 * rq]), "+v"(accp[j][4 * rq + 1]), "+v"(accp[j][4 * rq + 2]), "+v"(accp[j][4 * rq + 3]));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // (gemm_epilogue.h epi1 with a folded LayerNorm, alpha == 1, no activation - bit for bit)
            const float v = rs4[c] * (accp[j][4 * rq + c] - dm4[c] * csump[j]);       // rs4 = rstd cs, csump = csum / cs
            x[c] = v + biasp[j];
        }
        // (one class test per slice: a NaN / inf among the four makes their sum NaN / inf - the kernel is instruction-issue bound)
        bad |= __builtin_amdgcn_classf((x[0] + x[1]) + (x[2] + x[3]), 0x207);
        as_quad_transpose(x, o);
        const int m = m0 + wm * 32 + 8 * rq + 4 * h + (lane & 3);
        const int n = tile_p * AS_BN + wn * 64 + j * 32 + ((lane & 31) >> 2) * 4;
        if constexpr (PL) {
            // (the sub-tile's 32 columns are on one side of c_planes_from - a multiple of 32: a scalar branch)
            const int nsub = tile_p * AS_BN + wn * 64 + j * 32;
            float* cr = Cb + (long long)m * g.sCm;
            if (nsub >= g.c_planes_from) {
                unsigned a0, a1, b0, b1;
                split2b(o[0], o[1], a0, a1);
                split2b(o[2], o[3], b0, b1);
                const int rel = n - g.c_planes_from, gi = (int)(((unsigned)(rel >> 4) * 0xAAABu) >> 17), ch = rel - gi * 48;      // rel / 48
                unsigned short* cs = reinterpret_cast<unsigned short*>(cr + g.c_planes_from + gi * 48) + ch;
                if (!EDGE || m < g.M) {
                    *reinterpret_cast<u32x2*>(cs) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(cs + 48) = u32x2{a1, b1};
                }
            } else if (!EDGE || m < g.M) {
                *reinterpret_cast<f32x2*>(cr + n) = (f32x2){o[0], o[1]};
                *reinterpret_cast<f32x2*>(cr + n + 2) = (f32x2){o[2], o[3]};
            }
        } else
#ifdef ABX_AS_NT_STORE     // (probe build: non-temporal stores of the write-once projection output)
        if (ABL != 1 && (!EDGE || m < g.M)) __builtin_nontemporal_store((f32x4){o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(Cb + (long long)m * g.sCm + n));
#else
        if (ABL != 1 && (!EDGE || m < g.M)) *reinterpret_cast<f32x4*>(Cb + (long long)m * g.sCm + n) = (f32x4){o[0], o[1], o[2], o[3]};
#endif
    };
    // GLU: four value slices (row quad rq: the lane's 4 consecutive rows of its channel -> 8 bytes per plane), then two store slices:
    // v_permlane32_swap hands the h = 0 lane of a channel the rows 8 A .. 8 A + 7 and the h = 1 lane the rows 8 B .. 8 B + 7 of the row
    // octets (A, B) = (0, 1) / (2, 3): 16 contiguous bytes per lane and plane, the two lanes of a channel 32, the two slices its 64
    unsigned gp0[8], gp1[8];
    auto glu_value = [&](auto rq_) __attribute__((always_inline)) {
        constexpr int rq = decltype(rq_)::value;
        const int rl = wm * 32 + 8 * rq + 4 * h;
        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(st + 128 + rl);
        const f32x4 dm4 = *reinterpret_cast<const f32x4*>(st_w + 8 * rq), rs4 = *reinterpret_cast<const f32x4*>(st_w + 64 + 8 * rq);
        float v[4];
        asm volatile("" : "+v"(accp[0][4 * rq]), "+v"(accp[0][4 * rq + 1]), "+v"(accp[0][4 * rq + 2]), "+v"(accp[0][4 * rq + 3]));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = rs4[c] * (accp[0][4 * rq + c] - dm4[c] * csump[0]);        // rs4 = rstd cs, csump = csum / cs
            float gt = rs4[c] * (accp[1][4 * rq + c] - dm4[c] * csump[1]);
            a = a + biasp[0];
            gt = gt + biasp[1];
            v[c] = a * sigmoidf_(gt) * sc4[c];
        }
        bad |= __builtin_amdgcn_classf((v[0] + v[1]) + (v[2] + v[3]), 0x207);      // (one class test per slice, as plain_slice)
        // (the dispatch admits c_split_nA % 32 == 0: the wave's 32 channels are on one side - a scalar branch)
        if (tile_p * 64 + wn * 32 < g.c_split_nA) {
            split2h(v[0], v[1], gp0[2 * rq], gp1[2 * rq]);
            split2h(v[2], v[3], gp0[2 * rq + 1], gp1[2 * rq + 1]);
        } else {
            split2b(v[0], v[1], gp0[2 * rq], gp1[2 * rq]);
            split2b(v[2], v[3], gp0[2 * rq + 1], gp1[2 * rq + 1]);
        }
    };
    auto glu_round = [&](auto R_) __attribute__((always_inline)) {
        constexpr int R = decltype(R_)::value;                   // octet pair (2 R, 2 R + 1)
        constexpr int A = 2 * R, B = 2 * R + 1;
        const int n = tile_p * 64 + wn * 32 + (lane & 31);
        const int oct = h ? B : A;
        const int m = m0 + wm * 32 + oct * 8;                    // first of 8 consecutive rows: one i, 8 consecutive k
        int ii, kk;
        pair_tile_decode(m, g.c_split_L, ii, kk);
        unsigned short* cp0 = g.C_split + (long long)b * g.sCb + (long long)n * g.sCm + (kk >> 4) * g.sCk + ii * 16 + (kk & 15);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned* gp = p ? gp1 : gp0;
            // words (2 A, 2 A + 1) = rows 8 A + 4 h .. + 3, words (2 B, 2 B + 1) = rows 8 B + 4 h .. + 3
            const auto s0 = __builtin_amdgcn_permlane32_swap(gp[2 * A], gp[2 * B], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(gp[2 * A + 1], gp[2 * B + 1], false, false);
            // h = 0: (own A words | partner's A words), h = 1: (partner's B words | own B words)
            const u32x4 v = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
            unsigned short* cp = cp0 + p * g.sCp;
            if (ABL == 1) { if (v[0] == 0x12345u) *reinterpret_cast<u32x4*>(cp) = v; }
            else if (!EDGE) *reinterpret_cast<u32x4*>(cp) = v;     // (non-temporal here: 14.2 vs 11.2 ms - the 16-byte plane pieces need the L2 to combine)
            else if (m < g.M && ii < g.pair_L) {
                if (kk + 8 <= g.c_split_L) *reinterpret_cast<u32x4*>(cp) = v;
                else if (kk + 4 <= g.c_split_L) *reinterpret_cast<u32x2*>(cp) = u32x2{v[0], v[1]};
            }
        }
    };
    // SIDE: this wave's first sub-tile holds columns 0 .. 31 of s2 for its 32 rows: (b, n, m) store, 4 consecutive rows per register quad
    auto side_store = [&]() __attribute__((always_inline)) {
        const int n = lane & 31;
        if (n < s2.N) {
            const float cS = s2.ln_csum[n], bS = s2.bias ? s2.bias[n] : 0.f;
            const long long LLs = s2.M;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int rl = wm * 32 + 8 * rq + 4 * h;
                const f32x4 dm4 = *reinterpret_cast<const f32x4*>(st_w + 8 * rq), rs4 = *reinterpret_cast<const f32x4*>(st_w + 64 + 8 * rq);
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = accp[0][4 * rq + c] * csS;
                    x = (rs4[c] * inv_cs) * (x - dm4[c] * cS);
                    v[c] = (x + bS) * s2.alpha;
                    bad |= __builtin_amdgcn_classf(v[c], 0x207);
                }
                const long long mg = (long long)m0 + rl;        // global row = sb * LLs + ms
                const long long sb = mg / LLs, ms = mg - sb * LLs;
                float* op = s2.C + sb * s2.sCb + (long long)n * s2.sCm + ms;
                if ((LLs & 3) == 0 && (!EDGE || mg + 4 <= g.M) && (reinterpret_cast<uintptr_t>(op) & 15) == 0) *reinterpret_cast<f32x4*>(op) = v;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const long long mc = mg + c;
                        if (mc < g.M) {
                            const long long sbc = mc / LLs;
                            s2.C[sbc * s2.sCb + (long long)n * s2.sCm + (mc - sbc * LLs)] = v[c];
                        }
                    }
                }
            }
        }
    };

#if defined(ABX_AS_STAMP) && ABX_AS_STAMP >= 2
    // probe build: shader-clock ticks wave 0 spends in the rendezvous of the walk (perturbs the walk: s_memtime twice per step)
#define AS_STAMP_PRE unsigned long long ts0_ = __builtin_amdgcn_s_memtime();
#define AS_STAMP_POST t_wait += __builtin_amdgcn_s_memtime() - ts0_;
#else
#define AS_STAMP_PRE
#define AS_STAMP_POST
#endif
    // ---- one k-step.  set = kt & 1 holds this step's fragments; the next step's go into the other set
    // The wave's ROLE (its column wn) is a compile-time constant of the step: the walk is instantiated once per role below, so that a
    // step is one basic block and the compiler can lay the role's extra work (DMA issue / split) between the MFMAs.
    auto step = [&](int tile, auto kt_, auto first_, auto role_) __attribute__((always_inline)) {
        constexpr int kt = decltype(kt_)::value;
        constexpr bool first = decltype(first_)::value != 0;
        constexpr int role = decltype(role_)::value;
        constexpr int set = kt & 1;
        AS_STAMP_PRE
#if defined(ABX_AS_STAMP) && ABX_AS_STAMP >= 3
        if constexpr (role == 1 && !EDGE) {
            // probe build: the streaming wave's wait for its DMA (and, in issue order, for its older stores) apart from the barrier
            const unsigned long long ta_ = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)"
                         : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0][0]), "+v"(fb[set][0][1]), "+v"(fb[set][1][0]), "+v"(fb[set][1][1])
                         : "n"(Sched::allow(kt, first)) : "memory");
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            t_vm += tb_ - ta_;
            t_bar += __builtin_amdgcn_s_memtime() - tb_;
        } else
#endif
        if constexpr (role == 0) as_rendezvous_free(fa[set], fb[set]);
        else if constexpr (EDGE) as_rendezvous<0>(fa[set], fb[set]);             // (predicated stores: the instruction counts are not exact)
        else as_rendezvous<Sched::allow(kt, first)>(fa[set], fb[set]);
        AS_STAMP_POST
        if constexpr (role == 1) {
            {   // weight tile of step s + 3 -> the ring slot step s has just left
                constexpr int k3 = (kt + 3) % AS_NK;
                issue_b(kt + 3 >= AS_NK ? tile + 1 : tile, k3, kt % 3, ABL == 3);
            }
            if constexpr (kt == 0) issue_consts(tile);
        } else if constexpr (first && kt < AS_NK - 2) {
            // first tile, column 0: statistics + split of the k-tile two steps ahead (its fragments are read one step ahead, behind
            // the next rendezvous); this wave's vector-memory queue holds nothing but the burst
            as_wait_vm<2 * (AS_NK - 3 - kt)>();
            as_conv_ktile<false>(cv, a_half + (kt + 2) * AS_KT);
            if constexpr (kt == AS_NK - 3) as_conv_finish(cv, st, wm, g.ln_eps, cs);
        }
        if constexpr (kt == 1 && !first) read_consts();
        read_frags((kt + 1) & 1, (kt + 1) % AS_NK, (kt + 1) % 3);
        if (ABL == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_bit_cast(float, fa[set][0][e] ^ fb[set][j][0][e] ^ fb[set][j][1][e] ^ fa[set][1][e]);
        } else {   // (the idle wave column of a ragged last tile computes on duplicate columns: no branch in the step)
            u32x4 p2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) p2[j] = f16x8_lo(fb[set][j][0]);
            // a1 p2, a0 p1, a0 p0 (smallest first); the first term of a tile starts from zero
            if constexpr (kt == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = mfma_split(fa[set][1], p2[j], z);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = mfma_split(fa[set][1], p2[j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mfma_split(fa[set][0], fb[set][j][1], acc[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = mfma_split(fa[set][0], fb[set][j][0], acc[j]);
        }
        if constexpr (!first) {
            if constexpr (EPI == AS_EPI_PLAIN) {
                if constexpr (kt >= 2 && kt < 10) plain_slice(IC<kt - 2>{});
            } else {
                if constexpr (kt >= 2 && kt < 6) glu_value(IC<kt - 2>{});
                if constexpr (kt >= 6 && kt < 8) glu_round(IC<kt - 6>{});
            }
        }
    };
    auto rotate = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) accp[j] = acc[j];
        tile_p = tile;
    };

    // ---- the walk
#ifdef ABX_AS_STAMP
    const unsigned long long t_walk0 = __builtin_amdgcn_s_memtime();
    unsigned long long t_walk1 = 0;
#endif
    read_frags(0, 0, 0);                                        // (pieces of k-tile 0 written, weight stage 0 landed: the barrier above)
    auto walk = [&](auto role_) __attribute__((always_inline)) {
        as_static_for<0, AS_NK>([&](auto kt_) { step(0, kt_, IC<1>{}, role_); });
        rotate(0);
        for (int tile = 1; tile < ntiles; ++tile) {
            as_static_for<0, AS_NK>([&](auto kt_) { step(tile, kt_, IC<0>{}, role_); });
            rotate(tile);
        }
#ifdef ABX_AS_STAMP
        t_walk1 = __builtin_amdgcn_s_memtime();
#endif
        // the last tile's epilogue, un-overlapped (every DMA still in flight is a duplicate nobody reads; drained before the block ends)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        constexpr int role = decltype(role_)::value;
        if (SIDE && role == 1) side_store();
        else if (SIDE || !ragged || role == 0) {                // (the idle wave column of a ragged last tile stores nothing)
            read_consts();
            if constexpr (EPI == AS_EPI_PLAIN) as_static_for<0, 8>([&](auto q_) { plain_slice(q_); });
            else {
                as_static_for<0, 4>([&](auto q_) { glu_value(q_); });
                as_static_for<0, 2>([&](auto q_) { glu_round(q_); });
            }
        }
    };
    if (wn == 0) walk(IC<0>{});
    else walk(IC<1>{});
    if (g.range_flag && __any(bad) && lane == 0) atomicOr(g.range_flag, g.range_tag);
#ifdef ABX_AS_STAMP
    if (g.clock_probe && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        // one record per block (atomics on one line serialise at ~12 ns each and keep finished blocks resident)
        unsigned long long* rec = g.clock_probe + 16 + (size_t)blockIdx.x * 8;
        rec[0] = t_wait; rec[1] = t_walk1 - t_walk0; rec[2] = t_walk0 - t_begin; rec[3] = t_end - t_walk1;
        rec[4] = t_issued - t_begin; rec[5] = as_t_landed - t_issued; rec[6] = t_end - t_begin; rec[7] = 1ull;
    }
    if (g.clock_probe && threadIdx.x == 64) {                    // wave (0, 1): a streaming wave's record behind the blocks' first records
        unsigned long long* rec = g.clock_probe + 16 + ((size_t)gridDim.x + blockIdx.x) * 8;
        rec[0] = t_vm; rec[1] = t_bar; rec[7] = 1ull;
    }
#endif
}

template <int EPI, bool SIDE, int ABL = 0, bool PL = false>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(const AbxGemm g, const AbxGemm s2) {
    extern __shared__ __attribute__((aligned(16))) float as_smem[];
    char* lds = reinterpret_cast<char*>(as_smem);
    const ClockProbe probe(g.clock_probe);
    const unsigned ntm = (unsigned)((g.M + AS_BM - 1) / AS_BM);
    // XCD-aware remap (gemm3_kernel): every XCD gets a contiguous range of row tiles
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int b = (int)(wgid / ntm), mt = (int)(wgid - (unsigned)b * ntm);
    bool edge = (mt + 1) * AS_BM > g.M;
    if (EPI == AS_EPI_GLU) edge = edge || (g.pair_L & 7) != 0 || (g.c_split_L & 15) != 0;     // padded pair rows: predicated stores
    if (edge) as_block<EPI, SIDE, true, ABL, PL>(g, s2, lds, mt, b);
    else as_block<EPI, SIDE, false, ABL, PL>(g, s2, lds, mt, b);
    probe.finish();
}

}  // namespace

// Called by abx_gemm / abx_gemm_side (gemm.hip) with validated descriptors (vector flags filled).  side == nullptr: no side GEMM.
// Returns 1 when the problem is not served here (the caller goes on to the tile kernels of gemm3.hip).
int abx_gemm_as_dispatch(const AbxGemm& g, const AbxGemm* side, hipStream_t st, int* rc) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    static const bool off = getenv("ABX_NO_GEMM_AS") != nullptr;          // (A / B measurements: always the tile kernels)
    if (off || (g.tune & 2048)) return 1;                       // tune bit 11: the tile kernels for this call
    if (g.exact == 1 || !g.B_split || !g.b_f16 || g.A_split || !g.A || g.sAk != 1 || g.K != AS_NK * 16) return 1;
    if (g.A2 || g.out_ln_w || g.mlp || g.ln_stats || g.gate || g.resid || g.batch_inner || !g.ln_csum || !g.bias) return 1;
    if (g.act != 0 || g.alpha != 1.0f) return 1;
    if (!al16(g.A) || g.sAm % 4 != 0 || g.sAb % 4 != 0 || !al16(g.B_split) || g.sB3n % 8 != 0 || g.sB3p % 8 != 0 || g.sB3k % 8 != 0) return 1;
    if (g.b_exp < -100 || g.b_exp > 100) return 1;
    if (((g.a_pair_transpose > 0 || g.a_pair) ? (long long)g.M * g.sAm : 64LL * g.sAm) >= (1LL << 30)) return 1;
    if ((long long)(g.K / 16) * g.sB3k >= (1LL << 31)) return 1;
    const long long ntm = ((long long)g.M + AS_BM - 1) / AS_BM;
    if (ntm * g.batch >= (1LL << 31)) return 1;
    // the walk pays from two N-tiles on; small problems keep the tile kernels (more blocks than CUs matter more there).
    // INVARIANT: this choice depends on the LAUNCH size, also under exact == 2 where the caller has fixed the arithmetic class so that a
    // sample's bits do not depend on how many samples share a launch - batch / chunk / shard invariance of the network therefore rests on
    // this kernel being BIT-IDENTICAL to the tile kernels of gemm3.hip (test_gemm_as_*: edge and non-edge shapes, with / without a side,
    // a_relu, and test_gemm_as_dispatch_threshold_is_bit_invariant across this very threshold).  A change that gives that up must gate
    // the kernel on L (the complex) instead of M (the launch).
    if (g.N < 256 || ntm * g.batch < 1024) return 1;
    const bool glu = g.glu != 0;
    if (glu) {
        if (side || !g.C_split || !g.c_split_tile || !g.c_transposed || g.N % 128 != 0 || !g.a_pair || g.pair_Lp <= 0 || g.c_split_L != g.pair_Lp || g.c_split_nA % 32 != 0) return 1;
        if (!g.c_vec_ok || (reinterpret_cast<uintptr_t>(g.C_split) & 15) != 0 || g.sCb % 8 != 0 || g.sCm % 8 != 0 || g.sCk % 8 != 0 || g.sCp % 8 != 0) return 1;
        if (g.sB3b != 0) return 1;
    } else {
        if (g.c_transposed || g.C_split || g.rowscale || g.a_pair_transpose > 0 || g.pair_Lp != 0 || g.batch != 1) return 1;
        if (!g.c_vec_ok || g.N % 64 != 0) return 1;
        // plane output of the columns from c_planes_from on: this kernel's PL instantiation only (with the side projection: the q | k | v launch)
        if (g.c_planes_from > 0 && (!side || g.c_planes_group != 48 || g.c_planes_from % 32 != 0 || (g.N - g.c_planes_from) % 48 != 0 || g.N >= (1 << 16))) return 1;
        if (side) {
            const AbxGemm& s = *side;
            if (g.N % 128 != 64) return 1;                      // the side rides in the free half of the ragged last tile
            if (!s.B_split || !s.b_f16 || s.A_split || s.A != g.A || s.sAk != 1 || s.K != g.K || s.sAm != g.sAm || !s.c_transposed || s.N > 32 ||
                s.exact == 1 || (long long)s.M * s.batch != (long long)g.M || (s.batch > 1 && s.sAb != (long long)s.M * s.sAm) ||
                (s.ln_csum == nullptr) != (g.ln_csum == nullptr) || s.ln_stats || s.gate || s.resid || s.rowscale || s.act != 0 || s.glu || s.C_split ||
                s.A2 || s.out_ln_w || s.mlp || s.a_pair_transpose > 0 || s.pair_Lp != 0 || s.a_relu != g.a_relu ||
                !al16(s.B_split) || s.sB3n % 8 != 0 || s.sB3p % 8 != 0 || s.sB3k % 8 != 0 || s.b_exp < -100 || s.b_exp > 100 ||
                (long long)(s.K / 16) * s.sB3k >= (1LL << 31) || (g.ln_csum && s.ln_eps != g.ln_eps))
                return 1;
        }
    }
    static const AbxGemm none = {};
    const dim3 grid((unsigned)(ntm * g.batch)), block(256);
#define AS_LAUNCH(EPI, SIDE, ABL, ...)                                                                                                   \
    do {                                                                                                                                 \
        if (int e = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_as_kernel<EPI, SIDE, ABL, ##__VA_ARGS__>), AS_LDS, "abx_gemm(as)")) { \
            *rc = e;                                                                                                                     \
            return 0;                                                                                                                    \
        }                                                                                                                                \
        hipLaunchKernelGGL((gemm_as_kernel<EPI, SIDE, ABL, ##__VA_ARGS__>), grid, block, AS_LDS, st, g, side ? *side : none);            \
    } while (0)
#ifdef ABX_AS_ABLATE
    const int abl = (g.tune >> 12) & 3;
    if (abl == 1) { if (glu) AS_LAUNCH(AS_EPI_GLU, false, 1); else if (side) AS_LAUNCH(AS_EPI_PLAIN, true, 1); else AS_LAUNCH(AS_EPI_PLAIN, false, 1); }
    else if (abl == 2) { if (glu) AS_LAUNCH(AS_EPI_GLU, false, 2); else if (side) AS_LAUNCH(AS_EPI_PLAIN, true, 2); else AS_LAUNCH(AS_EPI_PLAIN, false, 2); }
    else if (abl == 3) { if (glu) AS_LAUNCH(AS_EPI_GLU, false, 3); else if (side) AS_LAUNCH(AS_EPI_PLAIN, true, 3); else AS_LAUNCH(AS_EPI_PLAIN, false, 3); }
    else
#endif
    if (glu) AS_LAUNCH(AS_EPI_GLU, false, 0);
    else if (side && g.c_planes_from > 0) AS_LAUNCH(AS_EPI_PLAIN, true, 0, true);
    else if (side) AS_LAUNCH(AS_EPI_PLAIN, true, 0);
    else AS_LAUNCH(AS_EPI_PLAIN, false, 0);
#undef AS_LAUNCH
    *rc = abx_check_launch("abx_gemm(as)");
    return 0;
}
