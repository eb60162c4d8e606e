// Pair-representation assembly AND the sequence attention's pair bias in one pass over the pair rows (round 6):
//     z0[b,i,j,:]      = [pair_static | t_emb | t_emb] + LayerNorm(prev_pair) + Embedding(prev_pos)        (seqformer.py:193-223)
//     bias[b,h,i,j]    = Linear(LayerNorm(z0[b,i,j,:]))[h]                                                  (seqformer.py:324-333: proj_pair(pair_norm(pair)))
// `abx_assemble_pair` wrote z0 (9.5 GB at 100 samples of L = 352) and the 192 -> 32 projection read it back in a launch of its own at 4.4 TB/s
// (2.1 - 2.3 ms): the rows are here in registers anyway.  One workgroup owns a (b, i) ROW of the pair tensor; a wave takes 32 positions j at a time
// in the FRAGMENT layout of the matrix cores - lane = (position, k half), 8 consecutive channels per 16-channel k-tile - so that the assembled row
// is at once what is stored (64 contiguous bytes per position and k-tile) and, shifted and split into its two f16 pieces, the A operand of the
// projection: 12 k-steps x 3 product terms against the LayerNorm-folded weight planes in LDS (24 KB), the folded-LayerNorm epilogue of the GEMMs
// (rstd (acc - dmean csum) + bias'), the 32 x 128 bias tile of a round staged in LDS and written as 512-byte row segments of (b, h, i, :).
// The arithmetic of the projection is the split-f16 product of abx_gemm (A pieces of x 2^-4, weight planes with their exponent, three exact terms,
// fp32 accumulate, shifted statistics); the assembly is fp32 as before (LayerNorm of prev_pair two-pass, centred).  HBM traffic: prev_pair read,
// z0 written, the bias written - the second read of z0 is gone.
// MEASURED SLOWER than the two launches it replaces (7.14 vs 5.99 ms at 100 samples of L = 352, 2.9 TB/s: profiles/r06q_kb_assemble_bias.txt): in the
// fragment layout a load / store instruction touches 32 cache lines with two 16-byte pieces each (the 16-lanes-per-row kernel of embed.hip: four
// lines of 256 contiguous bytes), and the texture path, not HBM, sets the pace; a version that lands the rows by DMA in a row-major LDS image and
// re-reads them as fragments would need 24 KB per wave tile.  Kept as a tested entry point (test_assemble_pair_with_the_seq_attention_bias), not
// used by the model unless ABX_ASSEMBLE_BIAS is set.
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int AB_W = 192, AB_C = 128, AB_E = 32, AB_H = 32, AB_NK = AB_W / 16;
constexpr int AB_IMG = 2 * AB_H * 32;                        // one k-tile of the weight image: [2 planes][32 heads][16 k] float16 = 2 048 bytes
constexpr int AB_BT = 129;                                   // row stride (floats) of the staged bias tile [32 heads][128 positions]: conflict-free columns
constexpr int AB_OFF_BT = AB_NK * AB_IMG;                    // 24 576
constexpr int AB_OFF_ST = AB_OFF_BT + AB_H * AB_BT * 4;      // + 16 512: per wave [32 rows][2] (dmean, rstd) of the projection's LayerNorm
constexpr int AB_OFF_CONST = AB_OFF_ST + 4 * 32 * 2 * 4;     // + 1 024: gamma 192 | beta 192 | csum 32 | bias 32
constexpr int AB_LDS = AB_OFF_CONST + (2 * AB_W + 2 * AB_H) * 4;      // 43 904 bytes; two workgroups per CU (the 96 registers of the prev_pair row: 168 VGPRs spill)

__device__ __forceinline__ int ab_plane_off(int plane, int row, int half) {
    return plane * (AB_H * 32) + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
}

__global__ __launch_bounds__(256, 2) void assemble_pair_bias_kernel(const float* __restrict__ pair_static, long long ps_b, const float* __restrict__ temb,
                                                                    const float* __restrict__ prev, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const long long* __restrict__ prev_pos,
                                                                    const float* __restrict__ pos_table, float* __restrict__ out,
                                                                    const unsigned short* __restrict__ Wp, int w_exp, const float* __restrict__ csum,
                                                                    const float* __restrict__ pbias, float ln_eps, float* __restrict__ biasT, int B, int L,
                                                                    int* range_flag, int range_tag) {
    extern __shared__ __attribute__((aligned(16))) float ab_smem[];
    char* lds = reinterpret_cast<char*>(ab_smem);
    float* bt = reinterpret_cast<float*>(lds + AB_OFF_BT);
    float* stw = reinterpret_cast<float*>(lds + AB_OFF_ST);
    float* cst = reinterpret_cast<float*>(lds + AB_OFF_CONST);          // gamma | beta | csum | bias'
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, r = lane & 31;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    const int b = (int)(wgid / (unsigned)L), i = (int)(wgid - (unsigned)b * (unsigned)L);
    const long long LL = (long long)L * L;

    // ---- prologue: the weight image (k-tiled planes [kt][plane][head][16] as abx_split_weights_f16 writes them -> the swizzled LDS image) and constants
    for (int it = tid; it < AB_NK * 2 * AB_H * 2; it += 256) {          // 16-byte pieces: (kt, plane, head, k half)
        const int hh = it & 1, n = (it >> 1) & (AB_H - 1), p = (it >> 6) & 1, kt = it >> 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Wp + ((long long)(kt * 2 + p) * AB_H + n) * 16 + hh * 8);
        *reinterpret_cast<u32x4*>(lds + kt * AB_IMG + ab_plane_off(p, n, hh)) = v;
    }
    for (int it = tid; it < 2 * AB_W + 2 * AB_H; it += 256)
        cst[it] = it < AB_W ? (gamma ? gamma[it] : 1.f) : it < 2 * AB_W ? (beta ? beta[it - AB_W] : 0.f) : it < 2 * AB_W + AB_H ? csum[it - 2 * AB_W] : (pbias ? pbias[it - 2 * AB_W - AB_H] : 0.f);
    __syncthreads();

    const float cs = __builtin_ldexpf(1.0f, -ABX_F16_A_EXP - w_exp);       // accumulator scale of the split product
    const long long row0 = ((long long)b * L + i) * L;                     // first pair row of (b, i)
    const float* st_row = pair_static + (long long)b * ps_b + (long long)i * L * AB_C;
    const float* te = temb + (long long)b * AB_E;
    bool bad = false;
    const int nround = (L + 127) / 128;
    for (int rd = 0; rd < nround; ++rd) {
        const int j0 = rd * 128 + wave * 32;
        if (j0 < L) {
            const int j = min(j0 + r, L - 1);
            const long long row = row0 + j;
            // ---- LayerNorm(prev_pair row): the lane pair (r, h = 0 / 1) holds the row, 8 channels per k-tile each
            f32x4 x[AB_NK][2];
            float mean = 0.f, rstd = 0.f;
            if (prev) {
                const float* pr = prev + row * AB_W + 8 * h;
                float s = 0.f;
#pragma unroll
                for (int kt = 0; kt < AB_NK; ++kt) {
                    x[kt][0] = *reinterpret_cast<const f32x4*>(pr + kt * 16);
                    x[kt][1] = *reinterpret_cast<const f32x4*>(pr + kt * 16 + 4);
                    s += ((x[kt][0][0] + x[kt][0][1]) + (x[kt][0][2] + x[kt][0][3])) + ((x[kt][1][0] + x[kt][1][1]) + (x[kt][1][2] + x[kt][1][3]));
                }
                s += __shfl_xor(s, 32, 64);
                mean = s / (float)AB_W;
                float qq = 0.f;
#pragma unroll
                for (int kt = 0; kt < AB_NK; ++kt)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float d = x[kt][u][c] - mean;
                            qq = fmaf(d, d, qq);
                        }
                qq += __shfl_xor(qq, 32, 64);
                rstd = 1.0f / sqrtf(qq / (float)AB_W + 1e-5f);
            }
            const float* pt = prev_pos ? pos_table + prev_pos[row] * AB_W + 8 * h : nullptr;
            float* op = out + row * AB_W + 8 * h;
            // ---- per k-tile: assemble, store, shift, statistics of the projection's LayerNorm, split, three products
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            float lshift = 0.f, ls = 0.f, lq = 0.f;
#pragma unroll
            for (int kt = 0; kt < AB_NK; ++kt) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c0 = kt * 16 + 8 * h + 4 * u;                  // first of 4 consecutive channels
                    f32x4 a;
                    if (kt < AB_C / 16) a = *reinterpret_cast<const f32x4*>(st_row + (long long)j * AB_C + c0);
                    else a = *reinterpret_cast<const f32x4*>(te + ((c0 - AB_C) & (AB_E - 1)));
                    if (prev) {
                        const f32x4 ga = *reinterpret_cast<const f32x4*>(cst + c0), be = *reinterpret_cast<const f32x4*>(cst + AB_W + c0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) a[c] += (x[kt][u][c] - mean) * rstd * ga[c] + be[c];
                    }
                    if (pt) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pt + kt * 16 + 4 * u);
#pragma unroll
                        for (int c = 0; c < 4; ++c) a[c] += pv[c];
                    }
                    if (j0 + r < L) *reinterpret_cast<f32x4*>(op + kt * 16 + 4 * u) = a;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[4 * u + c] = a[c];
                }
                // (gemm3_mainloop's inline LayerNorm: shifted by the row's first element, partial sums per lane, the two k halves folded at the end)
                if (kt == 0) lshift = __shfl(v[0], r, 64);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] -= lshift;
                    ls += v[e];
                    lq = fmaf(v[e], v[e], lq);
                }
                unsigned q0[4], q1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2h(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
                const u32x4 a0 = {q0[0], q0[1], q0[2], q0[3]}, a1 = {q1[0], q1[1], q1[2], q1[3]};
                const u32x4 p0 = *reinterpret_cast<const u32x4*>(lds + kt * AB_IMG + ab_plane_off(0, r, h));
                const u32x4 p1 = *reinterpret_cast<const u32x4*>(lds + kt * AB_IMG + ab_plane_off(1, r, h));
                acc = mfma_split(a1, f16x8_lo(p0), acc);                    // a1 p2, a0 p1, a0 p0: smallest first
                acc = mfma_split(a0, p1, acc);
                acc = mfma_split(a0, p0, acc);
            }
            {   // row statistics of the projection's LayerNorm -> the wave's strip (the accumulators hold 16 ROWS per lane)
                const float sm = ls + __shfl_xor(ls, 32, 64), sq = lq + __shfl_xor(lq, 32, 64);
                if (h == 0) {
                    const float dm = sm / (float)AB_W;
                    stw[(wave * 32 + r) * 2] = dm;
                    stw[(wave * 32 + r) * 2 + 1] = 1.0f / sqrtf(fmaxf(sq / (float)AB_W - dm * dm, 0.f) + ln_eps);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- epilogue: lane = head n (r), registers = rows 8 rq + 4 h + c of the tile -> the staged bias tile [head][position of the round]
            const float csn = cst[2 * AB_W + r], bn = cst[2 * AB_W + AB_H + r];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = 8 * (e >> 2) + 4 * h + (e & 3);
                const float dm = stw[(wave * 32 + rl) * 2], rs = stw[(wave * 32 + rl) * 2 + 1];
                const float vv = rs * (acc[e] * cs - dm * csn) + bn;
                bad |= __builtin_amdgcn_classf(vv, 0x207);
                bt[r * AB_BT + wave * 32 + rl] = vv;
            }
        }
        __syncthreads();
        {   // the round's bias tile: 32 heads x up to 128 positions, 512 contiguous bytes per head
            const int jr0 = rd * 128, nj = min(128, L - jr0);
            for (int it = tid; it < AB_H * 128; it += 256) {
                const int n = it >> 7, jj = it & 127;
                if (jj < nj) biasT[((long long)b * AB_H + n) * LL + (long long)i * L + jr0 + jj] = bt[n * AB_BT + jj];
            }
        }
        __syncthreads();
    }
    if (range_flag && __any(bad) && lane == 0) atomicOr(range_flag, range_tag);
}

}  // namespace

extern "C" int abx_assemble_pair_bias(const float* pair_static, long long ps_b, const float* temb, const float* prev_pair, const float* gamma,
                                      const float* beta, const long long* prev_pos, const float* pos_table, float* out,
                                      const unsigned short* w_planes, int w_exp, const float* csum, const float* bias, float ln_eps, float* biasT,
                                      int B, int L, int* range_flag, int range_tag, hipStream_t st) {
    ABX_REQUIRE(pair_static && temb && out && w_planes && csum && biasT && B > 0 && L > 0, "abx_assemble_pair_bias: bad args");
    ABX_REQUIRE(!prev_pair || (gamma && beta), "abx_assemble_pair_bias: LN params missing");
    ABX_REQUIRE(!prev_pos || pos_table, "abx_assemble_pair_bias: pos table missing");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    ABX_REQUIRE(al16(pair_static) && ps_b % 4 == 0 && al16(temb) && al16(out) && al16(w_planes) && (!prev_pair || al16(prev_pair)) &&
                    (!prev_pos || al16(pos_table)),
                "abx_assemble_pair_bias: 16-byte aligned operands (C = 128, E = 32, 32 heads)");
    ABX_REQUIRE(w_exp >= -100 && w_exp <= 100 && (long long)B * L < (1LL << 31), "abx_assemble_pair_bias: bad exponent / grid too large");
    if (ln_eps <= 0.f) ln_eps = 1e-5f;
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(assemble_pair_bias_kernel), AB_LDS, "abx_assemble_pair_bias")) return rc;
    hipLaunchKernelGGL(assemble_pair_bias_kernel, dim3((unsigned)((long long)B * L)), dim3(256), AB_LDS, st, pair_static, ps_b, temb, prev_pair, gamma,
                       beta, prev_pos, pos_table, out, w_planes, w_exp, csum, bias, ln_eps, biasT, B, L, range_flag, range_tag);
    return abx_check_launch("abx_assemble_pair_bias");
}
