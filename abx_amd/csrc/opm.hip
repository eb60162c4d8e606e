// OuterProductMean, its feature tensor never in memory (reference abx/model/seqformer.py:395-411):
//     z[b,i,j,:] += out_proj([l_j * r_i | l_j - r_i]) = l_j . U_i + c_i          l = left[b,j,0:64], r = right[b,i,0:64] (both masked)
//     U_i = diag(r_i) W1 + W2   (64 x 192)        c_i = bias - r_i . W2   (192)        W = out_proj weight^T [128][192] = [W1 ; W2]
// Until round 5 `opm_features` wrote the (B, L, L, 128) feature tensor (6.3 GB at 100 samples of L = 352) and a K = 128 GEMM read it back next to
// the 9.5 GB of z it updates: 31.7 GB of HBM traffic per pass for 19 GB of pair rows (DESIGN 4d item 9 priced a plane-contraction form of the
// identity above on the existing tile kernel: the per-tile fixed costs at K = 80 ate the saving).  Here one workgroup owns a (b, i) ROW of the pair
// tensor:
//   prologue  U_i as the B-side operand image in LDS ([4 k-tiles][2 planes][192][16] float16 of 16 x value: 48 KB; W comes from the L2, 98 KB per
//             352 pair rows) and c_i;
//   main      a wave takes 32 positions j at a time: its A operand is l_j - 64 fp32 per position straight from the L2 (the (L, 64) table of a
//             sample is shared by its L workgroups), split into the two f16 pieces in registers; 4 k-steps x 6 column tiles x 3 product terms =
//             72 MFMAs per 32 x 192 tile, in two column halves of 48 accumulator registers;
//   epilogue  + c_i + z (the residual), a 4 x 4 DPP transpose per lane quad -> 16-byte loads / stores, 128 contiguous bytes per row segment.
// HBM traffic = z read + z written.  Half the matrix products of the feature form (K = 64 instead of 128).  The arithmetic is the split-f16 product
// of the contraction kernels (A pieces of x 2^-4, B planes of y 2^4, three exact terms, fp32 accumulate): |U| < 4095, else NaN -> range word.
// Mathematically the reference's expression regrouped: sum_k (l r) W1 + sum_k (l - r) W2 = l . (diag(r) W1 + W2) - r . W2.
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int OPM_C = 64, OPM_N = 192, OPM_KT = OPM_C / 16;
constexpr int OPM_IMG = 2 * OPM_N * 32;                     // one k-tile of the U image: [2 planes][192 columns][16 k] float16 = 12 288 bytes
constexpr int OPM_LDS = OPM_KT * OPM_IMG + OPM_N * 4 + OPM_C * 4;      // U image 49 152 + c_i 768 + r_i 256

// byte offset of (plane, row, 16-byte half) in a [2][ROWS][16] 16-bit tile image (the layout of gemm3.hip: conflict-free 16-byte fragment reads)
__device__ __forceinline__ int opm_plane_off(int plane, int row, int half) {
    return plane * (OPM_N * 32) + row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
}

// 4 x 4 transpose inside every lane quad (gemm_as.hip as_quad_transpose): lane q of a quad holds x[0 .. 3] = rows 0 .. 3 of column q and receives
// o[0 .. 3] = columns 0 .. 3 of row q
__device__ __forceinline__ void opm_quad_transpose(const float (&x)[4], float (&o)[4]) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"
        "v_cndmask_b32_dpp %4, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %6, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\ts_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
        "v_cndmask_b32_dpp %5, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b32 vcc_lo, 0x33333333\n\ts_mov_b32 vcc_hi, 0x33333333\n\t"
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b32 vcc_lo, 0xcccccccc\n\ts_mov_b32 vcc_hi, 0xcccccccc\n\t"
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])
        : "vcc");
}

__global__ __launch_bounds__(256, 3) void opm_out_kernel(const float* __restrict__ lr, long long ld, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, float* __restrict__ z, int B, int L,
                                                         int* range_flag, int range_tag) {
    extern __shared__ __attribute__((aligned(16))) float opm_smem[];
    char* Us = reinterpret_cast<char*>(opm_smem);
    float* cs = reinterpret_cast<float*>(Us + OPM_KT * OPM_IMG);           // [192]  c_i
    float* rs = cs + OPM_N;                                                 // [64]   r_i
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    // XCD-aware remap: every XCD gets a contiguous range of (b, i) rows, so the (L, 64) table of a sample is fetched into one L2
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const unsigned wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    const int b = (int)(wgid / (unsigned)L), i = (int)(wgid - (unsigned)b * (unsigned)L);
    const float* lrb = lr + (long long)b * L * ld;

    // ---- prologue: r_i, then U_i = diag(r_i) W1 + W2 as plane image and c_i = bias - r_i . W2
    if (tid < OPM_C) rs[tid] = lrb[(long long)i * ld + OPM_C + tid];
    __syncthreads();
    for (int it = tid; it < OPM_N * OPM_KT * 2; it += 256) {               // item = (column n, k-tile kt, k half hh): 8 consecutive k
        const int n = it % OPM_N, kh = it / OPM_N, kt = kh >> 1, hh = kh & 1;
        const int k0 = kt * 16 + hh * 8;
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = fmaf(rs[k0 + e], Wt[(long long)(k0 + e) * OPM_N + n], Wt[(long long)(OPM_C + k0 + e) * OPM_N + n]);
        unsigned p0[4], p1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2b(u[2 * e], u[2 * e + 1], p0[e], p1[e]);
        char* d = Us + kt * OPM_IMG + opm_plane_off(0, n, hh);
        *reinterpret_cast<u32x4*>(d) = u32x4{p0[0], p0[1], p0[2], p0[3]};
        *reinterpret_cast<u32x4*>(d + OPM_N * 32) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    }
    if (tid < OPM_N) {
        float acc = bias ? bias[tid] : 0.f;
#pragma unroll 8
        for (int k = 0; k < OPM_C; ++k) acc = fmaf(-rs[k], Wt[(long long)(OPM_C + k) * OPM_N + tid], acc);
        cs[tid] = acc;
    }
    __syncthreads();

    bool bad = false;
    float* zrow0 = z + ((long long)b * L + i) * L * OPM_N;
    for (int j0 = wave * 32; j0 < L; j0 += 128) {
        // ---- A operand of this wave's 32 positions: l_j, 8 consecutive channels per (lane, k-tile) -> pieces a0, a1
        const int j = min(j0 + (lane & 31), L - 1);
        const float* lp = lrb + (long long)j * ld + 8 * h;
        u32x4 fa[OPM_KT][2];
#pragma unroll
        for (int kt = 0; kt < OPM_KT; ++kt) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(lp + kt * 16), hi = *reinterpret_cast<const f32x4*>(lp + kt * 16 + 4);
            unsigned q0[4], q1[4];
            split2h(lo[0], lo[1], q0[0], q1[0]);
            split2h(lo[2], lo[3], q0[1], q1[1]);
            split2h(hi[0], hi[1], q0[2], q1[2]);
            split2h(hi[2], hi[3], q0[3], q1[3]);
            fa[kt][0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
            fa[kt][1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // the residual rows of this half are requested before its matrix work (12 x 16 bytes per lane in flight under 36 MFMAs): as loads
            // in front of every store the epilogue ran at 4.3 TB/s
            f32x4 zres[3][4];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int jr = min(j0 + 8 * rq + 4 * h + (lane & 3), L - 1);
                    zres[t][rq] = *reinterpret_cast<const f32x4*>(zrow0 + (long long)jr * OPM_N + half * 96 + t * 32 + ((lane & 31) >> 2) * 4);
                }
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < OPM_KT; ++kt) {
                u32x4 fb[3][2];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        fb[t][p] = *reinterpret_cast<const u32x4*>(Us + kt * OPM_IMG + opm_plane_off(p, half * 96 + t * 32 + (lane & 31), h));
                // a1 p2, a0 p1, a0 p0 (smallest first), consecutive MFMAs on different accumulators
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma_split(fa[kt][1], f16x8_lo(fb[t][0]), acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma_split(fa[kt][0], fb[t][1], acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma_split(fa[kt][0], fb[t][0], acc[t]);
            }
            // ---- epilogue of the half: + c_i + z, 16 bytes per lane (the accumulators hold rows 8 rq + 4 h + c of column lane & 31)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float cn = cs[half * 96 + t * 32 + (lane & 31)];
                const int n4 = half * 96 + t * 32 + ((lane & 31) >> 2) * 4;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float x[4], o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = acc[t][4 * rq + c] + cn;
                    opm_quad_transpose(x, o);
                    const int jr = j0 + 8 * rq + 4 * h + (lane & 3);
                    if (jr < L) {
                        float* zp = zrow0 + (long long)jr * OPM_N + n4;
                        const f32x4 zv = zres[t][rq];
                        f32x4 v = {o[0] + zv[0], o[1] + zv[1], o[2] + zv[2], o[3] + zv[3]};
                        bad |= __builtin_amdgcn_classf((v[0] + v[1]) + (v[2] + v[3]), 0x207);
                        *reinterpret_cast<f32x4*>(zp) = v;
                    }
                }
            }
        }
    }
    if (range_flag && __any(bad) && lane == 0) atomicOr(range_flag, range_tag);
}

}  // namespace

extern "C" int abx_opm_out_fwd(const float* lr, long long ld, const float* Wt, const float* bias, float* z, int B, int L, int* range_flag,
                               int range_tag, hipStream_t st) {
    ABX_REQUIRE(lr && Wt && z && B > 0 && L > 0, "abx_opm_out_fwd: bad args");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    ABX_REQUIRE(al16(lr) && al16(z) && ld % 4 == 0 && ld >= 2 * OPM_C, "abx_opm_out_fwd: lr rows of [left 64 | right 64] floats, 16-byte aligned");
    ABX_REQUIRE((long long)B * L < (1LL << 31), "abx_opm_out_fwd: grid too large");
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(opm_out_kernel), OPM_LDS, "abx_opm_out_fwd")) return rc;
    hipLaunchKernelGGL(opm_out_kernel, dim3((unsigned)((long long)B * L)), dim3(256), OPM_LDS, st, lr, ld, Wt, bias, z, B, L, range_flag, range_tag);
    return abx_check_launch("abx_opm_out_fwd");
}
