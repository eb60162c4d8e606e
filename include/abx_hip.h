/* libabx_hip — C ABI of the MI355X (gfx950) kernels behind AbX's reverse-diffusion sampling hot path.
 *
 * The reference (CarbonMatrixLab/AbX) is pure PyTorch: it has no FFI of its own, so the "binding" a maintainer
 * adds is a ctypes stub that replaces the ATen op groups listed in SURVEY.md §2.1 / §8(a).
 * Every entry below cites the reference site (file:line under the reference root) it replaces.
 *
 * Conventions:
 *   - raw DEVICE pointers + explicit sizes/strides (in ELEMENTS) + hipStream_t; no torch types;
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library keeps no mutable global state;
 *   - every entry returns int: 0 = ok, <0 = argument check failed, >0 = hipError_t of the launch;
 *     abx_last_error_string() gives the thread-local message;
 *   - asynchronous on the given stream, no internal synchronisation (hipGraph-capturable), re-entrant.
 */
#ifndef ABX_HIP_H
#define ABX_HIP_H

#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ABX_HIP_ABI_VERSION 1

int abx_version(void);
const char* abx_last_error_string(void);
/* device properties sanity check: returns 0 when the current device is gfx950 */
int abx_init(int device);
/* diagnostics: fill the LDS of every CU with `pattern` (results must not depend on stale LDS contents) */
int abx_debug_poison_lds(unsigned pattern, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Dense contractions.  Replaces every torch Linear / LayerNorm->Linear / einsum on the path:
 * abx/model/common_modules.py:11-44 (Linear), abx/model/seqformer.py:358-376 (Transition), :380-411 (OPM out_proj),
 * :443-504 (TriangleMultiplication projections + 'bikc,bjkc->bijc' / 'bkic,bkjc->bijc'), :260-312 (q/k/v/gate/out),
 * abx/model/score_network.py:117-137, abx/model/folding.py:69-132 (IPA projections), abx/model/head.py:147-160,207-220.
 *   acc[b][m][n] = sum_k relu?(A[b][m][k]) * B[b][k][n]
 *   LayerNorm over k is applied algebraically in the epilogue: pass B scaled by gamma (B[k][n] = gamma[k] W[n][k]),
 *   ln_csum[n] = sum_k B[k][n], bias[n] = sum_k beta[k] W[n][k] + b[n] and the row statistics:  ln(v) = rstd[m] (v - mean[m] csum[n])
 *   epi(v) = ((ln(v) + bias[n]) * alpha) -> act -> * rowscale[b][m] -> * (sigmoid?)(gate[b][m][n]) -> + resid[b][m][n]
 *
 * MODES.  Beside the plain GEMM the descriptor selects one of the fused forms below; abx_gemm_check_modes (called by abx_gemm first,
 * callable on its own without a GPU) rejects every pair marked x with a negative code and a message that names both modes.
 *   glu       glu = 1: (value, gate) column pairs -> value * sigmoid(gate)             (tri-mul projections, seqformer.py:480-485)
 *   mlp       mlp = 1 + B2_split: relu(LN(A) B + b) B2 + b2 (+ resid), hidden on-chip   (pair Transition, seqformer.py:358-376)
 *   dual      A2 + B2_split: epi(A B) * sigmoid(LN(A2) B2 + b2) (+ resid)               (tri-mul tail, seqformer.py:496-503)
 *   c_split   C_split: output as the f16 operand image of the following contraction     (tri-mul projections)
 *   out_ln    out_ln_w: LayerNorm over the OUTPUT row                                    (IpaScore pair init, score_network.py:117-120)
 *   a_split   A_split (+ B_split of activations): plane x plane contraction             ('bikc,bjkc->bijc', seqformer.py:490-493)
 *   pair      pair_Lp > 0 / a_pair / c_pair / a_pair_transpose: padded or transposed pair-row maps
 *   exact     exact = 1: the exact fp32-MFMA kernels (plain epilogue only)
 *
 *              glu   mlp   dual  c_split out_ln a_split pair  exact
 *   glu         .     x     x     ok      x      x      ok    x
 *   mlp         x     .     x     x       x      x      x     x
 *   dual        x     x     .     x       x      x      ok    x
 *   c_split     ok    x     x     .       x      x      ok    x
 *   out_ln      x     x     x     x       .      x      x     x
 *   a_split     x     x     x     x       x      .      x     x
 *   pair        ok    x     ok    ok      x      x      .     x
 *   exact       x     x     x     x       x      x      x     .
 * Further requirements of a single mode (also checked, same function): glu and c_split need c_transposed; glu: N % 128 == 0, no gate;
 * mlp: act = 1, folded LayerNorm (ln_csum, no ln_stats), N2 <= 192, no gate / rowscale / c_transposed; dual: ln2_csum, no c_transposed;
 * out_ln: N <= 128, out_ln_b, no c_transposed; a_split: no LayerNorm, no a_relu (there is no fp32 row to normalise).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct AbxGemm {
    const float* A; long long sAb, sAm, sAk;       /* one of sAm / sAk must be 1 */
    const float* B; long long sBb, sBk, sBn;       /* one of sBk / sBn must be 1 */
    float* C;       long long sCb, sCm;            /* n-contiguous; if c_transposed: (m,n) at C + b*sCb + n*sCm + m */
    int M, N, K, batch;
    int c_transposed;
    const float* ln_stats; long long sSb;          /* (mean,rstd) pairs, row index b*sSb + m; NULL with ln_csum set = the kernel
                                                      computes the row statistics itself from the A operand stream */
    const float* ln_csum;                          /* [N] column sums of the gamma-scaled B; NULL = no LayerNorm */
    float ln_eps;                                  /* inline statistics: epsilon (0 -> 1e-5) */
    int a_relu;
    const float* bias;                             /* [N] or NULL */
    float alpha;                                   /* use 1.0f for none */
    int act;                                       /* 0 none, 1 relu, 2 sigmoid */
    const float* rowscale; long long sRSb;         /* [b*sRSb + m] or NULL */
    const float* gate; long long sGb, sGm; int gate_sigmoid;   /* addressed like C: (m,n) at gate + b*sGb + m*sGm + n, or */
    const float* resid; long long sRb, sRm;        /* + n*sGm + m when c_transposed; resid likewise and may alias C */
    /* Split-f16 operands and output (csrc/gemm3.hip): every fp32 product of the large contractions is evaluated on the float16
     * matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate) from 3 exact partial products:
     *     x y  ~  a1 p2 + a0 p1 + a0 p0      A side (two pieces):  x' = x 2^-4,  a0 = f16(x'),  a1 = f16((x' - a0) 2^11)
     *                                        B side (two planes):   y' = y 2^e,   p0 = f16(y'),  p1 = f16(y' - p0);  p2 = f16(p0 2^-11)
 *                                                               is derived in registers (one rounding of an exact value)
     * A piece pair carries 23 significant bits (|x' - a0 - a1 2^-11| <= 2^-23 |x'| worst case, 2^-25 on average), the dropped term
     * a1 p1 2^-11 is <= 2^-22 |x y| with mean zero: measured as accurate as the exact fp32 MFMA kernel
     * (tests/test_gpu_kernels.py::test_gemm_split_accuracy_vs_exact).  A-side range: |x| < 2^20, full relative precision from
     * |x| = 2^-9, 2^-32 absolute below; an operand beyond the range becomes inf - inf = NaN in its output row, never a silently wrong
     * number (remedy: exact = 1).  fp32 A operands are split in registers by the kernels.
     *   Weights (B of an fp32-A GEMM): abx_split_weights_f16 planes with a per-tensor e = b_exp (max |y'| in [2^13, 2^14)): b_f16 = 1;
     *     the accumulators are multiplied by 2^(4 - b_exp) before the epilogue.
     *   Activations on both sides (A_split and B_split: the TriangleMultiplication contraction): images written by the C_split
     *     epilogue of the projection GEMM, B side with the fixed e = 4 (|y| < 4095; 2^-29 absolute below 2^-6); b_f16 = 0.
     * The planes are k-TILED: element (row, k) of plane p at base + batch*sXb + (k/16)*sXk + p*sXp + row*sXr + k%16 (16-bit units),
     * i.e. the 16 k of one k-tile are contiguous (abx_split_weights_f16 writes [Kp/16][2][N][16]: sB3k = 2*N*16, sB3p = N*16,
     * sB3n = 16). */
    const unsigned short* B_split; long long sB3p, sB3n, sB3k, sB3b;   /* B as planes (sB3b = 0: shared weights); used instead of B */
    const unsigned short* A_split; long long sA3p, sA3m, sA3k, sA3b;   /* A as planes (pieces a0, a1), used instead of A:
                                                      the TriangleMultiplication contraction takes both operands this way (K % 16 == 0) */
    int batch_inner; long long sA3i, sB3i;         /* batch_inner > 0: two-level batch of the plane operands, entry b sits at
                                                      (b / batch_inner) * sX3b + (b % batch_inner) * sX3i (left / right channels of
                                                      one sample inside a wider channel tensor) */
    int c_split_tile;                              /* with C_split, pair_Lp > 0 and a_pair: the M = ceil8(pair_L) * ceil16(pair_Lp) GEMM rows are
                                                      pair positions (i, k) in blocks of (8 i x 16 k), m = ((i/8) * KT + k/16) * 128 + (i%8) * 16
                                                      + k%16 (KT = ceil(pair_Lp / 16)): 64 contiguous plane bytes per store instruction and
                                                      channel.  rowscale stays indexed [i * pair_Lp + k] */
    int c_split_nA;                                /* with C_split: output channels n < c_split_nA are written as the A side of the
                                                      following contraction (pieces a0, a1), the others as its B side (planes p0, p1): two planes per channel */
    unsigned short* C_split; long long sCp, sCk; int c_split_L;   /* write the output as planes instead of C, laid out as the
                                                      k-tiled OPERAND of the following contraction: with m = i*L + k (L =
                                                      c_split_L, M % L == 0, transposed store only), element (m, n) of plane p goes to
                                                      C_split + b*sCb + n*sCm + (k/16)*sCk + p*sCp + i*16 + k%16 */
    int glu;                                       /* transposed store only: the N columns are (value, gate) pairs of 32-column blocks
                                                      [v0 | g0 | v1 | g1 | ...] of the same N/2 output channels (weights packed that
                                                      way); out = epi(value) * sigmoid(epi(gate)), C / C_split have N/2 channels.
                                                      Needs a kernel whose wave tile holds both blocks (split-f16 128x128 tiles) */
    int a_pair_transpose;                          /* L > 0: M == L*L rows per batch are pair positions (i,k); row i*L + k of
                                                      the GEMM reads source row k*L + i (k-contiguous A, split-f16 path only) */
    int pair_L, pair_Lp;                           /* pair_Lp > 0: the M rows of the GEMM are PADDED pair positions m = i*pair_Lp + j
                                                      (i < pair_L, j < pair_Lp, pair_Lp % 4 == 0, M == pair_L*pair_Lp): any residue
                                                      count L keeps 16-byte aligned pair rows inside the triangle multiplication
                                                      (split-f16 path only).  With C_split, c_split_L == pair_Lp */
    int a_pair;                                    /* A rows live in the UNpadded pair tensor: GEMM row (i,j) reads source row
                                                      i*pair_L + min(j, pair_L-1), or min(j, pair_L-1)*pair_L + i when
                                                      a_pair_transpose != 0 (k-contiguous fp32 A) */
    int c_pair;                                    /* C / gate / resid rows live in the UNpadded pair tensor: GEMM row (i,j) is
                                                      stored at row i*pair_L + j, rows with j >= pair_L are dropped (plain store) */
    /* Dual GEMM (split-f16 path, plain store): out = epi(A' B) * sigmoid(LN(A2) B2 + bias2) (+ resid) - the tail of the
     * TriangleMultiplication (seqformer.py:496-503: proj_out(final_norm(x)) * sigmoid(final_gate(norm(z)))) in one kernel.
     * A2: k-contiguous fp32 rows (K2 % 16 == 0), statistics inline; with pair_Lp > 0 its rows live in the UNpadded pair tensor. */
    const float* A2; long long sA2b, sA2m; int K2;
    const unsigned short* B2_split; long long sB23p, sB23n, sB23k;   /* gate weights as k-tiled planes (abx_split_weights_f16, b2_exp) */
    const float* ln2_csum; const float* bias2;     /* [N] column sums of the gamma-scaled gate weights, folded bias */
    /* Fused two-layer transition (split-f16 path, plain store; seqformer.py:358-376 LayerNorm -> Linear -> ReLU -> Linear + residual):
     * mlp != 0:  C = relu(LN(A) B + bias) B2 + bias2 (+ resid), the N-wide hidden activations never leave the CU.  A k-contiguous
     * fp32 with inline LayerNorm (ln_csum, ln_stats NULL), act = 1; N % 16 == 0 = hidden width; C / resid have N2 <= 192 columns;
     * B2_split = planes of the second layer's weights [N/16][2][N2][16] (strides sB23k / sB23p / sB23n) whose 16 k of every k-tile
     * are stored in the order 0-3, 8-11, 4-7, 12-15 (the accumulator layout of the first GEMM feeds the second one from
     * registers); C may alias resid and A (a block reads its rows before it writes them).
     * mlp = 2, the gated tail of the TriangleAttention (seqformer.py:300-312) in the same shape:
     *     C = (sigmoid(LN(A) B + bias) * gate) B2 + bias2 (+ resid)
     * act = 2; gate = the attention output [M][N] fp32 rows (sGm, sGb; 16-byte aligned), N = gate channels (N % 16 == 0); B2_split
     * in the permuted k order as above; C may alias resid and A. */
    int mlp, N2;
    /* LayerNorm over the N OUTPUT columns (gamma, beta; eps), applied right after bias / alpha / act and before rowscale / gate /
     * resid: Linear -> LayerNorm without a round trip of the rows through HBM (score_network.py:117-120).  Needs a kernel whose
     * wave tile holds whole rows: split-f16 path only (its own 128x128-tile instantiation), N <= 128, k-contiguous fp32 A, plain store */
    const float* out_ln_w; const float* out_ln_b; float out_ln_eps;
    int exact;                                     /* 0: large problems run on the float16 matrix cores from split operands
                                                      (3 exact products per fp32 product, fp32 accumulate - csrc/gemm3.hip);
                                                      1: always the exact fp32 MFMA kernel (v_mfma_f32_32x32x2_f32);
                                                      2: the split-f16 kernels whatever the problem size (falls back to the exact
                                                      kernel only on shape / alignment grounds): callers that need results
                                                      independent of the batch size fix the arithmetic per op with 1 or 2 */
    /* float16 weight planes: see "Split-f16 operands" above.  b_exp / b2_exp: exponents of B_split / B2_split (|.| <= 100) */
    int b_f16, b_exp, b2_exp;
    int tune;                                      /* 0 = library default; kernel-variant selector for benchmarking */
    int c_planes_from, c_planes_group;             /* plain store, c_planes_from > 0: the output columns n >= c_planes_from leave as the B-side
                                                      OPERAND IMAGE of a following split-f16 product instead of fp32 - two float16 planes of
                                                      16 x value (p0 = f16(x'), p1 = f16(x' - p0): the same 4 bytes per element) in groups of
                                                      c_planes_group channels: channel n = c_planes_from + gi * G + ch of row m, plane p, sits at
                                                      byte  4 * (m * sCm + c_planes_from + gi * G) + p * 2 G + 2 ch  of C.  The k | v columns of
                                                      TriangleAttention's q | k | v projection (seqformer.py:520-531) are written this way
                                                      (from = 192, G = 48 = one head): the attention kernel stages them by DMA instead of loading,
                                                      splitting and writing them with a producer wave (AbxTriAttn.kv_planes).  from, G and N
                                                      multiples of 4, (N - from) % G == 0; no gate / glu / C_split / transposed store */
    int* range_flag; int range_tag;                /* range safety of the split-f16 kernels, optional DEVICE [1]: a workgroup whose accumulators
                                                      are not finite - what an operand beyond the split ranges above turns into, inf - inf,
                                                      and what a non-finite input gives too - ORs range_tag into *range_flag (one atomic per
                                                      wave, only then).  The caller clears the word, gives every call site its own bit, reads
                                                      it after a network call and repeats the call with exact = 1 (abx_amd/model/abx.py does) */
    unsigned long long* clock_probe;               /* diagnostics, optional DEVICE [2]: every workgroup of the split-f16 kernels adds
                                                      its elapsed shader-clock ticks (s_memtime) to [0] and its elapsed constant
                                                      100 MHz ticks (s_memrealtime) to [1]: 100 MHz * [0] / [1] = the shader clock
                                                      the kernel actually ran at (tools/probes/clock_probe.py) */
    int a_vec_ok, b_vec_ok, fast_ok;               /* filled by the library */
    int c_vec_ok, g_vec_ok, r_vec_ok, rs_vec_ok;   /* filled by the library (16-byte epilogue accesses possible) */
} AbxGemm;
int abx_gemm(const AbxGemm* desc, hipStream_t stream);
/* Two GEMMs over the SAME rows in one launch: `main_gemm` a plain-store split-f16 problem with N % 128 == 0 (k-contiguous fp32 A), `side_gemm` a
 * skinny (N <= 32) transposed-store projection - TriangleAttention's q | k | v | gate projection and its pair bias, which both read
 * LayerNorm(z) (seqformer.py:520-531).  The side tiles are dealt between the main tiles of the grid, so the skinny projection reads its
 * A panel from the L2 the main tiles have just filled instead of streaming the 9.5 GB pair tensor from HBM in a launch of its own.
 * Bit-identical to abx_gemm(main_gemm) + abx_gemm(side_gemm), which is also what it does when the pair does not qualify. */
int abx_gemm_side(const AbxGemm* main_gemm, const AbxGemm* side_gemm, hipStream_t stream);
/* 1 when an abx_gemm_side launch over M rows takes the kernel that can write plane output (AbxGemm.c_planes_from), else 0 */
int abx_gemm_planes_ok(long long M);
/* the mode table above, without a launch: 0 or a negative code (abx_last_error_string names the offending pair / requirement) */
int abx_gemm_check_modes(const AbxGemm* desc);
/* fp32 weights W[n][k] -> out[Kp/16][2][N][16] float16 planes (p0, p1) of w * 2^scale_exp (see AbxGemm.b_f16); the caller picks
 * scale_exp = 14 - e with max|w| = m * 2^e, 0.5 <= m < 1 (so that max|w| * 2^scale_exp is in [2^13, 2^14)) */
int abx_split_weights_f16(const float* w, long long s_n, long long s_k, int N, int K, int scale_exp, unsigned short* out, hipStream_t stream);
/* The tail of an IPA layer on the single representation s (M = B*L rows of C = 256 channels), reference score_network.py:126-163 /
 * folding.py (per layer, after the attention): s <- LN1(s + feat W_final + b_final); s <- LN2(s + relu(relu(s W0 + b0) W2 + b2) W4 + b4)
 * (attention_module.final_proj + attention_layer_norm + transition_module.0/.2/.4 + transition_layer_norm) in ONE launch: the
 * intermediate activations never leave the CU (split-f16 arithmetic of AbxGemm; weights as abx_split_weights_f16 planes
 * [K/16][2][256][16] with their exponents).  s is updated in place. */
typedef struct AbxIpaTail {
    const float* feat; long long s_feat;           /* IPA features (M, K1), row stride in floats (K1 % 16 == 0) */
    float* s; long long s_s;                       /* (M, 256) in / out */
    int M, K1, C;                                  /* C must be 256 */
    const unsigned short* W_final; int e_final; const float* b_final;
    const float* ln1_w; const float* ln1_b;
    const unsigned short* W_t0; int e_t0; const float* b_t0;
    const unsigned short* W_t2; int e_t2; const float* b_t2;
    const unsigned short* W_t4; int e_t4; const float* b_t4;
    const float* ln2_w; const float* ln2_b;
    float ln_eps;
    /* optional (W_aff != NULL): affine_update (W_aff [256][6] fp32, b_aff [6]) of the new s and the frame update of abx_rigid_update
     * (same operands: fixed_mask, init / current frames, accumulated quaternion, position scale) in the same launch */
    const float* W_aff; const float* b_aff;
    const int* fixed; const float* init_q; const float* init_t;
    float* cur_q; float* cur_t; float* cur_R; float* delta_q; float pscale;
    int* range_flag; int range_tag;                /* see AbxGemm.range_flag: set when the layer's output rows are not finite */
    /* optional (partial != NULL): feat W_final arrives as n_partial K-slice products (each (M, 256) fp32, s_partial floats apart) of a
     * split-K abx_gemm launched before (batch = slice: sAb = K1 / n_partial, sB3b = (K1 / n_partial / 16) * sB3k, sCb = s_partial); the kernel adds
     * them in slice order instead of walking K1 itself - a 32-row block's 132-k-step DMA chain at K1 = 2 112 is what a layer costs at
     * small batches.  feat / W_final / K1 are then unused. */
    const float* partial; int n_partial; long long s_partial;
} AbxIpaTail;
int abx_ipa_tail(const AbxIpaTail* desc, hipStream_t stream);
/* The per-residue heads on the final single representation in ONE launch (split-f16 arithmetic of AbxGemm, activations resident on the
 * CU): TorsionModule (abx/model/sidechain.py:28-62: proj_act + proj_init_act, two ResNet blocks, projection -> 14 unnormalised
 * sin / cos), SequenceHead.net (abx/model/head.py:143-163: LayerNorm, 256 -> 128 -> 128 -> 20) and, when W_p1 != NULL,
 * PredictedLDDTHead.net (head.py:206-226: LayerNorm, 256 -> 128 -> 128 -> 50).  Weights as abx_split_weights_f16 planes
 * [K/16][2][128][16] of the (K, 128) transposed weight with their exponents; the 14 / 20 / 50-column projections ZERO-PADDED to 128
 * columns (planes and [128] biases).  Thirteen (eighteen) abx_gemm / abx_layernorm calls otherwise. */
typedef struct AbxHeadsTail {
    const float* s; long long s_s;                 /* (M, 256) structure-module output, row stride in floats */
    const float* s0; long long s_s0;               /* (M, 256) initial single representation (init_seq_layer_norm output) */
    int M;
    const unsigned short* W_act; int e_act; const float* b_act;        /* torsion_module.proj_act.1       256 -> 128 */
    const unsigned short* W_init; int e_init; const float* b_init;     /* torsion_module.proj_init_act.1  256 -> 128 */
    const unsigned short* W_r0; int e_r0; const float* b_r0;           /* blocks.0.net.1 */
    const unsigned short* W_r1; int e_r1; const float* b_r1;           /* blocks.0.net.3 */
    const unsigned short* W_r2; int e_r2; const float* b_r2;           /* blocks.1.net.1 */
    const unsigned short* W_r3; int e_r3; const float* b_r3;           /* blocks.1.net.3 */
    const unsigned short* W_proj; int e_proj; const float* b_proj;     /* projection 128 -> 14 (padded) */
    float* un;                                                         /* (M, 14) out */
    const float* lns_w; const float* lns_b;                            /* sequence_module.net.0 (LayerNorm [256]) */
    const unsigned short* W_s1; int e_s1; const float* b_s1;           /* net.1 256 -> 128 */
    const unsigned short* W_s3; int e_s3; const float* b_s3;           /* net.3 128 -> 128 */
    const unsigned short* W_s5; int e_s5; const float* b_s5;           /* net.5 128 -> 20 (padded) */
    float* logits;                                                     /* (M, 20) out */
    const float* lnp_w; const float* lnp_b;                            /* predicted_lddt.net.0; the head is optional (W_p1 == NULL: skipped) */
    const unsigned short* W_p1; int e_p1; const float* b_p1;
    const unsigned short* W_p3; int e_p3; const float* b_p3;
    const unsigned short* W_p5; int e_p5; const float* b_p5;           /* 128 -> 50 (padded) */
    float* pl;                                                         /* (M, 50) out */
    float ln_eps;
    int* range_flag; int range_tag;                /* see AbxGemm.range_flag: set when an output value is not finite */
} AbxHeadsTail;
int abx_heads_tail(const AbxHeadsTail* desc, hipStream_t stream);
/* diagnostics: resident workgroups per CU of the main split-f16 GEMM instantiations (0: 128x192, 1: 128x128,
 * 2: 128x128 transposed store, 3: 128x192 plane operands); negative on error */
int abx_gemm3_occupancy(int which);

/* LayerNorm statistics (mean, rstd) per row for the LN-on-load GEMM prologue (torch.nn.LayerNorm, eps 1e-5).
 * s_k == 1: rows dense over batch (row stride s_row); else channel-major: element (b,row,k) at x + b*s_b + k*s_k + row. */
int abx_row_stats(const float* x, long long s_b, long long s_row, long long s_k, int batch, int rows, int K, float eps,
                  float* stats, hipStream_t stream);
/* materialised LayerNorm over contiguous rows, out = (res?) + LN(x)  (seqformer.py:216-221 prev_*_norm, score_network.py:119-133) */
int abx_layernorm(const float* x, long long s_row, long long rows, int K, const float* gamma, const float* beta, float eps,
                  float* out, long long s_out, const float* res, long long s_res, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Attention kernels
 * ---------------------------------------------------------------------------------------------------------- */
/* Flash-style triangle attention (seqformer.py:506-550 + Attention.forward split_first=True :272-312):
 * per (b, row s, head h): softmax_k( q.k * scale + bias[b,h,q,k] + keymask ) v, then * sigmoid(gate).
 * q,k,v,gate: element (b,s,l,h,d) at ptr + b*sb + s*ss + l*sl + h*D + d.  per_column orientation = swapped ss/sl. */
typedef struct AbxTriAttn {
    const float* q; const float* k; const float* v; const float* gate;
    long long sb, ss, sl;
    const float* bias; long long bias_sb, bias_sh, bias_sq, bias_sk;
    const float* keymask; long long km_sb;          /* float 0/1 [b*km_sb + k] or NULL */
    float* out; long long ob, os, ol;               /* out (b,s,l,h*D+d) */
    int B, S, L, H, D;                              /* D must be 48 */
    float scale;
    int exact;                                      /* 0: split-f16 matrix-core kernel (csrc/attention.hip tri_attn8_kernel; any L): keys / values
                                                       staged as two float16 planes of 16 x (|k|, |v| < 4095; 2^-29 absolute below 2^-6),
                                                       queries as two pieces of q * scale * log2(e) * 8 (|q * scale| < 5600), softmax weights
                                                       as two pieces of P * 2^8: 3 exact float16 products per fp32 product, fp32 accumulate;
                                                       an operand beyond its range gives NaN in its output rows, never a wrong number;
                                                       1: exact fp32 MFMA kernel (v_mfma_f32_16x16x4_f32; keys in chunks: any L) */
    unsigned long long* clock_probe;                /* diagnostics, optional DEVICE [2] (see AbxGemm.clock_probe; split-f16 kernels) */
    int* range_flag; int range_tag;                 /* see AbxGemm.range_flag: set when a query row's output is not finite (split-f16 kernels) */
    int tune;                                       /* 0 = library default (persistent workgroups, two query tiles per wave walked together);
                                                       bit 1: the other key-chunk size (128 <-> 192; bit-identical results); bit 2: the round-3
                                                       kernel tri_attn4 (one workgroup per row, one query tile at a time; with bit 0: without
                                                       its producer wave) - benchmarking and cross-checks: its accumulation order differs */
    int bias_log2;                                  /* 1: `bias` holds ABX_TRI_BIAS_LOG2 x the pair bias - the projection that wrote it applied the
                                                       factor (AbxGemm.alpha), so the split-f16 kernel adds it to its accumulators as it is: the
                                                       same bits as the multiplication in the kernel (one rounding of the same product), 32 vector
                                                       instructions less per pair of query tiles and key tile.  abx_tri_attn_block_fwd does this. */
    int kv_planes;                                  /* 1: k and v do not point to fp32 rows but to the operand images AbxGemm.c_planes_from writes: per
                                                       key and head [plane p0: 48 float16 | plane p1: 48 float16] of 16 x value at the byte address
                                                       the fp32 head slice would have (192 bytes either way: same pointers, same strides).  Split-f16
                                                       kernel tri_attn8 only (exact = 0, tune = 0): its producer wave turns into a DMA issuer
                                                       (global_load_lds straight into the chunk buffers: no registers, no split, no LDS writes) */
    int q_parts, row_groups;                        /* filled by the library */
} AbxTriAttn;
/* float(log2 e) * 2^7: base-2 logits in the accumulator units of tri_attn8_kernel */
#define ABX_TRI_BIAS_LOG2 (1.4426950408889634f * 128.0f)
int abx_tri_attn_fwd(const AbxTriAttn* desc, hipStream_t stream);

/* Sequence attention with pair bias (seqformer.py:314-356 + Attention.forward split_first=False :278-312).
 * qkv: [B*L][H*3*D] with per-head layout [q D | k D | v D]; bias [b][h][q][k]; gate pre-activation [B*L][H*D];
 * out [B*L][H*D] = softmax(...) v * sigmoid(gate).  D <= 32. */
int abx_seq_attn_fwd(const float* qkv, const float* bias, const float* keymask, const float* gate, float* out, int B, int L,
                     int H, int D, float scale, hipStream_t stream);

/* Invariant Point Attention core (folding.py:47-132).  abx_ipa_pack turns the fused projection rows
 * [q_scalar 192 | kv_scalar 384 | q_point_local 144 | kv_point_local 432] into global-frame packs
 * (r3.rigids_apply, r3.py:9-16); abx_ipa_attn does logits (scalar + point distance + pair bias), mask, softmax_j and the
 * three outputs (scalar, points back in the local frame + norms, attention over the pair slab), writing the
 * 2112-wide concat [scalar | points (r n) | norms | pair] that final_proj consumes.  Two launches: the attention weights
 * (kept in attn_ws [B][L][3 head groups][L][4] fp32, abx_ipa_attn_workspace_bytes) with the scalar / point outputs, then the stream over
 * the pair slab z [B][L][L][128]. */
int abx_ipa_pack(const float* proj, const float* rots, const float* trans, float* qpack, float* kpack, float* vpack,
                 int B, int L, float scalar_weight, hipStream_t stream);
int abx_ipa_attn(const float* qpack, const float* kpack, const float* vpack, const float* bias2d, const float* z,
                 const float* mask, const float* rots, const float* trans, const float* point_weights /* [12] */,
                 float* attn_ws, float* feat, int B, int L, hipStream_t stream);
/* the two launches of abx_ipa_attn, separately callable (profiling; feat rows: weights writes [0, 576), pair writes [576, 2112)) */
int abx_ipa_weights(const float* qpack, const float* kpack, const float* vpack, const float* bias2d, const float* mask,
                    const float* rots, const float* trans, const float* point_weights, float* attn_ws, float* feat, int B, int L,
                    hipStream_t stream);
int abx_ipa_pair(const float* attn_ws, const float* z, float* feat, int B, int L, hipStream_t stream);
long long abx_ipa_attn_workspace_bytes(int B, int L);
long long abx_ipa_qpack_bytes(int B, int L);   /* size of qpack: query rows padded to blocks of 12, pairs interleaved */

/* ------------------------------------------------------------------------------------------------------------
 * Embedding assembly (seqformer.py:49-119,170-223) and small pair-stack helpers
 * ---------------------------------------------------------------------------------------------------------- */
/* sinusoidal embedding of t*10000 (double product, then float) -> [B][dim]; freqs [dim/2] = exp(arange * -ln(1e4)/(dim/2-1)) */
int abx_timestep_embedding(const double* t, const float* freqs, int B, int dim, float* out, hipStream_t stream);
/* seq_act[b,l] = [ seq_static[b,l] (+ aa_table[seq_t[b,l]] for l < Lab) | temb[b] ] + LN(prev_seq[b,l]) */
int abx_assemble_seq(const float* seq_static, long long ss_b, const float* aa_table, const long long* seq_t, int Lab,
                     const float* temb, const float* prev_seq, const float* gamma, const float* beta, float* out, int B,
                     int L, int C, int E, hipStream_t stream);
/* pair_act[b,i,j] = [ pair_static[b,i,j] | temb[b] | temb[b] ] + LN(prev_pair[b,i,j]) + pos_table[prev_pos[b,i,j]];
 * stats_out (optional): LayerNorm (mean, rstd) of every assembled row for the first consumer */
int abx_assemble_pair(const float* pair_static, long long ps_b, const float* temb, const float* prev_pair,
                      const float* gamma, const float* beta, const long long* prev_pos, const float* pos_table, float* out,
                      float* stats_out, int B, int L, int C, int E, hipStream_t stream);
/* OuterProductMean features (seqformer.py:400-409): feat[b,i,j] = [ left[b,j]*right[b,i] | left[b,j]-right[b,i] ];
 * left/right rows have stride ld floats */
/* Pair-representation assembly AND the sequence attention's pair bias in one pass over the pair rows (round 6; seqformer.py:193-223 + :324-333):
 * out = z0 as abx_assemble_pair writes it (C = 128, E = 32: 192 channels) and biasT[b][h][i*L + j] = Linear(LayerNorm(z0[b,i,j,:]))[h] for 32 heads -
 * what abx_gemm computes from z0 with a folded LayerNorm (w_planes = abx_split_weights_f16 planes [12][2][32][16] of the gamma-scaled weight with
 * exponent w_exp, csum its column sums, bias the folded bias; split-f16 arithmetic, range_tag as AbxGemm.range_flag).  One workgroup per (b, i)
 * row of the pair tensor; the 9.5 GB second read of z0 by a projection launch of its own is gone. */
int abx_assemble_pair_bias(const float* pair_static, long long ps_b, const float* temb, const float* prev_pair, const float* gamma,
                           const float* beta, const long long* prev_pos, const float* pos_table, float* out, const unsigned short* w_planes,
                           int w_exp, const float* csum, const float* bias, float ln_eps, float* biasT, int B, int L, int* range_flag,
                           int range_tag, hipStream_t stream);
int abx_opm_features(const float* left, const float* right, long long ld, float* feat, int B, int L, int C,
                     hipStream_t stream);
/* OuterProductMean without its feature tensor (seqformer.py:395-411; round 6): z[b,i,j,:] += out_proj([l_j * r_i | l_j - r_i]) evaluated as
 * l_j . (diag(r_i) W1 + W2) + (bias - r_i . W2) - one workgroup per (b, i) row of the pair tensor builds the 64 x 192 operand image of its row
 * in LDS and streams the L positions j; HBM traffic = z read + z written (the (B, L, L, 128) feature tensor of abx_opm_features + the K = 128
 * abx_gemm over it: 31.7 GB per call at 100 samples of L = 352 instead of 19 GB).  lr: [B*L][ld] rows [left 64 | right 64] (already masked),
 * Wt: out_proj.weight^T [128][192] fp32 (rows 0..63 against the products, 64..127 against the differences), bias [192] or NULL, z updated
 * in place.  Split-f16 arithmetic (3 exact products per fp32 product, fp32 accumulate; |diag(r) W1 + W2| < 4095, else NaN and range_tag is
 * OR-ed into *range_flag - see AbxGemm.range_flag; the caller's exact route is abx_opm_features + abx_gemm with exact = 1). */
int abx_opm_out_fwd(const float* lr, long long ld, const float* Wt, const float* bias, float* z, int B, int L, int* range_flag, int range_tag,
                    hipStream_t stream);
/* out[n][a][b] = in[n][b][a] (transpose != 0) or in[n][a][b] for nmat L x L matrices, output rows padded to Lp >= L floats (pad
 * columns = 0).  The triangle attention reads its (4, L, L) pair bias key-contiguously in rows of Lp % 4 == 0 floats (16-byte
 * loads for any residue count); the ending-node orientation reads it transposed ('b i j c -> b j i c', seqformer.py:536) */
int abx_transpose_last2(const float* in, float* out, int nmat, int L, int Lp, int transpose, hipStream_t stream);
/* pair mask[b,i,j] = mask[b,i]*mask[b,j], rows of Lp >= L entries (columns j >= L are written as 0: the padded pair rows of
 * AbxGemm.pair_Lp) */
int abx_pair_mask(const float* mask, float* out, int B, int L, int Lp, hipStream_t stream);

/* Trajectory-invariant encoders (encoder.py:123-269, seqformer.py:177-206): gather/concat stages; the MLPs run on abx_gemm. */
int abx_pair_embed_features(const long long* aa, const int* chain_id, const int* residx, const float* atom14,
                            const unsigned char* atom14_exists, const float* aa_pair_embed, const float* relpos_embed,
                            const float* distcoef, const float* dgram_embed, const float* sq_breaks /* [14] */,
                            float* feat512, float* dist196, int B, int L, hipStream_t stream);
int abx_relpos_block(const int* residx, const float* table, float* out, int B, int L, int Lab, int C, int max_rel,
                     hipStream_t stream);
int abx_gather_rows(const float* table, const long long* idx, const float* rowscale, float* out, long long s_out, long long n,
                    int C, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Rigid frames, torsions, heads (score_network.py:100-194, quat_affine.py, r3.py, atom.py, sidechain.py:64-90, head.py:162-226)
 * ---------------------------------------------------------------------------------------------------------- */
/* rigids_t (B,L,7) f32 or f64 -> init_q, init_t (unscaled), cur_q, cur_t (= init_t/position_scale), cur_R, delta_q = identity */
int abx_frames_init(const void* rigids_t, int is_f64, float* init_q, float* init_t, float* cur_q, float* cur_t, float* cur_R,
                    float* delta_q, int n, float position_scale, hipStream_t stream);
/* one IPA layer tail: quat_precompose_vec on delta and current, translation update with the OLD rotation, fixed-mask restore,
 * quat_to_rot (score_network.py:137-149) */
int abx_rigid_update(const float* upd6, const int* fixed_mask, const float* init_q, const float* init_t, float* cur_q,
                     float* cur_t, float* cur_R, float* delta_q, int n, float position_scale, hipStream_t stream);
/* final quaternion, igso3 rot score (cached table lookup, so3_diffuser.py:264-297), r3 trans score (r3_diffuser.py:158-164),
 * rigids (B,L,7).  t is (B,) double; t_is_f32 selects the reference's fp32 arithmetic of the warm-up call.
 * trans_score is written as double when !t_is_f32 else float. */
typedef struct AbxScoreArgs {
    const float* init_q; const float* init_t; const float* delta_q; const float* cur_t; const int* fixed_mask;
    const double* t; int t_is_f32;
    const float* score_norms; int num_sigma, num_omega; const float* discrete_sigma; const float* discrete_omega;
    float exp_max_sigma, exp_min_sigma;     /* fp32 exp(1.5), exp(0.1) as torch computes them */
    float min_b, bdiff;                     /* fp32 0.1, 19.9 */
    float coord_scale;                      /* fp32 0.1 */
    float position_scale;
    float* rot_score; void* trans_score; float* rigids;
    int B, L;
} AbxScoreArgs;
int abx_scores(const AbxScoreArgs* a, hipStream_t stream);
/* torsion head tail: l2-normalise (eps 1e-12) and take ground-truth torsions at fixed residues (sidechain.py:67-72) */
int abx_torsion_finalize(const float* unnorm, const float* gt_sincos, const int* fixed_mask, float* angles, int n,
                         hipStream_t stream);
/* SequenceHead tail (head.py:165-199): seq_0 = argmax(logits) merged with seq_t at fixed positions; frames from torsions;
 * atom14 and atom37 with the seq_0 residue tables. tables: default_frames (21,8,4,4), group_idx (21,14) int, lit_pos (21,14,3) */
int abx_seq_head_atoms(const float* logits, const int* fixed_mask, const long long* seq_t, const float* rigids,
                       const float* angles, const long long* atom37_to_atom14, const float* default_frames,
                       const int* group_idx, const float* lit_pos, long long* seq_0, float* atom14, float* atom37, int n,
                       hipStream_t stream);
/* get_prev (abx.py:17-26): virtual C-beta from N,CA,C (common_modules.py:62-83) and 15-bin distogram -> int64 (B,L,L) */
int abx_prev_pos(const float* atom37, const float* sq_breaks, int num_breaks, long long* prev_pos, int B, int L,
                 hipStream_t stream);
/* pLDDT = 100 * sum softmax(logits) * bin centres (utils.py:158-171) */
int abx_plddt(const float* logits, float* out, int n, int bins, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Diffuser (diffuser/so3_diffuser.py, r3_diffuser.py, discrete_diffuser.py, full_diffuser.py:174-227)
 * ---------------------------------------------------------------------------------------------------------- */
/* IGSO(3) series tables (so3_diffuser.py:15-49,72-112,153-166): pdf, cdf, score_norms [num_sigma][num_omega] */
int abx_igso3_tables(const float* sigma, const float* omega, int num_sigma, int num_omega, int L_terms, float* pdf, float* cdf,
                     float* score_norms, hipStream_t stream);
/* One reverse step of all three processes in float64 with injected or device-generated noise, mask merge after all three.
 * rigid_in f32 or f64 (B,L,7); rot_score f32; trans_score f64 (or f32 when ts_is_f32); logits f32 (B,L,20); t (B,) double.
 * Noise: z_rot,z_trans f32 (B,L,3), jumps f32 (B,L,20) when given; otherwise Philox4x32-10 keyed by (seed, sample id, residue, step).
 * The Poisson jump counts (discrete_diffuser.py:182-183) are a pure function of one uniform each: the inverse cdf of
 * Poisson(rate*dt) (fp32 pmf recurrence from (float)exp(-(double)lam), see diffuser.hip::poisson_icdf) evaluated at
 * u_jumps[b,l,s] in (0,1) when given (jumps == NULL), else at the Philox uniform of (sample id, residue, step, draw s).
 * Outputs rigid_out f64 (B,L,7), seq_out int64 (B,L).  Optional: rates_out (B,L,20) = the Poisson rates * dt,
 * jumps_out (B,L,20) = the jump counts that were applied. */
typedef struct AbxReverseArgs {
    const void* rigid_in; int rigid_is_f64;
    const long long* seq_in;
    const float* rot_score; const void* trans_score; int ts_is_f32; const float* logits;
    const int* diffuse_mask; const double* t; float dt;
    const float* z_rot; const float* z_trans; const float* jumps;
    const float* u_jumps;                    /* optional (B,L,20) uniforms in (0,1) driving the Poisson inverse cdf (jumps == NULL) */
    const float* dt_dev;                     /* optional DEVICE float: overrides `dt` (the reference's loop passes dt as a 0-dim
                                                device tensor, inference.py:198-199: no device-to-host read on the step path) */
    unsigned long long seed; const long long* sample_ids; int step;
    const int* step_dev;                     /* optional DEVICE int: overrides `step` (a hipGraph-captured reverse step reads the
                                                current step index at replay time instead of a value frozen at capture) */
    float exp_max_sigma, exp_min_sigma, min_b, bdiff, coord_scale, rate_const;
    float noise_scale; int center;
    double* rigid_out; long long* seq_out; float* rates_out; float* jumps_out;
    int B, L;
} AbxReverseArgs;
int abx_reverse_step(const AbxReverseArgs* a, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Guidance terms (north_star "guidance pairwise-distance / clash terms").  NOT in the reference's sampler
 * (its loop runs under no_grad): an opt-in extension built from the violation material the reference ships -
 * van-der-Waals radii / overlap tolerance (abx/common/residue_constants.py:381-386,483-525, config/config_model.json:116,213-214)
 * and the C-N peptide-bond term of eval/metric_scripts/cal_vio.py:29-74.  Per sample:
 *   energy[b] = { w_clash * sum_pairs w_ij relu(r_a + r_b - overlap_tolerance - d_ab),
 *                 w_bond  * sum_i relu(|d(C_i,N_i+1) - l0| - tol * sigma),
 *                 w_angle * sum_i [ relu(|cos(CA_i,C_i,N_i+1) - c1| - tol * s1) + relu(|cos(C_i,N_i+1,CA_i+1) - c2| - tol * s2) ] }
 *   (|.| smoothed as sqrt(1e-6 + .^2); l0 / sigma proline-aware; c1 = -0.4473 +- 0.0311, c2 = -0.5203 +- 0.0353: cal_vio.py:76-105,
 *   abx/common/residue_constants.py:475-480)
 *   grad_atom = dE/dx (B,L,14,3);  grad_trans = sum_a g_a;  grad_rot = sum_a (x_a - frame_trans) x g_a   (B,L,3) each.
 * Pairs of one residue, the peptide bond C(i)-N(i+1) of linked neighbours and SG-SG pairs are excluded from the clash term; w_ij =
 * between_chain_factor for atoms of different chains.  Residues i, i+1 are linked when they share a chain id and, if residx is
 * given, residx[i+1] == residx[i] + 1 (residx == NULL: the chain-only rule of cal_vio.py:51).
 * radius: [21][14] van-der-Waals radius of every atom14 slot (0 for empty slots).
 * The caller allocates the workspace (abx_clash_grad_workspace_bytes). */
typedef struct AbxGuidanceArgs {
    const float* atom14; const unsigned char* atom_mask; const long long* aatype; const int* chain_id;
    const int* residx;                              /* optional residue numbers (B,L) */
    const float* radius; const float* frame_trans;
    float overlap_tolerance, between_chain_factor, bond_tolerance_factor, w_clash, w_bond, w_angle;
    float* energy;                                  /* (B,3): clash, bond, angle */
    float* grad_atom; float* grad_trans; float* grad_rot;
    int B, L;
} AbxGuidanceArgs;
long long abx_clash_grad_workspace_bytes(int B, int L);
int abx_clash_grad(const AbxGuidanceArgs* a, void* workspace, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Op-group entry points: one call per reference module of the pair stack, for a maintainer who binds
 * abx/model/seqformer.py without the Python orchestration of abx_amd/model/forward.py.  Each is a fixed sequence of the launches above
 * (abx_gemm descriptors filled here exactly as forward.py fills them; same kernels, same bits), asynchronous on the stream, no
 * allocation: weights come as packs built once by abx_pack_linear into caller memory, scratch as a caller workspace
 * (abx_*_workspace_bytes).  exact = 0: the split-f16 kernels (needs L >= 64 and B * L * L >= 32768 rows: what AbxGemm.exact = 2 serves);
 * exact = 1: the exact fp32-MFMA kernels (any size; triangle attention: L <= 389).  range_flag / range_tag: see AbxGemm.range_flag
 * (every launch of a group ORs the same tag).
 * ---------------------------------------------------------------------------------------------------------- */
/* One Linear (optionally with the preceding LayerNorm folded in), packed for both arithmetic paths.  All pointers point into the
 * caller's buffer handed to abx_pack_linear. */
typedef struct AbxLinearPack {
    const float* Wt;                    /* [K][N] fp32, rows scaled by the LayerNorm gamma when folded */
    const float* csum;                  /* [N] column sums of Wt (folded LayerNorm) or NULL */
    const float* bias;                  /* [N]: beta @ W^T + b (folded) / b / NULL */
    const unsigned short* planes;       /* abx_split_weights_f16 image [Kp/16][2][N][16] of Wt (of its k-permuted copy with ABX_PACK_PERMUTE_K16) */
    int b_exp;                          /* exponent of the planes */
    int K, N;
} AbxLinearPack;
/* a group of weight rows that becomes output columns: W [rows][K] (torch Linear.weight), b [rows] or NULL; glu: 0 = the sources
 * are concatenated in order; 1 / 2 = value / gate rows of a gated projection: channel c of the concatenated value (gate) sources
 * lands in column (c / 32) * 64 + c % 32 (+ 32), the (value, gate) column pairs of AbxGemm.glu */
typedef struct AbxLinearSrc { const float* W; const float* b; int rows; int glu; } AbxLinearSrc;
#define ABX_PACK_PERMUTE_K16 1          /* planes in the k order of the fused transition's second layer (0-3, 8-11, 4-7, 12-15 per 16) */
long long abx_pack_linear_bytes(int K, int N);
/* gamma / beta: [K] of the LayerNorm to fold, or NULL.  Synchronises the stream once (the plane exponent needs max |w| on the host):
 * packing is set-up work, outside any graph capture.  buf: 256-byte aligned device memory of abx_pack_linear_bytes(K, N). */
int abx_pack_linear(const AbxLinearSrc* src, int nsrc, int K, const float* gamma, const float* beta, int flags, void* buf,
                    AbxLinearPack* out, hipStream_t stream);

/* pair Transition (seqformer.py:358-376): z <- z + W2 relu(W1 LN(z) + b1) + b2 on rows [M][C], in place.
 * l1: pack of transition.1 with transition.0 (LayerNorm) folded; l2: pack of transition.3 with ABX_PACK_PERMUTE_K16.
 * exact = 0: ONE kernel (AbxGemm.mlp; C <= 192, no workspace); exact = 1: two GEMMs, workspace = M * l1.N floats. */
long long abx_transition_workspace_bytes(long long M, int hidden, int exact);
int abx_transition_fwd(const AbxLinearPack* l1, const AbxLinearPack* l2, float* z, long long M, int exact, void* workspace,
                       int* range_flag, int range_tag, hipStream_t stream);

/* TriangleMultiplication (seqformer.py:443-504), outgoing ('bikc,bjkc->bijc') or incoming ('bkic,bkjc->bijc').
 * glu: [left_proj | right_proj] (glu = 1) and [left_gate | right_gate] (glu = 2) with `norm` folded; out: proj_out with
 * `final_norm` folded; gate: final_gate with `norm` folded.  z_in (B, L*L, 192) -> z_out (B, L*L, 192), z_out != z_in (the tail
 * reads all channels of a row of z_in while other tiles write theirs); mask (B, L) float 0/1.  Split-f16 path only (exact = 0):
 * three launches - gated projections written as f16 operand images, plane x plane contraction, output projection * final gate +
 * residual.  The image region of the workspace must read zero where the kernels never write (pad k-tiles when ceil4(L) % 16 != 0):
 * abx_tri_mul_workspace_init once per (workspace, B, L). */
typedef struct AbxTriMulPack { AbxLinearPack glu, out, gate; } AbxTriMulPack;
long long abx_tri_mul_workspace_bytes(int B, int L);
int abx_tri_mul_workspace_init(void* workspace, int B, int L, hipStream_t stream);
int abx_tri_mul_fwd(const AbxTriMulPack* w, const float* z_in, float* z_out, const float* mask, int B, int L, int outgoing,
                    void* workspace, int* range_flag, int range_tag, hipStream_t stream);

/* TriangleAttention block (seqformer.py:506-550 + Attention.forward :272-312), starting node (per_row = 1) or ending node:
 * z <- z + proj_out(sigmoid(gate(LN z)) * attention(LN(z))) in place on (B, L*L, 192).  qkv: [proj_q | proj_k | proj_v] with `norm` folded;
 * gate: attn.gate with `norm` folded; pair: proj_pair (192 -> 4 heads) with `norm` folded; out: attn.proj_out packed with
 * ABX_PACK_PERMUTE_K16.  Launches (exact = 0): q | k | v GEMM with the pair bias in its grid (abx_gemm_side), bias transpose / pad (ending
 * node or L % 4 != 0), abx_tri_attn_fwd without a gate, then ONE gated tail (AbxGemm.mlp = 2): the gate projection, sigmoid, the product
 * with the attention output, the output projection and the residual - the gate never exists in memory.  With exact GEMMs the tail is two
 * launches (gate projection * attention output into the workspace, output projection + residual).  exact here is two bits: bit 0 the
 * GEMMs, bit 1 the attention kernel (a small complex runs exact GEMMs - too few rows for the split tiles - with the split-f16
 * attention, which has no size limit: exact = 1; everything exact: exact = 3). */
typedef struct AbxTriAttnPack { AbxLinearPack qkv, gate, pair, out; } AbxTriAttnPack;
long long abx_tri_attn_block_workspace_bytes(int B, int L);
int abx_tri_attn_block_fwd(const AbxTriAttnPack* w, float* z, const float* mask, int B, int L, int per_row, int exact, void* workspace,
                           int* range_flag, int range_tag, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ABX_HIP_H */
