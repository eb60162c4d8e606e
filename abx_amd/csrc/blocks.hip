// Op-group entry points of the C ABI (include/abx_hip.h, "Op-group entry points"): one call per reference module of the pair stack
// (pair Transition seqformer.py:358-376, TriangleMultiplication :443-504, TriangleAttention :506-550), each a fixed sequence of the
// launches of gemm3.hip / attention.hip / embed.hip with the descriptors filled exactly as abx_amd/model/forward.py fills them, and the
// weight packing those launches need (LayerNorm fold, (value, gate) column pairs, k-permuted planes) as abx_pack_linear.
// No allocation, no synchronisation in the forward calls; abx_pack_linear synchronises once (set-up work).
#include <math.h>
#include <string.h>

#include <stdlib.h>

#include "common.h"
#include "abx_hip.h"

namespace {

constexpr long long ALIGN = 256;
inline long long up(long long x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

// rows of one source -> rows of the concatenated [N][K] matrix (glu: (value, gate) column pairs in blocks of 32 channels)
__global__ void pack_rows_kernel(const float* __restrict__ W, const float* __restrict__ b, int rows, int K, int c0, int glu,
                                 float* __restrict__ Wcat, float* __restrict__ bcat) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * K) return;
    const int r = (int)(idx / K), k = (int)(idx - (long long)r * K);
    const int c = c0 + r;
    const int dst = glu ? (c / 32) * 64 + (glu == 2 ? 32 : 0) + c % 32 : c;
    Wcat[(long long)dst * K + k] = W[idx];
    if (k == 0) bcat[dst] = b ? b[r] : 0.f;
}

// LayerNorm folded into the Linear (float64, like the host packing of forward.py): Wt[k][n] = gamma[k] W[n][k], csum[n] = sum_k of
// those, bias[n] = sum_k beta[k] W[n][k] + b[n]; without gamma: the plain transpose.  One thread per output column.  amax: bit
// pattern of max |Wt| (non-negative floats order like unsigned integers); +inf when any weight is not finite.
__global__ void fold_kernel(const float* __restrict__ Wcat, const float* __restrict__ bcat, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int N, int K, float* __restrict__ Wt, float* __restrict__ csum,
                            float* __restrict__ bias, unsigned* __restrict__ amax) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double cs = 0.0, bs = 0.0;
    float mx = 0.f;
    bool bad = false;
    for (int k = 0; k < K; ++k) {
        const double w = (double)Wcat[(long long)n * K + k];
        const double ws = gamma ? (double)gamma[k] * w : w;
        const float wf = (float)ws;
        Wt[(long long)k * N + n] = wf;
        cs += ws;
        if (beta) bs += (double)beta[k] * w;
        mx = fmaxf(mx, fabsf(wf));
        bad |= !(fabsf(wf) <= 3.0e38f);
    }
    csum[n] = (float)cs;
    bias[n] = (float)(bs + (double)bcat[n]);
    atomicMax(amax, bad ? 0x7f800000u : __float_as_uint(mx));
}

// rows of every 16-block reordered 0-3, 8-11, 4-7, 12-15 (AbxGemm.mlp feeds its second GEMM from the accumulators of the first)
__global__ void permute_k16_kernel(const float* __restrict__ Wt, int K, int N, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)K * N) return;
    const int k = (int)(idx / N), n = (int)(idx - (long long)k * N);
    const int j = k & 15;
    const int src = (k & ~15) | ((j & 3) | ((j & 4) << 1) | ((j & 8) >> 1));        // swaps bits 2 and 3 of the position
    out[idx] = Wt[(long long)src * N + n];
}

// ---- descriptor helpers: what abx_amd/ops.py gemm() fills ------------------------------------------------------------------------
inline void set_weights(AbxGemm& g, const AbxLinearPack& p, bool ln, int exact) {
    g.B = p.Wt; g.sBb = 0; g.sBk = p.N; g.sBn = 1;
    g.N = p.N; g.K = p.K;
    g.bias = p.bias;
    if (ln) { g.ln_csum = p.csum; g.ln_eps = 1e-5f; }
    g.B_split = p.planes; g.sB3k = 2LL * p.N * 16; g.sB3p = (long long)p.N * 16; g.sB3n = 16; g.sB3b = 0;
    g.b_f16 = 1; g.b_exp = p.b_exp;
    g.exact = exact ? 1 : 2;
    g.alpha = 1.0f;
}
inline void set_range(AbxGemm& g, int* flag, int tag, int exact) {
    if (!exact) { g.range_flag = flag; g.range_tag = tag; }
}

}  // namespace

extern "C" long long abx_pack_linear_bytes(int K, int N) {
    if (K <= 0 || N <= 0) return -1;
    const long long Kp = (K + 15) / 16 * 16;
    return up(4LL * K * N) + 2 * up(4LL * N) + up(2LL * Kp * 2 * N) + up(4LL * K * N) + up(4LL * N) + up(4);
}

extern "C" int abx_pack_linear(const AbxLinearSrc* src, int nsrc, int K, const float* gamma, const float* beta, int flags, void* buf,
                               AbxLinearPack* out, hipStream_t st) {
    ABX_REQUIRE(src && nsrc > 0 && nsrc <= 8 && K > 0 && buf && out, "abx_pack_linear: bad args");
    ABX_REQUIRE((reinterpret_cast<uintptr_t>(buf) & (ALIGN - 1)) == 0, "abx_pack_linear: buf must be 256-byte aligned");
    ABX_REQUIRE((gamma == nullptr) == (beta == nullptr), "abx_pack_linear: gamma and beta come together");
    int N = 0, nv = 0, ng = 0;
    bool any_b = false;
    for (int i = 0; i < nsrc; ++i) {
        ABX_REQUIRE(src[i].W && src[i].rows > 0 && src[i].glu >= 0 && src[i].glu <= 2, "abx_pack_linear: bad source");
        N += src[i].rows;
        nv += src[i].glu == 1 ? src[i].rows : 0;
        ng += src[i].glu == 2 ? src[i].rows : 0;
        any_b |= src[i].b != nullptr;
    }
    ABX_REQUIRE((nv == 0 && ng == 0) || (nv == ng && nv + ng == N && nv % 32 == 0),
                "abx_pack_linear: glu sources need as many value as gate rows, in multiples of 32, and nothing else");
    ABX_REQUIRE(!(flags & ABX_PACK_PERMUTE_K16) || K % 16 == 0, "abx_pack_linear: ABX_PACK_PERMUTE_K16 needs K % 16 == 0");
    const long long Kp = (K + 15) / 16 * 16;
    char* p = static_cast<char*>(buf);
    float* Wt = reinterpret_cast<float*>(p); p += up(4LL * K * N);
    float* csum = reinterpret_cast<float*>(p); p += up(4LL * N);
    float* bias = reinterpret_cast<float*>(p); p += up(4LL * N);
    unsigned short* planes = reinterpret_cast<unsigned short*>(p); p += up(2LL * Kp * 2 * N);
    float* Wcat = reinterpret_cast<float*>(p); p += up(4LL * K * N);
    float* bcat = reinterpret_cast<float*>(p); p += up(4LL * N);
    unsigned* amax = reinterpret_cast<unsigned*>(p);
    int cv = 0, cg = 0, c0 = 0;
    for (int i = 0; i < nsrc; ++i) {
        const long long tot = (long long)src[i].rows * K;
        const int base = src[i].glu == 1 ? cv : (src[i].glu == 2 ? cg : c0);
        hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src[i].W, src[i].b, src[i].rows, K, base,
                           src[i].glu, Wcat, bcat);
        (src[i].glu == 1 ? cv : (src[i].glu == 2 ? cg : c0)) += src[i].rows;
    }
    if (hipMemsetAsync(amax, 0, 4, st) != hipSuccess) { abx_set_error("abx_pack_linear: memset failed"); return ABX_ERR_ARG; }
    hipLaunchKernelGGL(fold_kernel, dim3((N + 63) / 64), dim3(64), 0, st, Wcat, bcat, gamma, beta, N, K, Wt, csum, bias, amax);
    if (int rc = abx_check_launch("abx_pack_linear")) return rc;
    unsigned bits = 0;
    if (hipMemcpyAsync(&bits, amax, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        abx_set_error("abx_pack_linear: reading max |w| failed");
        return ABX_ERR_ARG;
    }
    float mx;
    memcpy(&mx, &bits, 4);
    ABX_REQUIRE(mx <= 3.0e38f, "abx_pack_linear: non-finite weight");
    int e = 0;
    if (mx > 0.f) frexpf(mx, &e);
    int w_exp = mx > 0.f ? 14 - e : 0;
    w_exp = w_exp < -100 ? -100 : (w_exp > 100 ? 100 : w_exp);
    const float* plane_src = Wt;
    if (flags & ABX_PACK_PERMUTE_K16) {
        const long long tot = (long long)K * N;
        hipLaunchKernelGGL(permute_k16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Wt, K, N, Wcat);   // (Wcat is free again)
        plane_src = Wcat;
    }
    if (int rc = abx_split_weights_f16(plane_src, 1, N, N, K, w_exp, planes, st)) return rc;
    out->Wt = Wt;
    out->csum = gamma ? csum : nullptr;
    out->bias = (gamma || any_b) ? bias : nullptr;
    out->planes = planes;
    out->b_exp = w_exp;
    out->K = K;
    out->N = N;
    return ABX_OK;
}

// ---- pair Transition --------------------------------------------------------------------------------------------------------------
extern "C" long long abx_transition_workspace_bytes(long long M, int hidden, int exact) {
    return exact ? up(4LL * M * hidden) : 0;
}

extern "C" int abx_transition_fwd(const AbxLinearPack* l1, const AbxLinearPack* l2, float* z, long long M, int exact, void* workspace,
                                  int* range_flag, int range_tag, hipStream_t st) {
    ABX_REQUIRE(l1 && l2 && z && M > 0 && M < (1LL << 31), "abx_transition_fwd: bad args");
    ABX_REQUIRE(l1->csum && l1->N == l2->K && l2->N == l1->K, "abx_transition_fwd: l1 = LayerNorm-folded C -> hidden, l2 = hidden -> C");
    const int C = l1->K, Hd = l1->N;
    if (!exact) {
        AbxGemm g = {};
        g.A = z; g.sAm = C; g.sAk = 1;
        g.C = z; g.sCm = C;
        g.M = (int)M; g.batch = 1;
        set_weights(g, *l1, true, 0);
        g.act = 1;
        g.mlp = 1; g.N2 = C; g.b2_exp = l2->b_exp;
        g.B2_split = l2->planes; g.sB23k = 2LL * C * 16; g.sB23p = (long long)C * 16; g.sB23n = 16;
        g.bias2 = l2->bias;
        g.resid = z; g.sRm = C;
        set_range(g, range_flag, range_tag, 0);
        return abx_gemm(&g, st);
    }
    ABX_REQUIRE(workspace, "abx_transition_fwd: the exact path needs its workspace (abx_transition_workspace_bytes)");
    float* hid = static_cast<float*>(workspace);
    AbxGemm g = {};
    g.A = z; g.sAm = C; g.sAk = 1;
    g.C = hid; g.sCm = Hd;
    g.M = (int)M; g.batch = 1;
    set_weights(g, *l1, true, 1);
    g.act = 1;
    if (int rc = abx_gemm(&g, st)) return rc;
    AbxGemm h = {};
    h.A = hid; h.sAm = Hd; h.sAk = 1;
    h.C = z; h.sCm = C;
    h.M = (int)M; h.batch = 1;
    set_weights(h, *l2, false, 1);
    h.B_split = nullptr; h.b_f16 = 0;                      // (l2's planes are k-permuted: not an operand of a plain GEMM)
    h.resid = z; h.sRm = C;
    return abx_gemm(&h, st);
}

// ---- TriangleMultiplication ---------------------------------------------------------------------------------------------------------
namespace {
struct TriMulWs { unsigned short* lrp; float* pm; float* tt; long long lrp_bytes; };
inline TriMulWs tri_mul_ws(void* ws, int B, int L) {
    const long long Lp = (L + 3) / 4 * 4, KT = (Lp + 15) / 16;
    TriMulWs w;
    char* p = static_cast<char*>(ws);
    w.lrp_bytes = 2LL * B * 256 * KT * 2 * L * 16;
    w.lrp = reinterpret_cast<unsigned short*>(p); p += up(w.lrp_bytes);
    w.pm = reinterpret_cast<float*>(p); p += up(4LL * B * L * Lp);
    w.tt = reinterpret_cast<float*>(p);
    return w;
}
}  // namespace

extern "C" long long abx_tri_mul_workspace_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return -1;
    const long long Lp = (L + 3) / 4 * 4, KT = (Lp + 15) / 16;
    return up(2LL * B * 256 * KT * 2 * L * 16) + up(4LL * B * L * Lp) + up(4LL * B * 128 * L * Lp);
}

extern "C" int abx_tri_mul_workspace_init(void* workspace, int B, int L, hipStream_t st) {
    ABX_REQUIRE(workspace && B > 0 && L > 0, "abx_tri_mul_workspace_init: bad args");
    const TriMulWs w = tri_mul_ws(workspace, B, L);
    if (hipMemsetAsync(w.lrp, 0, (size_t)w.lrp_bytes, st) != hipSuccess) { abx_set_error("abx_tri_mul_workspace_init: memset failed"); return ABX_ERR_ARG; }
    return ABX_OK;
}

extern "C" int abx_tri_mul_fwd(const AbxTriMulPack* wp, const float* z_in, float* z_out, const float* mask, int B, int L, int outgoing,
                               void* workspace, int* range_flag, int range_tag, hipStream_t st) {
    ABX_REQUIRE(wp && z_in && z_out && z_in != z_out && mask && workspace && B > 0 && L > 0, "abx_tri_mul_fwd: bad args (z_out must not alias z_in)");
    const AbxLinearPack &glu = wp->glu, &out = wp->out, &gate = wp->gate;
    ABX_REQUIRE(glu.csum && out.csum && gate.csum && glu.N == 512 && out.K == 128 && gate.K == glu.K && out.N == gate.N && glu.K == out.N,
                "abx_tri_mul_fwd: packs: glu = LN-folded C -> 2 x 256 (value, gate) pairs, out = LN-folded 128 -> C, gate = LN-folded C -> C");
    const int C = glu.K;
    const long long LL = (long long)L * L;
    const int Lp = (L + 3) / 4 * 4, KT = (Lp + 15) / 16;
    const long long LLp = (long long)L * Lp;
    const TriMulWs w = tri_mul_ws(workspace, B, L);
    if (int rc = abx_pair_mask(mask, w.pm, B, L, Lp, st)) return rc;
    // 1: [left | right] projections * sigmoid(their gates) * pair mask -> f16 operand images of the contraction.  GEMM rows are padded
    //    pair positions in (8 i x 16 k) blocks; the incoming variant reads z pair-transposed
    {
        AbxGemm g = {};
        g.A = z_in; g.sAb = LL * C; g.sAm = C; g.sAk = 1;
        g.M = (int)(((long long)(L + 7) / 8 * 8) * ((long long)(Lp + 15) / 16 * 16));
        g.batch = B;
        set_weights(g, glu, true, 0);
        g.C_split = w.lrp; g.sCb = 256LL * KT * 2 * L * 16; g.sCm = (long long)KT * 2 * L * 16; g.sCk = 2LL * L * 16; g.sCp = (long long)L * 16;
        g.c_split_L = Lp; g.c_transposed = 1; g.c_split_nA = 128; g.c_split_tile = 1;
        g.pair_L = L; g.pair_Lp = Lp; g.a_pair = 1; g.a_pair_transpose = outgoing ? 0 : L;
        g.glu = 1;
        g.rowscale = w.pm; g.sRSb = LLp;
        set_range(g, range_flag, range_tag, 0);
        if (int rc = abx_gemm(&g, st)) return rc;
    }
    // 2: 'ik,jk->ij' per (sample, channel): left channels = A-side images, right channels = B-side images of the same tensor
    {
        AbxGemm g = {};
        const long long per_ch = (long long)KT * 2 * L * 16;
        g.A_split = w.lrp; g.sA3b = 256 * per_ch; g.sA3i = per_ch; g.sA3k = 2LL * L * 16; g.sA3p = (long long)L * 16; g.sA3m = 16;
        g.B_split = w.lrp + 128 * per_ch; g.sB3b = 256 * per_ch; g.sB3i = per_ch; g.sB3k = 2LL * L * 16; g.sB3p = (long long)L * 16; g.sB3n = 16;
        g.batch_inner = 128;
        g.C = w.tt; g.sCb = LLp; g.sCm = Lp;
        g.M = L; g.N = L; g.K = KT * 16; g.batch = B * 128;
        g.exact = 2; g.alpha = 1.0f;
        set_range(g, range_flag, range_tag, 0);
        if (int rc = abx_gemm(&g, st)) return rc;
    }
    // 3: proj_out(final_norm(product)) * sigmoid(final_gate(norm(z))) + z in one dual GEMM (A = the channel-major product)
    {
        AbxGemm g = {};
        g.A = w.tt; g.sAb = 128 * LLp; g.sAm = 1; g.sAk = LLp;
        g.M = (int)LLp; g.batch = B;
        set_weights(g, out, true, 0);
        g.C = z_out; g.sCb = LL * C; g.sCm = C;
        if (Lp != L) { g.pair_L = L; g.pair_Lp = Lp; g.c_pair = 1; }
        g.A2 = z_in; g.sA2b = LL * C; g.sA2m = C; g.K2 = C;
        g.B2_split = gate.planes; g.sB23k = 2LL * gate.N * 16; g.sB23p = (long long)gate.N * 16; g.sB23n = 16; g.b2_exp = gate.b_exp;
        g.ln2_csum = gate.csum; g.bias2 = gate.bias;
        g.resid = z_in; g.sRb = LL * C; g.sRm = C;
        set_range(g, range_flag, range_tag, 0);
        return abx_gemm(&g, st);
    }
}

// ---- TriangleAttention block --------------------------------------------------------------------------------------------------------
namespace {
struct TriAttnWs { float* qkv; float* hid; float* bT; float* bT2; float* o; };
inline TriAttnWs tri_attn_ws(void* ws, int B, int L) {
    const long long LL = (long long)L * L, Lp = (L + 3) / 4 * 4;
    TriAttnWs w;
    char* p = static_cast<char*>(ws);
    // (the head of the workspace stays 768 floats per pair row: model/forward.py uses it as its 768-wide scratch between the blocks)
    w.qkv = reinterpret_cast<float*>(p);                      // (B L L, 576) rows
    w.hid = w.qkv + (long long)B * LL * 576;                  // (B L L, 192): gate * attention output, exact GEMMs only
    p += up(4LL * B * LL * 768);
    w.bT = reinterpret_cast<float*>(p); p += up(4LL * B * 4 * LL);
    w.bT2 = reinterpret_cast<float*>(p); p += up(4LL * B * 4 * L * Lp);
    w.o = reinterpret_cast<float*>(p);
    return w;
}
}  // namespace

extern "C" long long abx_tri_attn_block_workspace_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return -1;
    const long long LL = (long long)L * L, Lp = (L + 3) / 4 * 4;
    return up(4LL * B * LL * 768) + up(4LL * B * 4 * LL) + up(4LL * B * 4 * L * Lp) + up(4LL * B * LL * 192);
}

extern "C" int abx_tri_attn_block_fwd(const AbxTriAttnPack* wp, float* z, const float* mask, int B, int L, int per_row, int exact,
                                      void* workspace, int* range_flag, int range_tag, hipStream_t st) {
    ABX_REQUIRE(wp && z && mask && workspace && B > 0 && L > 0, "abx_tri_attn_block_fwd: bad args");
    const AbxLinearPack &qkv = wp->qkv, &gate = wp->gate, &pair = wp->pair, &out = wp->out;
    ABX_REQUIRE(qkv.csum && gate.csum && pair.csum && qkv.N == 576 && gate.N == 192 && pair.N == 4 && qkv.K == 192 && gate.K == 192 && pair.K == 192 &&
                    out.K == 192 && out.N == 192,
                "abx_tri_attn_block_fwd: packs: qkv = LN-folded 192 -> 576, gate = LN-folded 192 -> 192, pair = LN-folded 192 -> 4, out = 192 -> 192 "
                "(ABX_PACK_PERMUTE_K16)");
    const long long LL = (long long)L * L, M2 = (long long)B * LL;
    ABX_REQUIRE(M2 < (1LL << 31), "abx_tri_attn_block_fwd: too many pair rows for one launch");
    ABX_REQUIRE(exact >= 0 && exact <= 3, "abx_tri_attn_block_fwd: exact is a 2-bit field");
    const int attn_exact = (exact >> 1) & 1;                  // bit 1: the attention kernel; bit 0: the GEMMs
    exact &= 1;
    const int Lp = (L + 3) / 4 * 4, C = 192;
    const TriAttnWs w = tri_attn_ws(workspace, B, L);
    // Round 6 experiment, OFF unless ABX_KV_PLANES is set: on the split-f16 route the projection can write its k | v columns
    // as the operand images of the attention (two float16 planes of 16 x value per head: the same bytes at the same addresses) and the
    // attention's producer wave then stages them by DMA.  Bit-identical to the fp32 route (the attention computes exactly these pieces from
    // fp32 k | v) - and measured SLOWER: the attention does not gain (17.19 -> 17.42 ms per launch at 100 samples: its producer wave was
    // never on the critical path), the projection pays for the split and the 8-byte stores (13.7 -> 15.6 ms): profiles/r06h_kb_kvplanes.txt.
    static const bool kv_planes_on = getenv("ABX_KV_PLANES") != nullptr;
    const int planes = (kv_planes_on && !exact && !attn_exact && abx_gemm_planes_ok(M2)) ? 1 : 0;
    {   // q | k | v, and the pair bias stored (b, h, i, j) in the same grid (abx_gemm_side: one launch on the split-f16 path - the side
        // rides in the free half of the projection's last column tile -, the two launches otherwise)
        AbxGemm g = {};
        g.A = z; g.sAm = C; g.sAk = 1;
        g.C = w.qkv; g.sCm = 576;
        g.M = (int)M2; g.batch = 1;
        set_weights(g, qkv, true, exact);
        set_range(g, range_flag, range_tag, exact);
        if (planes) { g.c_planes_from = 192; g.c_planes_group = 48; }
        AbxGemm s2 = {};
        s2.A = z; s2.sAb = LL * C; s2.sAm = C; s2.sAk = 1;
        s2.C = w.bT; s2.sCb = 4 * LL; s2.sCm = LL; s2.c_transposed = 1;
        s2.M = (int)LL; s2.batch = B;
        set_weights(s2, pair, true, exact);
        set_range(s2, range_flag, range_tag, exact);
        // the split-f16 attention adds the bias to accumulators that hold 2^7 x base-2 logits: the factor goes into this epilogue
        // (AbxTriAttn.bias_log2: the same product, rounded once, here instead of per key tile in the attention's hot loop)
        if (!attn_exact) s2.alpha = ABX_TRI_BIAS_LOG2;
        if (int rc = abx_gemm_side(&g, &s2, st)) return rc;
    }
    const float* bias = w.bT;
    if (!per_row || Lp != L) {      // key-contiguous rows of Lp floats; ending node: bias[b,h,q,k] = P[b,k,q,h]
        if (int rc = abx_transpose_last2(w.bT, w.bT2, B * 4, L, Lp, per_row ? 0 : 1, st)) return rc;
        bias = w.bT2;
    }
    {
        AbxTriAttn a = {};
        a.q = w.qkv; a.k = w.qkv + 192; a.v = w.qkv + 384; a.gate = nullptr;     // (the gate is applied by the tail)
        a.sb = LL * 576;
        a.ss = per_row ? (long long)L * 576 : 576;
        a.sl = per_row ? 576 : (long long)L * 576;
        a.bias = bias; a.bias_sb = 4LL * L * Lp; a.bias_sh = (long long)L * Lp; a.bias_sq = Lp; a.bias_sk = 1;
        a.keymask = mask; a.km_sb = L;
        a.out = w.o; a.ob = LL * C;
        a.os = per_row ? (long long)L * C : C;
        a.ol = per_row ? C : (long long)L * C;
        a.B = B; a.S = L; a.L = L; a.H = 4; a.D = 48;
        a.scale = 0.14433756729740643f;                     // 48^-0.5
        a.exact = attn_exact;
        a.bias_log2 = attn_exact ? 0 : 1;
        a.kv_planes = planes;
        if (!attn_exact) { a.range_flag = range_flag; a.range_tag = range_tag; }
        if (int rc = abx_tri_attn_fwd(&a, st)) return rc;
    }
    if (!exact) {
        // gated tail: z += (sigmoid(LN(z) Wg + bg) * o) Wo + bo in ONE kernel (AbxGemm.mlp = 2)
        AbxGemm g = {};
        g.A = z; g.sAm = C; g.sAk = 1;
        g.C = z; g.sCm = C;
        g.M = (int)M2; g.batch = 1;
        set_weights(g, gate, true, 0);
        g.act = 2;
        g.gate = w.o; g.sGm = C;
        g.mlp = 2; g.N2 = C; g.b2_exp = out.b_exp;
        g.B2_split = out.planes; g.sB23k = 2LL * C * 16; g.sB23p = (long long)C * 16; g.sB23n = 16;
        g.bias2 = out.bias;
        g.resid = z; g.sRm = C;
        set_range(g, range_flag, range_tag, 0);
        return abx_gemm(&g, st);
    }
    {   // exact GEMMs: hid = sigmoid(LN(z) Wg + bg) * o
        AbxGemm g = {};
        g.A = z; g.sAm = C; g.sAk = 1;
        g.C = w.hid; g.sCm = C;
        g.M = (int)M2; g.batch = 1;
        set_weights(g, gate, true, 1);
        g.act = 2;
        g.gate = w.o; g.sGm = C; g.gate_sigmoid = 0;
        if (int rc = abx_gemm(&g, st)) return rc;
    }
    {   // output projection + residual
        AbxGemm g = {};
        g.A = w.hid; g.sAm = C; g.sAk = 1;
        g.C = z; g.sCm = C;
        g.M = (int)M2; g.batch = 1;
        set_weights(g, out, false, 1);
        g.B_split = nullptr; g.b_f16 = 0;                   // (out's planes are k-permuted: not an operand of a plain GEMM)
        g.resid = z; g.sRm = C;
        return abx_gemm(&g, st);
    }
}
