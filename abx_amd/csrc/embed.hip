// Embedding assembly and the small gather/elementwise stages around the pair stack.
// Reference sites: abx/model/seqformer.py:49-119 (timestep embedding, Embedder), :170-223 (EmbeddingAndSeqformer.forward),
// :400-409 (OuterProductMean features), abx/model/encoder.py:123-269 (Residue/PairEmbedding gathers).
#include "common.h"
#include "abx_hip.h"

namespace {

__global__ void timestep_embedding_kernel(const double* __restrict__ t, const float* __restrict__ freqs, int B, int dim,
                                          float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx % half;
    // seqformer.py:56-62: timesteps * 10000 in t's own dtype (double in the loop), THEN .float(); the fp32 frequency
    // table exp(arange(half) * -log(10000)/(half-1)) comes from the host so that it is bit-identical to torch's
    const float tt = (float)(t[b] * 10000.0);
    const float arg = tt * freqs[i];
    out[b * dim + i] = sinf(arg);
    out[b * dim + half + i] = cosf(arg);
}

// one wave per (b,l) row; C = static channels (512), E = time-embedding width (32); LN over C+E of prev_seq.
__global__ __launch_bounds__(256) void assemble_seq_kernel(const float* __restrict__ seq_static, long long ss_b,
                                                           const float* __restrict__ aa_table,
                                                           const long long* __restrict__ seq_t, int Lab,
                                                           const float* __restrict__ temb, const float* __restrict__ prev,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ out, int B, int L, int C, int E) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * L) return;
    const int b = (int)(row / L), l = (int)(row % L);
    const int W = C + E;
    float mean = 0.f, rstd = 0.f;
    const float* pr = prev ? prev + row * W : nullptr;
    if (pr) {
        float s = 0.f;
        for (int k = lane; k < W; k += 64) s += pr[k];
        mean = wave_sum(s) / (float)W;
        float q = 0.f;
        for (int k = lane; k < W; k += 64) {
            const float d = pr[k] - mean;
            q += d * d;
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-5f);
    }
    const float* st = seq_static + (long long)b * ss_b + (long long)l * C;
    const float* aa = (l < Lab) ? aa_table + seq_t[row] * C : nullptr;
    for (int k = lane; k < W; k += 64) {
        float v;
        if (k < C) {
            v = st[k];
            if (aa) v = v + aa[k];
        } else {
            v = temb[b * E + (k - C)];
        }
        if (pr) v += (pr[k] - mean) * rstd * gamma[k] + beta[k];
        out[row * W + k] = v;
    }
}

// Model dimensions (C = 128, E = 32 -> W = 192): 16 lanes per (b,i,j) row, three float4 per lane, 4 rows per wave; every
// access is 16 bytes, the LayerNorm reductions stay inside the 16-lane group.
__global__ __launch_bounds__(256) void assemble_pair192_kernel(const float* __restrict__ pair_static, long long ps_b,
                                                               const float* __restrict__ temb, const float* __restrict__ prev,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const long long* __restrict__ prev_pos,
                                                               const float* __restrict__ pos_table, float* __restrict__ out,
                                                               long long rows, long long LL) {
    constexpr int C = 128, E = 32, W = 192;
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    const int b = (int)(row / LL);
    const long long ij = row % LL;
    f32x4 x[3];
    float mean = 0.f, rstd = 0.f;
    if (prev) {
        const f32x4* pr = reinterpret_cast<const f32x4*>(prev + row * W);
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            x[u] = pr[l16 + 16 * u];
            s += (x[u][0] + x[u][1]) + (x[u][2] + x[u][3]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mean = s / (float)W;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = x[u][c] - mean;
                q = fmaf(d, d, q);
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        rstd = 1.0f / sqrtf(q / (float)W + 1e-5f);
    }
    const f32x4* st = reinterpret_cast<const f32x4*>(pair_static + (long long)b * ps_b + ij * C);
    const f32x4* pt = prev_pos ? reinterpret_cast<const f32x4*>(pos_table + prev_pos[row] * W) : nullptr;
    f32x4* op = reinterpret_cast<f32x4*>(out + row * W);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int q4 = l16 + 16 * u;                     // float4 index inside the row: 0..31 static part, 32..47 the two time embeddings
        f32x4 v = q4 < C / 4 ? st[q4] : reinterpret_cast<const f32x4*>(temb + b * E)[(q4 - C / 4) % (E / 4)];
        if (prev) {
            const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[q4], be = reinterpret_cast<const f32x4*>(beta)[q4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += (x[u][c] - mean) * rstd * ga[c] + be[c];
        }
        if (pt) {
            const f32x4 pv = pt[q4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += pv[c];
        }
        op[q4] = v;
    }
}

// one wave per (b,i,j) row; W = C + 2E = 192 handled as W/64 = 3 elements per lane.
__global__ __launch_bounds__(256) void assemble_pair_kernel(const float* __restrict__ pair_static, long long ps_b,
                                                            const float* __restrict__ temb, const float* __restrict__ prev,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const long long* __restrict__ prev_pos,
                                                            const float* __restrict__ pos_table, float* __restrict__ out,
                                                            float* __restrict__ stats_out, long long rows, long long LL, int C,
                                                            int E) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / LL);
    const long long ij = row % LL;
    const int W = C + 2 * E;
    float x[4];                                   // W <= 256
    const float* pr = prev ? prev + row * W : nullptr;
    float mean = 0.f, rstd = 0.f;
    if (pr) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lane + u * 64;
            x[u] = k < W ? pr[k] : 0.f;
            s += x[u];
        }
        mean = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lane + u * 64;
            const float d = k < W ? x[u] - mean : 0.f;
            q += d * d;
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-5f);
    }
    const float* st = pair_static + (long long)b * ps_b + ij * C;
    const float* pt = prev_pos ? pos_table + prev_pos[row] * W : nullptr;
    float y[4];
    float ys = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = lane + u * 64;
        y[u] = 0.f;
        if (k >= W) continue;
        float v = k < C ? st[k] : temb[b * E + ((k - C) % E)];
        if (pr) v += (x[u] - mean) * rstd * gamma[k] + beta[k];
        if (pt) v += pt[k];
        out[row * W + k] = v;
        y[u] = v;
        ys += v;
    }
    if (stats_out) {      // LayerNorm statistics of the assembled row for the first consumer (seq_attn.pair_norm), two-pass
        const float om = wave_sum(ys) / (float)W;
        float oq = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lane + u * 64;
            const float d = k < W ? y[u] - om : 0.f;
            oq += d * d;
        }
        oq = wave_sum(oq) / (float)W;
        if (lane == 0) {
            stats_out[2 * row] = om;
            stats_out[2 * row + 1] = 1.0f / sqrtf(oq + 1e-5f);
        }
    }
}

__global__ __launch_bounds__(256) void opm_features_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                           long long ld, float* __restrict__ feat, int B, int L, int C) {
    // thread per (b,i,j,c): feat[...,c] = left[b,j,c]*right[b,i,c]; feat[...,C+c] = left[b,j,c]-right[b,i,c]
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * L * L * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long p = idx / C;
    const int j = (int)(p % L);
    const long long bi = p / L;
    const int b = (int)(bi / L);
    const float lv = left[((long long)b * L + j) * ld + c], rv = right[bi * ld + c];
    feat[p * 2 * C + c] = lv * rv;
    feat[p * 2 * C + C + c] = lv - rv;
}

// the same with 16-byte accesses: thread per (b, i, j, 4 channels); a pair row leaves as two contiguous C*4-byte segments
__global__ __launch_bounds__(256) void opm_features4_kernel(const float* __restrict__ left, const float* __restrict__ right,
                                                            long long ld, float* __restrict__ feat, int B, int L, int C4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * L * L * C4) return;
    const int c4 = (int)(idx % C4);
    const long long p = idx / C4;
    const int j = (int)(p % L);
    const long long bi = p / L;
    const int b = (int)(bi / L);
    const f32x4 lv = *reinterpret_cast<const f32x4*>(left + ((long long)b * L + j) * ld + c4 * 4);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(right + bi * ld + c4 * 4);
    f32x4 pr, df;
#pragma unroll
    for (int c = 0; c < 4; ++c) { pr[c] = lv[c] * rv[c]; df[c] = lv[c] - rv[c]; }
    float* o = feat + p * (8 * C4) + c4 * 4;
    *reinterpret_cast<f32x4*>(o) = pr;
    *reinterpret_cast<f32x4*>(o + 4 * C4) = df;
}

__global__ void pair_mask_kernel(const float* __restrict__ mask, float* __restrict__ out, int B, int L, int Lp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * L * Lp) return;
    const int j = (int)(idx % Lp);
    const long long bi = idx / Lp;
    const int b = (int)(bi / L);
    out[idx] = j < L ? mask[bi] * mask[(long long)b * L + j] : 0.f;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx,
                                                          const float* __restrict__ rowscale, float* __restrict__ out,
                                                          long long s_out, long long n, int C) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * C) return;
    const long long r = t / C;
    const int c = (int)(t % C);
    float v = table[idx[r] * C + c];
    if (rowscale) v *= rowscale[r];
    out[r * s_out + c] = v;
}

// relative-position block embedding (seqformer.py:181-206): antibody x antibody and antigen x antigen blocks, zero off-block.
__global__ __launch_bounds__(256) void relpos_block_kernel(const int* __restrict__ residx, const float* __restrict__ table,
                                                           float* __restrict__ out, int B, int L, int Lab, int C, int max_rel) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * L * L * C) return;
    const int c = (int)(t % C);
    const long long p = t / C;
    const int j = (int)(p % L);
    const long long bi = p / L;
    const int i = (int)(bi % L), b = (int)(bi / L);
    float v = 0.f;
    if ((i < Lab) == (j < Lab)) {
        int off = residx[(long long)b * L + j] - residx[(long long)b * L + i] + max_rel;
        off = min(max(off, 0), 2 * max_rel) + 1;
        v = table[(long long)off * C + c];
    }
    out[t] = v;
}

__device__ __forceinline__ void pseudo_beta(const float* n, const float* ca, const float* c, float* cb) {
    const float bx = ca[0] - n[0], by = ca[1] - n[1], bz = ca[2] - n[2];
    const float cx = c[0] - ca[0], cy = c[1] - ca[1], cz = c[2] - ca[2];
    const float ax = by * cz - bz * cy, ay = bz * cx - bx * cz, az = bx * cy - by * cx;
    cb[0] = -0.58273431f * ax + 0.56802827f * bx - 0.54067466f * cx + ca[0];
    cb[1] = -0.58273431f * ay + 0.56802827f * by - 0.54067466f * cy + ca[1];
    cb[2] = -0.58273431f * az + 0.56802827f * bz - 0.54067466f * cz + ca[2];
}

// PairEmbedding gather stage (encoder.py:231-262): one workgroup per (b,i,j).
//  feat512[0:128]   = aa_pair_embed[aa_i*23 + aa_j]
//  feat512[128:256] = relpos_embed[clamp(residx_i - residx_j, -32, 32) + 32] * same_chain
//  feat512[256:384] : left for the distance MLP (written by abx_gemm)
//  feat512[384:512] = dgram_embed[bin(|CB_i - CB_j|^2)]
//  dist196[a*14+a'] = exp(-softplus(coef[aa_pair][a*14+a']) * (|x_ia - x_ja'| / 10)^2) * CA_i_exists * CA_j_exists
__global__ __launch_bounds__(256) void pair_embed_features_kernel(
    const long long* __restrict__ aa, const int* __restrict__ chain_id, const int* __restrict__ residx,
    const float* __restrict__ atom14, const unsigned char* __restrict__ exists, const float* __restrict__ aa_pair_embed,
    const float* __restrict__ relpos_embed, const float* __restrict__ distcoef, const float* __restrict__ dgram_embed,
    const float* __restrict__ sq_breaks, float* __restrict__ feat512, float* __restrict__ dist196, int B, int L) {
    const long long p = blockIdx.x;                       // (b*L + i)*L + j
    const int j = (int)(p % L);
    const long long bi = p / L;
    const int b = (int)(bi / L);
    const long long bj = (long long)b * L + j;
    const int tid = threadIdx.x;
    const long long aap = aa[bi] * 23 + aa[bj];
    __shared__ float xi[42], xj[42];
    if (tid < 42) xi[tid] = atom14[bi * 42 + tid];
    else if (tid >= 64 && tid < 106) xj[tid - 64] = atom14[bj * 42 + (tid - 64)];
    __syncthreads();
    float* f = feat512 + p * 512;
    if (tid < 128) {
        f[tid] = aa_pair_embed[aap * 128 + tid];
    } else {
        const int c = tid - 128;
        int rel = residx[bi] - residx[bj];
        rel = min(max(rel, -32), 32) + 32;
        const float same = chain_id[bi] == chain_id[bj] ? 1.f : 0.f;
        f[128 + c] = relpos_embed[rel * 128 + c] * same;
        float cbi[3], cbj[3];
        pseudo_beta(xi, xi + 3, xi + 6, cbi);
        pseudo_beta(xj, xj + 3, xj + 6, cbj);
        const float dx = cbi[0] - cbj[0], dy = cbi[1] - cbj[1], dz = cbi[2] - cbj[2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        int bin = 0;
        // sq_breaks = square(linspace(3.375, 21.375, 14)) as torch computes it on the host (common_modules.py:108-109)
#pragma unroll
        for (int k = 0; k < 14; ++k) bin += d2 > sq_breaks[k] ? 1 : 0;
        f[384 + c] = dgram_embed[bin * 128 + c];
    }
    if (tid < 196) {
        const int a = tid / 14, a2 = tid % 14;
        const float dx = xi[a * 3] - xj[a2 * 3], dy = xi[a * 3 + 1] - xj[a2 * 3 + 1], dz = xi[a * 3 + 2] - xj[a2 * 3 + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz) / 10.0f;
        const float w = distcoef[aap * 196 + tid];
        const float sp = w > 20.f ? w : log1pf(expf(w));            // torch softplus (beta 1, threshold 20)
        const float m = (exists[bi * 14 + 1] && exists[bj * 14 + 1]) ? 1.f : 0.f;
        dist196[p * 196 + tid] = expf(-1.0f * sp * (d * d)) * m;
    }
}

// out[n][a][b] = in[n][b][a] (transpose) or in[n][a][b] (copy) for nmat L x L matrices, output rows padded to Lp >= L entries
// (pad columns written as 0): 32x32 LDS tiles, both sides coalesced
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ in, float* __restrict__ out, int L, int Lp,
                                                              int transpose) {
    __shared__ float tile[32][33];
    const long long ibase = (long long)blockIdx.z * L * L, obase = (long long)blockIdx.z * L * Lp;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;         // output tile: rows a0.., columns b0..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // transpose: read in[b][a] with a fastest; copy: read in[a][b] with b fastest
        const int y = ty + r * 8;
        const int ia = transpose ? a0 + tx : a0 + y, ib = transpose ? b0 + y : b0 + tx;
        float v = 0.f;
        if (ia < L && ib < L) v = transpose ? in[ibase + (long long)ib * L + ia] : in[ibase + (long long)ia * L + ib];
        if (transpose) tile[y][tx] = v;            // tile[b - b0][a - a0]
        else tile[tx][y] = v;                      // tile[b - b0][a - a0]
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = a0 + ty + r * 8, b = b0 + tx;
        if (a < L && b < Lp) out[obase + (long long)a * Lp + b] = tile[tx][ty + r * 8];
    }
}

}  // namespace

extern "C" int abx_transpose_last2(const float* in, float* out, int nmat, int L, int Lp, int transpose, hipStream_t st) {
    ABX_REQUIRE(in && out && in != out && nmat > 0 && nmat <= 65535 && L > 0 && Lp >= L, "abx_transpose_last2: bad args");
    hipLaunchKernelGGL(transpose_last2_kernel, dim3((Lp + 31) / 32, (L + 31) / 32, nmat), dim3(256), 0, st, in, out, L, Lp, transpose);
    return abx_check_launch("abx_transpose_last2");
}

extern "C" int abx_timestep_embedding(const double* t, const float* freqs, int B, int dim, float* out, hipStream_t st) {
    ABX_REQUIRE(t && freqs && out && B > 0 && dim >= 4 && dim % 2 == 0, "abx_timestep_embedding: bad args");
    const int n = B * dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 127) / 128), dim3(128), 0, st, t, freqs, B, dim, out);
    return abx_check_launch("abx_timestep_embedding");
}

extern "C" int abx_assemble_seq(const float* seq_static, long long ss_b, const float* aa_table, const long long* seq_t, int Lab,
                                const float* temb, const float* prev_seq, const float* gamma, const float* beta, float* out,
                                int B, int L, int C, int E, hipStream_t st) {
    ABX_REQUIRE(seq_static && aa_table && seq_t && temb && out && B > 0 && L > 0, "abx_assemble_seq: bad args");
    ABX_REQUIRE(!prev_seq || (gamma && beta), "abx_assemble_seq: LN params missing");
    const long long rows = (long long)B * L;
    hipLaunchKernelGGL(assemble_seq_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, seq_static, ss_b, aa_table, seq_t,
                       Lab, temb, prev_seq, gamma, beta, out, B, L, C, E);
    return abx_check_launch("abx_assemble_seq");
}

extern "C" int abx_assemble_pair(const float* pair_static, long long ps_b, const float* temb, const float* prev_pair,
                                 const float* gamma, const float* beta, const long long* prev_pos, const float* pos_table,
                                 float* out, float* stats_out, int B, int L, int C, int E, hipStream_t st) {
    ABX_REQUIRE(pair_static && temb && out && B > 0 && L > 0, "abx_assemble_pair: bad args");
    ABX_REQUIRE(C + 2 * E <= 256, "abx_assemble_pair: width > 256");
    ABX_REQUIRE(!prev_pair || (gamma && beta), "abx_assemble_pair: LN params missing");
    ABX_REQUIRE(!prev_pos || pos_table, "abx_assemble_pair: pos table missing");
    const long long rows = (long long)B * L * L;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (C == 128 && E == 32 && !stats_out && ps_b % 4 == 0 && al16(pair_static) && al16(temb) && al16(out) &&
        (!prev_pair || (al16(prev_pair) && al16(gamma) && al16(beta))) && (!prev_pos || al16(pos_table))) {
        hipLaunchKernelGGL(assemble_pair192_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, pair_static, ps_b, temb,
                           prev_pair, gamma, beta, prev_pos, pos_table, out, rows, (long long)L * L);
        return abx_check_launch("abx_assemble_pair");
    }
    hipLaunchKernelGGL(assemble_pair_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, pair_static, ps_b, temb,
                       prev_pair, gamma, beta, prev_pos, pos_table, out, stats_out, rows, (long long)L * L, C, E);
    return abx_check_launch("abx_assemble_pair");
}

extern "C" int abx_opm_features(const float* left, const float* right, long long ld, float* feat, int B, int L, int C,
                                hipStream_t st) {
    ABX_REQUIRE(left && right && feat && B > 0 && L > 0 && C > 0, "abx_opm_features: bad args");
    const long long total = (long long)B * L * L * C;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (C % 4 == 0 && ld % 4 == 0 && al16(left) && al16(right) && al16(feat)) {
        ABX_REQUIRE((total / 4 + 255) / 256 < (1LL << 31), "abx_opm_features: grid too large");
        hipLaunchKernelGGL(opm_features4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, left, right, ld, feat, B, L,
                           C / 4);
        return abx_check_launch("abx_opm_features");
    }
    ABX_REQUIRE((total + 255) / 256 < (1LL << 31), "abx_opm_features: grid too large");
    hipLaunchKernelGGL(opm_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, left, right, ld, feat, B, L, C);
    return abx_check_launch("abx_opm_features");
}

extern "C" int abx_pair_mask(const float* mask, float* out, int B, int L, int Lp, hipStream_t st) {
    ABX_REQUIRE(mask && out && B > 0 && L > 0 && Lp >= L, "abx_pair_mask: bad args");
    const long long total = (long long)B * L * Lp;
    hipLaunchKernelGGL(pair_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, mask, out, B, L, Lp);
    return abx_check_launch("abx_pair_mask");
}

extern "C" int abx_gather_rows(const float* table, const long long* idx, const float* rowscale, float* out, long long s_out,
                               long long n, int C, hipStream_t st) {
    ABX_REQUIRE(table && idx && out && n > 0 && C > 0, "abx_gather_rows: bad args");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * C + 255) / 256)), dim3(256), 0, st, table, idx, rowscale, out,
                       s_out, n, C);
    return abx_check_launch("abx_gather_rows");
}

extern "C" int abx_relpos_block(const int* residx, const float* table, float* out, int B, int L, int Lab, int C, int max_rel,
                                hipStream_t st) {
    ABX_REQUIRE(residx && table && out && B > 0 && L > 0, "abx_relpos_block: bad args");
    const long long total = (long long)B * L * L * C;
    hipLaunchKernelGGL(relpos_block_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, residx, table, out, B, L,
                       Lab, C, max_rel);
    return abx_check_launch("abx_relpos_block");
}

extern "C" int abx_pair_embed_features(const long long* aa, const int* chain_id, const int* residx, const float* atom14,
                                       const unsigned char* atom14_exists, const float* aa_pair_embed, const float* relpos_embed,
                                       const float* distcoef, const float* dgram_embed, const float* sq_breaks, float* feat512,
                                       float* dist196, int B, int L, hipStream_t st) {
    ABX_REQUIRE(aa && chain_id && residx && atom14 && atom14_exists && aa_pair_embed && relpos_embed && distcoef && dgram_embed &&
                    sq_breaks && feat512 && dist196 && B > 0 && L > 0, "abx_pair_embed_features: bad args");
    const long long blocks = (long long)B * L * L;
    ABX_REQUIRE(blocks < (1ll << 31), "abx_pair_embed_features: too many pairs");
    hipLaunchKernelGGL(pair_embed_features_kernel, dim3((unsigned)blocks), dim3(256), 0, st, aa, chain_id, residx, atom14,
                       atom14_exists, aa_pair_embed, relpos_embed, distcoef, dgram_embed, sq_breaks, feat512, dist196, B, L);
    return abx_check_launch("abx_pair_embed_features");
}
