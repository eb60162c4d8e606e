// Structural-violation guidance terms (BASELINE north_star "guidance pairwise-distance/clash terms").
//
// The reference has no sampling-time guidance; this is the opt-in extension built from the material it does
// ship: the van-der-Waals radii and overlap tolerance of its violation bounds (abx/common/residue_constants.py:381-386,483-525,
// config/config_model.json:116,213-214 `clash_overlap_tolerance` 1.5, `between_chain_factor` 0.2, `violation_tolerance_factor`
// 12) and the C-N peptide-bond term of eval/metric_scripts/cal_vio.py:29-74.  Energies (per sample):
//     E_clash = sum over atom pairs of DIFFERENT residues, each pair once,
//                   w_ij * relu(r_a + r_b - overlap_tolerance - |x_a - x_b|)          (w_ij = between_chain_factor across chains)
//               excluding the peptide bond C(i)-N(i+1) of chain neighbours and SG-SG disulfides (AlphaFold's exclusions)
//     E_bond  = sum over chain neighbours of relu(sqrt(1e-6 + (|C_i - N_{i+1}| - l0)^2) - tolerance * sigma)   (l0, sigma: proline-aware)
//     E_angle = the same flat-bottom violation of cos(CA_i, C_i, N_{i+1}) and cos(C_i, N_{i+1}, CA_{i+1}) (cal_vio.py:76-105)
// "Chain neighbours" = array neighbours with the same chain id and, when `residx` is given, consecutive residue numbers.
// Outputs: the three energies, dE/dx for every atom14 position, and the pull-back to the residue frames x_a = R_i p_a + t_i:
//     dE/dt_i = sum_a g_a ,   dE/d(rotation vector of R_i, world frame) = sum_a (x_a - t_i) x g_a .
//
// Kernel: an O(N^2) pair kernel with N = 14 L atoms per sample (12.8 M pairs at L = 256, 24.3 M at L = 352), LDS-tiled: a block owns
// 16 residues (224 atoms, one thread each), streams all residues through a 16-residue LDS tile (positions + radius + residue /
// chain tags as float4 + int), every thread accumulates the force on ITS atom in a fixed order (deterministic), the energy is
// reduced with wave shuffles and written per block (summed in fixed order by the frame kernel).  HBM traffic is the atom table
// itself (56 KB per sample), so the kernel is ALU / LDS bound and far from any roofline that matters for the step (< 0.1 ms).
#include "common.h"
#include "abx_hip.h"

namespace {

constexpr int RT = 16;                 // residues per tile
constexpr int AT = RT * 14;            // atoms per tile (224)

// residue r is peptide-bonded to its array predecessor r - 1: same chain (full chain ids) and, when residue numbers are given,
// consecutive numbers (a cropped antigen patch or a chain with missing residues keeps one chain id across the gap)
__device__ __forceinline__ bool linked_to_prev(const AbxGuidanceArgs& a, long long ab, int res) {
    if (res <= 0 || res >= a.L) return false;
    const long long r = ab + res;
    if (a.chain_id[r] != a.chain_id[r - 1]) return false;
    return !a.residx || a.residx[r] == a.residx[r - 1] + 1;
}

__global__ __launch_bounds__(256) void clash_kernel(const AbxGuidanceArgs a, float* __restrict__ epart) {
    __shared__ float4 tile[AT];        // x, y, z, radius (radius < 0: atom absent)
    __shared__ int tag[AT];            // residue index << 6 | linked to predecessor << 5 | SG << 4 | atom slot
    __shared__ int chn[AT];            // chain id
    __shared__ float ered[4];
    const int b = blockIdx.y, it = blockIdx.x, tid = threadIdx.x, L = a.L;
    const long long ab = (long long)b * L;
    auto load_atom = [&](int res, int slot, float4& p, int& t, int& c) {
        p = make_float4(0.f, 0.f, 0.f, -1.f);
        t = 0;
        c = 0;
        if (res < L) {
            const long long r = ab + res;
            const float* x = a.atom14 + (r * 14 + slot) * 3;
            long long aa = a.aatype[r];
            aa = aa < 0 ? 20 : (aa > 20 ? 20 : aa);
            const bool ok = a.atom_mask[r * 14 + slot] != 0;
            p = make_float4(x[0], x[1], x[2], ok ? a.radius[aa * 14 + slot] : -1.f);
            // SG of cysteine sits in atom14 slot 5: flagged for the disulfide exclusion
            const int sg = (aa == 4 && slot == 5) ? 1 : 0;
            t = (res << 6) | ((linked_to_prev(a, ab, res) ? 1 : 0) << 5) | (sg << 4) | slot;
            c = a.chain_id[r];
        }
    };
    // my atom
    const int mres = it * RT + tid / 14, mslot = tid % 14;
    float4 me = make_float4(0.f, 0.f, 0.f, -1.f);
    int mtag = 0, mchain = 0;
    if (tid < AT) load_atom(mres, mslot, me, mtag, mchain);
    const int msg = (mtag >> 4) & 1, mlink = (mtag >> 5) & 1;
    float gx = 0.f, gy = 0.f, gz = 0.f, e = 0.f;
    for (int jt = 0; jt < (L + RT - 1) / RT; ++jt) {
        __syncthreads();
        if (tid < AT) load_atom(jt * RT + tid / 14, tid % 14, tile[tid], tag[tid], chn[tid]);
        __syncthreads();
        if (tid < AT && me.w > 0.f) {
            for (int k = 0; k < AT; ++k) {
                const float4 o = tile[k];
                const int ot = tag[k];
                const int ores = ot >> 6;
                if (o.w <= 0.f || ores == mres) continue;
                const int oslot = ot & 15, ochain = chn[k];
                // peptide bond C(i) - N(i+1) of linked neighbours, SG - SG disulfide
                if ((ores == mres + 1 && mslot == 2 && oslot == 0 && ((ot >> 5) & 1)) || (mres == ores + 1 && oslot == 2 && mslot == 0 && mlink)) continue;
                if (msg && ((ot >> 4) & 1)) continue;
                const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
                const float d = sqrtf(1e-10f + dx * dx + dy * dy + dz * dz);
                const float ov = me.w + o.w - a.overlap_tolerance - d;
                if (ov > 0.f) {
                    const float w = (ochain == mchain ? 1.0f : a.between_chain_factor) * a.w_clash;
                    e += 0.5f * w * ov;                         // every pair is visited from both of its atoms
                    const float s = -w / d;                      // d(relu(c - d))/dx_me = -(x_me - x_o)/d
                    gx += s * dx; gy += s * dy; gz += s * dz;
                }
            }
        }
    }
    if (tid < AT && mres < L) {
        float* g = a.grad_atom + ((ab + mres) * 14 + mslot) * 3;
        g[0] = gx; g[1] = gy; g[2] = gz;
    }
    e = wave_sum(e);
    if ((tid & 63) == 0) ered[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) epart[(long long)b * gridDim.x + it] = (ered[0] + ered[1]) + (ered[2] + ered[3]);
}

// Flat-bottom term relu(sqrt(1e-6 + (v - v0)^2) - tol * sd): returns the energy, `slope` = dE/dv (0 inside the flat bottom)
__device__ __forceinline__ float flat_bottom(float v, float v0, float tol_sd, float& slope) {
    const float err = sqrtf(1e-6f + (v - v0) * (v - v0));
    const float e = err - tol_sd;
    slope = e > 0.f ? (v - v0) / err : 0.f;
    return e > 0.f ? e : 0.f;
}

// Peptide-geometry terms of the residue pair (l, l + 1) (eval/metric_scripts/cal_vio.py:29-110): the C-N bond length and the cosines
// of the CA-C-N and C-N-CA angles against their literature values (abx/common/residue_constants.py:475-480), each a flat-bottom
// violation.  g[0..3] receive dE/d(CA_l, C_l, N_u, CA_u) of THIS pair, eb / ea the bond / angle energies.
struct PairGrad { float g[4][3]; float eb, ea; };
__device__ __forceinline__ void peptide_pair(const AbxGuidanceArgs& a, long long ab, int l, PairGrad& o) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o.g[k][0] = o.g[k][1] = o.g[k][2] = 0.f;
    o.eb = o.ea = 0.f;
    if (l < 0 || !linked_to_prev(a, ab, l + 1)) return;
    const long long r = ab + l;
    const bool m_ca = a.atom_mask[r * 14 + 1] != 0, m_c = a.atom_mask[r * 14 + 2] != 0;
    const bool m_n = a.atom_mask[(r + 1) * 14 + 0] != 0, m_ca2 = a.atom_mask[(r + 1) * 14 + 1] != 0;
    if (!m_c || !m_n) return;
    const float* ca = a.atom14 + (r * 14 + 1) * 3;
    const float* c = a.atom14 + (r * 14 + 2) * 3;
    const float* n = a.atom14 + ((r + 1) * 14 + 0) * 3;
    const float* ca2 = a.atom14 + ((r + 1) * 14 + 1) * 3;
    const bool pro = a.aatype[r + 1] == 14;
    const float l0 = pro ? 1.341f : 1.329f, sd = pro ? 0.016f : 0.014f;
    // ---- bond: v = |C - N|
    const float bx = n[0] - c[0], by = n[1] - c[1], bz = n[2] - c[2];           // C -> N
    const float d = sqrtf(1e-6f + bx * bx + by * by + bz * bz);
    float sl;
    o.eb = a.w_bond * flat_bottom(d, l0, a.bond_tolerance_factor * sd, sl);
    {
        const float s = a.w_bond * sl / d;                                      // dE/dN = s * (N - C)
        o.g[2][0] += s * bx; o.g[2][1] += s * by; o.g[2][2] += s * bz;
        o.g[1][0] -= s * bx; o.g[1][1] -= s * by; o.g[1][2] -= s * bz;
    }
    if (a.w_angle == 0.f) return;
    // unit vector C -> N (l2_normalize: x / sqrt(max(|x|^2, 1e-12)), abx/model/utils.py:12-14)
    const float nb = sqrtf(fmaxf(bx * bx + by * by + bz * bz, 1e-12f));
    const float vx = bx / nb, vy = by / nb, vz = bz / nb;
    // cos(x, y) of unit vectors u = p / |p|, v = q / |q| about a vertex: d cos / dp = (v - cos u) / |p|, d cos / dq = (u - cos v) / |q|
    if (m_ca) {     // ---- CA_l - C_l - N_u  (vertex C): p = CA - C, q = N - C
        const float px = ca[0] - c[0], py = ca[1] - c[1], pz = ca[2] - c[2];
        const float np_ = sqrtf(fmaxf(px * px + py * py + pz * pz, 1e-12f));
        const float ux = px / np_, uy = py / np_, uz = pz / np_;
        const float cs = ux * vx + uy * vy + uz * vz;
        const float e = flat_bottom(cs, -0.4473f, a.bond_tolerance_factor * 0.0311f, sl);
        o.ea += a.w_angle * e;
        const float k = a.w_angle * sl;
        const float gp[3] = {k * (vx - cs * ux) / np_, k * (vy - cs * uy) / np_, k * (vz - cs * uz) / np_};
        const float gq[3] = {k * (ux - cs * vx) / nb, k * (uy - cs * vy) / nb, k * (uz - cs * vz) / nb};
#pragma unroll
        for (int x = 0; x < 3; ++x) { o.g[0][x] += gp[x]; o.g[2][x] += gq[x]; o.g[1][x] -= gp[x] + gq[x]; }
    }
    if (m_ca2) {    // ---- C_l - N_u - CA_u  (vertex N): p = C - N = -b, q = CA_u - N
        const float qx = ca2[0] - n[0], qy = ca2[1] - n[1], qz = ca2[2] - n[2];
        const float nq = sqrtf(fmaxf(qx * qx + qy * qy + qz * qz, 1e-12f));
        const float wx = qx / nq, wy = qy / nq, wz = qz / nq;
        const float ux = -vx, uy = -vy, uz = -vz;
        const float cs = ux * wx + uy * wy + uz * wz;
        const float e = flat_bottom(cs, -0.5203f, a.bond_tolerance_factor * 0.0353f, sl);
        o.ea += a.w_angle * e;
        const float k = a.w_angle * sl;
        const float gp[3] = {k * (wx - cs * ux) / nb, k * (wy - cs * uy) / nb, k * (wz - cs * uz) / nb};
        const float gq[3] = {k * (ux - cs * wx) / nq, k * (uy - cs * wy) / nq, k * (uz - cs * wz) / nq};
#pragma unroll
        for (int x = 0; x < 3; ++x) { o.g[1][x] += gp[x]; o.g[3][x] += gq[x]; o.g[2][x] -= gp[x] + gq[x]; }
    }
}

// Peptide bond / angle terms added to grad_atom, then the frame pull-back.  One thread per RESIDUE: it owns the backbone atoms
// N, CA, C of its residue and adds what the pairs (i - 1, i) and (i, i + 1) contribute to them (each pair is evaluated by both of
// its residues: no two threads write one atom, fixed summation order); the energies are counted by the lower residue of a pair.
__global__ __launch_bounds__(256) void bond_frames_kernel(const AbxGuidanceArgs a, const float* __restrict__ epart, int nparts) {
    extern __shared__ float esh[];                           // [256] bond + [256] angle energies of this sample
    const int b = blockIdx.x, tid = threadIdx.x, L = a.L;
    const long long ab = (long long)b * L;
    float eb = 0.f, ea = 0.f;
    for (int i = tid; i < L; i += 256) {
        const long long r = ab + i;
        PairGrad lo, hi;
        peptide_pair(a, ab, i - 1, lo);                      // this residue is the upper one: N (g[2]), CA (g[3])
        peptide_pair(a, ab, i, hi);                          // this residue is the lower one: CA (g[0]), C (g[1])
        eb += hi.eb;
        ea += hi.ea;
        float* gn = a.grad_atom + (r * 14 + 0) * 3;
        float* gca = a.grad_atom + (r * 14 + 1) * 3;
        float* gc = a.grad_atom + (r * 14 + 2) * 3;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            gn[x] += lo.g[2][x];
            gca[x] += lo.g[3][x] + hi.g[0][x];
            gc[x] += hi.g[1][x];
        }
    }
    esh[tid] = eb;
    esh[256 + tid] = ea;
    __syncthreads();
    if (tid == 0) {
        float sb = 0.f, sa = 0.f, ec = 0.f;
        for (int k = 0; k < 256; ++k) { sb += esh[k]; sa += esh[256 + k]; }
        for (int k = 0; k < nparts; ++k) ec += epart[(long long)b * nparts + k];
        a.energy[3 * b] = ec;
        a.energy[3 * b + 1] = sb;
        a.energy[3 * b + 2] = sa;
    }
    __threadfence_block();
    __syncthreads();
    // frame pull-back: translation gradient = sum of the residue's atom gradients, rotation gradient = torque about the frame origin
    for (int i = tid; i < L; i += 256) {
        const long long r = ab + i;
        const float* t = a.frame_trans + r * 3;
        float ft[3] = {0.f, 0.f, 0.f}, tq[3] = {0.f, 0.f, 0.f};
        for (int s = 0; s < 14; ++s) {
            if (!a.atom_mask[r * 14 + s]) continue;
            const float* x = a.atom14 + (r * 14 + s) * 3;
            const float* g = a.grad_atom + (r * 14 + s) * 3;
            const float rx = x[0] - t[0], ry = x[1] - t[1], rz = x[2] - t[2];
            ft[0] += g[0]; ft[1] += g[1]; ft[2] += g[2];
            tq[0] += ry * g[2] - rz * g[1];
            tq[1] += rz * g[0] - rx * g[2];
            tq[2] += rx * g[1] - ry * g[0];
        }
        for (int k = 0; k < 3; ++k) {
            a.grad_trans[r * 3 + k] = ft[k];
            a.grad_rot[r * 3 + k] = tq[k];
        }
    }
}

}  // namespace

extern "C" long long abx_clash_grad_workspace_bytes(int B, int L) {
    return (long long)B * ((L + RT - 1) / RT) * sizeof(float);
}

extern "C" int abx_clash_grad(const AbxGuidanceArgs* ap, void* workspace, hipStream_t st) {
    ABX_REQUIRE(ap != nullptr, "abx_clash_grad: null");
    const AbxGuidanceArgs a = *ap;
    ABX_REQUIRE(a.atom14 && a.atom_mask && a.aatype && a.chain_id && a.radius && a.frame_trans && a.energy && a.grad_atom &&
                    a.grad_trans && a.grad_rot && workspace, "abx_clash_grad: null operand");
    ABX_REQUIRE(a.B > 0 && a.L > 1 && a.B <= 65535 && a.L < (1 << 22), "abx_clash_grad: bad sizes");
    const int nparts = (a.L + RT - 1) / RT;
    float* epart = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(clash_kernel, dim3(nparts, a.B), dim3(256), 0, st, a, epart);
    int rc = abx_check_launch("abx_clash_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(bond_frames_kernel, dim3(a.B), dim3(256), 512 * sizeof(float), st, a, epart, nparts);
    return abx_check_launch("abx_clash_grad(bond, frames)");
}
