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
// Outputs: both energies, dE/dx for every atom14 position, and the pull-back to the residue frames x_a = R_i p_a + t_i:
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

__global__ __launch_bounds__(256) void clash_kernel(const AbxGuidanceArgs a, float* __restrict__ epart) {
    __shared__ float4 tile[AT];        // x, y, z, radius (radius < 0: atom absent)
    __shared__ int tag[AT];            // residue index << 8 | chain << 4 | atom slot
    __shared__ float ered[4];
    const int b = blockIdx.y, it = blockIdx.x, tid = threadIdx.x, L = a.L;
    const long long ab = (long long)b * L;
    auto load_atom = [&](int res, int slot, float4& p, int& t) {
        p = make_float4(0.f, 0.f, 0.f, -1.f);
        t = 0;
        if (res < L) {
            const long long r = ab + res;
            const float* x = a.atom14 + (r * 14 + slot) * 3;
            long long aa = a.aatype[r];
            aa = aa < 0 ? 20 : (aa > 20 ? 20 : aa);
            const bool ok = a.atom_mask[r * 14 + slot] != 0;
            p = make_float4(x[0], x[1], x[2], ok ? a.radius[aa * 14 + slot] : -1.f);
            // SG of cysteine sits in atom14 slot 5: flagged for the disulfide exclusion
            const int sg = (aa == 4 && slot == 5) ? 1 : 0;
            t = (res << 9) | (sg << 8) | ((a.chain_id[r] & 15) << 4) | slot;
        }
    };
    // my atom
    const int mres = it * RT + tid / 14, mslot = tid % 14;
    float4 me = make_float4(0.f, 0.f, 0.f, -1.f);
    int mtag = 0;
    if (tid < AT) load_atom(mres, mslot, me, mtag);
    const int mchain = (mtag >> 4) & 15, msg = (mtag >> 8) & 1;
    float gx = 0.f, gy = 0.f, gz = 0.f, e = 0.f;
    for (int jt = 0; jt < (L + RT - 1) / RT; ++jt) {
        __syncthreads();
        if (tid < AT) load_atom(jt * RT + tid / 14, tid % 14, tile[tid], tag[tid]);
        __syncthreads();
        if (tid < AT && me.w > 0.f) {
            for (int k = 0; k < AT; ++k) {
                const float4 o = tile[k];
                const int ot = tag[k];
                const int ores = ot >> 9;
                if (o.w <= 0.f || ores == mres) continue;
                const int oslot = ot & 15, ochain = (ot >> 4) & 15;
                // peptide bond C(i) - N(i+1) inside a chain, SG - SG disulfide
                if (ochain == mchain && ((ores == mres + 1 && mslot == 2 && oslot == 0) || (mres == ores + 1 && oslot == 2 && mslot == 0))) continue;
                if (msg && ((ot >> 8) & 1)) continue;
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

// C(i) - N(i+1) flat-bottom bond term (cal_vio.py:29-74), added to grad_atom; one thread per residue pair, then the frame pull-back
__global__ __launch_bounds__(256) void bond_frames_kernel(const AbxGuidanceArgs a, const float* __restrict__ epart, int nparts) {
    extern __shared__ float ebond[];                         // per-thread bond energies of this sample
    const int b = blockIdx.x, tid = threadIdx.x, L = a.L;
    const long long ab = (long long)b * L;
    float e = 0.f;
    for (int i = tid; i + 1 < L; i += 256) {
        const long long r = ab + i;
        if (a.chain_id[r] != a.chain_id[r + 1] || !a.atom_mask[r * 14 + 2] || !a.atom_mask[(r + 1) * 14 + 0]) continue;
        const float* c = a.atom14 + (r * 14 + 2) * 3;
        const float* n = a.atom14 + ((r + 1) * 14 + 0) * 3;
        const bool pro = a.aatype[r + 1] == 14;
        const float l0 = pro ? 1.341f : 1.329f, sd = pro ? 0.016f : 0.014f;
        const float dx = c[0] - n[0], dy = c[1] - n[1], dz = c[2] - n[2];
        const float d = sqrtf(1e-6f + dx * dx + dy * dy + dz * dz);
        const float err = sqrtf(1e-6f + (d - l0) * (d - l0));
        const float v = err - a.bond_tolerance_factor * sd;
        if (v > 0.f) {
            e += a.w_bond * v;
            const float s = a.w_bond * ((d - l0) / err) / d;     // dE/dC = s * (C - N)
            // each (i, i+1) pair is owned by one thread and every atom takes part in at most one bond as C and one as N:
            // C atoms are written by pair i, N atoms by pair i (of residue i + 1): no two threads touch the same atom
            float* gc = a.grad_atom + (r * 14 + 2) * 3;
            float* gn = a.grad_atom + ((r + 1) * 14 + 0) * 3;
            gc[0] += s * dx; gc[1] += s * dy; gc[2] += s * dz;
            gn[0] -= s * dx; gn[1] -= s * dy; gn[2] -= s * dz;
        }
    }
    ebond[tid] = e;
    __syncthreads();
    if (tid == 0) {
        float eb = 0.f, ec = 0.f;
        for (int k = 0; k < 256; ++k) eb += ebond[k];
        for (int k = 0; k < nparts; ++k) ec += epart[(long long)b * nparts + k];
        a.energy[2 * b] = ec;
        a.energy[2 * b + 1] = eb;
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
    hipLaunchKernelGGL(bond_frames_kernel, dim3(a.B), dim3(256), 256 * sizeof(float), st, a, epart, nparts);
    return abx_check_launch("abx_clash_grad(bond, frames)");
}
