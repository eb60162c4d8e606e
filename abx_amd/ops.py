"""Tensor-level wrappers over the C ABI (abx_amd/_lib.py).  torch is used for device memory and the current
stream only; every function launches HIP kernels from libabx_hip.so and raises if the library is unavailable."""
import ctypes as C
import math

import torch

from abx_amd import _lib
from abx_amd._lib import (AbxGemm, AbxTriAttn, AbxIpaTail, AbxHeadsTail, AbxScoreArgs, AbxReverseArgs, AbxGuidanceArgs, AbxLinearPack, AbxLinearSrc,
                           AbxTriMulPack, AbxTriAttnPack, check)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, 'libabx_hip works on device memory only'
    return t.data_ptr()


def _f32(t):
    assert t.dtype == torch.float32, t.dtype
    return t


GEMM_TUNE = int(__import__('os').environ.get('ABX_GEMM_TUNE', '0'))      # kernel-variant selector (benchmarking only: AbxGemm.tune of every descriptor-level launch)

# ---- range safety of the split-f16 kernels (include/abx_hip.h, AbxGemm.range_flag) ----------------------------------------------
# Every split-f16 launch made through this module carries the address of one device word per GPU and a bit that names its call-site
# class; a kernel whose accumulators are not finite - what an operand beyond the split ranges turns into - ORs its bit into the word.
# abx_amd.model.abx.ScoreNetwork clears the word before a network call, reads it after, and repeats the call on the exact fp32-MFMA
# kernels when it is set, so the range contract of the fast path never reaches a caller as a wrong or non-finite result.
RANGE_TAGS = {'gemm': 1, 'contraction': 2, 'plane_projection': 4, 'tri_mul_tail': 8, 'pair_transition': 16, 'ipa_pair_init': 32,
              'tri_attn': 64, 'ipa_tail': 128, 'heads_tail': 256, 'gemm_late': 512}     # gemm / gemm_late: the plain GEMMs in front of / behind the pair stack
RANGE_CHECK = not bool(__import__('os').environ.get('ABX_NO_RANGE_CHECK'))     # (A / B measurements of the probe's cost)
_range_words = {}


RANGE_SLOTS = 4
RANGE_SLOT = 0       # which of the device's range words the launches issued now report to (ScoreNetwork: one word per network pass, so that
                     # the pass - and with ops.RANGE_ORDER the op class - that left the range FIRST is known from one read-back per call)


def range_words(device):
    """The int32 [RANGE_SLOTS] range words of a device (created on first use, zero)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    w = _range_words.get(idx)
    if w is None:
        w = torch.zeros(RANGE_SLOTS, dtype=torch.int32, device=torch.device('cuda', idx))
        _range_words[idx] = w
    return w


def range_word(device):
    """The first range word (int32 [1] view): what every launch reports to unless ops.RANGE_SLOT says otherwise."""
    return range_words(device)[0:1]


def range_ptr(device):
    """Device address of the range word in use (ops.RANGE_SLOT)."""
    return range_words(device).data_ptr() + 4 * RANGE_SLOT


def range_names(bits):
    return [n for n, b in RANGE_TAGS.items() if bits & b]


# order in which the op classes appear in a network pass: a class that left its range hands NaN rows to every class after it, so of the
# bits a call sets only the FIRST one names an op that needs the exact kernels (abx_amd.model.abx.ScoreNetwork.forward)
RANGE_ORDER = ('gemm', 'plane_projection', 'contraction', 'tri_mul_tail', 'tri_attn', 'pair_transition', 'ipa_pair_init', 'gemm_late', 'ipa_tail', 'heads_tail')


def first_range_tag(bits, skip=0):
    """The bit of the earliest op class of a pass among `bits` that is not in `skip`, or 0."""
    for n in RANGE_ORDER:
        if bits & RANGE_TAGS[n] & ~skip:
            return RANGE_TAGS[n]
    return 0


GEMM_EXACT = False   # True: every GEMM on the exact fp32 MFMA kernel (v_mfma_f32_32x32x2_f32); default: large problems on the
                     # split-f16 kernels of csrc/gemm3.hip (fp32-accurate)


def gemm_kernel_name(M, N, K, batch, a_kcontig=True, b_ncontig=True, transposed=False, split=False, exact=None, a_split=False, dual=False, out_ln=False):
    """Name of the kernel instantiation abx_gemm launches for a problem (mirror of the selection in csrc/gemm.hip and
    csrc/gemm3.hip); used by bench.py to aggregate per KERNEL exactly like `rocprofv3 --stats` does."""
    b = lambda x: 'true' if x else 'false'
    if dual:
        return 'gemm3_dual_kernel<128, 96, 32, 96, 3>'
    if out_ln:
        return 'gemm3_oln_kernel<128, 128, 32, 128, 4>'
    exact = 1 if GEMM_EXACT else int(exact or 0)
    blocks128 = ((M + 127) // 128) * ((N + 127) // 128) * batch
    wide192 = ((N + 191) // 192) * 192 <= ((N + 127) // 128) * 128
    if exact != 1 and split and N <= 32 and a_kcontig and not a_split and K % 16 == 0:
        return f'gemm3_kernel<128, 32, 32, 32, 0, {b(transposed)}, 6>'
    if exact != 1 and split and N > 64 and (blocks128 >= 256 or exact == 2) and K % 16 == 0 and (a_kcontig or a_split or M % 4 == 0):
        amode = 2 if a_split else (0 if a_kcontig else 1)
        pad128, pad192 = ((N + 127) // 128) * 128, ((N + 191) // 192) * 192
        wide = pad192 <= pad128 if a_split else (wide192 and N % 128 != 0)
        cfg = (128, 192, 32, 192, 3) if wide else (128, 128, 32, 128, 4)
        if not wide and blocks128 < 384 and M > 64:
            cfg = (64, 128, 32, 64, 4)
        return f'gemm3_kernel<{cfg[0]}, {cfg[1]}, {cfg[2]}, {cfg[3]}, {amode}, {b(transposed)}, {cfg[4]}>'
    if N <= 32:
        cfg = (128, 32, 32, 32, 3)
    elif N <= 64:
        cfg = (128, 64, 32, 64, 3)
    elif blocks128 < 512:
        cfg = (64, 64, 32, 32, 3)
    elif wide192:
        cfg = (128, 192, 64, 96, 2)
    else:
        cfg = (128, 128, 64, 64, 3)
    return f'gemm_kernel<{cfg[0]}, {cfg[1]}, {cfg[2]}, {cfg[3]}, 16, {b(a_kcontig)}, {b(b_ncontig)}, {b(transposed)}, {cfg[4]}>'


def gemm_as_kernel_name(g, side=None):
    """Name of the A-stationary kernel instantiation (csrc/gemm_as.hip) that abx_gemm / abx_gemm_side launch for a filled descriptor, or
    None when the problem goes to the tile kernels (mirror of abx_gemm_as_dispatch; alignment is taken for granted: the tensors of
    model/forward.py are 16-byte aligned)."""
    if GEMM_EXACT or (g.tune & 2048) or g.exact == 1 or not g.B_split or not g.b_f16 or g.A_split or not g.A or g.sAk != 1 or g.K != 192:
        return None
    if g.A2 or g.out_ln_w or g.mlp or g.ln_stats or g.gate or g.resid or g.batch_inner or not g.ln_csum or not g.bias or g.act != 0 or g.alpha != 1.0:
        return None
    ntm = (g.M + 63) // 64
    if g.N < 256 or ntm * g.batch < 1024:
        return None
    if g.glu:
        if side is not None or not g.C_split or not g.c_split_tile or not g.c_transposed or g.N % 128 != 0 or not g.a_pair or g.pair_Lp <= 0:
            return None
        return 'gemm_as_kernel<1, false, 0, false>'
    if g.c_transposed or g.C_split or g.rowscale or g.a_pair_transpose > 0 or g.pair_Lp != 0 or g.batch != 1 or g.N % 64 != 0:
        return None
    if side is not None:
        s2 = side
        if g.N % 128 != 64 or not s2.B_split or not s2.c_transposed or s2.N > 32 or s2.K != g.K or s2.M * s2.batch != g.M or s2.act != 0 or s2.exact == 1:
            return None
        return 'gemm_as_kernel<0, true, 0, true>' if g.c_planes_from > 0 else 'gemm_as_kernel<0, true, 0, false>'
    return 'gemm_as_kernel<0, false, 0, false>'


SPLIT_MIN_L = 64


def gemm_mode(L):
    """Arithmetic class of the GEMMs of a network pass, fixed by the complex (residue count L) and NOT by how many samples share a
    launch: 2 (split-f16 kernels) from L = 64, 1 (exact fp32 MFMA) below.  Chunking a batch, sharding it over GPUs or running
    one sample alone therefore gives bit-identical results."""
    return 1 if (GEMM_EXACT or L < SPLIT_MIN_L) else 2


def gemm_split_eligible(M, N, K, batch=1):
    """True when abx_gemm serves an (aligned) problem of this size on the split-f16 kernels (mirror of
    abx_gemm3_dispatch in csrc/gemm3.hip)."""
    return (not GEMM_EXACT) and N > 64 and K % 16 == 0 and ((M + 127) // 128) * ((N + 127) // 128) * batch >= 256


def pack_glu_weights(Wv, Wg, bv=None, bg=None):
    """Column order of a glu GEMM: 32-column blocks [v0 | g0 | v1 | g1 | ...] of value weights Wv (K, C) and gate weights
    Wg (K, C) for the same C output channels (C % 64 == 0).  Returns (W (K, 2C), bias (2C) or None)."""
    K, C_ = Wv.shape
    assert Wg.shape == (K, C_) and C_ % 64 == 0
    W = torch.stack([Wv.reshape(K, C_ // 32, 32), Wg.reshape(K, C_ // 32, 32)], dim=2).reshape(K, 2 * C_).contiguous()
    b = None
    if bv is not None or bg is not None:
        z = torch.zeros(C_, device=Wv.device, dtype=Wv.dtype)
        b = torch.stack([(bv if bv is not None else z).reshape(C_ // 32, 32), (bg if bg is not None else z).reshape(C_ // 32, 32)],
                        dim=1).reshape(2 * C_).contiguous()
    return W, b


_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def permute_k16(Wt):
    """Wt (K, N), K % 16 == 0 -> rows reordered inside every block of 16 as 0-3, 8-11, 4-7, 12-15: the k order in which the fused
    transition (AbxGemm.mlp) feeds its second GEMM from the accumulator registers of the first (csrc/gemm3.hip)."""
    K, N = Wt.shape
    assert K % 16 == 0
    idx = torch.tensor(_PERM16, device=Wt.device)
    return Wt.reshape(K // 16, 16, N)[:, idx, :].reshape(K, N).contiguous()


class WeightPlanes(torch.Tensor):
    """float16 tensor [Kp/16][2][N][16] of split weight planes + the power-of-two exponent they were scaled with (`w_exp`)."""
    w_exp = 0
    __torch_function__ = torch._C._disabled_torch_function_impl     # a plain data carrier: no dispatch overhead on .stride() / .shape


def split_weights(Wt):
    """Wt (K, N) packed weight (n-contiguous) -> WeightPlanes [Kp/16][2][N][16] float16: the k-tiled operand image of the split-f16
    weight GEMMs (AbxGemm.b_f16, include/abx_hip.h): with w' = w * 2^w_exp, max|w'| in [2^13, 2^14):
    p0 = f16(w'), p1 = f16(w' - p0) (the kernels derive p2 = f16(p0 * 2^-11) in registers);  w' = p0 + p1 up to 2^-23 |w'| (+ 2^-25
    absolute).  One host sync (the
    maximum) per call: weights are packed once, outside any graph capture.  Kp = K rounded up to 16."""
    K, N = Wt.shape
    _f32(Wt)
    Kp = (K + 15) // 16 * 16
    amax = float(Wt.abs().max())
    if not math.isfinite(amax):
        raise ValueError('split_weights: non-finite weight')
    w_exp = 14 - math.frexp(amax)[1] if amax > 0 else 0
    w_exp = max(-100, min(100, w_exp))
    out = torch.empty(Kp // 16, 2, N, 16, device=Wt.device, dtype=torch.float16)
    check(_lib.load().abx_split_weights_f16(_p(Wt), Wt.stride(1), Wt.stride(0), N, K, w_exp, _p(out), _stream()), 'abx_split_weights_f16')
    out = out.as_subclass(WeightPlanes)
    out.w_exp = w_exp
    return out


def weights_to_float(w3):
    """WeightPlanes -> fp32 (Kp, N) value the kernels multiply with: (p0 + p1) * 2^-w_exp."""
    v = (w3[:, 0].float() + w3[:, 1].float()) * 2.0 ** (-w3.w_exp)
    return v.permute(0, 2, 1).reshape(-1, w3.shape[2])


def _weight_planes(w3, N, K=None, what='B3'):
    assert isinstance(w3, WeightPlanes) and w3.dtype == torch.float16 and w3.is_contiguous() and w3.shape[1:] == (2, N, 16), \
        f'{what}: WeightPlanes (ops.split_weights) of N = {N} expected'
    assert K is None or w3.shape[0] * 16 >= K
    return w3


def gemm(A, B, Cout, *, ln=None, a_relu=False, bias=None, alpha=1.0, act=0, rowscale=None, gate=None, gate_sigmoid=True,
         resid=None, tune=None, B3=None, exact=None, a_pair_transpose=0, glu=False, pair=None, a_pair=False, c_pair=False, dual=None, out_ln=None, clock_probe=None, mlp=None, c_split_nA=0, c_split_tile=False,
         defer=False, range_class=None, c_plane_cols=None):
    """Cout[b] = epi(A'[b] @ B[b]).  A (b,M,K) or (M,K); B (b,K,N) or (K,N) (shared); Cout (b,M,N) or (M,N) logical tensors.
    Strides decide the kernel variant: A k- or m-contiguous, B n- or k-contiguous, Cout n-contiguous or (if its last-but-one
    stride is 1) stored transposed.  ln = (stats (rows,2) | None, csum).  rowscale (b,M)|(M,), gate/resid (b,M,N)|(M,N) logical
    tensors laid out like Cout (n-contiguous, or m-contiguous when Cout is stored transposed).
    Split-f16 operands (include/abx_hip.h, "Split-f16 operands"): B3 (K/16,2,N,16) = split_weights(B), float16 weight planes;
    A (b,K/16,2,M,16) and B (b,K/16,2,N,16) both int16 tensors of activation images: the TriangleMultiplication contraction (A: the
    pieces a0, a1; B: the planes p0, p1); Cout (b,N,L/16,2,L,16) int16 (M = L*L pair rows, m = i*L + k): the output is written
    as the operand image [n][k/16][plane][i][16] of that contraction (transposed store), channels n < c_split_nA as its A side, the
    others as its B side.  a_pair_transpose=L: GEMM row i*L+k reads A row k*L+i.
    glu=True: B holds (value, gate) column pairs (pack_glu_weights); Cout has N/2 channels = value * sigmoid(gate).
    pair=(L, Lp): the M rows are padded pair positions i*Lp + j (Lp % 4 == 0, any L); a_pair: A is the UNpadded (b, L*L, K) pair
    tensor; c_pair: Cout / gate / resid are UNpadded (b, L*L, N) pair tensors (pad rows dropped).  rowscale is indexed by GEMM row.
    dual=(A2, B3_2, csum2, bias2): Cout = epi(A' B) * sigmoid(LN(A2) @ W2 + bias2) (+ resid): A2 (b, rows, K2) k-contiguous fp32 (the
    UNpadded pair tensor when pair is given), B3_2 = split_weights of the gamma-scaled gate weights (K2, N), csum2 their column sums.
    out_ln=(gamma, beta[, eps]): LayerNorm over the N output columns right after bias / alpha / act (split-f16 path only, N <= 128).
    defer=True: nothing is launched, the filled AbxGemm is returned (for gemm_side).
    mlp=(B3_2, bias2): fused two-layer transition Cout = relu(LN(A) @ B + bias) @ W2 + bias2 (+ resid); B (K, N) is the first layer
    (N = hidden width, act must be 1, ln given), B3_2 = split_weights(permute_k16(W2t)) of the second layer W2t (N, N2), Cout / resid
    have N2 <= 192 columns and may alias A.  With `gate` (M, N) fp32 rows and act = 2 the same call is the gated tail of the triangle attention
    (AbxGemm.mlp = 2): Cout = (sigmoid(LN(A) @ B + bias) * gate) @ W2 + bias2 (+ resid)."""
    lib = _lib.load()
    g = AbxGemm()
    a_planes, b_planes, c_planes = A.dtype == torch.int16, B.dtype == torch.int16, Cout.dtype == torch.int16
    if a_planes and A.dim() == 6:      # (Bo, Bi, K/16, 2, M, 16): two-level batch (channel slices of a wider per-sample tensor)
        assert B.dtype == torch.int16 and B.dim() == 6 and B.shape[:2] == A.shape[:2]
        g.batch_inner, g.sA3i, g.sB3i = A.shape[1], A.stride(1), B.stride(1)
        bo_a, bo_b = A.stride(0), B.stride(0)
        A = A.as_strided((A.shape[0] * A.shape[1],) + tuple(A.shape[2:]), (0,) + tuple(A.stride()[2:]), A.storage_offset())
        B = B.as_strided((B.shape[0] * B.shape[1],) + tuple(B.shape[2:]), (0,) + tuple(B.stride()[2:]), B.storage_offset())
    else:
        bo_a = bo_b = None
    if a_planes:
        assert A.dim() == 5 and A.shape[2] == 2 and A.shape[4] == 16 and A.stride(4) == 1
        nb, KT, _, M, _ = A.shape
        K = KT * 16
        g.A_split, g.sA3b, g.sA3k, g.sA3p, g.sA3m = _p(A), (A.stride(0) if nb > 1 else 0), A.stride(1), A.stride(2), A.stride(3)
        if bo_a is not None:
            g.sA3b = bo_a
    else:
        if A.dim() == 2:
            A = A.unsqueeze(0)
        nb, M, K = A.shape
        _f32(A)
        g.A, g.sAb, g.sAm, g.sAk = _p(A), A.stride(0) if nb > 1 else 0, A.stride(1), A.stride(2)
        if a_pair:
            assert pair is not None and M == pair[0] * pair[0], (A.shape, pair)
            M = pair[0] * pair[1]
            if c_split_tile:        # GEMM rows = pair positions in (8 i x 16 k) blocks (AbxGemm.c_split_tile)
                M = ((pair[0] + 7) // 8 * 8) * ((pair[1] + 15) // 16 * 16)
    if b_planes:
        assert B.dim() == 5 and B.shape[2] == 2 and B.shape[4] == 16 and B.stride(4) == 1 and B.shape[1] * 16 == K and B.shape[0] == nb
        N = B.shape[3]
        g.B_split, g.sB3b, g.sB3k, g.sB3p, g.sB3n = _p(B), (B.stride(0) if nb > 1 else 0), B.stride(1), B.stride(2), B.stride(3)
        if bo_b is not None:
            g.sB3b = bo_b
    else:
        if B.dim() == 2:
            B = B.unsqueeze(0)
        assert B.shape[1] == K, (A.shape, B.shape)
        N = B.shape[2]
        _f32(B)
        g.B, g.sBb, g.sBk, g.sBn = _p(B), (B.stride(0) if B.shape[0] > 1 else 0), B.stride(1), B.stride(2)
    No = N // 2 if glu else N
    if mlp is not None:
        B32m, bias2m = mlp
        No = B32m.shape[2]
        _weight_planes(B32m, No, what='mlp B3_2')
        assert B32m.shape[0] * 16 == N
        g.mlp, g.N2, g.b2_exp = (2 if gate is not None else 1), No, B32m.w_exp
        g.B2_split, g.sB23k, g.sB23p, g.sB23n = _p(B32m), B32m.stride(0), B32m.stride(1), B32m.stride(2)
        g.bias2 = _p(bias2m)
        assert act == (2 if gate is not None else 1) and ln is not None and ln[0] is None
    if c_planes:
        L = Cout.shape[4]
        Lp = pair[1] if pair is not None else L
        assert Cout.dim() == 6 and Cout.shape == (nb, No, (Lp + 15) // 16, 2, L, 16) and (c_split_tile or M == L * Lp), (Cout.shape, nb, M, N)
        assert not c_split_tile or (a_pair and pair is not None), 'c_split_tile: the A rows come through the pair-row map (a_pair, pair=(L, Lp))'
        assert Cout.stride(5) == 1 and Cout.stride(4) == 16
        g.C_split, g.sCb, g.sCm, g.sCk, g.sCp, g.c_split_L = _p(Cout), (Cout.stride(0) if nb > 1 else 0), Cout.stride(1), Cout.stride(2), Cout.stride(3), Lp
        g.c_transposed, g.c_split_nA, g.c_split_tile = 1, int(c_split_nA), int(bool(c_split_tile))
    else:
        if Cout.dim() == 2:
            Cout = Cout.unsqueeze(0)
        assert Cout.shape == (nb, pair[0] * pair[0] if c_pair else M, No), (Cout.shape, (nb, M, No))
        _f32(Cout)
        g.C = _p(Cout)
        Cl = Cout
        g.sCb = Cl.stride(0) if nb > 1 else 0
        if Cl.stride(2) == 1:
            g.c_transposed, g.sCm = 0, Cl.stride(1)
        else:
            assert Cl.stride(1) == 1, 'Cout must be n-contiguous or m-contiguous'
            g.c_transposed, g.sCm = 1, Cl.stride(2)
    g.M, g.N, g.K, g.batch = M, N, K, nb
    g.a_pair_transpose = int(a_pair_transpose)
    if pair is not None:
        g.pair_L, g.pair_Lp, g.a_pair, g.c_pair = int(pair[0]), int(pair[1]), int(bool(a_pair)), int(bool(c_pair))
        assert c_split_tile or M == pair[0] * pair[1], (M, pair)
    g.glu = int(bool(glu))
    if ln is not None:
        stats, csum = ln                     # stats None: the kernel derives (mean, rstd) from its own A stream
        assert csum.numel() == N
        g.ln_csum, g.ln_eps = _p(_f32(csum)), 1e-5
        if stats is not None:
            assert stats.numel() == 2 * nb * M, (stats.shape, nb, M)
            g.ln_stats, g.sSb = _p(_f32(stats)), M
    g.a_relu = 1 if a_relu else 0
    g.exact = 1 if GEMM_EXACT else int(exact or 0)        # 0 by problem size, 1 exact fp32 MFMA, 2 split-f16 whenever the shape allows
    if B3 is not None:
        _weight_planes(B3, N, K)
        assert not a_planes, 'weight planes go with an fp32 A (the plane x plane contraction takes activation images on both sides)'
        g.B_split, g.sB3k, g.sB3p, g.sB3n, g.sB3b = _p(B3), B3.stride(0), B3.stride(1), B3.stride(2), 0
        g.b_f16, g.b_exp = 1, B3.w_exp
    g.tune = GEMM_TUNE if tune is None else tune
    if c_plane_cols is not None:      # (from, group): the output columns n >= from as two float16 planes of 16 x value per group (AbxGemm.c_planes_from)
        g.c_planes_from, g.c_planes_group = int(c_plane_cols[0]), int(c_plane_cols[1])
    if RANGE_CHECK and g.exact != 1:
        g.range_flag = range_ptr(Cout.device)
        g.range_tag = RANGE_TAGS[('tri_attn' if gate is not None else 'pair_transition') if mlp is not None else 'tri_mul_tail' if dual is not None else 'ipa_pair_init' if out_ln is not None
                                 else 'plane_projection' if c_planes else 'contraction' if a_planes else (range_class or 'gemm')]
    if dual is not None:
        A2, B32, csum2, bias2 = dual
        if A2.dim() == 2:
            A2 = A2.unsqueeze(0)
        _f32(A2)
        assert A2.stride(2) == 1 and A2.shape[0] == nb and A2.shape[1] == (pair[0] * pair[0] if pair is not None else M)
        K2 = A2.shape[2]
        _weight_planes(B32, N, what='dual B3_2')
        assert B32.shape[0] * 16 == K2 and csum2.numel() == N
        g.b2_exp = B32.w_exp
        g.A2, g.sA2b, g.sA2m, g.K2 = _p(A2), (A2.stride(0) if nb > 1 else 0), A2.stride(1), K2
        g.B2_split, g.sB23k, g.sB23p, g.sB23n = _p(B32), B32.stride(0), B32.stride(1), B32.stride(2)
        g.ln2_csum, g.bias2 = _p(_f32(csum2)), _p(bias2)
        if g.ln_eps == 0:
            g.ln_eps = 1e-5
    if out_ln is not None:
        assert out_ln[0].numel() == N and out_ln[1].numel() == N
        g.out_ln_w, g.out_ln_b = _p(_f32(out_ln[0])), _p(_f32(out_ln[1]))
        g.out_ln_eps = float(out_ln[2]) if len(out_ln) > 2 else 1e-5
    if clock_probe is not None:       # diagnostics: uint64[2] device accumulators (shader ticks, 100 MHz ticks)
        assert clock_probe.dtype == torch.int64 and clock_probe.numel() >= 2
        g.clock_probe = _p(clock_probe)
    g.bias = _p(bias)
    g.alpha = float(alpha)
    g.act = int(act)
    if rowscale is not None:
        rs_rows = pair[0] * pair[1] if c_split_tile else M
        assert rowscale.numel() == nb * rs_rows and rowscale.is_contiguous()
        g.rowscale, g.sRSb = _p(_f32(rowscale)), rs_rows
    if gate is not None:
        if gate.dim() == 2:
            gate = gate.unsqueeze(0)
        cd, sd = (1, 2) if g.c_transposed else (2, 1)
        assert gate.shape == (nb, M if c_planes else Cout.shape[1], N) and gate.stride(cd) == 1, 'gate must be laid out like Cout (mlp: (M, N) rows)'
        g.gate, g.sGb, g.sGm, g.gate_sigmoid = _p(_f32(gate)), (gate.stride(0) if nb > 1 else 0), gate.stride(sd), int(gate_sigmoid)
    if resid is not None:
        if resid.dim() == 2:
            resid = resid.unsqueeze(0)
        cd, sd = (1, 2) if g.c_transposed else (2, 1)
        assert resid.shape == (nb, M if c_planes else Cout.shape[1], No if mlp is not None else N) and resid.stride(cd) == 1, 'resid must be laid out like Cout'
        g.resid, g.sRb, g.sRm = _p(_f32(resid)), (resid.stride(0) if nb > 1 else 0), resid.stride(sd)
    if defer:           # the filled descriptor instead of a launch (gemm_side); the caller keeps the operand tensors alive
        return g
    check(lib.abx_gemm(C.byref(g), _stream()), 'abx_gemm')
    return Cout


def gemm_side(g_main, g_side):
    """abx_gemm_side: two GEMMs over the same rows in one launch (descriptors from gemm(..., defer=True)): a plain-store split-f16
    problem with N % 128 == 0 and a skinny (N <= 32) transposed-store projection whose tiles ride in its grid and read their A panel
    from the L2; bit-identical to the two launches, which is what the library issues when the pair does not qualify."""
    check(_lib.load().abx_gemm_side(C.byref(g_main), C.byref(g_side), _stream()), 'abx_gemm_side')


def planes_to_float(p, a_side, dim=0):
    """int16 activation images (size 2 along `dim`) of the contraction -> the float32 value they stand for: A side (p0 + p1 2^-11) 2^4,
    B side (p0 + p1) 2^-4."""
    h = p.view(torch.float16).float()
    p0, p1 = h.select(dim, 0), h.select(dim, 1)
    return (p0 + p1 / 2048.0) * 16.0 if a_side else (p0 + p1) / 16.0


def row_stats(x, out=None, eps=1e-5):
    """(mean, rstd) per row.  x (rows,K) k-contiguous (dense rows), or x (b, K, rows) channel-major given as a
    (b, rows, K) logical view whose row stride is 1."""
    lib = _lib.load()
    _f32(x)
    if x.dim() == 2:
        rows, K = x.shape
        assert x.stride(1) == 1
        if out is None:
            out = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        check(lib.abx_row_stats(_p(x), 0, x.stride(0), 1, 1, rows, K, eps, _p(out), _stream()), 'abx_row_stats')
    else:
        nb, rows, K = x.shape
        assert x.stride(1) == 1, 'channel-major layout expected'
        if out is None:
            out = torch.empty(nb * rows, 2, device=x.device, dtype=torch.float32)
        check(lib.abx_row_stats(_p(x), x.stride(0), 1, x.stride(2), nb, rows, K, eps, _p(out), _stream()), 'abx_row_stats')
    return out


def layernorm(x, gamma, beta, out=None, res=None, eps=1e-5):
    lib = _lib.load()
    rows, K = x.shape
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty(rows, K, device=x.device, dtype=torch.float32)
    check(lib.abx_layernorm(_p(_f32(x)), x.stride(0), rows, K, _p(gamma), _p(beta), eps, _p(out), out.stride(0),
                            _p(res), res.stride(0) if res is not None else 0, _stream()), 'abx_layernorm')
    return out


def transpose_last2(x, out, transpose=True):
    """x (n, L, L) -> out (n, L, Lp): out[n, a, b] = x[n, b, a] (or x[n, a, b] when transpose=False), pad columns b >= L zeroed."""
    L, Lp = x.shape[-1], out.shape[-1]
    assert x.shape[-2] == L and x.is_contiguous() and out.is_contiguous() and out.shape[:-1] == x.shape[:-1] and Lp >= L
    check(_lib.load().abx_transpose_last2(_p(_f32(x)), _p(out), x.numel() // (L * L), L, Lp, int(bool(transpose)), _stream()),
          'abx_transpose_last2')
    return out


# ---- op-group entry points of the C ABI (include/abx_hip.h, csrc/blocks.hip) ---------------------------------------------------------
class LinearPack:
    """A Linear (optionally with its LayerNorm folded in) packed by abx_pack_linear: .c is the AbxLinearPack, the device buffer it points
    into is kept alive here; .Wt / .csum / .bias / .planes are tensor views of the same memory for the descriptor-level calls."""

    def __init__(self, sources, K, ln=None, permute_k16=False, device=None):
        """sources: list of (W (rows, K), b (rows) or None, glu in {0, 1, 2}); ln: (gamma, beta) of the LayerNorm to fold or None."""
        lib = _lib.load()
        dev = device if device is not None else sources[0][0].device
        N = sum(int(W.shape[0]) for W, _, _ in sources)
        nbytes = int(lib.abx_pack_linear_bytes(K, N))
        self.buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-self.buf.data_ptr()) % 256
        base = self.buf[off:off + nbytes]
        keep = []
        arr = (AbxLinearSrc * len(sources))()
        for i, (W, b, glu) in enumerate(sources):
            W = _f32(W).contiguous()
            assert W.shape[1] == K
            keep.append(W)
            arr[i].W, arr[i].rows, arr[i].glu = _p(W), int(W.shape[0]), int(glu)
            if b is not None:
                b = _f32(b).contiguous()
                keep.append(b)
                arr[i].b = _p(b)
        gamma = beta = None
        if ln is not None:
            gamma, beta = _f32(ln[0]).contiguous(), _f32(ln[1]).contiguous()
        self.c = AbxLinearPack()
        check(lib.abx_pack_linear(arr, len(sources), K, _p(gamma), _p(beta), 1 if permute_k16 else 0, base.data_ptr(), C.byref(self.c), _stream()),
              'abx_pack_linear')
        self.K, self.N, self.w_exp = K, N, int(self.c.b_exp)

        def view(ptr, numel, dtype):
            if not ptr:
                return None
            o = ptr - base.data_ptr()
            return base[o:o + numel * torch.empty(0, dtype=dtype).element_size()].view(dtype)

        Kp = (K + 15) // 16 * 16
        self.Wt = view(self.c.Wt, K * N, torch.float32).view(K, N)
        self.csum = view(self.c.csum, N, torch.float32)
        self.bias = view(self.c.bias, N, torch.float32)
        pl = view(self.c.planes, Kp * 2 * N, torch.float16).view(Kp // 16, 2, N, 16).as_subclass(WeightPlanes)
        pl.w_exp = self.w_exp
        self.planes = pl


def _range_args(exact, name):
    if RANGE_CHECK and not exact:
        return range_ptr(torch.device('cuda', torch.cuda.current_device())), RANGE_TAGS[name]
    return None, 0


def transition_fwd(l1, l2, z, exact=False, workspace=None):
    """pair Transition in place on z (M, C) through abx_transition_fwd (one launch on the split-f16 path)."""
    lib = _lib.load()
    M = z.shape[0]
    assert z.is_contiguous() and z.shape[1] == l1.K
    if exact:
        need = int(lib.abx_transition_workspace_bytes(M, l1.N, 1))
        assert workspace is not None and workspace.numel() * workspace.element_size() >= need
    rf, rt = _range_args(exact, 'pair_transition')
    check(lib.abx_transition_fwd(C.byref(l1.c), C.byref(l2.c), _p(z), M, int(bool(exact)), _p(workspace), rf, rt, _stream()), 'abx_transition_fwd')
    return z


def tri_mul_pack(glu, out, gate):
    p = AbxTriMulPack()
    p.glu, p.out, p.gate = glu.c, out.c, gate.c
    p._keep = (glu, out, gate)
    return p


def tri_mul_workspace(B, L, device):
    """(workspace tensor, initialised) for abx_tri_mul_fwd: the operand-image region is zero-filled once."""
    lib = _lib.load()
    ws = torch.empty(int(lib.abx_tri_mul_workspace_bytes(B, L)) + 256, dtype=torch.uint8, device=device)
    ws = ws[(-ws.data_ptr()) % 256:]
    check(lib.abx_tri_mul_workspace_init(_p(ws), B, L, _stream()), 'abx_tri_mul_workspace_init')
    return ws


def tri_mul_workspace_init(ws, B, L):
    """Zero the operand-image region of a tri-mul workspace for the layout of B samples (a buffer reused with another chunk size)."""
    check(_lib.load().abx_tri_mul_workspace_init(_p(ws), B, L, _stream()), 'abx_tri_mul_workspace_init')


def tri_mul_fwd(pack, z_in, z_out, mask_f, B, L, outgoing, workspace):
    lib = _lib.load()
    assert z_in.is_contiguous() and z_out.is_contiguous() and z_in.data_ptr() != z_out.data_ptr()
    rf, rt = _range_args(False, 'contraction')
    check(lib.abx_tri_mul_fwd(C.byref(pack), _p(z_in), _p(z_out), _p(_f32(mask_f)), B, L, int(bool(outgoing)), _p(workspace), rf, rt, _stream()),
          'abx_tri_mul_fwd')
    return z_out


def tri_attn_pack(qkv, gate, pair, out):
    p = AbxTriAttnPack()
    p.qkv, p.gate, p.pair, p.out = qkv.c, gate.c, pair.c, out.c
    p._keep = (qkv, gate, pair, out)
    return p


def tri_attn_block_workspace(B, L, device):
    ws = torch.empty(int(_lib.load().abx_tri_attn_block_workspace_bytes(B, L)) + 256, dtype=torch.uint8, device=device)
    return ws[(-ws.data_ptr()) % 256:]


def tri_attn_block_fwd(pack, z, mask_f, B, L, per_row, workspace, exact=False, attn_exact=None):
    """exact: the three GEMMs on the exact kernels; attn_exact: the attention kernel (default: the global GEMM_EXACT switch, as tri_attn)."""
    lib = _lib.load()
    ax = bool(GEMM_EXACT if attn_exact is None else attn_exact)
    ex = int(bool(exact)) | (2 if ax else 0)
    rf, rt = _range_args(ex == 3, 'tri_attn')
    check(lib.abx_tri_attn_block_fwd(C.byref(pack), _p(z), _p(_f32(mask_f)), B, L, int(bool(per_row)), ex, _p(workspace), rf, rt, _stream()),
          'abx_tri_attn_block_fwd')
    return z


def tri_attn_kernel_name(L, exact=None, bias_vec=True):
    """Name of the kernel abx_tri_attn_fwd launches (mirror of the selection in csrc/attention.hip), for per-kernel aggregation."""
    if GEMM_EXACT if exact is None else exact:
        return 'tri_attn_kernel'
    kc = 128 if (L + 127) // 128 == (L + 191) // 192 else 192
    return 'tri_attn8_kernel<%d, 768, %s>' % (kc, 'true' if bias_vec else 'false')


# float(log2 e) * 2^7 (include/abx_hip.h ABX_TRI_BIAS_LOG2): the factor the pair-bias projection applies (gemm(..., alpha=)) when the attention
# that reads it runs with bias_log2=True
TRI_BIAS_LOG2 = float(__import__('numpy').float32(1.4426950408889634) * __import__('numpy').float32(128.0))


KV_PLANES = bool(__import__('os').environ.get('ABX_KV_PLANES'))       # round-6 experiment, off by default (measured slower: profiles/r06h_kb_kvplanes.txt)


def kv_planes_ok(M):
    """True when a q | k | v projection over M pair rows may write its k | v columns as operand images (gemm(..., c_plane_cols=(192, 48)) through
    gemm_side) for tri_attn(..., kv_planes=True): the launch takes the A-stationary kernel (abx_gemm_planes_ok)."""
    return bool(_lib.load().abx_gemm_planes_ok(int(M))) and not GEMM_EXACT and not (GEMM_TUNE & 2048)


def tri_attn(qkvg, biasT, keymask, out, B, L, per_row, H=4, D=48, bias_is_qk=False, exact=None, clock_probe=None, tune=0, bias_log2=False, kv_planes=False):
    """qkvg (B*L*L, 4*H*D) = [q|k|v|gate], or (B*L*L, 3*H*D) = [q|k|v]: no gate (the gated tail applies it: gemm(..., mlp=, gate=));
    biasT (B,H,L,L) projected from the UNtransposed pair tensor (bias_is_qk=False) or
    already laid out [b,h,q,k] for this orientation (bias_is_qk=True; then (B,H,L,Lp) with rows padded to Lp % 4 == 0 floats gives
    the kernel 16-byte bias loads for any L); out (B*L*L, H*D)."""
    lib = _lib.load()
    W = qkvg.shape[1]
    assert W in (3 * H * D, 4 * H * D) and qkvg.is_contiguous() and out.is_contiguous() and biasT.is_contiguous()
    a = AbxTriAttn()
    es = qkvg.element_size()
    base = qkvg.data_ptr()
    a.q, a.k, a.v = base, base + H * D * es, base + 2 * H * D * es
    if W == 4 * H * D:
        a.gate = base + 3 * H * D * es
    a.sb = L * L * W
    a.ss, a.sl = (L * W, W) if per_row else (W, L * W)
    a.bias = _p(biasT)
    Lp = biasT.shape[-1] if (biasT.dim() == 4 and bias_is_qk) else L       # (B,H,L,Lp): key-contiguous rows padded to Lp floats
    assert biasT.numel() == B * H * L * Lp
    a.bias_sb, a.bias_sh = H * L * Lp, L * Lp
    a.bias_sq, a.bias_sk = (Lp, 1) if (per_row or bias_is_qk) else (1, L)
    if keymask is not None:
        a.keymask, a.km_sb = _p(_f32(keymask)), L
    C_ = H * D
    a.out, a.ob = _p(out), L * L * C_
    a.os, a.ol = (L * C_, C_) if per_row else (C_, L * C_)
    a.B, a.S, a.L, a.H, a.D = B, L, L, H, D
    a.scale = float(D ** (-0.5))
    a.exact = int(GEMM_EXACT if exact is None else exact)
    a.tune = int(tune)
    a.bias_log2 = int(bool(bias_log2))              # biasT = TRI_BIAS_LOG2 x the pair bias (AbxTriAttn.bias_log2)
    a.kv_planes = int(bool(kv_planes))              # k | v columns hold the operand images of AbxGemm.c_planes_from, not fp32 (AbxTriAttn.kv_planes)
    if RANGE_CHECK and not a.exact:
        a.range_flag, a.range_tag = range_ptr(out.device), RANGE_TAGS['tri_attn']
    if clock_probe is not None:
        assert clock_probe.dtype == torch.int64 and clock_probe.numel() >= 2
        a.clock_probe = _p(clock_probe)
    check(lib.abx_tri_attn_fwd(C.byref(a), _stream()), 'abx_tri_attn_fwd')
    return out


def seq_attn(qkv, biasT, keymask, gate, out, B, L, H=32, D=17):
    lib = _lib.load()
    assert qkv.is_contiguous() and biasT.is_contiguous() and gate.is_contiguous() and out.is_contiguous()
    check(lib.abx_seq_attn_fwd(_p(qkv), _p(biasT), _p(keymask), _p(gate), _p(out), B, L, H, D, float(D ** (-0.5)), _stream()),
          'abx_seq_attn_fwd')
    return out


def ipa_weights(qpack, kpack, vpack, bias2d, mask, rots, trans, pw, attn_ws, feat, B, L):
    """First launch of the IPA core: attention weights -> attn_ws, scalar / point outputs -> feat[:, :576]."""
    assert attn_ws.numel() >= B * L * L * 12 and attn_ws.is_contiguous()
    check(_lib.load().abx_ipa_weights(_p(qpack), _p(kpack), _p(vpack), _p(bias2d), _p(mask), _p(rots), _p(trans), _p(pw), _p(attn_ws),
                                      _p(feat), B, L, _stream()), 'abx_ipa_weights')


def ipa_pair(attn_ws, z, feat, B, L):
    """Second launch: attention over the pair slab z (B*L*L, 128) -> feat[:, 576:]."""
    check(_lib.load().abx_ipa_pair(_p(attn_ws), _p(z), _p(feat), B, L, _stream()), 'abx_ipa_pair')


def gemm_splitk(A, w3, partial, range_class='gemm'):
    """K-slice products of A (M, K) fp32 rows against split_weights planes w3 of a (K, N) weight: partial[s] = A[:, s*Ks:(s+1)*Ks] @
    W[s*Ks:(s+1)*Ks] for the S = partial.shape[0] slices (Ks = K / S a multiple of 16), ONE abx_gemm launch with batch = slice (operand
    windows by batch strides; split-f16 arithmetic).  The caller adds the slices in a fixed order (abx_ipa_tail's `partial` input): a
    long-K, few-row GEMM becomes S times as many blocks with a 1 / S as long k loop."""
    S, M, N = partial.shape
    K = A.shape[1]
    assert A.shape[0] == M and A.stride(1) == 1 and partial.is_contiguous() and K % (16 * S) == 0
    _weight_planes(w3, N, K, what='gemm_splitk')
    assert w3.shape[0] * 16 == K
    Ks = K // S
    g = AbxGemm()
    g.A, g.sAb, g.sAm, g.sAk = _p(_f32(A)), Ks, A.stride(0), 1
    g.B_split, g.sB3b, g.sB3k, g.sB3p, g.sB3n = _p(w3), (Ks // 16) * w3.stride(0), w3.stride(0), w3.stride(1), w3.stride(2)
    g.b_f16, g.b_exp = 1, w3.w_exp
    g.C, g.sCb, g.sCm = _p(_f32(partial)), M * N, N
    g.M, g.N, g.K, g.batch = M, N, Ks, S
    g.alpha, g.exact = 1.0, 2
    if RANGE_CHECK:
        g.range_flag, g.range_tag = range_ptr(A.device), RANGE_TAGS[range_class]
    check(_lib.load().abx_gemm(C.byref(g), _stream()), 'abx_gemm(split-K)')
    return partial


def ipa_tail(feat, s, w_final, ln1, w_t0, w_t2, w_t4, ln2, eps=1e-5, affine=None, rigid=None, partial=None):
    """The tail of an IPA layer in one launch (csrc/gemm3.hip ipa_tail_kernel; reference score_network.py:126-163):
    s <- LN1(s + feat @ W_final + b_final);  s <- LN2(s + relu(relu(s @ W0 + b0) @ W2 + b2) @ W4 + b4), in place.
    feat (M, K1) and s (M, 256) fp32 rows; w_* = (WeightPlanes of the (K, 256) weight, bias (256)); ln* = (gamma, beta).
    affine = (Wt (256, 6) fp32, bias (6)) with rigid = (fixed_i32, init_q, init_t, cur_q, cur_t, cur_R, delta_q, position_scale): also
    affine_update of the new s and the frame update of rigid_update() in the same launch.
    partial (S, M, 256): feat @ W_final already computed as S K-slice products (gemm_splitk); the kernel adds them instead of walking K1."""
    M, K1 = feat.shape
    assert s.shape == (M, 256) and feat.stride(1) == 1 and s.stride(1) == 1 and K1 % 16 == 0
    a = AbxIpaTail()
    a.feat, a.s_feat, a.s, a.s_s, a.M, a.K1, a.C = _p(_f32(feat)), feat.stride(0), _p(_f32(s)), s.stride(0), M, K1, 256
    if partial is not None:       # feat @ W_final as K-slice products (gemm_splitk), summed by the kernel in slice order
        assert partial.dim() == 3 and partial.shape[1:] == (M, 256) and partial.is_contiguous()
        a.partial, a.n_partial, a.s_partial = _p(_f32(partial)), partial.shape[0], M * 256
    for tag, (w3, bias), K in (('final', w_final, K1), ('t0', w_t0, 256), ('t2', w_t2, 256), ('t4', w_t4, 256)):
        _weight_planes(w3, 256, K, what='ipa_tail ' + tag)
        assert w3.shape[0] * 16 == K and bias.numel() == 256
        setattr(a, 'W_' + tag, _p(w3)); setattr(a, 'e_' + tag, w3.w_exp); setattr(a, 'b_' + tag, _p(_f32(bias)))
    a.ln1_w, a.ln1_b, a.ln2_w, a.ln2_b = _p(_f32(ln1[0])), _p(_f32(ln1[1])), _p(_f32(ln2[0])), _p(_f32(ln2[1]))
    a.ln_eps = float(eps)
    if affine is not None:
        wa, ba = affine
        fixed, init_q, init_t, cur_q, cur_t, cur_R, delta_q, pscale = rigid
        assert wa.shape == (256, 6) and wa.is_contiguous() and ba.numel() == 6 and fixed.dtype == torch.int32 and fixed.numel() == M
        a.W_aff, a.b_aff, a.fixed = _p(_f32(wa)), _p(_f32(ba)), _p(fixed)
        a.init_q, a.init_t, a.cur_q, a.cur_t, a.cur_R, a.delta_q = [_p(_f32(t)) for t in (init_q, init_t, cur_q, cur_t, cur_R, delta_q)]
        a.pscale = float(pscale)
    if RANGE_CHECK:
        a.range_flag, a.range_tag = range_ptr(s.device), RANGE_TAGS['ipa_tail']
    check(_lib.load().abx_ipa_tail(C.byref(a), _stream()), 'abx_ipa_tail')
    return s


def pad_planes_128(wt, bias):
    """(WeightPlanes, bias[128]) of a (K, n <= 128) weight zero-padded to 128 columns: the narrow last projections of abx_heads_tail."""
    K, n = wt.shape
    w = torch.zeros(K, 128, device=wt.device, dtype=torch.float32)
    w[:, :n] = wt
    b = torch.zeros(128, device=wt.device, dtype=torch.float32)
    if bias is not None:
        b[:n] = bias
    return split_weights(w), b


def heads_tail(s, s0, torsion, seq_head, plddt_head, un, logits, pl=None, eps=1e-5):
    """The per-residue heads in one launch (csrc/gemm3.hip heads_tail_kernel; reference sidechain.py:28-62, head.py:143-226).
    s, s0 (M, 256) fp32 rows; torsion = 7 x (WeightPlanes (K, 128), bias [128]): proj_act, proj_init_act, the four ResNet linears, the
    projection padded to 128 columns (pad_planes_128); seq_head / plddt_head = ((gamma, beta), (planes, bias) x 3), last one padded;
    un (M, 14), logits (M, 20), pl (M, 50) contiguous outputs (pl None: the pLDDT head is skipped)."""
    M = s.shape[0]
    assert s.shape == (M, 256) and s0.shape == (M, 256) and s.stride(1) == 1 and s0.stride(1) == 1
    assert un.shape == (M, 14) and un.is_contiguous() and logits.shape == (M, 20) and logits.is_contiguous()
    a = AbxHeadsTail()
    a.s, a.s_s, a.s0, a.s_s0, a.M = _p(_f32(s)), s.stride(0), _p(_f32(s0)), s0.stride(0), M
    keep = []

    def put(tag, wb, K):
        w3, bias = wb
        _weight_planes(w3, 128, K, what='heads_tail ' + tag)
        assert w3.shape[0] * 16 == K and bias.numel() == 128
        setattr(a, 'W_' + tag, _p(w3)); setattr(a, 'e_' + tag, w3.w_exp); setattr(a, 'b_' + tag, _p(_f32(bias)))
        keep.append((w3, bias))
    for tag, wb, K in zip(('act', 'init', 'r0', 'r1', 'r2', 'r3', 'proj'), torsion, (256, 256, 128, 128, 128, 128, 128)):
        put(tag, wb, K)
    (g_s, b_s), *lin_s = seq_head
    a.lns_w, a.lns_b = _p(_f32(g_s)), _p(_f32(b_s))
    for tag, wb, K in zip(('s1', 's3', 's5'), lin_s, (256, 128, 128)):
        put(tag, wb, K)
    a.un, a.logits = _p(_f32(un)), _p(_f32(logits))
    if pl is not None:
        assert pl.shape == (M, 50) and pl.is_contiguous()
        (g_p, b_p), *lin_p = plddt_head
        a.lnp_w, a.lnp_b = _p(_f32(g_p)), _p(_f32(b_p))
        for tag, wb, K in zip(('p1', 'p3', 'p5'), lin_p, (256, 128, 128)):
            put(tag, wb, K)
        a.pl = _p(_f32(pl))
    a.ln_eps = float(eps)
    if RANGE_CHECK:
        a.range_flag, a.range_tag = range_ptr(s.device), RANGE_TAGS['heads_tail']
    check(_lib.load().abx_heads_tail(C.byref(a), _stream()), 'abx_heads_tail')


def ipa_qpack_numel(B, L):
    """floats in the Q pack of abx_ipa_pack (query rows padded to blocks of 12)."""
    return _lib.load().abx_ipa_qpack_bytes(B, L) // 4


def ipa_pack(proj, rots, trans, qpack, kpack, vpack, B, L, w_s):
    assert qpack.numel() >= ipa_qpack_numel(B, L)
    check(_lib.load().abx_ipa_pack(_p(proj), _p(rots), _p(trans), _p(qpack), _p(kpack), _p(vpack), B, L, float(w_s), _stream()),
          'abx_ipa_pack')


def ipa_attn(qpack, kpack, vpack, bias2d, z, mask, rots, trans, pw, feat, B, L, attn_ws=None):
    """attn_ws: (B*L*L, 12) fp32 scratch for the attention weights (allocated here when the caller has no workspace)."""
    if attn_ws is None:
        attn_ws = torch.empty(_lib.load().abx_ipa_attn_workspace_bytes(B, L) // 4, device=feat.device, dtype=torch.float32)
    assert attn_ws.numel() >= B * L * L * 12 and attn_ws.is_contiguous()
    check(_lib.load().abx_ipa_attn(_p(qpack), _p(kpack), _p(vpack), _p(bias2d), _p(z), _p(mask), _p(rots), _p(trans), _p(pw),
                                   _p(attn_ws), _p(feat), B, L, _stream()), 'abx_ipa_attn')


_FREQS = {}


def timestep_embedding(t64, dim, out):
    import math
    assert t64.dtype == torch.float64
    key = (dim, str(t64.device))
    if key not in _FREQS:       # seqformer.py:58-59, computed by torch on the host exactly as the reference does
        half = dim // 2
        _FREQS[key] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(t64.device)
    check(_lib.load().abx_timestep_embedding(_p(t64), _p(_FREQS[key]), t64.shape[0], dim, _p(out), _stream()),
          'abx_timestep_embedding')
    return out


def assemble_seq(seq_static, aa_table, seq_t, Lab, temb, prev_seq, gamma, beta, out, B, L, C_, E):
    ss_b = 0 if seq_static.shape[0] == 1 else seq_static.stride(0)
    assert seq_t.dtype == torch.int64 and seq_t.is_contiguous()
    check(_lib.load().abx_assemble_seq(_p(seq_static), ss_b, _p(aa_table), _p(seq_t), Lab, _p(temb), _p(prev_seq), _p(gamma),
                                       _p(beta), _p(out), B, L, C_, E, _stream()), 'abx_assemble_seq')
    return out


def assemble_pair(pair_static, temb, prev_pair, gamma, beta, prev_pos, pos_table, out, B, L, C_, E, stats_out=None):
    ps_b = 0 if pair_static.shape[0] == 1 else pair_static.stride(0)
    if prev_pos is not None:
        assert prev_pos.dtype == torch.int64 and prev_pos.is_contiguous()
    check(_lib.load().abx_assemble_pair(_p(pair_static), ps_b, _p(temb), _p(prev_pair), _p(gamma), _p(beta), _p(prev_pos),
                                        _p(pos_table), _p(out), _p(stats_out), B, L, C_, E, _stream()), 'abx_assemble_pair')
    return out


def assemble_pair_bias(pair_static, temb, prev_pair, gamma, beta, prev_pos, pos_table, out, w3, csum, bias, biasT, B, L, range_class='gemm'):
    """abx_assemble_pair_bias: z0 = assemble_pair(...) -> out (B, L, L, 192) and the sequence attention's pair bias
    biasT (B, 32, L*L) = Linear(LayerNorm(z0)) from the SAME pass over the pair rows (w3 = split_weights of the gamma-scaled (192, 32) weight, csum
    its column sums, bias the folded bias: Packed.ln_linear + split_narrow)."""
    ps_b = 0 if pair_static.shape[0] == 1 else pair_static.stride(0)
    assert out.is_contiguous() and out.shape[-1] == 192 and pair_static.shape[-1] == 128 and temb.shape[-1] == 32
    assert tuple(w3.shape) == (12, 2, 32, 16) and biasT.is_contiguous() and biasT.numel() == B * 32 * L * L and csum.numel() == 32
    if prev_pos is not None:
        assert prev_pos.dtype == torch.int64 and prev_pos.is_contiguous()
    flag, tag = (range_ptr(out.device), RANGE_TAGS[range_class]) if RANGE_CHECK else (None, 0)
    check(_lib.load().abx_assemble_pair_bias(_p(pair_static), ps_b, _p(temb), _p(prev_pair), _p(gamma), _p(beta), _p(prev_pos), _p(pos_table), _p(out),
                                             _p(w3), w3.w_exp, _p(_f32(csum)), _p(bias), 1e-5, _p(biasT), B, L, flag, tag, _stream()),
          'abx_assemble_pair_bias')
    return out


def opm_features(lr, feat, B, L, C_=64):
    """lr (B*L, 2*C) = [left | right] (already masked)."""
    es = lr.element_size()
    check(_lib.load().abx_opm_features(lr.data_ptr(), lr.data_ptr() + C_ * es, lr.stride(0), _p(feat), B, L, C_, _stream()),
          'abx_opm_features')
    return feat


def opm_out(lr, Wt, bias, z, B, L, range_class='gemm'):
    """OuterProductMean without its feature tensor (abx_opm_out_fwd): z (B*L*L, 192) += out_proj([l_j * r_i | l_j - r_i]) with lr (B*L, 128) =
    [left | right] (already masked) and Wt = out_proj.weight^T (128, 192); split-f16 arithmetic, range-tagged like the GEMMs."""
    assert lr.shape[1] == 128 and tuple(Wt.shape) == (128, 192) and Wt.is_contiguous() and z.is_contiguous() and z.shape[-1] == 192
    flag, tag = (range_ptr(z.device), RANGE_TAGS[range_class]) if RANGE_CHECK else (None, 0)
    check(_lib.load().abx_opm_out_fwd(_p(_f32(lr)), lr.stride(0), _p(_f32(Wt)), _p(bias), _p(_f32(z)), B, L, flag, tag, _stream()), 'abx_opm_out_fwd')
    return z


def pair_mask(mask_f, out, B, L, Lp=None):
    """out (B, L, Lp) = mask_i * mask_j, zero in the pad columns j >= L (Lp defaults to L)."""
    check(_lib.load().abx_pair_mask(_p(mask_f), _p(out), B, L, L if Lp is None else Lp, _stream()), 'abx_pair_mask')
    return out


def gather_rows(table, idx, out, rowscale=None):
    """out[r, :C] = table[idx[r]] (* rowscale[r]); out may be a column window of a wider buffer."""
    idx = idx.reshape(-1)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and table.is_contiguous()
    n, C_ = idx.numel(), table.shape[1]
    check(_lib.load().abx_gather_rows(_p(table), _p(idx), _p(rowscale), _p(out), out.stride(0), n, C_, _stream()), 'abx_gather_rows')
    return out


def relpos_block(residx, table, out, B, L, Lab, max_rel):
    assert residx.dtype == torch.int32 and residx.is_contiguous()
    check(_lib.load().abx_relpos_block(_p(residx), _p(table), _p(out), B, L, Lab, table.shape[1], max_rel, _stream()),
          'abx_relpos_block')
    return out


def pair_embed_features(aa, chain_id, residx, atom14, exists_u8, aa_pair_embed, relpos_embed, distcoef, dgram_embed, sq_breaks,
                        feat512, dist196, B, L):
    assert aa.dtype == torch.int64 and chain_id.dtype == torch.int32 and residx.dtype == torch.int32 and exists_u8.dtype == torch.uint8
    check(_lib.load().abx_pair_embed_features(_p(aa), _p(chain_id), _p(residx), _p(atom14), _p(exists_u8), _p(aa_pair_embed),
                                              _p(relpos_embed), _p(distcoef), _p(dgram_embed), _p(sq_breaks), _p(feat512),
                                              _p(dist196), B, L, _stream()), 'abx_pair_embed_features')


def frames_init(rigids_t, init_q, init_t, cur_q, cur_t, cur_R, delta_q, n, pscale):
    assert rigids_t.is_contiguous() and rigids_t.dtype in (torch.float32, torch.float64)
    check(_lib.load().abx_frames_init(_p(rigids_t), int(rigids_t.dtype == torch.float64), _p(init_q), _p(init_t), _p(cur_q),
                                      _p(cur_t), _p(cur_R), _p(delta_q), n, float(pscale), _stream()), 'abx_frames_init')


def rigid_update(upd6, fixed_i32, init_q, init_t, cur_q, cur_t, cur_R, delta_q, n, pscale):
    assert fixed_i32.dtype == torch.int32
    check(_lib.load().abx_rigid_update(_p(upd6), _p(fixed_i32), _p(init_q), _p(init_t), _p(cur_q), _p(cur_t), _p(cur_R),
                                       _p(delta_q), n, float(pscale), _stream()), 'abx_rigid_update')


def scores(**kw):
    a = AbxScoreArgs()
    for k, v in kw.items():
        setattr(a, k, _p(v) if torch.is_tensor(v) else v)
    check(_lib.load().abx_scores(C.byref(a), _stream()), 'abx_scores')


def torsion_finalize(unnorm, gt, fixed_i32, angles, n):
    check(_lib.load().abx_torsion_finalize(_p(unnorm), _p(gt), _p(fixed_i32), _p(angles), n, _stream()), 'abx_torsion_finalize')


def seq_head_atoms(logits, fixed_i32, seq_t, rigids, angles, a37to14, dframes, gidx, lit, seq_0, atom14, atom37, n):
    assert seq_t.dtype == torch.int64 and a37to14.dtype == torch.int64 and gidx.dtype == torch.int32
    check(_lib.load().abx_seq_head_atoms(_p(logits), _p(fixed_i32), _p(seq_t), _p(rigids), _p(angles), _p(a37to14), _p(dframes),
                                         _p(gidx), _p(lit), _p(seq_0), _p(atom14), _p(atom37), n, _stream()), 'abx_seq_head_atoms')


def prev_pos(atom37, sq_breaks, out, B, L):
    assert out.dtype == torch.int64
    check(_lib.load().abx_prev_pos(_p(atom37), _p(sq_breaks), sq_breaks.numel(), _p(out), B, L, _stream()), 'abx_prev_pos')
    return out


def plddt(logits, out, n, bins):
    check(_lib.load().abx_plddt(_p(logits), _p(out), n, bins, _stream()), 'abx_plddt')
    return out


def igso3_tables(sigma, omega, pdf, cdf, score_norms, L_terms=1000):
    check(_lib.load().abx_igso3_tables(_p(sigma), _p(omega), sigma.numel(), omega.numel(), L_terms, _p(pdf), _p(cdf),
                                       _p(score_norms), _stream()), 'abx_igso3_tables')


def reverse_step(**kw):
    a = AbxReverseArgs()
    for k, v in kw.items():
        setattr(a, k, _p(v) if torch.is_tensor(v) else v)
    check(_lib.load().abx_reverse_step(C.byref(a), _stream()), 'abx_reverse_step')


_VDW = {'C': 1.7, 'N': 1.55, 'O': 1.52, 'S': 1.8}       # abx/common/residue_constants.py:381-386
_RADIUS = {}


def vdw_radius_table(device):
    """(21, 14) van-der-Waals radius of every atom14 slot by element (0 for empty slots)."""
    key = str(device)
    if key not in _RADIUS:
        from abx_amd import residue_constants as rc
        t = torch.zeros(21, 14)
        for i, r in enumerate(rc.restypes):
            for j, name in enumerate(rc.restype_name_to_atom14_names[rc.restype_1to3[r]]):
                if name:
                    t[i, j] = _VDW[name[0]]
        _RADIUS[key] = t.to(device).contiguous()
    return _RADIUS[key]


_A14MASK = {}


def atom14_mask_table(device):
    """(21, 14) bool: which atom14 slots a residue type fills (cached per device: no host-to-device copy on the step path)."""
    key = str(device)
    if key not in _A14MASK:
        from abx_amd import residue_constants as rc
        _A14MASK[key] = torch.as_tensor(rc.restype_atom14_mask).to(device=device, dtype=torch.bool).contiguous()
    return _A14MASK[key]


def clash_grad(atom14, atom_mask, aatype, chain_id, frame_trans, overlap_tolerance=1.5, between_chain_factor=0.2,
               bond_tolerance_factor=12.0, w_clash=1.0, w_bond=1.0, w_angle=1.0, residx=None):
    """Violation energies and gradients (abx_clash_grad).  atom14 (B,L,14,3) f32, atom_mask (B,L,14), aatype (B,L) int64,
    chain_id (B,L) int32, frame_trans (B,L,3); residx (B,L) int32 or None (None: neighbours are linked by chain id alone, the
    rule of cal_vio.py:51).  -> energy (B,3) [clash, bond, angle], grad_atom (B,L,14,3), grad_trans, grad_rot (B,L,3)."""
    lib = _lib.load()
    B, L = aatype.shape
    dev = atom14.device
    a = AbxGuidanceArgs()
    x = _f32(atom14).contiguous()
    m = atom_mask.to(torch.uint8).contiguous()
    aa = aatype.to(torch.int64).contiguous()
    ch = chain_id.to(torch.int32).contiguous()
    ft = _f32(frame_trans).contiguous()
    energy = torch.empty(B, 3, device=dev)
    g_atom = torch.empty(B, L, 14, 3, device=dev)
    g_t = torch.empty(B, L, 3, device=dev)
    g_r = torch.empty(B, L, 3, device=dev)
    ws = torch.empty(max(int(lib.abx_clash_grad_workspace_bytes(B, L)) // 4, 1), device=dev)
    a.atom14, a.atom_mask, a.aatype, a.chain_id, a.radius, a.frame_trans = _p(x), _p(m), _p(aa), _p(ch), _p(vdw_radius_table(dev)), _p(ft)
    if residx is not None:
        ri = residx.to(torch.int32).contiguous()
        a.residx = _p(ri)
    a.overlap_tolerance, a.between_chain_factor, a.bond_tolerance_factor = float(overlap_tolerance), float(between_chain_factor), float(bond_tolerance_factor)
    a.w_clash, a.w_bond, a.w_angle = float(w_clash), float(w_bond), float(w_angle)
    a.energy, a.grad_atom, a.grad_trans, a.grad_rot = _p(energy), _p(g_atom), _p(g_t), _p(g_r)
    a.B, a.L = B, L
    check(lib.abx_clash_grad(C.byref(a), _p(ws), _stream()), 'abx_clash_grad')
    return energy, g_atom, g_t, g_r
