"""Per-kernel parity: every C-ABI entry of libabx_hip.so (called through abx_amd.ops) against the oracle / an fp64 torch
restatement of the same op on identical seeded inputs.  Needs an MI355X: run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import load_npz, tt

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, tol, name):
    assert a.shape == b.shape, (name, a.shape, b.shape)
    e = rel_err(a, b)
    assert np.isfinite(e) and e <= tol, f'{name}: rel err {e:.3e} > {tol}'


@pytest.fixture(scope='module')
def ops():
    from abx_amd import ops as _ops, _lib
    lib = _lib.load()
    assert lib.abx_init(0) == 0, lib.abx_last_error_string()
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def fold_ln(W, b, gamma, beta):
    """LayerNorm folded into the Linear (abx_gemm algebraic-LN contract): Wt' = gamma*Wt, csum, bias' = beta@Wt + b."""
    wt = W.double().t()
    wts = gamma.double()[:, None] * wt
    bias = beta.double() @ wt + (b.double() if b is not None else 0.0)
    return wts.float().contiguous().to(DEV), wts.sum(0).float().contiguous().to(DEV), bias.float().contiguous().to(DEV)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 70, 52), (37, 20, 256), (513, 192, 192), (64, 6, 256), (96, 448, 192)])
def test_gemm_plain_bias_act(ops, M, N, K):
    A = torch.randn(M, K, generator=g(1))
    W = torch.randn(N, K, generator=g(2)) / K ** 0.5
    b = torch.randn(N, generator=g(3))
    for act in (0, 1, 2):
        out = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(A.to(DEV), W.t().contiguous().to(DEV), out, bias=b.to(DEV), act=act)
        ref = A.double() @ W.double().t() + b.double()
        ref = torch.relu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
        check(out, ref, 2e-6, f'gemm act={act} {M}x{N}x{K}')


def test_gemm_ln_gate_resid_rowscale(ops):
    M, N, K = 300, 192, 192
    A = torch.randn(M, K, generator=g(4)) * 2 + 0.5
    W = torch.randn(N, K, generator=g(5)) / K ** 0.5
    b = torch.randn(N, generator=g(6))
    ga, be = torch.randn(K, generator=g(7)), torch.randn(K, generator=g(8))
    gate = torch.randn(M, N + 8, generator=g(9))
    res = torch.randn(M, N, generator=g(10))
    rs = (torch.rand(M, generator=g(11)) > 0.3).float()
    Ad = A.to(DEV)
    stats = ops.row_stats(Ad)
    ref_ln = torch.nn.functional.layer_norm(A.double(), (K,), ga.double(), be.double(), 1e-5)
    mean, rstd = A.double().mean(-1), 1 / torch.sqrt(A.double().var(-1, unbiased=False) + 1e-5)
    check(stats[:, 0], mean, 1e-6, 'row_stats mean')
    check(stats[:, 1], rstd, 1e-6, 'row_stats rstd')
    resd = res.to(DEV).clone()
    gd = gate.to(DEV)
    Wt, csum, bias2 = fold_ln(W, b, ga, be)
    ops.gemm(Ad, Wt, resd, bias=bias2, ln=(stats, csum), alpha=0.5, rowscale=rs.to(DEV), gate=gd[:, :N], resid=resd)
    ref = ((ref_ln @ W.double().t() + b.double()) * 0.5) * rs.double()[:, None] * torch.sigmoid(gate[:, :N].double()) + res.double()
    check(resd, ref, 3e-6, 'gemm LN+gate+resid (in place)')
    resd = res.to(DEV).clone()
    ops.gemm(Ad, Wt, resd, bias=bias2, ln=(None, csum), alpha=0.5, rowscale=rs.to(DEV), gate=gd[:, :N], resid=resd)
    check(resd, ref, 3e-6, 'gemm inline-LN+gate+resid (in place)')
    # inline statistics with a large common offset (mean >> std): the shifted accumulation must not cancel
    Aoff = (A * 0.05 + 300.0).to(DEV)          # |mean| / sigma = 3000
    outo = torch.empty(M, N, device=DEV)
    ops.gemm(Aoff, Wt, outo, bias=bias2, ln=(None, csum))
    refo = torch.nn.functional.layer_norm((A.double() * 0.05 + 300.0).float().double(), (K,), ga.double(), be.double(), 1e-5) @ W.double().t() + b.double()
    check(outo, refo, 2e-5, 'gemm inline-LN with mean >> std')
    # ragged / unaligned epilogue operands: N not a multiple of 4, gate and resid views that start off a 16-byte boundary
    for Nn, Mm in ((190, 300), (67, 130), (129, 257)):
        gate_u = torch.randn(Mm, Nn + 3, generator=g(90))[:, 1:1 + Nn]
        res_u = torch.randn(Mm, Nn + 1, generator=g(91))[:, 1:]
        outb = torch.full((Mm, Nn + 5), float('nan'), device=DEV)
        out = outb[:, 3:3 + Nn]
        Wn = W[:Nn] if Nn <= N else torch.randn(Nn, K, generator=g(92)) / K ** 0.5
        An = torch.randn(Mm, K, generator=g(93))
        ops.gemm(An.to(DEV), Wn.t().contiguous().to(DEV), out, gate=gate_u.to(DEV), resid=res_u.to(DEV), gate_sigmoid=False)
        r2 = (An.double() @ Wn.double().t()) * gate_u.double() + res_u.double()
        check(out, r2, 3e-6, f'gemm unaligned gate/resid/out N={Nn}')
        assert torch.isnan(outb[:, :3]).all() and torch.isnan(outb[:, 3 + Nn:]).all(), 'store outside the output view'
    # materialised layernorm, in place and with residual
    x = A.to(DEV).clone()
    ops.layernorm(x, ga.to(DEV), be.to(DEV), out=x)
    check(x, ref_ln, 2e-6, 'layernorm in place')
    # K = 128 takes the 16-byte-vector kernel (in place, with residual, strided rows, odd row count)
    x128 = torch.randn(1003, 160, generator=g(97)) * 3 + 1
    g128, b128, r128 = torch.randn(128, generator=g(98)), torch.randn(128, generator=g(99)), torch.randn(1003, 128, generator=g(100))
    xv = x128.to(DEV)[:, 16:144]                              # row stride 160, 16-byte aligned start
    o128 = ops.layernorm(xv, g128.to(DEV), b128.to(DEV), res=r128.to(DEV))
    ref128 = torch.nn.functional.layer_norm(x128[:, 16:144].double(), (128,), g128.double(), b128.double(), 1e-5) + r128.double()
    check(o128, ref128, 2e-6, 'layernorm K=128 vector kernel')
    ops.layernorm(xv, g128.to(DEV), b128.to(DEV), out=xv)
    check(xv, ref128 - r128.double(), 2e-6, 'layernorm K=128 in place')


def _host_planes(x, a_side):
    """fp32 tensor (..., rows, K) -> int16 k-tiled f16 operand image (..., K/16, 2, rows, 16) of the plane x plane contraction
    (include/abx_hip.h): A side x' = x / 16, (a0, (x' - a0) 2^11); B side x' = 16 x, (p0, x' - p0); RNE pieces."""
    xs = x / 16 if a_side else x * 16
    p0 = xs.half()
    r = xs - p0.float()
    p1 = (r * 2048).half() if a_side else r.half()
    pl = torch.stack([p0.view(torch.int16), p1.view(torch.int16)], dim=-3)                            # (..., 2, rows, K)
    rows, K = x.shape[-2:]
    assert K % 16 == 0
    pl = pl.reshape(*x.shape[:-2], 2, rows, K // 16, 16)
    return pl.movedim(-2, -4).contiguous()                                                           # (..., K/16, 2, rows, 16)


def _planes_to_f64(pl, a_side):
    """value of an operand image: (..., K/16, 2, rows, 16) int16 -> float64 (..., rows, K)."""
    h = pl.view(torch.float16).double()
    f = (h.select(-3, 0) + h.select(-3, 1) / 2048) * 16 if a_side else (h.select(-3, 0) + h.select(-3, 1)) / 16   # (..., K/16, rows, 16)
    return f.movedim(-3, -2).reshape(*f.shape[:-3], f.shape[-2], -1)


def test_gemm_split_f16_weight_image(ops):
    """abx_split_weights_f16 (the operand image of the split-f16 weight GEMMs): with w' = w 2^w_exp, max|w'| in [2^13, 2^14):
    p0 = f16(w'), p1 = f16(w' - p0), bit for bit the host's round-to-nearest conversions; p0 + p1 = w' to
    2^-23 |w'| + 2^-25 (two 11-bit pieces and a sign: 23 significant bits, worst case); k-tiled, zero padded rows."""
    for K, N, scale in ((192, 768, 1.0), (52, 70, 1e-3), (128, 192, 300.0)):
        Wt = (scale * torch.randn(K, N, generator=g(200)) * torch.logspace(-6, 0, N)[None]).to(DEV)
        w3 = ops.split_weights(Wt)
        Kp = (K + 15) // 16 * 16
        assert w3.shape == (Kp // 16, 2, N, 16) and w3.dtype == torch.float16
        wp = torch.zeros(N, Kp); wp[:, :K] = Wt.cpu().t() * 2.0 ** w3.w_exp
        assert 2.0 ** 13 <= float(wp.abs().max()) < 2.0 ** 14
        p0 = wp.half(); p1 = (wp - p0.float()).half()
        host = torch.stack([p0, p1], 0).reshape(2, N, Kp // 16, 16).permute(2, 0, 1, 3)
        assert torch.equal(torch.Tensor(w3.cpu()).view(torch.int16), host.contiguous().view(torch.int16))
        back = p0.double() + p1.double()
        assert ((back - wp.double()).abs() <= 2.0 ** -23 * wp.double().abs() + 2.0 ** -25).all()
        assert torch.equal(ops.weights_to_float(w3).cpu(), ((p0.float() + p1.float()) * 2.0 ** -w3.w_exp).t())


def test_gemm_split_f16_activation_range(ops):
    """Range contract of the split-f16 weight GEMMs (include/abx_hip.h, AbxGemm.b_f16): rows of magnitude 1e-6 ... 1e5 - with and
    without the folded LayerNorm - are as accurate as the exact fp32 MFMA kernel (elements below 2^-9: to 2^-32 absolute, which the LayerNorm's eps = 1e-5 amplifies to
    at most 7e-8); an activation beyond 2^20 gives NaN in its output
    row (never a silently wrong number) and leaves the other rows alone."""
    M, N, K = 40000, 192, 128
    A = torch.randn(M, K, generator=g(211)) + 0.3
    mag = 10.0 ** (torch.rand(M, generator=g(212)) * 11.0 - 6.0)                     # 1e-6 ... 1e5 per row (|x| < 6e5)
    A = A * mag[:, None]
    W = torch.randn(N, K, generator=g(213)) / K ** 0.5
    b = torch.randn(N, generator=g(214))
    Ad, Wt, bd = A.to(DEV), W.t().contiguous().to(DEV), b.to(DEV)
    w3 = ops.split_weights(Wt)
    # plain Linear: error relative to the row's own scale
    ref = A.double() @ W.double().t()
    o_x = torch.empty(M, N, device=DEV); o_s = torch.empty(M, N, device=DEV)
    ops.gemm(Ad, Wt, o_x, exact=True)
    ops.gemm(Ad, Wt, o_s, B3=w3, exact=2)
    den = (A.double().abs() @ W.double().abs().t())
    e_x = ((o_x.cpu().double() - ref).abs() / den); e_s = ((o_s.cpu().double() - ref).abs() / den)
    full = mag >= 1e-2                                      # |x| >= 2^-9: full relative precision; below: 2^-32 absolute per element
    assert float(e_s[full].max()) <= 1.5 * float(e_x[full].max()) and float(e_s[full].mean()) <= 1.2 * float(e_x[full].mean()), \
        (e_x[full].max(), e_s[full].max(), e_x[full].mean(), e_s[full].mean())
    floor = 2.0 ** -31 * W.double().abs().sum(1)[None]      # (2^-32 per element of the row, twice for slack)
    assert ((o_s.cpu().double() - ref).abs() <= 1.5 * float(e_x.max()) * den + floor).all()
    # folded LayerNorm: rows of any magnitude normalise to O(1)
    csum = Wt.sum(0).contiguous()
    refn = torch.nn.functional.layer_norm(A.double(), (K,)) @ W.double().t() + b.double()
    ops.gemm(Ad, Wt, o_x, bias=bd, ln=(None, csum), exact=True)
    ops.gemm(Ad, Wt, o_s, bias=bd, ln=(None, csum), B3=w3, exact=2)
    big = mag > 1e-2                                                                  # (below, eps = 1e-5 of the LayerNorm matters: compare kernels only)
    ex, es = (o_x.cpu().double() - refn).abs()[big], (o_s.cpu().double() - refn).abs()[big]
    assert float(es.max()) <= 1.5 * float(ex.max()) + 1e-7 and float(es.mean()) <= 1.2 * float(ex.mean()) + 1e-9, (ex.max(), es.max(), ex.mean(), es.mean())
    assert float((o_s - o_x).abs().max()) < 2e-5
    # overflow is loud
    A2 = Ad[:512].clone(); A2[7, 5] = 2.0 ** 20 * 1.01; A2[9, 100] = -3e7
    o2 = torch.empty(512, N, device=DEV)
    ops.gemm(A2, Wt, o2, B3=w3, exact=2)
    bad = ~torch.isfinite(o2).all(1)
    assert bad[7] and bad[9] and int(bad.sum()) == 2
    A2[7, 5] = 2.0 ** 20 * 0.99; A2[9, 100] = -1e6
    ops.gemm(A2, Wt, o2, B3=w3, exact=2)
    assert torch.isfinite(o2).all()
    ref2 = A2.cpu().double() @ W.double().t()
    assert float((o2.cpu().double() - ref2).abs()[7].max()) < 1e-6 * 2.0 ** 20


def _err(x, ref):
    d = (x.detach().cpu().double() - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.mean() / ref.abs().mean())


def test_gemm_split_accuracy_vs_exact(ops):
    """The split-f16 kernels (gemm3.hip) against float64, next to the exact fp32 MFMA kernel on the same problem: the
    split path must be as accurate as native fp32 (max error within 1.5x, mean error within 1.2x of the exact kernel)."""
    for (M, N, K) in ((33000, 192, 192), (66000, 768, 192), (33000, 192, 768), (40000, 128, 192), (70000, 190, 100), (70001, 190, 192),
                      (50003, 322, 128)):
        A = (torch.randn(M, K, generator=g(201)) * 3 + 0.7)
        W = torch.randn(N, K, generator=g(202)) / K ** 0.5
        b = torch.randn(N, generator=g(203))
        ref = A.double() @ W.double().t() + b.double()
        Ad, Wt, bd = A.to(DEV), W.t().contiguous().to(DEV), b.to(DEV)
        w3 = ops.split_weights(Wt)
        o_exact = torch.empty(M, N, device=DEV); o_split = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(Ad, Wt, o_exact, bias=bd, exact=True)
        ops.gemm(Ad, Wt, o_split, bias=bd, B3=w3, exact=False)
        e_x, e_s = _err(o_exact, ref), _err(o_split, ref)
        assert e_s[0] <= 1.5 * e_x[0] + 1e-8 and e_s[1] <= 1.2 * e_x[1] + 1e-9, (M, N, K, e_x, e_s)
        assert e_s[0] < 3e-6
        if K % 16 == 0:
            if ((M + 127) // 128) * ((N + 127) // 128) >= 256:
                assert not torch.equal(o_split, o_exact), 'the split-f16 kernel did not run'


def test_gemm_split_f16_error_model_heavy_tails(ops):
    """The error model of the split-f16 product on operands that are NOT well scaled: log-normal magnitudes over
    eight decades inside one row of A and inside the weights, random signs (heavy cancellation).  Every output must stay within
    the same multiple of 2^-24 * sum_k |a_k w_k| as the exact fp32-MFMA kernel reaches on the same problem (max <= 1.5x, mean <= 1.2x)."""
    for (M, N, K) in ((40000, 256, 128), (33000, 192, 768), (36000, 256, 2112)):
        A = torch.randn(M, K, generator=g(260)) * torch.exp(torch.randn(M, K, generator=g(261)) * 2.0)
        W = torch.randn(N, K, generator=g(262)) * torch.exp(torch.randn(N, K, generator=g(263)) * 2.0) / K ** 0.5
        Ad, Wt = A.to(DEV), W.t().contiguous().to(DEV)
        ref = A.double() @ W.double().t()
        den = A.double().abs() @ W.double().abs().t()
        o_x, o_s = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        ops.gemm(Ad, Wt, o_x, exact=True)
        ops.gemm(Ad, Wt, o_s, B3=ops.split_weights(Wt), exact=2)
        e_x = (o_x.cpu().double() - ref).abs() / den / 2.0 ** -24
        e_s = (o_s.cpu().double() - ref).abs() / den / 2.0 ** -24
        assert float(A.abs().max()) < 2.0 ** 20
        assert float(e_s.max()) <= 1.5 * float(e_x.max()) and float(e_s.mean()) <= 1.2 * float(e_x.mean()), \
            (M, N, K, float(e_x.max()), float(e_s.max()), float(e_x.mean()), float(e_s.mean()))


def test_gemm_split_operand_larger_than_4gb(ops):
    """The pair-stack GEMMs of a 20-sample chunk at L = 352 have A operands beyond 4 GB (2.48 M rows x 768): the DMA offsets
    are tile-relative 32-bit values, so such problems must still run on the split kernels and be right at both ends."""
    M, N, K = 1_450_000, 192, 768                    # M * K * 4 B = 4.45 GB
    gen = torch.Generator(device=DEV).manual_seed(77)
    A = torch.randn(M, K, device=DEV, generator=gen)
    W = torch.randn(K, N, device=DEV, generator=gen) / K ** 0.5
    res = torch.randn(M, N, device=DEV, generator=gen)
    out = res.clone()
    ops.gemm(A, W, out, resid=out, B3=ops.split_weights(W))
    oe = res.clone()
    ops.gemm(A, W, oe, resid=oe, exact=True)
    assert not torch.equal(out[:4096], oe[:4096]), 'the split-f16 kernel did not run'
    for sl in (slice(0, 4096), slice(M // 2, M // 2 + 4096), slice(M - 4096, M)):
        ref = A[sl].double() @ W.double() + res[sl].double()
        check(out[sl], ref, 3e-6, f'>4 GB operand rows {sl.start}')


def test_gemm_split_fused_paths(ops):
    """LayerNorm (inline statistics, mean >> sigma), relu-on-load, gate / residual, transposed store and the channel-major A
    operand on the split-f16 kernels, against float64."""
    M, N, K = 40 * 1000, 192, 192
    A = torch.randn(M, K, generator=g(204)) * 2 + 0.5
    W = torch.randn(N, K, generator=g(205)) / K ** 0.5
    b = torch.randn(N, generator=g(206)); ga = torch.randn(K, generator=g(207)); be = torch.randn(K, generator=g(208))
    gate = torch.randn(M, N, generator=g(209)); res = torch.randn(M, N, generator=g(210))
    rs = (torch.rand(M, generator=g(211)) > 0.3).float()
    Wt, csum, bias2 = fold_ln(W, b, ga, be)
    w3 = ops.split_weights(Wt)
    ln = torch.nn.functional.layer_norm(A.double(), (K,), ga.double(), be.double(), 1e-5)
    ref = ((ln @ W.double().t() + b.double()) * 0.5) * rs.double()[:, None] * torch.sigmoid(gate.double()) + res.double()
    out = res.to(DEV).clone()
    ops.gemm(A.to(DEV), Wt, out, bias=bias2, ln=(None, csum), alpha=0.5, rowscale=rs.to(DEV), gate=gate.to(DEV), resid=out, B3=w3)
    check(out, ref, 3e-6, 'split gemm inline-LN + gate + resid')
    stats = ops.row_stats(A.to(DEV))
    out2 = res.to(DEV).clone()
    ops.gemm(A.to(DEV), Wt, out2, bias=bias2, ln=(stats, csum), alpha=0.5, rowscale=rs.to(DEV), gate=gate.to(DEV), resid=out2, B3=w3)
    check(out2, ref, 3e-6, 'split gemm LN(stats) + gate + resid')
    Aoff = A * 0.05 + 300.0
    refo = torch.nn.functional.layer_norm(Aoff.double(), (K,), ga.double(), be.double(), 1e-5) @ W.double().t() + b.double()
    outo = torch.empty(M, N, device=DEV)
    ops.gemm(Aoff.to(DEV), Wt, outo, bias=bias2, ln=(None, csum), B3=w3)
    check(outo, refo, 2e-5, 'split gemm inline-LN with mean >> std')
    # relu on load + relu out
    W2 = W.t().contiguous().to(DEV)
    ops.gemm(A.to(DEV), W2, outo, bias=b.to(DEV), a_relu=True, act=1, B3=ops.split_weights(W2))
    check(outo, torch.relu(torch.relu(A.double()) @ W.double().t() + b.double()), 3e-6, 'split gemm relu-on-load')
    # transposed store with LN, mask and channel-major gates; ragged pair length (L = 203: M = 41209 rows per sample)
    B_, L_ = 2, 203
    LL = L_ * L_
    Z = torch.randn(B_, LL, K, generator=g(212))
    Wp = torch.randn(128, K, generator=g(213)) / K ** 0.5
    bp = torch.randn(128, generator=g(214))
    GT = torch.rand(B_, 128, LL, generator=g(215))
    pm = (torch.rand(B_ * LL, generator=g(216)) > 0.2).float()
    Wt2, csum2, bias3 = fold_ln(Wp, bp, ga, be)
    outT = torch.full((B_, 128, LL), float('nan'), device=DEV)
    ops.gemm(Z.to(DEV), Wt2, outT.transpose(1, 2), bias=bias3, ln=(None, csum2), rowscale=pm.to(DEV),
             gate=GT.to(DEV).transpose(1, 2), gate_sigmoid=False, B3=ops.split_weights(Wt2))
    lnz = torch.nn.functional.layer_norm(Z.double(), (K,), ga.double(), be.double(), 1e-5)
    refT = ((lnz @ Wp.double().t() + bp.double()) * pm.double().view(B_, LL, 1) * GT.double().transpose(1, 2)).transpose(1, 2)
    check(outT, refT, 3e-6, 'split gemm transposed store')
    # channel-major A (m-contiguous) with inline LN + gate + residual in place: the tri-mul output projection
    B_, L_ = 2, 204
    LL = L_ * L_
    T = torch.randn(B_, 128, LL, generator=g(217)) * 4
    Wo = torch.randn(192, 128, generator=g(218)) / 128 ** 0.5
    bo = torch.randn(192, generator=g(219)); g2 = torch.randn(128, generator=g(220)); b2 = torch.randn(128, generator=g(221))
    Wt3, csum3, bias4 = fold_ln(Wo, bo, g2, b2)
    Gf = torch.rand(B_, LL, 192, generator=g(222))
    z3 = torch.randn(B_, LL, 192, generator=g(223))
    zd = z3.to(DEV).clone()
    ops.gemm(T.to(DEV).transpose(1, 2), Wt3, zd, bias=bias4, ln=(None, csum3), gate=Gf.to(DEV), gate_sigmoid=False, resid=zd,
             B3=ops.split_weights(Wt3))
    lnt = torch.nn.functional.layer_norm(T.double().transpose(1, 2), (128,), g2.double(), b2.double(), 1e-5)
    check(zd, (lnt @ Wo.double().t() + bo.double()) * Gf.double() + z3.double(), 3e-6, 'split gemm channel-major A')


def test_gemm_split_contractions(ops):
    """TriangleMultiplication einsum (seqformer.py:490-493) on the split-f16 kernels: both operands as f16 operand images."""
    for nb, L_ in ((640, 80), (512, 128), (260, 208), (130, 352)):
        X = torch.randn(nb, L_, L_, generator=g(230)) * 2
        Y = torch.randn(nb, L_, L_, generator=g(231))
        Xd, Yd = X.to(DEV), Y.to(DEV)
        out = torch.full((nb, L_, L_), float('nan'), device=DEV); oe = torch.empty(nb, L_, L_, device=DEV)
        r_out = torch.einsum('bik,bjk->bij', X.double(), Y.double())
        ops.gemm(_host_planes(X, True).to(DEV), _host_planes(Y, False).to(DEV), out)
        ops.gemm(Xd, Yd.transpose(1, 2), oe, exact=True)
        e_s, e_x = _err(out, r_out), _err(oe, r_out)
        assert e_s[0] <= 1.5 * e_x[0] + 1e-8 and e_s[1] <= 1.2 * e_x[1] + 1e-9 and e_s[0] < 3e-6, ('NT', nb, L_, e_s, e_x)


def test_gemm_split_plane_output_and_pair_transpose(ops):
    """The projection GEMMs of the triangle multiplication: transposed store as f16 operand images (C_split) and the pair-transposed
    row gather (a_pair_transpose) of the incoming variant."""
    B_, L_, K, C_ = 5, 120, 192, 128
    LL = L_ * L_
    Z = torch.randn(B_, L_, L_, K, generator=g(240)) * 1.5 + 0.3
    W = torch.randn(C_, K, generator=g(241)) / K ** 0.5
    bias = torch.randn(C_, generator=g(242)); ga = torch.randn(K, generator=g(243)); be = torch.randn(K, generator=g(244))
    GT = torch.rand(B_, C_, LL, generator=g(245))
    pm = (torch.rand(B_, L_, L_, generator=g(246)) > 0.2).float()
    Wt, csum, bias2 = fold_ln(W, bias, ga, be)
    w3 = ops.split_weights(Wt)
    ln = torch.nn.functional.layer_norm(Z.double(), (K,), ga.double(), be.double(), 1e-5)
    proj = (ln @ W.double().t() + bias.double()) * pm.double()[..., None]                     # (B, L, L, C)
    for transpose in (False, True):
        zsrc = Z.to(DEV).view(B_, LL, K)
        planes = torch.zeros(B_, C_, (L_ + 15) // 16, 2, L_, 16, dtype=torch.int16, device=DEV)
        # gates / mask are indexed by the GEMM row (i.e. already in the transposed order when a_pair_transpose is used)
        pmv = (pm.transpose(1, 2) if transpose else pm).reshape(-1).contiguous().to(DEV)
        ops.gemm(zsrc, Wt, planes, bias=bias2, ln=(None, csum), rowscale=pmv, c_split_nA=64,
                 gate=GT.to(DEV).transpose(1, 2), gate_sigmoid=False, B3=w3, a_pair_transpose=L_ if transpose else 0)
        pc = planes.cpu()
        got = torch.cat([_planes_to_f64(pc[:, :64], True), _planes_to_f64(pc[:, 64:], False)], 1)[..., :L_].reshape(B_, C_, LL)
        want = (proj.transpose(1, 2) if transpose else proj).reshape(B_, LL, C_).transpose(1, 2) * GT.double()
        e = float((got - want).abs().max() / want.abs().max())
        assert e < 3e-6, (transpose, e)
        # the images are the host's split of the fp32 values the same GEMM stores as fp32, bit for bit; the k padding stays 0
        o32 = torch.empty(B_, C_, LL, device=DEV)
        ops.gemm(zsrc, Wt, o32.transpose(1, 2), bias=bias2, ln=(None, csum), rowscale=pmv,
                 gate=GT.to(DEV).transpose(1, 2), gate_sigmoid=False, B3=w3, a_pair_transpose=L_ if transpose else 0)
        o4 = o32.cpu().view(B_, C_, L_, L_)
        Kp = pc.shape[2] * 16
        o4p = torch.zeros(B_, C_, L_, Kp); o4p[..., :L_] = o4
        assert torch.equal(pc[:, :64], _host_planes(o4p[:, :64], True)) and torch.equal(pc[:, 64:], _host_planes(o4p[:, 64:], False)), \
            'plane output is not the split of the fp32 output'
        assert float((got.view(B_, C_, L_, L_) - o4.double()).abs().max() / o4.abs().max()) < 2.0 ** -22


def test_gemm_split_glu_and_two_level_batch(ops):
    """The gated projections of the triangle multiplication as ONE glu GEMM (value * sigmoid(gate) from (value, gate) column
    pairs, plane output, pair mask, pair-transposed rows) and the contraction over channel slices of that tensor
    (two-level batch of the plane operands)."""
    B_, L_, K, C_ = 5, 120, 192, 256
    LL = L_ * L_
    Z = torch.randn(B_, L_, L_, K, generator=g(250)) * 1.5 + 0.3
    Wv = torch.randn(C_, K, generator=g(251)) / K ** 0.5; Wg = torch.randn(C_, K, generator=g(252)) / K ** 0.5
    bv = torch.randn(C_, generator=g(253)); bg = torch.randn(C_, generator=g(254))
    ga = torch.randn(K, generator=g(255)); be = torch.randn(K, generator=g(256))
    pm = (torch.rand(B_, L_, L_, generator=g(257)) > 0.2).float()
    pm = pm * pm.transpose(1, 2)
    Wp, bp = ops.pack_glu_weights(Wv.t().contiguous(), Wg.t().contiguous(), bv, bg)
    assert Wp.shape == (K, 2 * C_) and torch.equal(Wp[:, 0:32], Wv.t()[:, 0:32]) and torch.equal(Wp[:, 32:64], Wg.t()[:, 0:32])
    Wt, csum, bias2 = fold_ln(Wp.t().contiguous(), bp, ga, be)
    w3 = ops.split_weights(Wt)
    ln = torch.nn.functional.layer_norm(Z.double(), (K,), ga.double(), be.double(), 1e-5)
    val = (ln @ Wv.double().t() + bv.double()) * torch.sigmoid(ln @ Wg.double().t() + bg.double()) * pm.double()[..., None]   # (B,L,L,C)
    KT = (L_ + 15) // 16
    for transpose in (False, True):
        planes = torch.zeros(B_, C_, KT, 2, L_, 16, dtype=torch.int16, device=DEV)
        ops.gemm(Z.to(DEV).view(B_, LL, K), Wt, planes, bias=bias2, ln=(None, csum), rowscale=pm.reshape(-1).contiguous().to(DEV),
                 glu=True, B3=w3, a_pair_transpose=L_ if transpose else 0, c_split_nA=128)
        pc = planes.cpu()
        got = torch.cat([_planes_to_f64(pc[:, :128], True), _planes_to_f64(pc[:, 128:], False)], 1)[..., :L_]      # (B, C, i, k)
        want = (val.transpose(1, 2) if transpose else val).permute(0, 3, 1, 2)
        e = float((got - want).abs().max() / want.abs().max())
        assert e < 3e-6, (transpose, e)
        # the same GEMM with its rows in (8 i x 16 k) block order (AbxGemm.c_split_tile: what the model runs): identical images
        planes_t = torch.zeros(B_, C_, KT, 2, L_, 16, dtype=torch.int16, device=DEV)
        ops.gemm(Z.to(DEV).view(B_, LL, K), Wt, planes_t, bias=bias2, ln=(None, csum), rowscale=pm.reshape(-1).contiguous().to(DEV),
                 glu=True, B3=w3, a_pair_transpose=L_ if transpose else 0, c_split_nA=128, c_split_tile=True, pair=(L_, L_), a_pair=True)
        assert torch.equal(planes_t, planes), 'tile-ordered rows changed the plane images'
        # contraction over the channel halves: out[b,c,i,j] = sum_k left[b,c,i,k] right[b,c,j,k]
        out = torch.full((B_ * 128, L_, L_), float('nan'), device=DEV)
        ops.gemm(planes[:, 0:128], planes[:, 128:256], out)
        ref = torch.einsum('bcik,bcjk->bcij', got[:, :128], got[:, 128:])
        e = float((out.cpu().double().view(B_, 128, L_, L_) - ref).abs().max() / ref.abs().max())
        assert e < 3e-6, ('contraction', transpose, e)


def test_gemm_layouts_batched_transposed(ops):
    nb, M, N, K = 5, 72, 72, 72       # L = 72 triangle contraction shapes (not multiples of the tiles)
    X = torch.randn(nb, M, K, generator=g(12))
    Y = torch.randn(nb, N, K, generator=g(13))
    Xd, Yd = X.to(DEV), Y.to(DEV)
    out = torch.empty(nb, M, N, device=DEV)
    ops.gemm(Xd, Yd.transpose(1, 2), out)                                    # A k-contig, B k-contig
    check(out, torch.einsum('bik,bjk->bij', X.double(), Y.double()), 2e-6, 'NT contraction')
    ops.gemm(Xd.transpose(1, 2), Yd, out)                                    # A m-contig, B n-contig: sum_k X[k,i] Y[k,j]
    check(out, torch.einsum('bki,bkj->bij', X.double(), Y.double()), 2e-6, 'TN contraction')
    # odd sizes -> scalar (unaligned) load paths
    nb, M, N, K = 3, 25, 25, 25
    X = torch.randn(nb, M, K, generator=g(14)); Y = torch.randn(nb, N, K, generator=g(15))
    out = torch.empty(nb, M, N, device=DEV)
    ops.gemm(X.to(DEV), Y.to(DEV).transpose(1, 2), out)
    check(out, torch.einsum('bik,bjk->bij', X.double(), Y.double()), 2e-6, 'NT odd')
    ops.gemm(X.to(DEV).transpose(1, 2), Y.to(DEV), out)
    check(out, torch.einsum('bki,bkj->bij', X.double(), Y.double()), 2e-6, 'TN odd')
    # transposed store (channel-major output) with LN, rowscale and gate, batched over samples
    B, LL, C, Cout = 2, 40 * 40, 192, 128
    Z = torch.randn(B, LL, C, generator=g(16))
    W = torch.randn(Cout, C, generator=g(17)) / C ** 0.5
    bias = torch.randn(Cout, generator=g(18))
    G = torch.randn(B, LL, 448, generator=g(19))
    pm = (torch.rand(B * LL, generator=g(20)) > 0.2).float()
    ga, be = torch.randn(C, generator=g(21)), torch.randn(C, generator=g(22))
    Zd = Z.to(DEV)
    stats = ops.row_stats(Zd.view(B * LL, C))
    outT = torch.full((B, Cout, LL), float('nan'), device=DEV)
    Wt, csum, bias2 = fold_ln(W, bias, ga, be)
    GT = G.to(DEV).transpose(1, 2).contiguous()                 # channel-major gates, like the transposed output
    gview = GT[:, 128:256].transpose(1, 2)
    ops.gemm(Zd, Wt, outT.transpose(1, 2), bias=bias2, ln=(stats, csum), rowscale=pm.to(DEV), gate=gview)
    ln = torch.nn.functional.layer_norm(Z.double(), (C,), ga.double(), be.double(), 1e-5)
    ref = (ln @ W.double().t() + bias.double()) * pm.double().view(B, LL, 1) * torch.sigmoid(G[:, :, 128:256].double())
    check(outT, ref.transpose(1, 2), 3e-6, 'transposed store')
    outT.fill_(float('nan'))
    ops.gemm(Zd, Wt, outT.transpose(1, 2), bias=bias2, ln=(None, csum), rowscale=pm.to(DEV), gate=gview)
    check(outT, ref.transpose(1, 2), 3e-6, 'transposed store, inline LN')
    # ragged transposed store (M = 37*37 not a multiple of 4, odd N) with a transposed residual, sigmoid at the producer
    B2, LL2, N2 = 2, 37 * 37, 45
    Z2 = torch.randn(B2, LL2, C, generator=g(94))
    W2 = torch.randn(N2, C, generator=g(95)) / C ** 0.5
    R2 = torch.randn(B2, N2, LL2, generator=g(96))
    o2 = torch.full((B2, N2, LL2), float('nan'), device=DEV)
    ops.gemm(Z2.to(DEV), W2.t().contiguous().to(DEV), o2.transpose(1, 2), act=2, resid=R2.to(DEV).transpose(1, 2))
    check(o2, (torch.sigmoid(Z2.double() @ W2.double().t()) + R2.double().transpose(1, 2)).transpose(1, 2), 3e-6, 'ragged transposed store')
    # channel-major A (m-contiguous) with channel-major LN stats, back to channel-last with residual
    T = torch.randn(B, Cout, LL, generator=g(23))
    Td = T.to(DEV)
    tcm = Td.transpose(1, 2)
    st2 = ops.row_stats(tcm)
    W2 = torch.randn(C, Cout, generator=g(24)) / Cout ** 0.5
    ga2, be2 = torch.randn(Cout, generator=g(25)), torch.randn(Cout, generator=g(26))
    res = Zd.clone()
    Wt2, csum2, bias22 = fold_ln(W2, None, ga2, be2)
    ops.gemm(tcm, Wt2, res, bias=bias22, ln=(st2, csum2), resid=res)
    ln2 = torch.nn.functional.layer_norm(T.double().transpose(1, 2), (Cout,), ga2.double(), be2.double(), 1e-5)
    check(res, ln2 @ W2.double().t() + Z.double(), 3e-6, 'channel-major A + LN')
    res = Zd.clone()
    ops.gemm(tcm, Wt2, res, bias=bias22, ln=(None, csum2), resid=res)
    check(res, ln2 @ W2.double().t() + Z.double(), 3e-6, 'channel-major A + inline LN')


def test_gemm_small_n_and_relu_input(ops):
    M, K = 150, 128
    A = torch.randn(M, K, generator=g(27))
    for N in (4, 12, 14, 32, 50):
        W = torch.randn(N, K, generator=g(28 + N)) / K ** 0.5
        out = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), W.t().contiguous().to(DEV), out, a_relu=True)
        check(out, torch.relu(A.double()) @ W.double().t(), 2e-6, f'gemm N={N} a_relu')
    # unaligned K (1538 = residue-embedding MLP input) and strided output window
    K = 1538
    A = torch.randn(40, K, generator=g(90)); W = torch.randn(64, K, generator=g(91)) / K ** 0.5
    wide = torch.zeros(40, 200, device=DEV)
    ops.gemm(A.to(DEV), W.t().contiguous().to(DEV), wide[:, 100:164])
    check(wide[:, 100:164], A.double() @ W.double().t(), 3e-6, 'gemm K=1538 window')
    assert float(wide[:, :100].abs().max()) == 0 and float(wide[:, 164:].abs().max()) == 0


def test_range_word_names_the_op_that_left_the_split_range(ops):
    """Range safety at the source (include/abx_hip.h, AbxGemm.range_flag; the reference's plain fp32 contractions have no operand range,
    seqformer.py:260-312, 443-504): every split-f16 kernel ORs the bit of its call-site class into the device's range word when an
    accumulator is not finite.  An in-range problem leaves the word clear; an A element beyond 2^20, a key / value beyond 4095, a query
    beyond 5600 / scale, and a pair row beyond 2^20 in front of the fused transition (whose hidden ReLU must not swallow the NaN) set
    the bit of THEIR op, turn exactly the affected rows into NaN and leave every other row bit-identical."""
    word = ops.range_word(DEV)
    T = ops.RANGE_TAGS
    # ---- weight GEMM
    M, N, K = 4096, 192, 192
    A = torch.randn(M, K, generator=g(801)).to(DEV)
    Wt = (torch.randn(N, K, generator=g(802)) / K ** 0.5).t().contiguous().to(DEV)
    w3 = ops.split_weights(Wt)
    o0, o1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    word.zero_()
    ops.gemm(A, Wt, o0, B3=w3, exact=2)
    assert int(word.item()) == 0 and torch.isfinite(o0).all()
    A2 = A.clone(); A2[77, 5] = 2.0 ** 20 * 1.5
    ops.gemm(A2, Wt, o1, B3=w3, exact=2)
    assert int(word.item()) == T['gemm']
    bad = ~torch.isfinite(o1).all(1)
    assert bad[77] and int(bad.sum()) == 1 and torch.equal(o1[~bad], o0[~bad])
    # the exact kernel never touches the word
    word.zero_()
    ops.gemm(A2, Wt, o1, exact=True)
    assert int(word.item()) == 0 and torch.isfinite(o1).all()
    # ---- fused pair transition: LayerNorm -> 768 -> ReLU -> 192 + residual (the NaN must survive the hidden ReLU)
    M = 128 * 300
    z = torch.randn(M, 192, generator=g(803)).to(DEV)
    W1 = (torch.randn(192, 768, generator=g(804)) / 14).to(DEV); W2 = (torch.randn(768, 192, generator=g(805)) / 28).to(DEV)
    b1, b2, cs = torch.randn(768, generator=g(806)).to(DEV), torch.randn(192, generator=g(807)).to(DEV), W1.sum(0).contiguous()
    W13, W23 = ops.split_weights(W1), ops.split_weights(ops.permute_k16(W2))
    r0 = torch.empty_like(z); r1 = torch.empty_like(z)
    word.zero_()
    ops.gemm(z, W1, r0, bias=b1, ln=(None, cs), B3=W13, act=1, resid=z, exact=2, mlp=(W23, b2))
    assert int(word.item()) == 0
    z2 = z.clone(); z2[1000, 17] = 3e6
    ops.gemm(z2, W1, r1, bias=b1, ln=(None, cs), B3=W13, act=1, resid=z2, exact=2, mlp=(W23, b2))
    assert int(word.item()) == T['pair_transition']
    bad = ~torch.isfinite(r1).all(1)
    assert bad[1000] and int(bad.sum()) == 1 and torch.equal(r1[~bad], r0[~bad])
    # ---- triangle attention: key, value, query
    B, L, H, D = 1, 96, 4, 48
    C = H * D
    x = torch.randn(B * L * L, 4 * C, generator=g(808)).to(DEV)
    bT = torch.randn(B, H, L, L, generator=g(809)).to(DEV)
    mask = torch.ones(B, L, device=DEV)
    ref = torch.empty(B * L * L, C, device=DEV)
    word.zero_()
    ops.tri_attn(x, bT, mask, ref, B, L, True)
    assert int(word.item()) == 0 and torch.isfinite(ref).all()
    row = lambda s_, l_: s_ * L + l_
    for what, col, val in (('key', C + 1 * D + 7, 5000.0), ('value', 2 * C + 2 * D + 40, -4500.0), ('query', 3 * D + 3, 1e5)):
        x2 = x.clone(); x2[row(5, 9), col] = val
        o = torch.empty_like(ref)
        word.zero_()
        ops.tri_attn(x2, bT, mask, o, B, L, True)
        assert int(word.item()) == T['tri_attn'], what
        nanmask = ~torch.isfinite(o)
        h = (col % C) // D
        expect = torch.zeros_like(nanmask)
        if what == 'query':
            expect[row(5, 9), h * D:(h + 1) * D] = True                  # that query only
        elif what == 'key':
            expect.view(L, L, C)[5, :, h * D:(h + 1) * D] = True         # every query of row 5, head h
        else:
            expect.view(L, L, C)[5, :, col % C] = True                   # one output channel of row 5
        assert torch.equal(nanmask, expect), what
        assert torch.equal(o[~nanmask], ref[~nanmask]), what
        oe = torch.empty_like(ref)
        word.zero_()
        ops.tri_attn(x2, bT, mask, oe, B, L, True, exact=True)
        assert int(word.item()) == 0 and torch.isfinite(oe).all(), what


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('exact', [False, True])
@pytest.mark.parametrize('L,per_row', [(40, True), (40, False), (64, True), (97, False), (200, True), (212, False), (261, True), (300, False)])
def test_tri_attn(ops, L, per_row, exact):
    """exact=False: split-f16 kernel (K/V in double-buffered chunks of 128 keys: L = 200 / 212 cross one chunk boundary, 261 / 300
    two, and carry the online softmax state over them); exact=True: fp32 MFMA kernel."""
    from oracle import abx_oracle as O
    B, H, D = (2 if L < 128 else 1), 4, 48
    C = H * D
    x = torch.randn(B, L, L, 4 * C, generator=g(30))            # [q|k|v|gate] in natural (i,j) layout
    P = torch.randn(B, L, L, H, generator=g(31))                # bias projection in natural layout
    mask = torch.rand(B, L, generator=g(32)) > 0.15
    mask[:, 0] = True
    out = torch.full((B * L * L, C), float('nan'), device=DEV)
    biasT = P.permute(0, 3, 1, 2).contiguous()                 # (B,H,i,j)
    ops.tri_attn(x.view(B * L * L, 4 * C).to(DEV), biasT.to(DEV), mask.float().to(DEV), out, B, L, per_row, exact=exact)
    xx = x if per_row else x.transpose(1, 2)
    PP = P if per_row else P.transpose(1, 2)
    q, k, v, gt = [t.reshape(B, L, L, H, D).permute(0, 1, 3, 2, 4).double() for t in torch.split(xx, C, dim=-1)]
    o = O._attention(q, k, v, PP.permute(0, 3, 1, 2).double(), mask[:, None, :], D)
    o = o * torch.sigmoid(xx[..., 3 * C:].double())
    if not per_row:
        o = o.transpose(1, 2)
    check(out.view(B, L, L, C), o, 5e-6, f'tri_attn L={L} per_row={per_row}')
    # same through the key-contiguous bias copy the model uses for the ending node
    Lp = (L + 3) // 4 * 4           # key-contiguous bias rows padded to a multiple of 4 floats, as model/forward.py passes them
    b2 = ops.transpose_last2(biasT.to(DEV).view(B * H, L, L), torch.full((B * H, L, Lp), float('nan'), device=DEV),
                             transpose=not per_row).view(B, H, L, Lp)
    assert torch.equal(b2[..., :L].cpu(), biasT if per_row else biasT.transpose(-1, -2)) and bool((b2[..., L:] == 0).all())
    out2 = torch.full((B * L * L, C), float('nan'), device=DEV)
    ops.tri_attn(x.view(B * L * L, 4 * C).to(DEV), b2, mask.float().to(DEV), out2, B, L, per_row, bias_is_qk=True, exact=exact)
    check(out2.view(B, L, L, C), o, 5e-6, f'tri_attn (qk bias) L={L} per_row={per_row}')


@pytest.mark.parametrize('L', [97, 352, 368])
def test_tri_attn_variants_bit_identical(ops, L):
    """The paired-tile kernel (tri_attn8: the library's choice) on 192-key and on 128-key chunks (AbxTriAttn.tune bit 1) evaluates every
    query with the same arithmetic: equal bits, masked keys and a ragged last key chunk included; so do the two staging variants of the
    round-3 kernel (tune 4 / 5: one query tile at a time).  Between the two kernels the accumulation order differs (the bias is the
    initial value of the S^T accumulators, 5 instead of 6 products per sub-block): same tolerance against the exact kernel."""
    B, H, D = 1, 4, 48
    C = H * D
    gen = torch.Generator(device=DEV).manual_seed(300 + L)
    x = torch.randn(B * L * L, 4 * C, device=DEV, generator=gen)
    biasT = torch.randn(B, H, L, L, device=DEV, generator=gen)
    mask = (torch.rand(B, L, device=DEV, generator=gen) > 0.1).float()
    mask[:, 0] = 1
    outs = {}
    for tune in (0, 2, 4, 5):
        o = torch.full((B * L * L, C), float('nan'), device=DEV)
        ops.tri_attn(x, biasT, mask, o, B, L, True, tune=tune)
        outs[tune] = o
    assert torch.equal(outs[0], outs[2])
    assert torch.equal(outs[4], outs[5])
    oe = torch.empty(B * L * L, C, device=DEV)
    ops.tri_attn(x, biasT, mask, oe, B, L, True, exact=True)
    for tune in (0, 4):
        assert float((outs[tune] - oe).abs().max()) < 5e-6 * float(oe.abs().max()) + 5e-6, tune


@pytest.mark.parametrize('L', [97, 352])
def test_tri_attn_bias_in_accumulator_units(ops, L):
    """AbxTriAttn.bias_log2: a bias the projection already multiplied by ABX_TRI_BIAS_LOG2 (float(log2 e) * 2^7, AbxGemm.alpha) gives the
    paired-tile kernel the bits of the plain bias (the same product, rounded once, outside the kernel's hot loop); the exact kernel and
    the round-3 kernel scale it back to base-2 logits: their results move by the rounding of the product only."""
    B, H, D = 1, 4, 48
    C = H * D
    gen = torch.Generator(device=DEV).manual_seed(400 + L)
    x = torch.randn(B * L * L, 3 * C, device=DEV, generator=gen)
    biasT = torch.randn(B, H, L, L, device=DEV, generator=gen) * 3
    mask = (torch.rand(B, L, device=DEV, generator=gen) > 0.1).float()
    mask[:, 0] = 1
    scaled = biasT * ops.TRI_BIAS_LOG2
    for tune in (0, 2):
        o0 = torch.full((B * L * L, C), float('nan'), device=DEV)
        o1 = torch.full((B * L * L, C), float('nan'), device=DEV)
        ops.tri_attn(x, biasT, mask, o0, B, L, True, tune=tune)
        ops.tri_attn(x, scaled, mask, o1, B, L, True, tune=tune, bias_log2=True)
        assert torch.equal(o0, o1), tune
    for kw in (dict(tune=4), dict(exact=True)):
        o0 = torch.empty(B * L * L, C, device=DEV)
        o1 = torch.empty(B * L * L, C, device=DEV)
        ops.tri_attn(x, biasT, mask, o0, B, L, True, **kw)
        ops.tri_attn(x, scaled, mask, o1, B, L, True, bias_log2=True, **kw)
        assert float((o0 - o1).abs().max()) < 2e-6 * float(o0.abs().max()) + 2e-6, kw


@pytest.mark.parametrize('L,per_row', [(416, True), (560, False), (752, True)])
def test_tri_attn_long_rows(ops, L, per_row):
    """Rows with more than 24 query tiles are dealt to several workgroups (AbxTriAttn.q_parts: 416 -> 2 x 13 tiles and 560 -> 2 x 18
    on the producer-wave variant, 752 -> 2 x 24 on the shared-staging one); no instantiation carries more than two tiles of softmax
    state per wave, whatever L.  Reference: the same attention in float64 torch ops on the GPU, one head at a time."""
    B, H, D = 1, 4, 48
    C = H * D
    gen = torch.Generator(device=DEV).manual_seed(400 + L)
    x = torch.randn(B, L, L, 4 * C, device=DEV, generator=gen)
    biasT = torch.randn(B, H, L, L, device=DEV, generator=gen)             # (b, h, q, k)
    mask = (torch.rand(B, L, device=DEV, generator=gen) > 0.1)
    mask[:, 0] = True
    out = torch.full((B * L * L, C), float('nan'), device=DEV)
    ops.tri_attn(x.view(B * L * L, 4 * C), biasT, mask.float(), out, B, L, per_row)
    out = out.view(B, L, L, C)
    xx = x if per_row else x.transpose(1, 2)
    got = out if per_row else out.transpose(1, 2)
    neg = torch.finfo(torch.float32).min
    worst = 0.0
    for h in range(H):
        q, k, v, gt = [xx[0, :, :, j * C + h * D:j * C + (h + 1) * D].double() for j in range(4)]      # (s, l, d)
        bias_h = biasT[0, h] if per_row else biasT[0, h].t()                  # (the ending-node orientation reads the bias transposed)
        logits = torch.einsum('sqd,skd->sqk', q, k) * D ** -0.5 + bias_h.double()[None]
        logits = torch.where(mask[0][None, None, :], logits, torch.full_like(logits, neg))
        o = torch.einsum('sqk,skd->sqd', torch.softmax(logits, -1), v) * torch.sigmoid(gt)
        worst = max(worst, float((got[0, :, :, h * D:(h + 1) * D].double() - o).abs().max() / o.abs().max()))
        del logits, o
    assert worst < 5e-6, (L, per_row, worst)


@pytest.mark.parametrize('exact', [False, True])
@pytest.mark.parametrize('L,spike', [(80, 70), (200, 190)])
def test_tri_attn_all_keys_masked_row_and_spike(ops, exact, L, spike):
    """Fully masked keys give the uniform softmax of finfo.min logits; a spiked key forces the online-softmax rescale (in the 2nd
    key tile of the first chunk / in the second chunk: the rescale factors cross from the S^T columns to the O rows)."""
    from oracle import abx_oracle as O
    B, H, D = 1, 4, 48
    C = H * D
    x = torch.randn(B, L, L, 4 * C, generator=g(33))
    x[0, 3, spike, C:2 * C] *= 30.0                             # this key of row 3 dominates -> the running max jumps late
    P = torch.zeros(B, L, L, H)
    for mask in (torch.zeros(B, L, dtype=torch.bool), torch.ones(B, L, dtype=torch.bool)):
        out = torch.empty(B * L * L, C, device=DEV)
        ops.tri_attn(x.view(-1, 4 * C).to(DEV), P.permute(0, 3, 1, 2).contiguous().to(DEV), mask.float().to(DEV), out, B, L, True,
                     exact=exact)
        q, k, v, gt = [t.reshape(B, L, L, H, D).permute(0, 1, 3, 2, 4).double() for t in torch.split(x, C, dim=-1)]
        o = O._attention(q, k, v, None, mask[:, None, :], D) * torch.sigmoid(x[..., 3 * C:].double())
        check(out.view(B, L, L, C), o, 5e-6, f'tri_attn mask all={bool(mask.all())}')


@pytest.mark.parametrize('M', [352, 4224 + 5, 31])
def test_ipa_tail(ops, M):
    """abx_ipa_tail: final_proj + residual + LayerNorm + the three-layer transition + residual + LayerNorm of an IPA layer in one launch
    (reference score_network.py:126-163) against float64, and against the same chain as six launches (the path it replaces)."""
    K1, Cc = 2112, 256
    gen = lambda i: g(500 + i)
    feat = torch.randn(M, K1, generator=gen(0)) * 1.5
    s0 = torch.randn(M, Cc, generator=gen(1))
    Wf = torch.randn(K1, Cc, generator=gen(2)) / K1 ** 0.5; W0 = torch.randn(Cc, Cc, generator=gen(3)) / 16
    W2 = torch.randn(Cc, Cc, generator=gen(4)) / 16; W4 = torch.randn(Cc, Cc, generator=gen(5)) / 16
    bs = [torch.randn(Cc, generator=gen(6 + i)) * 0.3 for i in range(4)]
    ln = [(1 + 0.2 * torch.randn(Cc, generator=gen(10 + i)), 0.2 * torch.randn(Cc, generator=gen(12 + i))) for i in range(2)]
    d = lambda t: t.to(DEV).contiguous()
    LN = lambda x, p: torch.nn.functional.layer_norm(x, (Cc,), p[0].double(), p[1].double(), 1e-5)
    x = LN(s0.double() + feat.double() @ Wf.double() + bs[0].double(), ln[0])
    hdn = torch.relu(torch.relu(x @ W0.double() + bs[1].double()) @ W2.double() + bs[2].double())
    ref = LN(x + hdn @ W4.double() + bs[3].double(), ln[1])
    featd, sd = d(feat), d(s0)
    wd = [(ops.split_weights(d(W)), d(b)) for W, b in zip((Wf, W0, W2, W4), bs)]
    lnd = [(d(a), d(b)) for a, b in ln]
    ops.ipa_tail(featd, sd, wd[0], lnd[0], wd[1], wd[2], wd[3], lnd[1])
    check(sd, ref, 5e-6, f'ipa_tail M={M}')
    # the six-launch path on the same split-f16 GEMM kernels
    s6, h1, h2 = d(s0), torch.empty(M, Cc, device=DEV), torch.empty(M, Cc, device=DEV)
    ops.gemm(featd, d(Wf), s6, bias=wd[0][1], B3=wd[0][0], resid=s6, exact=2)
    ops.layernorm(s6, *lnd[0], out=s6)
    ops.gemm(s6, d(W0), h1, bias=wd[1][1], B3=wd[1][0], act=1, exact=2)
    ops.gemm(h1, d(W2), h2, bias=wd[2][1], B3=wd[2][0], act=1, exact=2)
    ops.gemm(h2, d(W4), s6, bias=wd[3][1], B3=wd[3][0], resid=s6, exact=2)
    ops.layernorm(s6, *lnd[1], out=s6)
    assert float((sd - s6).abs().max()) < 2e-5
    # final_proj in front of the tail as a split-K GEMM of 11 K-slices (ops.gemm_splitk, one launch), summed by the tail in slice order
    part = ops.gemm_splitk(featd, wd[0][0], torch.full((11, M, Cc), float('nan'), device=DEV))
    refp = torch.stack([feat[:, i * 192:(i + 1) * 192].double() @ Wf[i * 192:(i + 1) * 192].double() for i in range(11)])
    check(part, refp, 5e-6, f'split-K slices M={M}')
    sk = d(s0)
    ops.ipa_tail(featd, sk, wd[0], lnd[0], wd[1], wd[2], wd[3], lnd[1], partial=part)
    check(sk, ref, 5e-6, f'ipa_tail on split-K partials M={M}')
    # with the affine_update + frame update in the same launch: against abx_gemm (exact kernel, N = 6) + abx_rigid_update on the new s
    Wa, ba = d(torch.randn(Cc, 6, generator=gen(20)) * 0.05), d(torch.randn(6, generator=gen(21)) * 0.05)
    q0 = torch.nn.functional.normalize(torch.randn(M, 4, generator=gen(22)), dim=-1)
    t0 = torch.randn(M, 3, generator=gen(23)) * 10
    fixed = d((torch.rand(M, generator=gen(24)) < 0.3).to(torch.int32))
    frames = []
    for fused in (True, False):
        iq, it, cq, ct, cR, dq = (torch.empty(M, 4, device=DEV), torch.empty(M, 3, device=DEV), torch.empty(M, 4, device=DEV),
                                  torch.empty(M, 3, device=DEV), torch.empty(M, 9, device=DEV), torch.empty(M, 4, device=DEV))
        ops.frames_init(d(torch.cat([q0, t0], -1)), iq, it, cq, ct, cR, dq, M, 10.0)
        s7 = d(s0)
        if fused:
            ops.ipa_tail(featd, s7, wd[0], lnd[0], wd[1], wd[2], wd[3], lnd[1], affine=(Wa, ba), rigid=(fixed, iq, it, cq, ct, cR, dq, 10.0))
            assert torch.equal(s7, sd)
        else:
            upd = torch.empty(M, 6, device=DEV)
            ops.gemm(sd, Wa, upd, bias=ba, exact=1)
            ops.rigid_update(upd, fixed, iq, it, cq, ct, cR, dq, M, 10.0)
        frames.append((cq, ct, cR, dq))
    for x1, x2, nm in zip(frames[0], frames[1], ('cur_q', 'cur_t', 'cur_R', 'delta_q')):
        assert float((x1 - x2).abs().max()) < 1e-5, nm


@pytest.mark.parametrize('L,Bc', [(72, 5), (65, 6)])
def test_gemm_side_equals_the_two_launches(ops, L, Bc):
    """abx_gemm_side: the q | k | v | gate projection (LayerNorm folded, N = 768, plain store) with the triangle attention's pair bias
    (N = 4, transposed (b, h, i, j) store) as side tiles of its grid: bit-identical to the two separate launches; L = 65: ragged last
    row tiles on both sides (L * L % 128 != 0)."""
    ge = g(900 + L)
    LL, K = L * L, 192
    z = (torch.randn(Bc, LL, K, generator=ge) * 1.3 + 0.2).to(DEV)
    Wq, Wb = (torch.randn(K, 768, generator=ge) / K ** 0.5).to(DEV), (torch.randn(K, 4, generator=ge) / K ** 0.5).to(DEV)
    bq, bb = torch.randn(768, generator=ge).to(DEV), torch.randn(4, generator=ge).to(DEV)
    csq, csb = Wq.sum(0).contiguous(), Wb.sum(0).contiguous()
    W3q, W3b = ops.split_weights(Wq), ops.split_weights(Wb)
    outs = []
    for side in (True, False):
        q = torch.full((Bc * LL, 768), float('nan'), device=DEV)
        bT = torch.full((Bc, 4, LL), float('nan'), device=DEV)
        kq = dict(bias=bq, ln=(None, csq), B3=W3q, exact=2)
        kb = dict(bias=bb, ln=(None, csb), B3=W3b, exact=2)
        if side:
            ops.gemm_side(ops.gemm(z.view(Bc * LL, K), Wq, q, defer=True, **kq), ops.gemm(z, Wb, bT.transpose(1, 2), defer=True, **kb))
        else:
            ops.gemm(z.view(Bc * LL, K), Wq, q, **kq)
            ops.gemm(z, Wb, bT.transpose(1, 2), **kb)
        outs.append((q, bT))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    x = z.double().cpu()
    ln = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    check(outs[0][1], (ln @ Wb.double().cpu() + bb.double().cpu()).transpose(1, 2), 5e-6, f'side bias L={L}')
    check(outs[0][0], (ln @ Wq.double().cpu() + bq.double().cpu()).reshape(Bc * LL, 768), 5e-6, f'main projection L={L}')


@pytest.mark.parametrize('M,with_plddt', [(352, True), (4224 + 5, False), (31, True)])
def test_heads_tail(ops, M, with_plddt):
    """abx_heads_tail: TorsionModule (sidechain.py:28-62), SequenceHead.net and PredictedLDDTHead.net (head.py:143-226) in one launch
    against float64; ragged last row block, with and without the pLDDT head, untouched pLDDT output when it is skipped."""
    gen = lambda i: g(700 + i)
    W = lambda i, K, N: torch.randn(K, N, generator=gen(i)) / K ** 0.5
    bv = lambda i, N: torch.randn(N, generator=gen(i)) * 0.3
    d = lambda t: t.to(DEV).contiguous()
    s, s0 = torch.randn(M, 256, generator=gen(0)) * 1.5, torch.randn(M, 256, generator=gen(1)) + 0.3
    tw = [(W(2, 256, 128), bv(3, 128)), (W(4, 256, 128), bv(5, 128))] + [(W(6 + 2 * i, 128, 128), bv(7 + 2 * i, 128)) for i in range(4)] + \
         [(W(14, 128, 14), bv(15, 14))]
    heads = []
    for k, n in ((0, 20), (1, 50)):
        ln = (1 + 0.2 * torch.randn(256, generator=gen(20 + 10 * k)), 0.2 * torch.randn(256, generator=gen(21 + 10 * k)))
        heads.append((ln, (W(22 + 10 * k, 256, 128), bv(23 + 10 * k, 128)), (W(24 + 10 * k, 128, 128), bv(25 + 10 * k, 128)),
                      (W(26 + 10 * k, 128, n), bv(27 + 10 * k, n))))
    D = lambda t: t.double()
    relu = torch.relu
    t = relu(D(s)) @ D(tw[0][0]) + D(tw[0][1]) + relu(D(s0)) @ D(tw[1][0]) + D(tw[1][1])
    for blk in range(2):
        t = t + relu(relu(t) @ D(tw[2 + 2 * blk][0]) + D(tw[2 + 2 * blk][1])) @ D(tw[3 + 2 * blk][0]) + D(tw[3 + 2 * blk][1])
    ref_un = relu(t) @ D(tw[6][0]) + D(tw[6][1])
    refs = []
    for ln, l1, l3, l5 in heads:
        hx = torch.nn.functional.layer_norm(D(s), (256,), D(ln[0]), D(ln[1]), 1e-5)
        refs.append(relu(relu(hx @ D(l1[0]) + D(l1[1])) @ D(l3[0]) + D(l3[1])) @ D(l5[0]) + D(l5[1]))
    full = lambda wb: (ops.split_weights(d(wb[0])), d(wb[1]))
    tors = [full(wb) for wb in tw[:-1]] + [ops.pad_planes_128(d(tw[-1][0]), d(tw[-1][1]))]
    hd = [((d(ln[0]), d(ln[1])), full(l1), full(l3), ops.pad_planes_128(d(l5[0]), d(l5[1]))) for ln, l1, l3, l5 in heads]
    un = torch.full((M, 14), float('nan'), device=DEV)
    logits = torch.full((M, 20), float('nan'), device=DEV)
    pl = torch.full((M, 50), 7.0, device=DEV)
    ops.heads_tail(d(s), d(s0), tors, hd[0], hd[1], un, logits, pl if with_plddt else None)
    check(un, ref_un, 5e-6, f'heads_tail torsions M={M}')
    check(logits, refs[0], 5e-6, f'heads_tail logits M={M}')
    if with_plddt:
        check(pl, refs[1], 5e-6, f'heads_tail pLDDT logits M={M}')
    else:
        assert bool((pl == 7.0).all())
    # the range word: a non-finite input row reaches the outputs and sets the kernel's bit
    if M == 352:
        word = ops.range_word(DEV); word.zero_()
        sb = d(s); sb[5, 17] = float('inf')
        ops.heads_tail(sb, d(s0), tors, hd[0], hd[1], un, logits, None)
        torch.cuda.synchronize()
        assert int(word.item()) & ops.RANGE_TAGS['heads_tail']
        word.zero_()


@pytest.mark.parametrize('L', [52, 131, 230, 402, 600])
def test_seq_attn(ops, L):
    """L = 52: one key per lane slot, partial; 131: three key slots (4-slot instantiation), one query block; 230: two query
    blocks of 128; 402: three query blocks, 7 key slots (8-slot instantiation); 600: the 12-slot / 4-wave instantiation, two query blocks.  One sample has every key
    masked but two."""
    from oracle import abx_oracle as O
    B, H, D = 2, 32, 17
    qkv = torch.randn(B, L, H, 3 * D, generator=g(34))
    bias = torch.randn(B, H, L, L, generator=g(35))
    gate = torch.randn(B, L, H * D, generator=g(36))
    mask = torch.rand(B, L, generator=g(37)) > 0.2
    mask[1, 2:] = False
    mask[1, :2] = True
    out = torch.full((B * L, H * D), float('nan'), device=DEV)
    ops.seq_attn(qkv.view(B * L, -1).to(DEV), bias.to(DEV), mask.float().to(DEV), gate.view(B * L, -1).to(DEV), out, B, L)
    t = qkv.permute(0, 2, 1, 3)[:, None].double()
    q, k, v = torch.chunk(t, 3, dim=-1)
    o = O._attention(q, k, v, bias.double(), mask[:, None, :], D)[:, 0] * torch.sigmoid(gate.double())
    check(out.view(B, L, -1), o, 5e-6, 'seq_attn')


@pytest.mark.parametrize('L,B', [(37, 2), (131, 2), (402, 2), (37, 13), (52, 9)])
def test_ipa_core(ops, params, cfg, L, B):
    """IPA core against the oracle; L = 37: one partial 12-query block tail and a single 64-key wave task per head,
    L = 131: three key tasks per head, 11 query blocks (last one 11 of 12); L = 402: 7 key tasks per head, 34 query blocks, 78 KB of
    logits per workgroup.  B = 13 / 9: one full round of 8 samples pinned to the XCDs + 5 / 1 samples dealt workgroup by workgroup
    (round 6: the weights kernel's grid for B % 8 != 0)."""
    from oracle import abx_oracle as O
    c = cfg.model.heads.diffusion_module.IPA
    s = torch.randn(B, L, 256, generator=g(40))
    z = torch.randn(B, L, L, 128, generator=g(41))
    quat = torch.nn.functional.normalize(torch.randn(B, L, 4, generator=g(42)), dim=-1)
    rots = O.quat_to_rot(quat)
    trans = torch.randn(B, L, 3, generator=g(43)) * 3
    mask = (torch.rand(B, L, generator=g(44)) > 0.15).float()
    p = dict(params)
    pre = O.P_IPA + 'attention_module.'
    p[pre + 'final_proj.weight'] = torch.eye(2112)
    p[pre + 'final_proj.bias'] = torch.zeros(2112)
    ref = O.ipa_attention(p, s, z, mask, rots, trans, c)                        # (B,L,2112) feature concat
    from abx_amd.model.forward import Packed
    P = Packed({k: v for k, v in params.items()}, DEV)
    M1 = B * L
    proj = torch.empty(M1, 1152, device=DEV)
    ops.gemm(s.view(M1, 256).to(DEV), P.wt[pre + 'proj'], proj, bias=P.b[pre + 'proj'])
    bias2d = torch.empty(B * L * L, 12, device=DEV)
    ops.gemm(z.view(-1, 128).to(DEV), P.wt[pre + 'proj_pair'], bias2d, bias=P.b[pre + 'proj_pair'], alpha=P.ipa_w2d)
    qp = torch.empty(ops.ipa_qpack_numel(B, L), device=DEV); kp = torch.empty(M1 * 12 * 28, device=DEV); vp = torch.empty(M1 * 12 * 40, device=DEV)
    Rd, td = rots.reshape(M1, 9).contiguous().to(DEV), trans.reshape(M1, 3).contiguous().to(DEV)
    ops.ipa_pack(proj, Rd, td, qp, kp, vp, B, L, P.ipa_ws)
    feat = torch.full((M1, 2112), float('nan'), device=DEV)
    ops.ipa_attn(qp, kp, vp, bias2d, z.to(DEV).contiguous(), mask.to(DEV), Rd, td, P.ipa_pw, feat, B, L)
    f, r = feat.view(B, L, 2112).cpu(), ref
    for name, sl in (('scalar', slice(0, 192)), ('points', slice(192, 480)), ('norms', slice(480, 576)), ('pair', slice(576, 2112))):
        check(f[..., sl], r[..., sl], 2e-5, 'ipa ' + name)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('L,B', [(37, 3), (128, 2), (230, 2), (352, 1)])
def test_opm_out_without_the_feature_tensor(ops, L, B):
    """Round 6: OuterProductMean's output projection without the (B, L, L, 128) feature tensor (seqformer.py:395-411): z += l_j . (diag(r_i) W1 + W2)
    + (b - r_i . W2), a workgroup per (b, i) row of the pair tensor (abx_opm_out_fwd) - against the reference's expression in fp64
    (out_proj([l_j * r_i | l_j - r_i]) + z) and against the two launches it replaces (abx_opm_features + the split-f16 GEMM); masked residues
    (zero rows of l / r), ragged 32-position groups, in place."""
    ge = g(300 + L)
    lr = torch.randn(B * L, 128, generator=ge)
    keep = (torch.rand(B * L, generator=ge) > 0.15).float()
    lr = lr * keep[:, None]
    W = torch.randn(192, 128, generator=ge) / 128 ** 0.5            # out_proj.weight
    bias = 0.1 * torch.randn(192, generator=ge)
    z = torch.randn(B * L * L, 192, generator=ge) * 2 + 0.3
    left, right = lr[:, :64].view(B, L, 64).double(), lr[:, 64:].view(B, L, 64).double()
    feat = torch.cat([left[:, None, :, :] * right[:, :, None, :], left[:, None, :, :] - right[:, :, None, :]], -1)      # [b, i, j, :]
    ref = feat.reshape(-1, 128) @ W.double().t() + bias.double() + z.double()
    Wt = W.t().contiguous().to(DEV)
    zd = z.clone().to(DEV)
    ops.range_words(DEV).zero_()
    ops.opm_out(lr.to(DEV), Wt, bias.to(DEV), zd, B, L)
    check(zd, ref, 3e-6, f'fused OPM vs fp64, L={L}')
    assert not any(ops.range_words(DEV).tolist())
    z2 = z.clone().to(DEV)
    f32 = torch.empty(B * L * L, 128, device=DEV)
    ops.opm_features(lr.to(DEV), f32, B, L, 64)
    ops.gemm(f32, Wt, z2, bias=bias.to(DEV), B3=ops.split_weights(Wt), resid=z2, exact=2)
    check(zd, z2, 3e-6, 'fused OPM vs features + GEMM')
    # a right-projection row beyond the operand range of the U image (|r W1| >= 4095): NaN in exactly that (b, i) row of the pair tensor, range word set
    lr2 = lr.clone()
    bb = min(B - 1, 1)
    lr2[bb * L + 1, 64:] = 3.0e5
    z3 = z.clone().to(DEV)
    ops.opm_out(lr2.to(DEV), Wt, bias.to(DEV), z3, B, L)
    assert ops.range_words(DEV).tolist()[ops.RANGE_SLOT] & ops.RANGE_TAGS['gemm']
    ops.range_words(DEV).zero_()
    bad = ~torch.isfinite(z3).all(-1).view(B, L, L)
    assert bad[bb, 1].all() and int(bad.sum()) == L, int(bad.sum())
    ok = ~bad.view(-1)
    assert torch.equal(z3[ok], zd[ok])


@pytest.mark.parametrize('L,B,shared', [(37, 3, False), (128, 2, True), (230, 2, True)])
def test_assemble_pair_with_the_seq_attention_bias(ops, L, B, shared):
    """Round 6: abx_assemble_pair_bias = abx_assemble_pair + the sequence attention's pair-bias projection (LayerNorm folded, 192 -> 32 heads,
    (b, h, i, j) store) from one pass over the pair rows: z0 against the separate assembly (fp32 LayerNorm, another summation order: 2e-6) and
    against fp64, the bias against the projection launch of the same z0 and against fp64; shared / per-sample static context, no prev_pair
    (first call), ragged position groups."""
    ge = g(400 + L)
    ps = torch.randn(1 if shared else B, L, L, 128, generator=ge).to(DEV)
    temb = torch.randn(B, 32, generator=ge).to(DEV)
    prev = (torch.randn(B, L, L, 192, generator=ge) * 2 + 0.5).to(DEV)
    ga, be = (1 + 0.1 * torch.randn(192, generator=ge)).to(DEV), (0.1 * torch.randn(192, generator=ge)).to(DEV)
    ppos = torch.randint(0, 15, (B, L, L), generator=ge).to(DEV)
    ptab = torch.randn(15, 192, generator=ge).to(DEV)
    Wp, bp = torch.randn(32, 192, generator=ge) / 192 ** 0.5, 0.1 * torch.randn(32, generator=ge)
    g2, b2 = 1 + 0.1 * torch.randn(192, generator=ge), 0.1 * torch.randn(192, generator=ge)
    w, cs, bi = fold_ln(Wp, bp, g2, b2)
    w3 = ops.split_weights(w)
    for with_prev in (True, False):
        z_ref = torch.empty(B, L, L, 192, device=DEV)
        ops.assemble_pair(ps, temb, prev if with_prev else None, ga, be, ppos if with_prev else None, ptab, z_ref, B, L, 128, 32)
        b_ref = torch.empty(B, 32, L * L, device=DEV)
        ops.gemm(z_ref.view(B, L * L, 192), w, b_ref.transpose(1, 2), bias=bi, ln=(None, cs), B3=w3, exact=2)
        z0 = torch.full((B, L, L, 192), float('nan'), device=DEV)
        bT = torch.full((B, 32, L * L), float('nan'), device=DEV)
        ops.assemble_pair_bias(ps, temb, prev if with_prev else None, ga, be, ppos if with_prev else None, ptab, z0, w3, cs, bi, bT, B, L)
        check(z0, z_ref, 2e-6, f'fused assembly vs abx_assemble_pair (prev={with_prev})')
        check(bT, b_ref, 5e-6, f'fused pair bias vs the projection launch (prev={with_prev})')
        zd = z_ref.double().cpu()
        ln = (zd - zd.mean(-1, keepdim=True)) / torch.sqrt(zd.var(-1, unbiased=False, keepdim=True) + 1e-5) * g2.double() + b2.double()
        ref = (ln @ Wp.double().t() + bp.double()).permute(0, 3, 1, 2).reshape(B, 32, L * L)
        check(bT, ref, 5e-6, 'fused pair bias vs fp64')


def test_embedding_assembly(ops, params):
    from oracle import abx_oracle as O
    B, L, Lab = 2, 20, 16
    t64 = torch.tensor([0.5050505050505051, 0.02], dtype=torch.float64)
    temb = torch.empty(B, 32, device=DEV)
    ops.timestep_embedding(t64.to(DEV), 32, temb)
    check(temb, O.timestep_embedding(t64, 32), 5e-7, 'timestep embedding (fp64 t)')
    t32 = torch.tensor([1.0, 0.3], dtype=torch.float32)
    ops.timestep_embedding(t32.double().to(DEV), 32, temb)
    check(temb, O.timestep_embedding(t32, 32), 5e-7, 'timestep embedding (fp32 t)')
    seq_static = torch.randn(B, L, 512, generator=g(50))
    pair_static = torch.randn(B, L, L, 128, generator=g(51))
    seq_t = torch.randint(0, 20, (B, L), generator=g(52))
    prev_seq = torch.randn(B, L, 544, generator=g(53)); prev_pair = torch.randn(B, L, L, 192, generator=g(54))
    prev_pos = torch.randint(0, 15, (B, L, L), generator=g(55))
    P = {k: v.to(DEV) for k, v in params.items()}
    so = torch.empty(B, L, 544, device=DEV); po = torch.empty(B, L, L, 192, device=DEV)
    ops.assemble_seq(seq_static.to(DEV), P[O.P_SEQF + 'proj_aa_type.weight'], seq_t.to(DEV), Lab, temb, prev_seq.to(DEV),
                     P[O.P_SEQF + 'prev_seq_norm.weight'], P[O.P_SEQF + 'prev_seq_norm.bias'], so, B, L, 512, 32)
    pst = torch.empty(B * L * L, 2, device=DEV)
    ops.assemble_pair(pair_static.to(DEV), temb, prev_pair.to(DEV), P[O.P_SEQF + 'prev_pair_norm.weight'],
                      P[O.P_SEQF + 'prev_pair_norm.bias'], prev_pos.to(DEV), P[O.P_SEQF + 'proj_prev_pos.weight'], po, B, L, 128, 32,
                      stats_out=pst)
    te = temb.cpu()
    sa = seq_static.clone()
    sa[:, :Lab] += params[O.P_SEQF + 'proj_aa_type.weight'][seq_t[:, :Lab]]
    sa = torch.cat([sa, te[:, None].expand(B, L, 32)], -1) + O.lnorm(params, O.P_SEQF + 'prev_seq_norm', prev_seq)
    pa = torch.cat([pair_static, te[:, None, None].expand(B, L, L, 32), te[:, None, None].expand(B, L, L, 32)], -1)
    pa = pa + O.lnorm(params, O.P_SEQF + 'prev_pair_norm', prev_pair) + params[O.P_SEQF + 'proj_prev_pos.weight'][prev_pos]
    check(so, sa, 2e-6, 'assemble_seq')
    check(po, pa, 2e-6, 'assemble_pair')
    pa2 = pa.double().view(-1, 192)
    check(pst[:, 0], pa2.mean(-1), 2e-6, 'assemble_pair fused stats mean')
    check(pst[:, 1], 1 / torch.sqrt(pa2.var(-1, unbiased=False) + 1e-5), 2e-6, 'assemble_pair fused stats rstd')
    # without the fused statistics the model dimensions (128 + 2*32) take the 16-byte-vector kernel
    po2 = torch.full_like(po, float('nan'))
    ops.assemble_pair(pair_static.to(DEV), temb, prev_pair.to(DEV), P[O.P_SEQF + 'prev_pair_norm.weight'],
                      P[O.P_SEQF + 'prev_pair_norm.bias'], prev_pos.to(DEV), P[O.P_SEQF + 'proj_prev_pos.weight'], po2, B, L, 128, 32)
    check(po2, pa, 2e-6, 'assemble_pair (vector kernel)')
    # shared (broadcast) static context
    ops.assemble_pair(pair_static[:1].contiguous().to(DEV), temb, None, None, None, None, None, po, B, L, 128, 32)
    check(po[1, ..., :128], pair_static[0], 0, 'assemble_pair broadcast')
    # OPM features + pair mask
    lr = torch.randn(B * L, 128, generator=g(56))
    feat = torch.empty(B * L * L, 128, device=DEV)
    ops.opm_features(lr.to(DEV), feat, B, L, 64)
    left, right = lr[:, :64].view(B, L, 64), lr[:, 64:].view(B, L, 64)
    ref = torch.cat([left[:, None, :, :] * right[:, :, None, :], left[:, None, :, :] - right[:, :, None, :]], -1)
    check(feat.view(B, L, L, 128), ref, 0, 'opm features')
    m = (torch.rand(B, L, generator=g(57)) > 0.3).float()
    pm = torch.empty(B * L * L, device=DEV)
    ops.pair_mask(m.to(DEV), pm, B, L)
    check(pm.view(B, L, L), m[:, :, None] * m[:, None, :], 0, 'pair mask')


def test_frames_scores_heads(ops, params, cfg, oracle_diffuser):
    from oracle import abx_oracle as O
    from abx_amd.model.forward import Packed
    B, L = 2, 33
    n = B * L
    ge = g(60)
    rig = torch.cat([torch.nn.functional.normalize(torch.randn(B, L, 4, generator=ge), dim=-1), torch.randn(B, L, 3, generator=ge) * 10], -1)
    fixed = (torch.rand(B, L, generator=ge) > 0.4).int()
    dev = lambda x: x.contiguous().to(DEV)
    bufs = [torch.empty(n, k, device=DEV) for k in (4, 3, 4, 3, 9, 4)]
    init_q, init_t, cur_q, cur_t, cur_R, delta_q = bufs
    for dt_ in (torch.float32, torch.float64):
        ops.frames_init(dev(rig.to(dt_)), *bufs, n, 10.0)
        check(cur_R.view(B, L, 3, 3), O.quat_to_rot(rig[..., :4]), 2e-6, 'frames_init R')
        check(cur_t.view(B, L, 3), rig[..., 4:] / 10, 1e-7, 'frames_init t')
    # three rigid updates against the oracle recursion
    q, t, R, dq = rig[..., :4].clone(), rig[..., 4:] / 10, O.quat_to_rot(rig[..., :4]), torch.zeros(B, L, 4)
    dq[..., 0] = 1
    dm = (1 - fixed[..., None]).float()
    for it in range(3):
        upd = torch.randn(B, L, 6, generator=ge) * 0.3
        ops.rigid_update(dev(upd.view(n, 6)), dev(fixed.view(-1)), init_q, init_t, cur_q, cur_t, cur_R, delta_q, n, 10.0)
        dq = O.quat_precompose_vec(dq, upd[..., :3])
        q = O.quat_precompose_vec(q, upd[..., :3])
        t = t + torch.einsum('...rd,...d->...r', R, upd[..., 3:])
        q = dm * q + (1 - dm) * rig[..., :4]
        t = dm * t + (1 - dm) * (rig[..., 4:] / 10)
        R = O.quat_to_rot(q)
    check(cur_q.view(B, L, 4), q, 3e-6, 'rigid_update q')
    check(cur_t.view(B, L, 3), t, 3e-6, 'rigid_update t')
    check(delta_q.view(B, L, 4), dq, 3e-6, 'rigid_update delta')
    # scores: fp64 t (loop) and fp32 t (warm-up)
    D = oracle_diffuser
    so3 = D.so3
    for tvals in (torch.tensor([0.5050505050505051, 0.02], dtype=torch.float64), torch.tensor([1.0, 0.37], dtype=torch.float32)):
        is32 = tvals.dtype == torch.float32
        rot = torch.empty(n, 3, device=DEV)
        ts = torch.empty(n, 3, device=DEV, dtype=torch.float32 if is32 else torch.float64)
        rigids = torch.empty(n, 7, device=DEV)
        ops.scores(init_q=init_q, init_t=init_t, delta_q=delta_q, cur_t=cur_t, fixed_mask=dev(fixed.view(-1)), t=dev(tvals.double()),
                   t_is_f32=int(is32), score_norms=dev(so3._score_norms), num_sigma=1000, num_omega=1000,
                   discrete_sigma=dev(so3.discrete_sigma), discrete_omega=dev(so3.discrete_omega),
                   exp_max_sigma=float(torch.exp(torch.tensor(1.5))), exp_min_sigma=float(torch.exp(torch.tensor(0.1))),
                   min_b=float(torch.tensor(0.1)), bdiff=float(torch.tensor(19.9)), coord_scale=float(torch.tensor(0.1)),
                   position_scale=10.0, rot_score=rot, trans_score=ts, rigids=rigids, B=B, L=L)
        q_fin = dm * O.quat_multiply(rig[..., :4], dq) + (1 - dm) * rig[..., :4]
        ref_ts = D.calc_trans_score(rig[..., 4:], t * 10, tvals)
        ref_rs = D.calc_quat_score(rig[..., :4], q_fin, tvals)
        assert ref_ts.dtype == ts.dtype
        check(ts.view(B, L, 3), ref_ts, 3e-6, f'trans_score f32={is32}')
        check(rigids.view(B, L, 7), torch.cat([q_fin, t * 10], -1), 3e-6, 'rigids')
        # fixed residues have q0^-1 q_t == identity: their rot_score is rounding noise / 2e-6 in the reference too and is
        # discarded by the mask merge of FullDiffuser.reverse -> compare the diffused residues only
        dif = fixed.bool().logical_not()
        bad = ((rot.view(B, L, 3).cpu() - ref_rs).abs() > 1e-4 + 1e-4 * ref_rs.abs()).any(-1)[dif].float().mean()
        assert bad <= 0.03, f'rot_score bucket mismatches {bad}'
        assert torch.isfinite(rot).all()
    # torsions
    un = torch.randn(n, 7, 2, generator=ge); gt = torch.randn(n, 7, 2, generator=ge)
    ang = torch.empty(n, 7, 2, device=DEV)
    ops.torsion_finalize(dev(un), dev(gt), dev(fixed.view(-1)), ang, n)
    ref = torch.where(fixed.view(n, 1, 1).bool(), gt, O.l2_normalize(un))
    check(ang, ref, 2e-6, 'torsion_finalize')
    # sequence head tail: argmax + frames + atoms
    P = Packed(dict(params), DEV)
    logits = torch.randn(B, L, 20, generator=ge)
    seq_t = torch.randint(0, 21, (B, L), generator=ge)
    a37 = torch.as_tensor(__import__('abx_amd.residue_constants', fromlist=['x']).restype_atom37_to_atom14)[torch.randint(0, 20, (B, L), generator=ge)].long()
    angles = O.l2_normalize(torch.randn(B, L, 7, 2, generator=ge))
    rg = torch.cat([q_fin, t * 10], -1)
    seq0 = torch.empty(n, dtype=torch.int64, device=DEV); a14 = torch.empty(n, 14, 3, device=DEV); a37o = torch.empty(n, 37, 3, device=DEV)
    ops.seq_head_atoms(dev(logits), dev(fixed.view(-1)), dev(seq_t), dev(rg), dev(angles), dev(a37), P.default_frames, P.group_idx,
                       P.lit_pos, seq0, a14, a37o, n)
    s0 = logits.argmax(-1) * (1 - fixed) + seq_t * fixed
    assert torch.equal(seq0.cpu().view(B, L), s0)
    fR, ft = O.torsion_angles_to_frames(s0, O.quat_to_rot(rg[..., :4]), rg[..., 4:], angles)
    ref14 = O.frames_to_atom14(s0, fR, ft)
    check(a14.view(B, L, 14, 3), ref14, 3e-6, 'atom14')
    check(a37o.view(B, L, 37, 3), O.atom14_to_atom37(ref14, a37), 3e-6, 'atom37')
    # get_prev distogram
    pp = cfg.model.embeddings_and_seqformer.prev_pos
    sq = torch.square(torch.linspace(pp.min_bin, pp.max_bin, steps=pp.num_bins - 1))
    out = torch.empty(B, L, L, dtype=torch.int64, device=DEV)
    atoms = O.atom14_to_atom37(ref14, a37)
    ops.prev_pos(dev(atoms), dev(sq), out, B, L)
    refb = O.dgram_from_positions(O.pseudo_beta_v2(atoms), **dict(pp))
    assert (out.cpu() != refb).float().mean() < 2e-3
    # pLDDT
    lg = torch.randn(n, 50, generator=ge)
    pl = torch.empty(n, device=DEV)
    ops.plddt(dev(lg), pl, n, 50)
    centers = torch.arange(start=0.01, end=1.0, step=0.02)
    check(pl, (torch.softmax(lg, -1) * centers).sum(-1) * 100, 2e-6, 'plddt')


def _igso3_truth(sig, om, L=1000):
    """fp64 series + a first-order rounding bound of the fp32 evaluation (the score norm is a quotient of two cancelling
    1000-term series: where the series ~ 0 the REFERENCE's own fp32 value is rounding noise, so parity is only
    meaningful up to this condition-aware bound)."""
    ls = torch.arange(L, dtype=torch.float64)[None, None]
    e, o = sig.double()[:, None, None], om.double()[None, :, None]
    w = (2 * ls + 1) * torch.exp(-ls * (ls + 1) * e ** 2 / 2)
    hi, dhi = torch.sin(o * (ls + 0.5)), (ls + 0.5) * torch.cos(o * (ls + 0.5))
    lo, dlo = torch.sin(o / 2), 0.5 * torch.cos(o / 2)
    t1, t2 = w * hi / lo, w * (lo * dhi - hi * dlo) / lo ** 2
    ex, ds = t1.sum(-1), t2.sum(-1)
    # argument rounding of sin/cos((l+1/2) omega) in fp32 (|arg| up to ~3e3) dominates: d(arg) ~ eps * arg
    eps = 6e-8
    arg = (o * (ls + 0.5)).abs()
    e1 = (eps * (t1.abs() + (w * dhi / lo).abs() * arg / (ls + 0.5))).sum(-1)
    e2 = (eps * (t2.abs() + (w / lo ** 2).abs() * ((ls + 0.5) * hi.abs() * lo.abs() + dhi.abs() * dlo.abs()) * arg)).sum(-1)
    score = ds / (ex + 1e-4)
    bound = (e2 + score.abs() * e1) / (ex + 1e-4).abs()
    pdf = ex * (1 - torch.cos(o[..., 0])) / np.pi
    return pdf, score, bound, e1 * (1 - torch.cos(o[..., 0])) / np.pi


def test_igso3_tables_kernel(ops):
    gd = load_npz('igso3_small.npz')
    sig, om = tt(gd['small_sigma']), tt(gd['small_omega'])
    pdf = torch.empty(40, 40, device=DEV); cdf = torch.empty(40, 40, device=DEV); sn = torch.empty(40, 40, device=DEV)
    ops.igso3_tables(sig.to(DEV), om.to(DEV), pdf, cdf, sn)
    tp, tsn, bound, pb = _igso3_truth(sig, om)
    for name, got in (('kernel', sn.cpu()), ('reference', tt(gd['small_score_norms']))):
        viol = ((got.double() - tsn).abs() > 8 * bound + 1e-4 + 2e-5 * tsn.abs()).float().mean()
        assert viol == 0, f'{name}: score norms outside the rounding bound ({viol})'
    well = bound < 1e-3                                   # well-conditioned entries: plain parity with the reference
    assert well.float().mean() > 0.5
    d = (sn.cpu() - tt(gd['small_score_norms'])).abs()
    assert float(d[well].max()) < 5e-3, f'well-conditioned score norms differ by {float(d[well].max())}'
    assert ((pdf.cpu().double() - tp).abs() <= 8 * pb + 1e-6 + 1e-5 * tp.abs()).all()
    check(pdf, tt(gd['small_pdf']), 1e-4, 'igso3 pdf')
    check(cdf, tt(gd['small_cdf']), 1e-4, 'igso3 cdf')
    # rows of the full 1000x1000 tables used by the step tests
    big_sig, big_om = tt(gd['big_sigma']), tt(gd['big_omega'])
    pdf = torch.empty(1000, 1000, device=DEV); cdf = torch.empty(1000, 1000, device=DEV); sn = torch.empty(1000, 1000, device=DEV)
    ops.igso3_tables(big_sig.to(DEV), big_om.to(DEV), pdf, cdf, sn)
    i, j = gd['spot_i'], gd['spot_j']
    check(pdf.cpu()[i, j], tt(gd['spot_pdf']), 1e-4, 'igso3 big pdf spots')
    check(cdf.cpu()[i, j], tt(gd['spot_cdf']), 1e-4, 'igso3 big cdf spots')
    rows = gd['rows']
    _, tsn, bound, _ = _igso3_truth(big_sig[rows], big_om)
    for name, got in (('kernel', sn.cpu()[rows]), ('reference', tt(gd['rows_score_norms']))):
        viol = ((got.double() - tsn).abs() > 8 * bound + 1e-4 + 2e-5 * tsn.abs()).float().mean()
        assert viol == 0, f'{name}: big-table score norms outside the rounding bound ({viol})'
    well = bound < 1e-3
    d = (sn.cpu()[rows] - tt(gd['rows_score_norms'])).abs()
    assert float(d[well].max()) < 5e-3, f'well-conditioned big score norms differ by {float(d[well].max())}'


def test_reverse_step_golden(ops, cfg):
    """Reference FullDiffuser.reverse (fp64 state, recorded noise) vs abx_reverse_step: tokens exact, rigids to 1e-9."""
    from abx_amd.diffuser.full_diffuser import FullDiffuser
    s = load_npz('step_tiny.npz')
    D = FullDiffuser(cfg.diffuser)
    D.set_tables(torch.zeros(1000, 1000), torch.zeros(1000, 1000), torch.zeros(1000, 1000), DEV)
    dm = tt(s['diffuse_mask']).to(DEV)
    for i in range(3):
        gg = lambda k: tt(s[f's{i}.{k}']).to(DEV)
        noise = dict(z_rot=gg('z_rot'), z_trans=gg('z_trans'), jumps=gg('jumps'))
        rig, seq = D.reverse(rigid_t=gg('rigid_in'), seq_t=gg('seq_in'), rot_score=gg('rot_score'), trans_score=gg('trans_score'),
                             logits_t=gg('logits'), t=gg('t'), dt=float(s['dt']), diffuse_mask=dm, noise=noise)
        assert rig.dtype == torch.float64 and seq.dtype == torch.int64
        assert torch.equal(seq.cpu(), tt(s[f's{i}.seq_out'])), f'step {i} tokens'
        ref = tt(s[f's{i}.rigid_out'])
        err = (rig.cpu() - ref).abs().max().item()
        # step 0 starts from float32 rigids (sample_ref): its quaternion <-> rotation-vector maps run in fp32 in the
        # reference, so parity is at fp32 rounding; later steps carry float64 state and agree to 1e-9
        tol = 2e-6 if s[f's{i}.rigid_in'].dtype == np.float32 else 1e-9
        assert err < tol * max(1.0, ref.abs().max().item()), f'step {i} rigids err {err}'
        ts = D.score_scaling(gg('t'))[1]
        assert (ts.cpu() - tt(s[f's{i}.trans_score_scaling'])).abs().max() < 1e-12


def _token_diffuser(cfg):
    from abx_amd.diffuser.full_diffuser import FullDiffuser
    D = FullDiffuser(cfg.diffuser)
    D.set_tables(torch.zeros(1000, 1000), torch.zeros(1000, 1000), torch.zeros(1000, 1000), DEV)
    return D


def _token_step(D, x_t, logits, t, dt, dm=None, noise=None, **kw):
    """abx_reverse_step with neutral rigid inputs: returns (seq_out, rates * dt, applied jump counts)."""
    B, L = x_t.shape
    rig = torch.zeros(B, L, 7, dtype=torch.float64, device=DEV)
    rig[..., 0] = 1.0
    z3 = torch.zeros(B, L, 3, device=DEV)
    rates = torch.full((B, L, 20), -1.0, device=DEV)
    jumps = torch.full((B, L, 20), -1.0, device=DEV)
    if dm is None:
        dm = torch.ones(B, L, dtype=torch.int32, device=DEV)
    _, seq = D.reverse(rig, x_t.to(DEV), z3, z3.double(), logits.to(DEV), torch.full((B,), float(t), dtype=torch.float64, device=DEV),
                       dt, dm, noise=noise, rates_out=rates, jumps_out=jumps, **kw)
    return seq.cpu(), rates.cpu(), jumps.cpu()


def test_reverse_step_token_rates_vs_reference_and_oracle(ops, cfg):
    """the reverse RATES of the default product path (softmax -> ratio -> inner -> rate * dt,
    diffuser.hip) against the argument of the reference's own torch.poisson call (rates_tiny.npz) and against the oracle's
    closed-form restatement: t in {1, 0.5, 0.02}, dt in {0.01, 0.1}, tokens 0 / 19 / out of range, a masked residue block.
    dt arrives once as a host float and once as the 0-dim device tensor of the reference's loop (inference.py:198-199)."""
    from oracle import abx_oracle as O
    z = load_npz('rates_tiny.npz')
    x_t, lg = tt(z['x_t']), tt(z['logits'])
    B, L = x_t.shape
    D = _token_diffuser(cfg)
    seq_o = O.OracleSeq({'rate_const': float(z['rate_const'])})
    dm = torch.ones(B, L, dtype=torch.int32)
    dm[:, 30:] = 0                                                            # fixed residues keep their token
    for ci, c in enumerate(z['cases']):
        t, dt = float(z[c + '.t']), tt(z[c + '.dt'])
        ref = tt(z[c + '.lam'])
        jn = tt(z[c + '.jumps'])
        dt_arg = dt.to(DEV) if ci % 2 else float(dt)
        seq, lam, ja = _token_step(D, x_t, lg, t, dt_arg, dm=dm.to(DEV), noise=dict(jumps=jn.to(DEV)))
        ro, _ = seq_o.reverse_rates(x_t, lg, torch.tensor(t))
        lo = ro * dt
        assert torch.equal(lam == 0, ref == 0) and torch.equal(lam == 0, lo == 0)
        pos = ref > 0
        rel_o = ((lam - lo).abs() / lo.abs().clamp_min(1e-30))[pos]
        rel_r = ((lam - ref).abs() / ref.abs().clamp_min(1e-30))[pos]
        assert float(rel_o.max()) < 2e-6, (c, 'vs oracle', float(rel_o.max()))   # same closed-form q_t0: fp32 rounding only
        assert float(rel_r.max()) < 1e-4, (c, 'vs reference', float(rel_r.max()))  # the reference's fp32 eigh route (3e-5 off fp64)
        assert torch.equal(ja, jn)
        want = tt(z[c + '.x_new']).long()
        want = torch.where(dm.bool(), want, x_t)
        assert torch.equal(seq, want), c


def _peaky_logits(B, L, x_t, gen):
    lg = 3.0 * torch.randn(B, L, 20, generator=gen)
    peaked = torch.rand(B, L, generator=gen) < 0.3
    tgt = (x_t.clamp(0, 19) + torch.randint(1, 20, (B, L), generator=gen)) % 20
    pk = torch.zeros(B, L, 20)
    pk.scatter_(2, tgt[..., None], 30.0)
    return torch.where(peaked[..., None], pk, lg)


def test_reverse_step_uniform_driven_poisson_exact_tokens(ops, cfg):
    """the Poisson draw is a pure function of one uniform.  On identical uniforms (2e6 draws, rates * dt up
    to ~5) the kernel's jump counts equal the oracle's restatement of the inverse cdf bit for bit, and the tokens equal the
    oracle's end-to-end tau-leaping step (own rates) except where a uniform sits within fp32 rounding of a cdf edge."""
    from oracle import abx_oracle as O
    B, L = 50, 1000
    ge = g(301)
    x_t = torch.randint(0, 20, (B, L), generator=ge)
    lg = _peaky_logits(B, L, x_t, ge)
    u = (torch.randint(0, 1 << 24, (B, L, 20), generator=ge).float() + 0.5) / float(1 << 24)
    D = _token_diffuser(cfg)
    seq_o = O.OracleSeq({'rate_const': D.rate_const})
    n_big = 0
    for t, dt in ((0.02, 0.1), (0.5, 0.01)):
        seq, lam, ja = _token_step(D, x_t, lg, t, dt, noise=dict(u_jumps=u.to(DEV)))
        n_big += int((lam > 3).sum())
        want_j = torch.from_numpy(O.poisson_icdf(lam.numpy(), u.numpy()))
        assert torch.equal(ja, want_j), f'{int((ja != want_j).sum())} jump counts differ from the inverse-cdf restatement'
        diffs = torch.arange(20).view(1, 1, 20) - x_t[..., None]
        want = torch.clamp(x_t + (want_j * diffs).sum(-1), 0, 19).long()
        assert torch.equal(seq, want)
        # end to end against the oracle's own rates: a count may differ only where the two fp32 rate evaluations straddle u
        ro, _ = seq_o.reverse_rates(x_t, lg, torch.tensor(t))
        jo = torch.from_numpy(O.poisson_icdf((ro * np.float32(dt)).numpy(), u.numpy()))
        bad = ja != jo
        assert int(bad.sum()) <= 8, int(bad.sum())
        if bad.any():
            from scipy.stats import poisson
            lb, ub, kb = lam[bad].double().numpy(), u[bad].double().numpy(), torch.minimum(ja, jo)[bad].numpy()
            assert np.all(np.abs(poisson.cdf(kb, lb) - ub) < 2e-6 * np.maximum(1.0, lb))
        so = seq_o.reverse(x_t, lg, torch.tensor(t), torch.tensor(np.float32(dt)), u_jumps=u).long()
        assert int((seq != so).sum()) <= int(bad.any(-1).sum())
    assert n_big > 1000                                                    # the vectors reach large rates


def test_reverse_step_philox_poisson_statistics(ops, cfg):
    """the device-RNG path.  With flat logits every off-token rate of a residue is the same lam; the Philox
    jump counts over 4e5 draws must have the Poisson pmf (chi-square against scipy), mean = var = lam, for lam ~ 0.0016, 0.5, 2.1."""
    from scipy.stats import poisson
    B, L = 20, 1000
    ge = g(302)
    x_t = torch.randint(0, 20, (B, L), generator=ge)
    lg = torch.zeros(B, L, 20)
    D = _token_diffuser(cfg)
    D.seed = 4242
    ids = torch.arange(B, device=DEV) + 100
    for t, dt in ((0.5, 0.01), (0.02, 0.2), (0.02, 0.85)):
        seq, lam, ja = _token_step(D, x_t, lg, t, dt, sample_ids=ids, step=3)
        off = lam > 0
        assert int(off.sum()) == B * L * 19
        lam0 = float(lam[off].double().mean())
        assert float(lam[off].max() - lam[off].min()) < 1e-5 * lam0
        k = ja[off].double().numpy()
        n = k.size
        assert abs(k.mean() - lam0) < 5 * np.sqrt(lam0 / n), (k.mean(), lam0)
        assert abs(k.var() - lam0) < 5 * np.sqrt((lam0 + 2 * lam0 ** 2) / n) + 1e-6, (k.var(), lam0)
        kmax = int(max(3, poisson.ppf(1 - 1e-4, lam0)))
        obs = np.array([(k == i).sum() for i in range(kmax)] + [(k >= kmax).sum()], dtype=np.float64)
        exp = np.array([poisson.pmf(i, lam0) for i in range(kmax)] + [poisson.sf(kmax - 1, lam0)]) * n
        keep = exp > 5
        chi2 = float((((obs - exp) ** 2) / exp)[keep].sum())
        assert chi2 < 40 + 3 * keep.sum(), (chi2, obs, exp)
        # the current token never jumps onto itself and the applied jumps reproduce the tokens
        assert float(ja[~off].abs().max()) == 0
        diffs = torch.arange(20).view(1, 1, 20) - x_t[..., None]
        assert torch.equal(seq, torch.clamp(x_t + (ja * diffs).sum(-1), 0, 19).long())
        # another step index / another sample id gives other draws; the same key the same draws
        _, _, jb = _token_step(D, x_t, lg, t, dt, sample_ids=ids, step=3)
        _, _, jc = _token_step(D, x_t, lg, t, dt, sample_ids=ids, step=4)
        assert torch.equal(ja, jb) and not torch.equal(ja, jc)


def test_reverse_step_device_rng_properties(ops, cfg):
    """Philox path: deterministic, batch-composition invariant (keyed by sample id), fixed residues untouched."""
    from abx_amd.diffuser.full_diffuser import FullDiffuser
    D = FullDiffuser(cfg.diffuser)
    D.set_tables(torch.zeros(1000, 1000), torch.zeros(1000, 1000), torch.zeros(1000, 1000), DEV)
    B, L = 4, 50
    ge = g(70)
    rig = torch.cat([torch.nn.functional.normalize(torch.randn(B, L, 4, generator=ge), dim=-1), torch.randn(B, L, 3, generator=ge) * 10], -1).double().to(DEV)
    seq = torch.randint(0, 20, (B, L), generator=ge).to(DEV)
    rs = torch.randn(B, L, 3, generator=ge).to(DEV); tsx = torch.randn(B, L, 3, generator=ge).double().to(DEV)
    lg = torch.randn(B, L, 20, generator=ge).to(DEV)
    dm = (torch.rand(B, L, generator=ge) > 0.5).int().to(DEV)
    t = torch.full((B,), 0.5, dtype=torch.float64, device=DEV)
    ids = torch.arange(B, device=DEV) + 10
    D.seed = 123
    a = D.reverse(rig, seq, rs, tsx, lg, t, 0.01, dm, sample_ids=ids, step=7)
    b = D.reverse(rig, seq, rs, tsx, lg, t, 0.01, dm, sample_ids=ids, step=7)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    sub = [2, 0]
    c = D.reverse(rig[sub], seq[sub], rs[sub], tsx[sub], lg[sub], t[sub], 0.01, dm[sub], sample_ids=ids[sub], step=7, center=False)
    d = D.reverse(rig, seq, rs, tsx, lg, t, 0.01, dm, sample_ids=ids, step=7, center=False)
    assert torch.equal(c[0], d[0][sub]) and torch.equal(c[1], d[1][sub])
    fixed = dm == 0
    assert torch.equal(a[1][fixed], seq[fixed])
    assert (a[0][..., 4:][fixed] - rig[..., 4:][fixed]).abs().max() == 0
    assert torch.isfinite(a[0]).all() and int(a[1].min()) >= 0 and int(a[1].max()) <= 19
    e = D.reverse(rig, seq, rs, tsx, lg, t, 0.01, dm, sample_ids=ids, step=8)
    assert not torch.equal(a[0], e[0])


def test_reverse_step_device_rng_samples_share_no_noise(ops, cfg):
    """Philox counter layout: the draw counter must not walk into the sample-id word.  With zero scores the Gaussian
    draws can be read back from the outputs: z_trans from the R^3 update, z_rot from q_t^-1 (x) q_{t-1}.  Neighbouring sample ids
    must not share any of them (the round-1 layout made z_trans[b, l, 1:3] == z_rot[b + 1, l, 0:2]), and a sample's noise must
    not depend on the batch it sits in."""
    from abx_amd.diffuser.full_diffuser import FullDiffuser, quat_multiply, quat_to_rotvec
    D = FullDiffuser(cfg.diffuser)
    D.set_tables(torch.zeros(1000, 1000), torch.zeros(1000, 1000), torch.zeros(1000, 1000), DEV)
    B, L = 6, 40
    ge = g(71)
    q = torch.nn.functional.normalize(torch.randn(B, L, 4, generator=ge), dim=-1)
    q = q * torch.sign(q[..., :1])
    rig = torch.cat([q, torch.randn(B, L, 3, generator=ge) * 10], -1).double().to(DEV)
    seq = torch.randint(0, 20, (B, L), generator=ge).to(DEV)
    zero3 = torch.zeros(B, L, 3, device=DEV)
    lg = torch.randn(B, L, 20, generator=ge).to(DEV)
    dm = torch.ones(B, L, dtype=torch.int32, device=DEV)
    tv = 0.5
    t = torch.full((B,), tv, dtype=torch.float64, device=DEV)
    ids = torch.arange(B, device=DEV) + 3
    D.seed = 99
    dt = float(np.float32(0.01))
    out, _ = D.reverse(rig, seq, zero3, zero3.double(), lg, t, dt, dm, sample_ids=ids, step=5, center=False)
    out = out.cpu()
    rig_c = rig.cpu()
    # R^3: x1 = x - ((-0.5 b x) dt + sqrt(b) dt z)
    bt = D.min_b_f32 + tv * D.bdiff_f32
    x = rig_c[..., 4:] * D.coord_scale_f32
    x1 = out[..., 4:] * D.coord_scale_f32
    z_t = (x - x1 + 0.5 * bt * x * dt) / (np.sqrt(bt) * dt)
    # SO(3): q1 = q_t (x) quat(g sqrt(dt) z)
    sig = np.log(tv * D.exp_max_sigma + (1 - tv) * D.exp_min_sigma)
    g_so3 = np.sqrt(2 * (D.exp_max_sigma - D.exp_min_sigma) * sig / np.exp(sig))
    qinv = torch.cat([rig_c[..., :1], -rig_c[..., 1:4]], -1)
    z_r = quat_to_rotvec(quat_multiply(qinv, out[..., :4])) / (g_so3 * np.sqrt(np.float32(dt)))
    assert z_t.abs().max() < 7 and z_r.abs().max() < 7 and z_t.std() > 0.8 and z_r.std() > 0.8      # they ARE unit Gaussians
    allz = torch.cat([z_r, z_t], -1)                                     # (B, L, 6) draws of (sample, residue)
    for b in range(B - 1):
        for l in range(L):
            d = (allz[b, l][:, None] - allz[b + 1, l][None, :]).abs()
            assert d.min() > 1e-7, f'samples {b} and {b + 1} share a Gaussian draw at residue {l}'
    # the same sample id in another batch composition sees the same noise
    sub = [4, 1]
    out2, seq2 = D.reverse(rig[sub], seq[sub], zero3[sub], zero3[sub].double(), lg[sub], t[sub], dt, dm[sub], sample_ids=ids[sub],
                           step=5, center=False)
    full, seqf = D.reverse(rig, seq, zero3, zero3.double(), lg, t, dt, dm, sample_ids=ids, step=5, center=False)
    assert torch.equal(out2, full[sub]) and torch.equal(seq2, seqf[sub])
    one, seq1 = D.reverse(rig[2:3], seq[2:3], zero3[2:3], zero3[2:3].double(), lg[2:3], t[2:3], dt, dm[2:3], sample_ids=ids[2:3],
                          step=5, center=False)
    assert torch.equal(one, full[2:3]) and torch.equal(seq1, seqf[2:3])


def test_clash_grad_vs_oracle_autograd(ops):
    """Row G: abx_clash_grad energies and ANALYTIC gradients against the fp64 torch restatement and its autograd gradient (which
    is the finite-difference limit), on a compact random two-chain + antigen complex where thousands of atom pairs overlap and
    peptide bonds / bond angles leave their flat bottoms; the frame pull-back (sum of atom gradients, torque about the frame origin)
    against the same gradients.  Both link rules: by chain id only (cal_vio.py:51) and by chain id + consecutive residue numbers."""
    from oracle import abx_oracle as O
    from abx_amd import residue_constants as rc
    B, L = 2, 70
    ge = g(90)
    aatype = torch.randint(0, 20, (B, L), generator=ge)
    aatype[0, 11] = 4; aatype[0, 40] = 4            # a cysteine pair (SG-SG excluded)
    aatype[:, 20] = 14                              # a proline after a peptide bond
    chain = torch.cat([torch.zeros(30), torch.ones(25), 17 * torch.ones(15)]).int()[None].repeat(B, 1)     # 17: chain ids are not 4-bit
    residx = torch.cat([torch.arange(30), torch.arange(25) + 512, torch.tensor([3, 4, 5, 9, 10, 11, 12, 40, 41, 42, 43, 44, 45, 46, 47])]).int()[None].repeat(B, 1)
    ca = torch.cumsum(1.6 * torch.randn(B, L, 3, generator=ge), dim=1)            # compact walk: many clashes
    x = ca[:, :, None] + 1.2 * torch.randn(B, L, 14, 3, generator=ge)
    mask = torch.as_tensor(rc.restype_atom14_mask)[aatype].clone()
    mask[1, 5] = False                               # a residue without atoms
    mask[0, 33, 4:] = False
    mask[1, 44, 1] = False                           # a missing CA: its angle terms drop out, the bond stays
    # put C(i) / N(i+1) of consecutive residues near bond length for half of the pairs so that both branches of the flat bottom occur
    x[:, 1:, 0] = x[:, :-1, 2] + torch.tensor([1.33, 0., 0.]) + 0.25 * torch.randn(B, L - 1, 3, generator=ge) * (torch.rand(B, L - 1, 1, generator=ge) > 0.5)
    # and near-ideal backbone angles on a third of the pairs: CA(i) and CA(i+1) placed at the literature cosines, slightly perturbed
    third = torch.rand(B, L - 1, 1, generator=ge) > 0.66
    x[:, :-1, 1] = torch.where(third, x[:, :-1, 2] + 1.52 * torch.tensor([-0.4473, 0.8944, 0.]) + 0.05 * torch.randn(B, L - 1, 3, generator=ge), x[:, :-1, 1])
    kw = dict(overlap_tolerance=1.5, between_chain_factor=0.2, bond_tolerance_factor=12.0, w_clash=0.7, w_bond=1.3, w_angle=0.9)
    t0 = x[:, :, 1].clone()
    for rx in (None, residx):
        e, ga, gt, gr = ops.clash_grad(x.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), residx=None if rx is None else rx.to(DEV), **kw)
        xd = x.double().requires_grad_(True)
        ec, eb, ea = O.violation_energy(xd, mask, aatype, chain, residx=rx, **kw)
        (ec.sum() + eb.sum() + ea.sum()).backward()
        assert float(ec.detach().min()) > 10 and float(eb.detach().min()) > 0.1 and float(ea.detach().min()) > 0.1, (ec, eb, ea)   # active terms
        check(e[:, 0].cpu(), ec.detach(), 2e-5, 'clash energy')
        check(e[:, 1].cpu(), eb.detach(), 2e-5, 'bond energy')
        check(e[:, 2].cpu(), ea.detach(), 2e-5, 'angle energy')
        gref = xd.grad * mask[..., None]
        check(ga.cpu() * mask[..., None], gref, 5e-5, 'atom gradients')
        check(gt.cpu(), gref.sum(2), 5e-5, 'frame translation gradient')
        tq = torch.cross(xd.detach() - t0.double()[:, :, None], gref, dim=-1).sum(2)
        check(gr.cpu(), tq, 5e-5, 'frame rotation gradient (torque)')
    # the two link rules differ exactly by the terms across the numbering gaps of the third chain
    e_chain = ops.clash_grad(x.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), **kw)[0]
    e_res = ops.clash_grad(x.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), residx=residx.to(DEV), **kw)[0]
    assert (e_chain[:, 1] >= e_res[:, 1]).all() and (e_chain[:, 1] > e_res[:, 1]).any()
    # the peptide terms alone (clash off): gradient of bond + angle energies only, against autograd
    kw2 = dict(kw, w_clash=0.0)
    e, ga = ops.clash_grad(x.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), residx=residx.to(DEV), **kw2)[:2]
    xd = x.double().requires_grad_(True)
    ec, eb, ea = O.violation_energy(xd, mask, aatype, chain, residx=residx, **kw2)
    (eb.sum() + ea.sum()).backward()
    assert float(e[:, 0].abs().max()) == 0
    check(ga.cpu() * mask[..., None], xd.grad * mask[..., None], 2e-5, 'peptide-term gradients')
    # central finite difference of the KERNEL's own fp32 energy along a random direction, on a looser geometry (a few hundred
    # overlapping pairs: with the compact complex above the fp32 rounding of an energy of ~1e5 would swamp the difference)
    ca2 = torch.cumsum(3.0 * torch.randn(B, L, 3, generator=ge), dim=1)
    x2 = ca2[:, :, None] + 1.0 * torch.randn(B, L, 14, 3, generator=ge)
    x2[:, 1:, 0] = x2[:, :-1, 2] + torch.tensor([1.6, 0., 0.])                    # every peptide bond stretched beyond the flat bottom
    dirn = torch.randn(B, L, 14, 3, generator=ge) * mask[..., None]
    kw_cb = dict(kw, w_angle=0.0)                                                  # clash + bond: piecewise linear in the distances
    run = lambda xx: ops.clash_grad(xx.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), **kw_cb)
    e2, ga2 = run(x2)[:2]
    assert 1.0 < float(e2.sum(1).min()) and float(e2.sum(1).max()) < 5e3, e2
    h = 5e-3                    # the energy is piecewise linear in the distances: a smaller step crosses fewer hinges
    fd = (run(x2 + h * dirn)[0].sum(1).cpu().double() - run(x2 - h * dirn)[0].sum(1).cpu().double()) / (2 * h)
    an = (ga2.cpu().double() * dirn.double()).sum((1, 2, 3))
    assert ((fd - an).abs() <= 4e-2 * an.abs() + 0.3).all(), (fd, an)
    # the angle terms alone (smooth between their hinges): a smaller step
    kw_a = dict(kw, w_clash=0.0, w_bond=0.0)
    run_a = lambda xx: ops.clash_grad(xx.to(DEV), mask.to(DEV), aatype.to(DEV), chain.to(DEV), t0.to(DEV), **kw_a)
    e3, ga3 = run_a(x2)[:2]
    assert float(e3[:, 2].min()) > 1.0 and float(e3[:, :2].abs().max()) == 0
    h = 1e-3
    fd = (run_a(x2 + h * dirn)[0].sum(1).cpu().double() - run_a(x2 - h * dirn)[0].sum(1).cpu().double()) / (2 * h)
    an = (ga3.cpu().double() * dirn.double()).sum((1, 2, 3))
    assert ((fd - an).abs() <= 4e-2 * an.abs() + 0.1).all(), (fd, an)


def test_clash_grad_peptide_terms_vs_reference_cal_vio(ops):
    """The bond / angle energies of abx_clash_grad against the UNMODIFIED reference's between_residue_bond_loss (cal_vio.py:29-110;
    vio_pdb.npz: its per-residue flat-bottom losses on the two shipped complexes and perturbed copies), summed over its masks."""
    z = load_npz('vio_pdb.npz')
    for c in z['cases']:
        code = str(c).split('.')[0]
        p = load_npz(f'pdb_{code}.npz')
        m, ch, aa = tt(p['batch.atom14_gt_exists']), tt(p['batch.chain_id']), tt(p['batch.seq'])
        x = tt(z[f'{c}.pos'])
        e = ops.clash_grad(x.to(DEV), m.to(DEV), aa.to(DEV), ch.to(DEV), x[:, :, 1].contiguous().to(DEV), w_clash=0.0)[0].cpu()
        link = tt(z[f'{c}.has_no_gap_mask']).float()
        mf = m.float()
        m_ca, m_c, m_n, m_ca2 = mf[:, :-1, 1], mf[:, :-1, 2], mf[:, 1:, 0], mf[:, 1:, 1]
        want_b = (tt(z[f'{c}.c_n_loss_per_residue']) * m_c * m_n * link).sum()
        want_a = (tt(z[f'{c}.ca_c_n_loss_per_residue']) * m_ca * m_c * m_n * link).sum() + (tt(z[f'{c}.c_n_ca_loss_per_residue']) * m_c * m_n * m_ca2 * link).sum()
        assert abs(float(e[0, 1]) - float(want_b)) < 1e-4 + 2e-5 * float(want_b), (c, float(e[0, 1]), float(want_b))
        assert abs(float(e[0, 2]) - float(want_a)) < 1e-4 + 2e-5 * float(want_a), (c, float(e[0, 2]), float(want_a))
    p = load_npz('pdb_6qd7.npz')
    x = tt(z['6qd7.s0.pos'])
    args = (x.to(DEV), tt(p['batch.atom14_gt_exists']).to(DEV), tt(p['batch.seq']).to(DEV), tt(p['batch.chain_id']).to(DEV), x[:, :, 1].contiguous().to(DEV))
    assert float(ops.clash_grad(*args, w_clash=0.0)[0][0, 1]) > 1.0                       # the chain-only rule: a bond across the patch gap
    assert float(ops.clash_grad(*args, w_clash=0.0, residx=tt(p['batch.residx']).to(DEV))[0][0, 1]) == 0.0


@pytest.mark.parametrize('L', [64, 70])
def test_gemm_dual_proj_out_times_gate(ops, L):
    """The tail of the TriangleMultiplication as one dual GEMM: LN(product, channel-major) @ Wo * sigmoid(LN(z) @ Wg + bg) + z,
    with padded pair rows when L % 4 != 0 (A2 / C / resid in the unpadded pair tensor), against fp64."""
    Bc = 3
    LL, Lp = L * L, (L + 3) // 4 * 4
    ge = g(120)
    z = torch.randn(Bc, LL, 192, generator=ge) * 2 + 0.5
    tt_full = torch.randn(Bc, 128, L, Lp, generator=ge)                  # channel-major product with padded rows
    Wo, bo = torch.randn(128, 192, generator=ge) / 11, torch.randn(192, generator=ge) * 0.1
    Wg, bg = torch.randn(192, 192, generator=ge) / 14, torch.randn(192, generator=ge) * 0.1
    g1, b1 = 1 + 0.1 * torch.randn(128, generator=ge), 0.1 * torch.randn(128, generator=ge)
    g2, b2 = 1 + 0.1 * torch.randn(192, generator=ge), 0.1 * torch.randn(192, generator=ge)
    wo, cso, bio = fold_ln(Wo.t().contiguous(), bo, g1, b1)
    wg, csg, big = fold_ln(Wg.t().contiguous(), bg, g2, b2)
    zd = z.clone().to(DEV)
    out = torch.full_like(zd, float('nan'))         # out of place: other column tiles still read the rows of z for their gate
    tt = tt_full.reshape(Bc, 128, L * Lp).to(DEV)
    pad = (L, Lp) if Lp != L else None
    ops.gemm(tt.transpose(1, 2), wo, out, bias=bio, ln=(None, cso), B3=ops.split_weights(wo), resid=zd,
             pair=pad, c_pair=pad is not None, dual=(zd, ops.split_weights(wg), csg, big), exact=2)
    zd = out
    x = tt_full[..., :L].permute(0, 2, 3, 1).reshape(Bc, LL, 128).double()
    ln = lambda v, ga, be: (v - v.mean(-1, keepdim=True)) / torch.sqrt(v.var(-1, unbiased=False, keepdim=True) + 1e-5) * ga.double() + be.double()
    ref = (ln(x, g1, b1) @ Wo.double() + bo.double()) * torch.sigmoid(ln(z.double(), g2, b2) @ Wg.double() + bg.double()) + z.double()
    check(zd, ref, 3e-6, f'dual gemm L={L}')


@pytest.mark.parametrize('L,Bc', [(64, 3), (70, 3), (128, 2), (118, 2)])
def test_gemm_dual_walk_variants_bit_identical(ops, L, Bc):
    """Round 6: the tri-mul tail on ONE block per 128 rows (gate walk over all 192 columns once, its sigmoid values kept in registers; the
    product rows against proj_out in two column passes; AbxGemm.tune bit 7 - measured slower, not the default) against the two-tile kernel
    of rounds 2 - 5: the same products in the same order into every accumulator, the same gate expression - equal bit for bit, padded
    pair rows (L % 4 != 0) and ragged last row tiles included."""
    LL, Lp = L * L, (L + 3) // 4 * 4
    ge = g(125 + L)
    z = (torch.randn(Bc, LL, 192, generator=ge) * 2 + 0.5).to(DEV)
    tt = torch.randn(Bc, 128, L * Lp, generator=ge).to(DEV)
    Wo, bo = torch.randn(128, 192, generator=ge) / 11, torch.randn(192, generator=ge) * 0.1
    Wg, bg = torch.randn(192, 192, generator=ge) / 14, torch.randn(192, generator=ge) * 0.1
    g1, b1 = 1 + 0.1 * torch.randn(128, generator=ge), 0.1 * torch.randn(128, generator=ge)
    g2, b2 = 1 + 0.1 * torch.randn(192, generator=ge), 0.1 * torch.randn(192, generator=ge)
    wo, cso, bio = fold_ln(Wo.t().contiguous(), bo, g1, b1)
    wg, csg, big = fold_ln(Wg.t().contiguous(), bg, g2, b2)
    wo3, wg3 = ops.split_weights(wo), ops.split_weights(wg)
    pad = (L, Lp) if Lp != L else None
    outs = []
    for tune in (0, 128):
        out = torch.full_like(z, float('nan'))
        ops.gemm(tt.transpose(1, 2), wo, out, bias=bio, ln=(None, cso), B3=wo3, resid=z, pair=pad, c_pair=pad is not None,
                 dual=(z, wg3, csg, big), exact=2, tune=tune)
        outs.append(out)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


@pytest.mark.parametrize('M,K,NH,N2,inplace', [(128 * 5, 192, 768, 192, True), (128 * 3 + 77, 192, 768, 192, False), (1000, 64, 272, 96, False),
                                                 (40 * 40 * 3, 192, 768, 192, True)])
def test_gemm_fused_transition(ops, M, K, NH, N2, inplace):
    """AbxGemm.mlp: LayerNorm -> Linear -> ReLU -> Linear (+ residual) in ONE kernel (the hidden never leaves the CU; seqformer.py:358-376)
    against fp64 and against the two-launch path of the same split-f16 kernels; ragged row tiles, a hidden width that is not a
    multiple of the 128-channel chunk, fewer than 192 output columns, in place over the input rows."""
    ge = g(130 + M % 7)
    z = torch.randn(M, K, generator=ge) * 2 + 0.7
    z[3] = 1e3 + torch.randn(K, generator=ge)                     # |mean| >> sigma: the shifted statistics
    W1, b1 = torch.randn(NH, K, generator=ge) / K ** 0.5, 0.1 * torch.randn(NH, generator=ge)
    W2, b2 = torch.randn(N2, NH, generator=ge) / NH ** 0.5, 0.1 * torch.randn(N2, generator=ge)
    ga, be = 1 + 0.1 * torch.randn(K, generator=ge), 0.1 * torch.randn(K, generator=ge)
    w1, cs1, bi1 = fold_ln(W1, b1, ga, be)
    w2t = W2.t().contiguous().to(DEV)
    zd = z.clone().to(DEV)
    resid = zd if N2 == K else torch.randn(M, N2, generator=ge).to(DEV)
    out = zd if (inplace and N2 == K) else torch.full((M, N2), float('nan'), device=DEV)
    rcpu = resid.cpu().double()
    w23 = ops.split_weights(ops.permute_k16(w2t))
    ops.gemm(zd, w1, out, bias=bi1, ln=(None, cs1), B3=ops.split_weights(w1), act=1, resid=resid, exact=2, mlp=(w23, b2.to(DEV)))
    x = z.double()
    ln = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * ga.double() + be.double()
    ref = torch.relu(ln @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double() + rcpu
    check(out, ref, 3e-6, 'fused transition vs fp64')
    # two launches of the same arithmetic class (hidden through HBM): agreement at fp32 rounding of the hidden sums
    if NH % 128 == 0:
        hid = torch.empty(M, NH, device=DEV)
        z2 = z.clone().to(DEV)
        ops.gemm(z2, w1, hid, bias=bi1, ln=(None, cs1), B3=ops.split_weights(w1), act=1, exact=2)
        out2 = torch.empty(M, N2, device=DEV)
        ops.gemm(hid, w2t, out2, bias=b2.to(DEV), B3=ops.split_weights(w2t), resid=resid if resid is not zd else z2, exact=2)
        check(out, out2, 2e-6, 'fused transition vs two launches')


@pytest.mark.parametrize('N,M,exact', [(128, 40 * 40 * 9 + 37, 2), (96, 129 * 130, 2)])
def test_gemm_output_layernorm(ops, N, M, exact):
    """Linear -> LayerNorm in the GEMM epilogue (out_ln; IpaScore's proj_init_pair_act + init_pair_layer_norm,
    score_network.py:117-120): split-f16 128x128 tiles with a full / partial (N = 96) row and a ragged last row tile; against fp64."""
    ge = g(131)
    K = 192
    x = torch.randn(M, K, generator=ge) * 1.5 + 0.3
    W, b = torch.randn(K, N, generator=ge) / 12, torch.randn(N, generator=ge) * 0.2
    ga, be = 1 + 0.2 * torch.randn(N, generator=ge), 0.1 * torch.randn(N, generator=ge)
    Wd = W.to(DEV)
    out = torch.full((M, N), float('nan'), device=DEV)
    ops.gemm(x.to(DEV), Wd, out, bias=b.to(DEV), B3=ops.split_weights(Wd) if exact == 2 else None, exact=exact,
             out_ln=(ga.to(DEV), be.to(DEV)))
    y = x.double() @ W.double() + b.double()
    ref = (y - y.mean(-1, keepdim=True)) / torch.sqrt(y.var(-1, unbiased=False, keepdim=True) + 1e-5) * ga.double() + be.double()
    check(out, ref, 5e-6, f'gemm out_ln N={N}')


@pytest.mark.parametrize('N,K,transposed,ln', [(4, 192, True, True), (32, 192, True, True), (12, 128, False, False)])
def test_gemm_narrow_split_tile(ops, N, K, transposed, ln):
    """The skinny pair-stack projections (triangle-attention bias 192 -> 4, sequence-attention pair bias 192 -> 32, IPA pair
    bias 128 -> 12) on the 128 x 32 tile of the split-f16 GEMM: LayerNorm folded (inline statistics) or plain with alpha,
    transposed (b, N, rows) or plain store, ragged last row tile; against fp64."""
    ge = g(140 + N)
    Bc, rows = 3, 37 * 37
    x = torch.randn(Bc, rows, K, generator=ge) * 1.7 + 0.4
    W, b = torch.randn(N, K, generator=ge) / 12, torch.randn(N, generator=ge) * 0.1
    ga, be = 1 + 0.1 * torch.randn(K, generator=ge), 0.1 * torch.randn(K, generator=ge)
    if ln:
        wt, cs, bias = fold_ln(W, b, ga, be)
        xn = (x.double() - x.double().mean(-1, keepdim=True)) / torch.sqrt(x.double().var(-1, unbiased=False, keepdim=True) + 1e-5) * ga.double() + be.double()
        ref = xn @ W.double().t() + b.double()
        kw = dict(ln=(None, cs), bias=bias)
        alpha = 1.0
    else:
        wt, bias, alpha = W.t().contiguous().to(DEV), b.to(DEV), 0.577
        ref = (x.double() @ W.double().t() + b.double()) * alpha
        kw = dict(bias=bias, alpha=alpha)
    w3 = ops.split_weights(wt)
    assert ops.gemm_kernel_name(rows, N, K, Bc, transposed=transposed, split=True, exact=2).startswith('gemm3_kernel<128, 32, 32, 32, 0')
    if transposed:
        out = torch.full((Bc, N, rows), float('nan'), device=DEV)
        ops.gemm(x.to(DEV), wt, out.transpose(1, 2), B3=w3, exact=2, **kw)
        got = out.transpose(1, 2)
    else:
        out = torch.full((Bc, rows, N), float('nan'), device=DEV)
        ops.gemm(x.to(DEV), wt, out, B3=w3, exact=2, **kw)
        got = out
    check(got, ref, 5e-6, f'narrow split gemm N={N}')


# ------------------------------------------------------------------------------------------------------------------
# Round 5: the A-stationary, N-walking GEMM (csrc/gemm_as.hip) and the gated tail of the triangle attention (AbxGemm.mlp = 2)
AS_OFF = 2048        # AbxGemm.tune bit 11: keep the tile kernels of gemm3.hip


@pytest.mark.parametrize('L,Bc', [(72, 13), (65, 16), (118, 5), (128, 4)])
def test_gemm_as_plain_and_side_equal_the_tile_kernels(ops, L, Bc):
    """The K = 192 LayerNorm projections on the A-stationary kernel (a block owns 64 rows, splits them once, walks all column tiles;
    seqformer.py:520-531): q | k | v (N = 576) with the pair bias (N = 4, (b, h, i, j) store) in the free half of its ragged last column tile,
    and a 768-wide projection without a side - BIT-IDENTICAL to the tile kernels (same pieces, same product order, same statistics), and
    against fp64.  L = 65 / 118: ragged last 64-row block (and a side whose 4-row store groups straddle samples for odd L * L)."""
    ge = g(1900 + L)
    LL, K = L * L, 192
    z = (torch.randn(Bc, LL, K, generator=ge) * 1.3 + 0.2)
    z[0, 5] = 1e3 + torch.randn(K, generator=ge)                  # |mean| >> sigma: the shifted statistics
    z = z.to(DEV)
    Wb = (torch.randn(K, 4, generator=ge) / K ** 0.5).to(DEV)
    bb, csb, W3b = torch.randn(4, generator=ge).to(DEV), Wb.sum(0).contiguous(), ops.split_weights(Wb)
    x = z.double().cpu()
    ln = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    for N, side in ((576, True), (768, False), (576, False)):
        Wq = (torch.randn(K, N, generator=ge) / K ** 0.5).to(DEV)
        bq, csq, W3q = torch.randn(N, generator=ge).to(DEV), Wq.sum(0).contiguous(), ops.split_weights(Wq)
        outs = []
        for tune in (0, AS_OFF):
            q = torch.full((Bc * LL, N), float('nan'), device=DEV)
            bT = torch.full((Bc, 4, LL), float('nan'), device=DEV)
            kq = dict(bias=bq, ln=(None, csq), B3=W3q, exact=2, tune=tune)
            if side:
                ops.gemm_side(ops.gemm(z.view(Bc * LL, K), Wq, q, defer=True, **kq),
                              ops.gemm(z, Wb, bT.transpose(1, 2), defer=True, bias=bb, ln=(None, csb), B3=W3b, exact=2, tune=tune))
            else:
                ops.gemm(z.view(Bc * LL, K), Wq, q, **kq)
            outs.append((q, bT))
        assert torch.equal(outs[0][0], outs[1][0]), (N, side, float((outs[0][0] - outs[1][0]).abs().max()))
        check(outs[0][0], (ln @ Wq.double().cpu() + bq.double().cpu()).reshape(Bc * LL, N), 5e-6, f'projection N={N} L={L}')
        if side:
            assert torch.equal(outs[0][1], outs[1][1])
            check(outs[0][1], (ln @ Wb.double().cpu() + bb.double().cpu()).transpose(1, 2), 5e-6, f'side bias L={L}')


def test_gemm_as_dispatch_threshold_is_bit_invariant(ops):
    """abx_gemm picks the A-stationary kernel by LAUNCH SIZE (>= 1024 blocks of 64 rows) even when the caller fixed the
    arithmetic class (exact = 2), so batch / chunk invariance of the network rests on the two kernel families being bit-identical.  The
    same rows through a launch of 1 024 blocks (A-stationary kernel) and of 1 023 blocks (tile kernels; the last block of rows dropped)
    must agree in every bit - with a side projection and without one, at a non-edge shape (M % 128 == 0).  (relu-on-load cannot reach
    the A-stationary kernel: it needs a folded LayerNorm, and abx_gemm rejects LayerNorm + a_relu - asserted below.)
    Any change to gemm_as.hip has to keep this test green, or gate the kernel on L instead of M."""
    ge = g(1930)
    K, N = 192, 576
    M1, M0 = 1024 * 64, 1023 * 64
    z = (torch.randn(M1, K, generator=ge) * 1.2 + 0.1).to(DEV)
    Wq = (torch.randn(K, N, generator=ge) / K ** 0.5).to(DEV)
    bq, csq, W3q = torch.randn(N, generator=ge).to(DEV), Wq.sum(0).contiguous(), ops.split_weights(Wq)
    Wb = (torch.randn(K, 4, generator=ge) / K ** 0.5).to(DEV)
    bb, csb, W3b = torch.randn(4, generator=ge).to(DEV), Wb.sum(0).contiguous(), ops.split_weights(Wb)
    for side in (False, True):
        outs = []
        for M in (M1, M0):
            q = torch.full((M, N), float('nan'), device=DEV)
            bT = torch.full((1, 4, M), float('nan'), device=DEV)
            kq = dict(bias=bq, ln=(None, csq), B3=W3q, exact=2)
            if side:
                ops.gemm_side(ops.gemm(z[:M], Wq, q, defer=True, **kq),
                              ops.gemm(z[:M].view(1, M, K), Wb, bT.transpose(1, 2), defer=True, bias=bb, ln=(None, csb), B3=W3b, exact=2))
            else:
                ops.gemm(z[:M], Wq, q, **kq)
            outs.append((q, bT))
        assert torch.isfinite(outs[0][0]).all()
        assert torch.equal(outs[0][0][:M0], outs[1][0]), (side, float((outs[0][0][:M0] - outs[1][0]).abs().max()))
        if side:
            assert torch.equal(outs[0][1][:, :, :M0], outs[1][1])
    from abx_amd._lib import AbxHipError
    with pytest.raises(AbxHipError, match='exclusive'):
        ops.gemm(z, Wq, torch.empty(M1, N, device=DEV), bias=bq, ln=(None, csq), B3=W3q, exact=2, a_relu=True)


@pytest.mark.parametrize('L,Bc', [(128, 4), (118, 5), (65, 16)])
def test_kv_operand_images_from_the_projection_equal_the_producer_wave(ops, L, Bc):
    """Round 6: the q | k | v projection writes its k | v columns as the operand images of the triangle attention
    (AbxGemm.c_planes_from = 192, groups of 48 = one head: per key and head [p0: 48 f16 | p1: 48 f16] of 16 x value at the byte address of the
    fp32 head slice) and the attention's producer wave stages them by DMA (AbxTriAttn.kv_planes) instead of loading, splitting and writing
    them.  (a) the q columns and the side bias are the fp32 launch's, bit for bit; the planes are the f16 split of the fp32 launch's k | v;
    (b) the attention on the images equals the attention on fp32 k | v bit for bit, both orientations, masked keys, ragged last chunk;
    (c) a launch too small for the kernel that writes planes is refused loudly, never served as fp32."""
    ge = g(2100 + L)
    LL, K, N = L * L, 192, 576
    z = (torch.randn(Bc, LL, K, generator=ge) * 1.3 + 0.2).to(DEV)
    Wq = (torch.randn(K, N, generator=ge) / K ** 0.5).to(DEV)
    bq, csq, W3q = torch.randn(N, generator=ge).to(DEV), Wq.sum(0).contiguous(), ops.split_weights(Wq)
    Wb = (torch.randn(K, 4, generator=ge) / K ** 0.5).to(DEV)
    bb, csb, W3b = torch.randn(4, generator=ge).to(DEV), Wb.sum(0).contiguous(), ops.split_weights(Wb)
    assert ops.kv_planes_ok(Bc * LL)
    outs = {}
    for planes in (False, True):
        q = torch.full((Bc * LL, N), float('nan'), device=DEV)
        bT = torch.full((Bc, 4, LL), float('nan'), device=DEV)
        ops.gemm_side(ops.gemm(z.view(Bc * LL, K), Wq, q, defer=True, bias=bq, ln=(None, csq), B3=W3q, exact=2, c_plane_cols=(192, 48) if planes else None),
                      ops.gemm(z, Wb, bT.transpose(1, 2), defer=True, bias=bb, ln=(None, csb), B3=W3b, exact=2, alpha=ops.TRI_BIAS_LOG2))
        outs[planes] = (q, bT)
    qf, qp = outs[False][0], outs[True][0]
    assert torch.equal(qf[:, :192], qp[:, :192]) and torch.equal(outs[False][1], outs[True][1])
    kv = qf[:, 192:].reshape(-1, 8, 48) * 16.0                      # (row, head slice of k | v, channel)
    p0 = kv.half()
    p1 = (kv - p0.float()).half()
    img = qp[:, 192:].contiguous().view(torch.int16).reshape(-1, 8, 2, 48)
    assert torch.equal(img[:, :, 0], p0.view(torch.int16)) and torch.equal(img[:, :, 1], p1.view(torch.int16))
    mask = torch.ones(Bc, L, device=DEV)
    mask[1, L - 9:] = 0
    for per_row in (True, False):
        bias = outs[False][1].view(Bc, 4, L, L)
        if not per_row or L % 4:
            Lp = (L + 3) // 4 * 4
            b2 = torch.zeros(Bc, 4, L, Lp, device=DEV)
            ops.transpose_last2(bias.reshape(Bc * 4, L, L), b2.view(Bc * 4, L, Lp), transpose=not per_row)
            bias = b2
        o = []
        for planes in (False, True):
            out = torch.full((Bc * LL, 192), float('nan'), device=DEV)
            ops.tri_attn(outs[planes][0], bias, mask, out, Bc, L, per_row, bias_is_qk=True, bias_log2=True, kv_planes=planes)
            o.append(out)
        assert torch.isfinite(o[0]).all()
        assert torch.equal(o[0], o[1]), (per_row, float((o[0] - o[1]).abs().max()))
    from abx_amd._lib import AbxHipError
    small = z[:1, :64 * 100].reshape(-1, K).contiguous()
    with pytest.raises(AbxHipError, match='A-stationary'):
        ops.gemm_side(ops.gemm(small, Wq, torch.empty(small.shape[0], N, device=DEV), defer=True, bias=bq, ln=(None, csq), B3=W3q, exact=2, c_plane_cols=(192, 48)),
                      ops.gemm(small.view(1, -1, K), Wb, torch.empty(1, 4, small.shape[0], device=DEV).transpose(1, 2), defer=True, bias=bb, ln=(None, csb), B3=W3b, exact=2))


@pytest.mark.parametrize('L,Bc', [(128, 5), (118, 6), (72, 14)])
def test_gemm_as_glu_equals_the_tile_kernel(ops, L, Bc):
    """The gated projections of the triangle multiplication (glu + plane output in (8 i x 16 k) row blocks + pair mask, both variants;
    seqformer.py:480-485) on the A-stationary kernel: plane images identical to the tile kernel's.  L = 118 / 72: padded pair rows
    (L % 8 != 0 or ceil4(L) % 16 != 0: predicated stores, the conservative waits)."""
    ge = g(1950 + L)
    K, C_ = 192, 256
    LL = L * L
    Lp = (L + 3) // 4 * 4
    KT = (Lp + 15) // 16
    z = (torch.randn(Bc, LL, K, generator=ge) * 1.5 + 0.3).to(DEV)
    W = (torch.randn(K, 2 * C_, generator=ge) / K ** 0.5).to(DEV)
    b, cs, W3 = torch.randn(2 * C_, generator=ge).to(DEV), W.sum(0).contiguous(), ops.split_weights(W)
    pm = (torch.rand(Bc * L * Lp, generator=ge) > 0.15).float().to(DEV)
    for outgoing in (True, False):
        imgs = []
        for tune in (0, AS_OFF):
            lrp = torch.zeros(Bc, C_, KT, 2, L, 16, device=DEV, dtype=torch.int16)
            ops.gemm(z, W, lrp, bias=b, ln=(None, cs), B3=W3, exact=2, tune=tune, rowscale=pm, glu=True, c_split_nA=128, c_split_tile=True,
                     a_pair_transpose=0 if outgoing else L, pair=(L, Lp), a_pair=True)
            imgs.append(lrp)
        assert torch.equal(imgs[0], imgs[1]), (L, outgoing, int((imgs[0] != imgs[1]).sum()))
        assert int((imgs[0] != 0).sum()) > 0.5 * imgs[0].numel()


@pytest.mark.parametrize('M,inplace', [(128 * 300, True), (128 * 257 + 37, False), (4224 + 5, True)])
def test_gemm_gated_tail(ops, M, inplace):
    """AbxGemm.mlp = 2, the tail of the triangle attention in ONE kernel (seqformer.py:300-312): out = (sigmoid(LN(z) Wg + bg) * o) Wo + bo + z
    - the gate never exists in memory - against fp64 and against the two-launch split-f16 path (gate projection * o, then the output
    projection); ragged last row tile, in place over z."""
    ge = g(1970 + M % 7)
    K = 192
    z = torch.randn(M, K, generator=ge) * 2 + 0.7
    z[3] = 1e3 + torch.randn(K, generator=ge)
    o = torch.randn(M, K, generator=ge) * 1.5
    Wg, bg = torch.randn(K, K, generator=ge) / K ** 0.5, 0.3 * torch.randn(K, generator=ge)
    Wo, bo = torch.randn(K, K, generator=ge) / K ** 0.5, 0.1 * torch.randn(K, generator=ge)
    ga, be = 1 + 0.1 * torch.randn(K, generator=ge), 0.1 * torch.randn(K, generator=ge)
    w1, cs1, bi1 = fold_ln(Wg, bg, ga, be)
    wot = Wo.t().contiguous().to(DEV)
    zd, od = z.clone().to(DEV), o.to(DEV)
    out = zd if inplace else torch.full((M, K), float('nan'), device=DEV)
    ops.gemm(zd, w1, out, bias=bi1, ln=(None, cs1), B3=ops.split_weights(w1), act=2, gate=od, resid=zd, exact=2,
             mlp=(ops.split_weights(ops.permute_k16(wot)), bo.to(DEV)))
    x = z.double()
    ln = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * ga.double() + be.double()
    ref = (torch.sigmoid(ln @ Wg.double().t() + bg.double()) * o.double()) @ Wo.double().t() + bo.double() + z.double()
    check(out, ref, 3e-6, 'gated tail vs fp64')
    z2 = z.clone().to(DEV)
    hid = torch.empty(M, K, device=DEV)
    ops.gemm(z2, w1, hid, bias=bi1, ln=(None, cs1), B3=ops.split_weights(w1), act=2, gate=od, gate_sigmoid=False, exact=2)
    out2 = torch.empty(M, K, device=DEV)
    ops.gemm(hid, wot, out2, bias=bo.to(DEV), B3=ops.split_weights(wot), resid=z2, exact=2)
    check(out, out2, 2e-6, 'gated tail vs two launches')


@pytest.mark.parametrize('M', [128 * 300, 128 * 257 + 37, 4224 + 5])
def test_gemm_gated_tail_walk_variants_bit_identical(ops, M):
    """Round 6: the gated tail walks the z rows ONCE (all 192 gate channels in one main loop, GEMM 2 in two passes over the output columns
    that replay the kept operand pieces) instead of once per gate chunk.  Every accumulator receives the same products in the same order
    as in the two-walk kernel of round 5 (AbxGemm.tune bit 6): the outputs must be equal bit for bit, ragged last row tile included."""
    ge = g(1985 + M % 5)
    K = 192
    z = (torch.randn(M, K, generator=ge) * 2 + 0.7).to(DEV)
    o = (torch.randn(M, K, generator=ge) * 1.5).to(DEV)
    Wg, bg = torch.randn(K, K, generator=ge) / K ** 0.5, 0.3 * torch.randn(K, generator=ge)
    Wo, bo = torch.randn(K, K, generator=ge) / K ** 0.5, 0.1 * torch.randn(K, generator=ge)
    ga, be = 1 + 0.1 * torch.randn(K, generator=ge), 0.1 * torch.randn(K, generator=ge)
    w1, cs1, bi1 = fold_ln(Wg, bg, ga, be)
    w13, wo3p = ops.split_weights(w1), ops.split_weights(ops.permute_k16(Wo.t().contiguous().to(DEV)))
    outs = []
    for tune in (0, 64):
        out = torch.full((M, K), float('nan'), device=DEV)
        ops.gemm(z, w1, out, bias=bi1, ln=(None, cs1), B3=w13, act=2, gate=o, resid=z, exact=2, mlp=(wo3p, bo.to(DEV)), tune=tune)
        outs.append(out)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def test_gated_tail_flags_an_operand_beyond_the_split_range(ops):
    """A gated attention output beyond the activation range of the tail's second GEMM (|gate * o| >= 4094) becomes NaN in exactly its
    row and sets the range word (the caller then repeats on the exact kernels); every other row is what the clean input gives."""
    ge = g(1990)
    M, K = 128 * 260, 192
    z, o = torch.randn(M, K, generator=ge).to(DEV), torch.randn(M, K, generator=ge).to(DEV)
    Wg, Wo = (torch.randn(K, K, generator=ge) / K ** 0.5).to(DEV), (torch.randn(K, K, generator=ge) / K ** 0.5).to(DEV)
    bg, bo, cs = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV), Wg.sum(0).contiguous()
    w3, wo3 = ops.split_weights(Wg), ops.split_weights(ops.permute_k16(Wo))
    word = ops.range_word(DEV)
    outs = []
    for bad in (False, True):
        od = o.clone()
        if bad:
            od[777, 5] = 3.0e4
        out = torch.empty(M, K, device=DEV)
        word.zero_()
        ops.gemm(z, Wg, out, bias=bg, ln=(None, cs), B3=w3, act=2, gate=od, resid=z, exact=2, mlp=(wo3, bo))
        outs.append(out)
        assert (int(word.item()) != 0) == bad
    assert bool(torch.isnan(outs[1][777]).all())
    keep = torch.ones(M, dtype=torch.bool, device=DEV)
    keep[777] = False
    assert torch.equal(outs[0][keep], outs[1][keep])
    word.zero_()
