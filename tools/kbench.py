#!/usr/bin/env python
"""Micro-benchmark of the individual libabx_hip kernels at the bench launch geometry (chunk of Bc samples, L residues).
Times each op with HIP events on the launch stream (median of R repeats).  Run on the GPU box:
    python tools/kbench.py [--bc 10] [--L 352] [--only gemm,tri]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abx_amd import ops  # noqa: E402

DEV = 'cuda:0'


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bc', type=int, default=10)
    ap.add_argument('--L', type=int, default=352)
    ap.add_argument('--only', default='')
    ap.add_argument('--exact', action='store_true', help='exact fp32 MFMA GEMM kernels instead of the split-f16 ones')
    a = ap.parse_args()
    ops.GEMM_EXACT = a.exact
    _gemm, _splits = ops.gemm, {}

    def gemm_with_split(A, B, Cout, **kw):       # weight GEMMs get their float16 weight planes, like model/forward.py does
        if not a.exact and B.dim() == 2 and B.stride(1) == 1 and B.shape[1] > 64 and 'B3' not in kw:
            key = (B.data_ptr(), tuple(B.shape))
            if key not in _splits:
                _splits[key] = ops.split_weights(B)
            kw['B3'] = _splits[key]
        return _gemm(A, B, Cout, **kw)
    ops.gemm = gemm_with_split
    Bc, L = a.bc, a.L
    LL, M2, M1 = L * L, Bc * L * L, Bc * L
    only = set(a.only.split(',')) if a.only else None
    r = lambda *s: torch.randn(*s, device=DEV)
    out = []

    def rec(name, ms, flops=0, bytes_=0):
        line = f'{name:46s} {ms:9.3f} ms'
        if flops:
            line += f'  {flops / ms / 1e9:8.1f} TFLOP/s'
        if bytes_:
            line += f'  {bytes_ / ms / 1e6:8.1f} GB/s'
        print(line, flush=True)
        out.append(line)

    if not only or 'gemm' in only:
        z = r(M2, 192)
        stats = ops.row_stats(z)
        for N, K, tag in ((768, 192, 'qkvg / transition-1 (LN)'), (448, 192, 'tri-mul gates (LN)'), (192, 192, 'tri-attn out (+resid)'),
                          (192, 768, 'transition-2 (+resid)'), (192, 128, 'opm out (+resid)'), (128, 192, 'ipa init pair'),
                          (32, 192, 'seq-attn bias (LN, T-store)')):
            A = r(M2, K)
            W = r(K, N) / K ** 0.5
            C = torch.empty(M2, N, device=DEV)
            bias, csum = r(N), r(N)
            if K == 192:
                ms = timeit(lambda: ops.gemm(A, W, C, bias=bias, ln=(None, csum)))
            else:
                ms = timeit(lambda: ops.gemm(A, W, C, bias=bias))
            rec(f'gemm {M2}x{N}x{K} {tag}', ms, 2.0 * M2 * N * K, 4.0 * M2 * (N + K))
        # tri-mul: transposed-store projection, contraction, channel-major final
        z3 = z.view(Bc, LL, 192)
        G = r(Bc, LL, 448)
        pm = torch.ones(M2, device=DEV)
        left = torch.empty(Bc, 128, LL, device=DEV)
        W = r(192, 128) / 14
        bias, csum = r(128), r(128)
        GT = torch.rand(Bc, 256, LL, device=DEV)
        ms = timeit(lambda: ops.gemm(z3, W, left.transpose(1, 2), bias=bias, ln=(None, csum), rowscale=pm,
                                     gate=GT[:, :128].transpose(1, 2), gate_sigmoid=False))
        rec('gemm tri-mul proj (LN, gate, transposed store)', ms, 2.0 * M2 * 128 * 192, 4.0 * M2 * (192 + 128 + 128))
        lz, rz, tz = left.view(Bc * 128, L, L), torch.randn_like(left).view(Bc * 128, L, L), torch.empty(Bc * 128, L, L, device=DEV)
        ms = timeit(lambda: ops.gemm(lz, rz.transpose(1, 2), tz))
        rec('bgemm contraction outgoing (NT)', ms, 2.0 * Bc * 128 * L ** 3, 4.0 * 3 * Bc * 128 * LL)
        ms = timeit(lambda: ops.gemm(lz.transpose(1, 2), rz, tz))
        rec('bgemm contraction incoming (TN)', ms, 2.0 * Bc * 128 * L ** 3, 4.0 * 3 * Bc * 128 * LL)
        tcm = left.transpose(1, 2)
        st2 = ops.row_stats(tcm)
        W2 = r(128, 192) / 11
        b2, c2 = r(192), r(192)
        Gf = torch.rand(Bc, LL, 192, device=DEV)
        ms = timeit(lambda: ops.gemm(tcm, W2, z3, bias=b2, ln=(None, c2), gate=Gf, gate_sigmoid=False, resid=z3))
        rec('gemm tri-mul out (channel-major A, LN, gate)', ms, 2.0 * M2 * 192 * 128, 4.0 * M2 * (128 + 192 * 3))
        ms = timeit(lambda: ops.row_stats(z, stats))
        rec('row_stats (M2 x 192)', ms, 0, 4.0 * M2 * 192)
        ms = timeit(lambda: ops.row_stats(tcm, st2))
        rec('row_stats channel-major (M2 x 128)', ms, 0, 4.0 * M2 * 128)
        # seq-track shapes
        for M, N, K in ((M1, 1632, 544), (M1, 2176, 544), (M1, 544, 2176), (M1, 1152, 256), (M1, 256, 2112), (M1, 256, 256)):
            A, W, C = r(M, K), r(K, N) / K ** 0.5, torch.empty(M, N, device=DEV)
            ms = timeit(lambda: ops.gemm(A, W, C))
            rec(f'gemm seq-track {M}x{N}x{K}', ms, 2.0 * M * N * K)
    if only and 'tune' in only:
        z = r(M2, 192)
        stats = ops.row_stats(z)
        for N, K in ((768, 192), (192, 768), (192, 192)):
            A, W, C = r(M2, K), r(K, N) / K ** 0.5, torch.empty(M2, N, device=DEV)
            bias, csum = r(N), r(N)
            for tune in range(8):
                kw = dict(bias=bias, tune=tune)
                if K == 192:
                    kw['ln'] = (stats, csum)
                ms = timeit(lambda: ops.gemm(A, W, C, **kw))
                rec(f'gemm {N}x{K} tune={tune} (noremap={tune & 1}, variant={tune >> 1})', ms, 2.0 * M2 * N * K)
    if only and 'ksweep' in only:
        for N in (192, 768, 128):
            for K in (16, 64, 192, 384, 768, 1536):
                A, W, C = r(M2, K), r(K, N) / K ** 0.5, torch.empty(M2, N, device=DEV)
                ms = timeit(lambda: ops.gemm(A, W, C))
                rec(f'ksweep N={N} K={K}', ms, 2.0 * M2 * N * K, 4.0 * M2 * (N + K))
    if only and 'shapes' in only:
        for N, K in ((768, 192), (192, 768), (192, 192), (192, 128), (448, 192)):
            A, W, C = r(M2, K), r(K, N) / K ** 0.5, torch.empty(M2, N, device=DEV)
            bias, csum = r(N), r(N)
            for tune, tag in ((0, 'default'), (2, '128x128'), (1, 'default, no xcd remap'), (3, '128x128 no remap')):
                ms = timeit(lambda: ops.gemm(A, W, C, bias=bias, ln=(None, csum), tune=tune))
                rec(f'shape N={N} K={K} {tag}', ms, 2.0 * M2 * N * K, 4.0 * M2 * (N + K))
            ms = timeit(lambda: ops.gemm(A, W, C, bias=bias, ln=(None, csum), exact=True))
            rec(f'shape N={N} K={K} exact fp32', ms, 2.0 * M2 * N * K, 4.0 * M2 * (N + K))
    if only and 'ablate' in only:
        for N, K in ((192, 1536), (768, 768)):
            A, W, C = r(M2, K), r(K, N) / K ** 0.5, torch.empty(M2, N, device=DEV)
            for tune, tag in ((0, 'full'), (16, 'no MFMA'), (32, 'no A dma'), (64, 'no W dma'), (96, 'no dma'), (128, 'no split'),
                              (16 + 128, 'no MFMA, no split'), (96 + 128, 'MFMA + LDS reads only'), (2, '128x128 tile')):
                ms = timeit(lambda: ops.gemm(A, W, C, tune=tune))
                rec(f'ablate N={N} K={K} {tag}', ms, 2.0 * M2 * N * K)
    if only and 'tm' in only:
        z = r(M2, 192); z3 = z.view(Bc, LL, 192)
        G = r(Bc, LL, 448)
        left = r(Bc, 128, LL)
        tcm = left.transpose(1, 2)
        W2 = r(128, 192) / 11
        b2, c2 = r(192), r(192)
        out3 = torch.empty(Bc, LL, 192, device=DEV)
        fl = 2.0 * M2 * 192 * 128
        rec('cm-A plain (no LN/gate/resid), out separate', timeit(lambda: ops.gemm(tcm, W2, out3, bias=b2)), fl)
        rec('cm-A + inline LN', timeit(lambda: ops.gemm(tcm, W2, out3, bias=b2, ln=(None, c2))), fl)
        rec('cm-A + LN + gate', timeit(lambda: ops.gemm(tcm, W2, out3, bias=b2, ln=(None, c2), gate=G[:, :, 256:448])), fl)
        rec('cm-A + LN + gate + resid(in place)', timeit(lambda: ops.gemm(tcm, W2, z3, bias=b2, ln=(None, c2), gate=G[:, :, 256:448], resid=z3)), fl)
        a_kc = r(Bc, LL, 128)
        rec('kc-A plain same shape', timeit(lambda: ops.gemm(a_kc, W2, out3, bias=b2)), fl)
        rec('kc-A + LN + gate + resid', timeit(lambda: ops.gemm(a_kc, W2, z3, bias=b2, ln=(None, c2), gate=G[:, :, 256:448], resid=z3)), fl)
        W = r(192, 128) / 14
        bias, csum = r(128), r(128)
        pm = torch.ones(M2, device=DEV)
        fl = 2.0 * M2 * 128 * 192
        outn = torch.empty(Bc, LL, 128, device=DEV)
        rec('proj normal store plain', timeit(lambda: ops.gemm(z3, W, outn, bias=bias)), fl)
        rec('proj normal store + LN + gate + mask', timeit(lambda: ops.gemm(z3, W, outn, bias=bias, ln=(None, csum), rowscale=pm, gate=G[:, :, :128])), fl)
        rec('proj T-store plain', timeit(lambda: ops.gemm(z3, W, left.transpose(1, 2), bias=bias)), fl)
        rec('proj T-store + LN', timeit(lambda: ops.gemm(z3, W, left.transpose(1, 2), bias=bias, ln=(None, csum))), fl)
        GT = r(Bc, 256, LL)
        rec('proj T-store + LN + T-gate(sigmoid) + mask', timeit(lambda: ops.gemm(z3, W, left.transpose(1, 2), bias=bias, ln=(None, csum), rowscale=pm, gate=GT[:, :128].transpose(1, 2))), fl)
        rec('proj T-store + LN + T-gate(plain) + mask', timeit(lambda: ops.gemm(z3, W, left.transpose(1, 2), bias=bias, ln=(None, csum), rowscale=pm, gate=GT[:, :128].transpose(1, 2), gate_sigmoid=False)), fl)
        W3 = r(192, 256) / 14
        b3, c3 = r(256), r(256)
        rec('lr_gates N=256 T-store + LN + sigmoid', timeit(lambda: ops.gemm(z3, W3, GT.transpose(1, 2), bias=b3, ln=(None, c3), act=2)), 2.0 * M2 * 256 * 192)
        W4 = r(192, 192) / 14
        rec('final_gate N=192 + LN + sigmoid', timeit(lambda: ops.gemm(z3, W4, out3, bias=b2, ln=(None, c2), act=2)), 2.0 * M2 * 192 * 192)
        W5 = r(192, 448) / 14
        b5, c5 = r(448), r(448)
        G2 = G.view(M2, 448)
        rec('old fused gates N=448 + LN', timeit(lambda: ops.gemm(z, W5, G2, bias=b5, ln=(None, c5))), 2.0 * M2 * 448 * 192)
    if not only or 'tri' in only:
        x = r(M2, 768)
        bT = r(Bc, 4, LL)
        mask = torch.ones(Bc, L, device=DEV)
        o = torch.empty(M2, 192, device=DEV)
        for per_row in (True, False):
            ms = timeit(lambda: ops.tri_attn(x, bT, mask, o, Bc, L, per_row, bias_is_qk=True))
            rec(f'tri_attn per_row={per_row}', ms, 4.0 * Bc * L * 4 * LL * 48, 4.0 * M2 * (768 + 192))
        bT2 = torch.empty_like(bT)
        ms = timeit(lambda: ops.transpose_last2(bT.view(Bc * 4, L, L), bT2.view(Bc * 4, L, L)))
        rec('transpose_last2 (bias)', ms, 0, 8.0 * Bc * 4 * LL)
    if not only or 'ipa' in only:
        qp, kp, vp = r(ops.ipa_qpack_numel(Bc, L)), r(M1 * 12 * 28), r(M1 * 12 * 40)
        bias2d, zz = r(M2, 12), r(M2, 128)
        mask = torch.ones(Bc, L, device=DEV)
        R, t = r(M1, 9), r(M1, 3)
        pw = -torch.rand(12, device=DEV) * 0.1
        feat = torch.empty(M1, 2112, device=DEV)
        ms = timeit(lambda: ops.ipa_attn(qp, kp, vp, bias2d, zz, mask, R, t, pw, feat, Bc, L))
        rec('ipa_attn', ms, 2.0 * M2 * 12 * (28 + 40 + 128), 4.0 * M2 * 140)
        proj = r(M1, 1152)
        ms = timeit(lambda: ops.ipa_pack(proj, R, t, qp, kp, vp, Bc, L, 0.14))
        rec('ipa_pack', ms)
    if not only or 'misc' in only:
        ps, temb = r(1, L, L, 128), r(Bc, 32)
        prev, ga, be = r(Bc, L, L, 192), r(192), r(192)
        pp = torch.randint(0, 15, (Bc, L, L), device=DEV)
        tab = r(15, 192)
        outp = torch.empty(Bc, L, L, 192, device=DEV)
        ms = timeit(lambda: ops.assemble_pair(ps, temb, prev, ga, be, pp, tab, outp, Bc, L, 128, 32))
        rec('assemble_pair', ms, 0, 4.0 * M2 * (192 * 2 + 2))
        lr = r(M1, 128)
        f = torch.empty(M2, 128, device=DEV)
        ms = timeit(lambda: ops.opm_features(lr, f, Bc, L, 64))
        rec('opm_features', ms, 0, 4.0 * M2 * 128)
        qkv, gate = r(M1, 32 * 51), r(M1, 544)
        biasT = r(Bc, 32, LL)
        o5 = torch.empty(M1, 544, device=DEV)
        ms = timeit(lambda: ops.seq_attn(qkv, biasT, torch.ones(Bc, L, device=DEV), gate, o5, Bc, L))
        rec('seq_attn', ms, 4.0 * Bc * 32 * LL * 17, 4.0 * Bc * 32 * LL)
    od = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out')
    os.makedirs(od, exist_ok=True)
    open(os.path.join(od, 'kbench.txt'), 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
    main()
