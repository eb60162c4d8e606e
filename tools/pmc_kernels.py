#!/usr/bin/env python
"""Launch ONE of the dominant kernels a few times at the bench geometry, for rocprofv3 --pmc passes (tools/pmc_run.sh).
    python tools/pmc_kernels.py <which> [Bc] [L]
which: qkvg (LN -> 768, gemm3 128x128 tiles: tune 2048)  qkv (round 5: LN -> q | k | v 576 + pair-bias side, A-stationary kernel)
       gtail (gated tail of the triangle attention, AbxGemm.mlp = 2)  mlp (fused transition)  trans2 (768 -> 192 + resid, gemm3 128x192)  glu (LN -> glu planes, transposed store)
       contract (plane x plane)  projout (channel-major A, gate, resid)  tri (triangle attention)  ipa (IPA attention)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abx_amd import ops  # noqa: E402

DEV = 'cuda:0'
which = sys.argv[1] if len(sys.argv) > 1 else 'qkvg'
Bc = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
REPS = 3
if which == 'qkvg':
    z, W = r(M2, 192), r(192, 768) / 14
    C, bias, csum, W3 = torch.empty(M2, 768, device=DEV), r(768), r(768), ops.split_weights(W)
    for _ in range(REPS):
        ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, tune=2048)
elif which == 'qkv':
    z, W, Wp = r(M2, 192), r(192, 576) / 14, r(192, 4) / 14
    C, bT = torch.empty(M2, 576, device=DEV), torch.empty(Bc, 4, LL, device=DEV)
    bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
    for _ in range(REPS):
        ops.gemm_side(ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True),
                      ops.gemm(z.view(Bc, LL, 192), Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True))
elif which == 'gtail':
    z, o, Wg, Wo = r(M2, 192), r(M2, 192), r(192, 192) / 14, r(192, 192) / 14
    w3, wo3p = ops.split_weights(Wg), ops.split_weights(ops.permute_k16(Wo))
    bg, cs, bo = r(192), Wg.sum(0).contiguous(), r(192)
    for _ in range(REPS):
        ops.gemm(z, Wg, z, bias=bg, ln=(None, cs), B3=w3, act=2, gate=o, resid=z, exact=2, mlp=(wo3p, bo))
elif which == 'mlp':
    z, W1, W2 = r(M2, 192), r(192, 768) / 14, r(768, 192) / 28
    W13, W23p = ops.split_weights(W1), ops.split_weights(ops.permute_k16(W2))
    b1, b2, cs = r(768), r(192), W1.sum(0).contiguous()
    for _ in range(REPS):
        ops.gemm(z, W1, z, bias=b1, ln=(None, cs), B3=W13, act=1, resid=z, exact=2, mlp=(W23p, b2))
elif which == 'trans2':
    h, W, z = r(M2, 768), r(768, 192) / 28, r(M2, 192)
    W3 = ops.split_weights(W)
    for _ in range(REPS):
        ops.gemm(h, W, z, bias=r(192), resid=z, B3=W3, exact=2)
elif which == 'glu':
    z3, Wv, Wg = r(Bc, LL, 192), r(192, 256) / 14, r(192, 256) / 14
    W, b = ops.pack_glu_weights(Wv, Wg, r(256), r(256))
    W3 = ops.split_weights(W)
    lrp = torch.zeros(Bc, 256, (L + 15) // 16, 2, L, 16, dtype=torch.int16, device=DEV)
    pm = torch.ones(Bc * LL, device=DEV)
    for _ in range(REPS):
        ops.gemm(z3, W, lrp, bias=b, ln=(None, W.sum(0).contiguous()), B3=W3, rowscale=pm, glu=True, exact=2, c_split_nA=128,
                 c_split_tile=True, pair=(L, L), a_pair=True)
elif which == 'contract':
    KT = (L + 15) // 16
    lrp = (torch.randn(Bc, 256, KT, 2, L, 16, device=DEV) * 100).to(torch.int16)
    tz = torch.empty(Bc * 128, L, L, device=DEV)
    for _ in range(REPS):
        ops.gemm(lrp[:, 0:128], lrp[:, 128:256], tz, exact=2)
elif which == 'projout':
    tt, W, z3, Gf = r(Bc, 128, LL), r(128, 192) / 11, r(Bc, LL, 192), torch.rand(Bc, LL, 192, device=DEV)
    W3 = ops.split_weights(W)
    for _ in range(REPS):
        ops.gemm(tt.transpose(1, 2), W, z3, bias=r(192), ln=(None, W.sum(0).contiguous()), B3=W3, gate=Gf, gate_sigmoid=False, resid=z3, exact=2)
elif which == 'tri':
    x, bT, mask, o = r(M2, 576), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)       # (q | k | v: no gate since round 5)
    for _ in range(REPS):
        ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True)
elif which == 'ipa':
    M1 = Bc * L
    qp, kp, vp = r(ops.ipa_qpack_numel(Bc, L)), r(M1 * 12 * 28), r(M1 * 12 * 40)
    bias2d, zi, mask = r(M2, 12), r(M2, 128), torch.ones(Bc, L, device=DEV)
    R = torch.eye(3, device=DEV).reshape(1, 9).repeat(M1, 1).contiguous()
    t, pw, feat = r(M1, 3), -torch.rand(12, device=DEV), torch.empty(M1, 2112, device=DEV)
    for _ in range(REPS):
        ops.ipa_attn(qp, kp, vp, bias2d, zi, mask, R, t, pw, feat, Bc, L)
else:
    raise SystemExit('unknown kernel ' + which)
torch.cuda.synchronize()
