"""Ablations of the A-stationary GEMM (probe library built with -DABX_AS_ABLATE; tune bits 12-13: 1 no slice stores, 2 no MFMA, 3 no weight DMA in the walk).
    python tools/ab_lib.py tools/probes/bin/libabx_hip_abl.so tools/probes/kb_as_abl.py [Bc]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit

DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
z3 = z.view(Bc, LL, 192)
W = r(192, 576) / 14
b, cs, W3 = r(576), W.sum(0).contiguous(), ops.split_weights(W)
Wp = r(192, 4) / 14
bp, csp, Wp3 = r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
out = torch.empty(M2, 576, device=DEV)
bT = torch.empty(Bc, 4, LL, device=DEV)
Wg = r(192, 512) / 14
bg, csg, Wg3 = r(512), Wg.sum(0).contiguous(), ops.split_weights(Wg)
KT = L // 16
lrp = torch.zeros(Bc, 256, KT, 2, L, 16, device=DEV, dtype=torch.int16)
pm = torch.ones(Bc * L * L, device=DEV)


def side(tune):
    g1 = ops.gemm(z, W, out, bias=b, ln=(None, cs), B3=W3, exact=2, tune=tune, defer=True)
    g2 = ops.gemm(z3, Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, tune=tune, defer=True)
    ops.gemm_side(g1, g2)


def glu(tune):
    ops.gemm(z3, Wg, lrp, bias=bg, ln=(None, csg), B3=Wg3, exact=2, tune=tune, rowscale=pm, glu=True, c_split_nA=128, c_split_tile=True,
             a_pair_transpose=0, pair=(L, L), a_pair=True)


for name, fn in (('side', side), ('glu', glu)):
    for abl, tag in ((0, 'full'), (1, 'no slice stores'), (2, 'no MFMA'), (3, 'half the weight DMA in the walk'), (0, 'full')):
        ms = timeit(lambda: fn(abl << 12), reps=7)
        print(f'{name:5s} {tag:28s} Bc={Bc} {ms:8.3f} ms', flush=True)
