"""The gated tail of the triangle attention (AbxGemm.mlp = 2) alone, and with its HBM streams collapsed onto one row (stride 0): what
the z rows (GEMM 1's A operand, read by both gate chunks) and the attention output o (the G stream of GEMM 2) cost.
    python tools/probes/kb_gtail.py [Bc]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit

DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 352
M2 = Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
z, o = r(M2, 192), r(M2, 192)
Wg, Wo = r(192, 192) / 14, r(192, 192) / 14
bg, cs, bo = r(192), Wg.sum(0).contiguous(), r(192)
w3, wo3, wo3p = ops.split_weights(Wg), ops.split_weights(Wo), ops.split_weights(ops.permute_k16(Wo))
out = torch.empty(M2, 192, device=DEV)
hid = torch.empty(M2, 192, device=DEV)
z1, o1 = z[:1].expand(M2, 192), o[:1].expand(M2, 192)


def fused(zz, oo, tune=0):
    ops.gemm(zz, Wg, out, bias=bg, ln=(None, cs), B3=w3, act=2, gate=oo, resid=z, exact=2, mlp=(wo3p, bo), tune=tune)


def two():
    ops.gemm(z, Wg, hid, bias=bg, ln=(None, cs), B3=w3, act=2, gate=o, gate_sigmoid=False, exact=2)
    ops.gemm(hid, Wo, out, bias=bo, B3=wo3, resid=z, exact=2)


def outproj():
    ops.gemm(hid, Wo, out, bias=bo, B3=wo3, resid=z, exact=2)


fl = 2.0 * M2 * 192 * 384
for name, fn in (('gated tail (one walk, round 6)', lambda: fused(z, o)), ('gated tail, two walks (round 5: tune 64)', lambda: fused(z, o, 64)),
                 ('gated tail (one walk, round 6)', lambda: fused(z, o)), ('gated tail, two walks (round 5: tune 64)', lambda: fused(z, o, 64)),
                 ('two walks, z rows collapsed', lambda: fused(z1, o, 64)), ('gated tail, o rows collapsed', lambda: fused(z, o1)), ('gated tail, z rows collapsed', lambda: fused(z1, o)),
                 ('gated tail, both collapsed', lambda: fused(z1, o1)), ('two launches (gate * o, out proj)', two), ('out proj + resid alone', outproj),
                 ('gated tail', lambda: fused(z, o))):
    ms = timeit(fn, reps=7)
    print(f'{name:42s} Bc={Bc} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
