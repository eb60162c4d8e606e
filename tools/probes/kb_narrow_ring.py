"""The skinny projections (seq-attention pair bias 192 -> 32, transposed store; IPA pair bias 128 -> 12) with a deeper operand ring (AbxGemm.tune bit 9 / 10:
3 stages at 4 blocks per CU / 4 stages at 3) against the default 2 stages at 6 blocks per CU.   python tools/probes/kb_narrow_ring.py [Bc]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 352
LL, M2 = L * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
z = r(Bc, LL, 192)
Ws, bs = r(192, 32) / 14, r(32)
Ws3, css = ops.split_weights(Ws), Ws.sum(0).contiguous()
zi = r(M2, 128)
Wb, bb = r(128, 12) / 11, r(12)
Wb3 = ops.split_weights(Wb)
outs = {}
for tune in (0, 512, 1024, 0, 512, 1024):
    bT = torch.empty(Bc, 32, LL, device=DEV)
    b12 = torch.empty(M2, 12, device=DEV)
    f1 = lambda: ops.gemm(z, Ws, bT.transpose(1, 2), bias=bs, ln=(None, css), B3=Ws3, exact=2, tune=tune)
    f2 = lambda: ops.gemm(zi, Wb, b12, bias=bb, B3=Wb3, exact=2, tune=tune)
    a, b = timeit(f1, reps=7), timeit(f2, reps=7)
    if tune not in outs: outs[tune] = (bT, b12)
    eq = torch.equal(outs[0][0], bT) and torch.equal(outs[0][1], b12)
    print(f'tune {tune:5d}: seq bias 192 -> 32 {a:6.3f} ms ({4.0 * M2 * 192 / a / 1e6:5.0f} GB/s) | IPA bias 128 -> 12 {b:6.3f} ms ({4.0 * M2 * 128 / b / 1e6:5.0f} GB/s) | equal to default: {eq}', flush=True)
