"""gemm_as q | k | v + bias and glu alone at the bench geometry (library under test: tools/ab_lib.py).  python tools/probes/kb_as_one.py [Bc]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops, _lib
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 352
LL, M2 = L * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
W, Wp = r(192, 576) / 14, r(192, 4) / 14
C, bT = torch.empty(M2, 576, device=DEV), torch.empty(Bc, 4, LL, device=DEV)
bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
def proj():
    ops.gemm_side(ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True),
                  ops.gemm(z.view(Bc, LL, 192), Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True))
Wv, Wg = r(192, 256) / 14, r(192, 256) / 14
Wglu, bglu = ops.pack_glu_weights(Wv, Wg, r(256), r(256))
Wglu3, csglu = ops.split_weights(Wglu), Wglu.sum(0).contiguous()
lrp = torch.zeros(Bc, 256, (L + 15) // 16, 2, L, 16, dtype=torch.int16, device=DEV)
pm = torch.ones(Bc * LL, device=DEV)
def glu():
    ops.gemm(z.view(Bc, LL, 192), Wglu, lrp, bias=bglu, ln=(None, csglu), B3=Wglu3, rowscale=pm, glu=True, exact=2, c_split_nA=128, c_split_tile=True, pair=(L, L), a_pair=True)
tag = _lib.LIB_PATH.split('/')[-1]
for rep in range(2):
    print(f'{tag:20s} q | k | v + bias {timeit(proj, reps=7):7.3f} ms | glu {timeit(glu, reps=7):7.3f} ms', flush=True)
