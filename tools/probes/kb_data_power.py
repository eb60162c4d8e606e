"""Under the board's power limit a kernel's time depends on its DATA: matrix-core operands that do not toggle draw less power, the clock rises.
The same launches on random operands, on zeros and on one repeated row (what the 'collapsed rows' and 'no producer' probes feed the MFMAs).
    python tools/probes/kb_data_power.py [Bc]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 352
LL, M2 = L * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
bT, mask, o = r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
x = r(M2, 576)
xz = x.clone(); xz[:, 192:] = 0                    # k | v = 0
x1 = x[:1].expand(M2, 576).contiguous()            # every row the same
for name, t in (('random q | k | v', x), ('k | v = 0 (q random)', xz), ('every row identical', x1), ('all zeros', torch.zeros_like(x))):
    ms = timeit(lambda: ops.tri_attn(t, bT.view(Bc, 4, L, L), mask, o, Bc, L, True, bias_is_qk=True, bias_log2=True), reps=7)
    print(f'tri_attn8   {name:24s} {ms:7.3f} ms', flush=True)
del x, xz, x1
z = r(M2, 192)
W, Wp = r(192, 576) / 14, r(192, 4) / 14
C = torch.empty(M2, 576, device=DEV)
bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
def proj(a):
    ops.gemm_side(ops.gemm(a, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True),
                  ops.gemm(a.view(Bc, LL, 192), Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True))
zc = z[:1].expand(M2, 192).contiguous()            # identical rows, but a full-size tensor: the HBM traffic of the normal launch
for name, a in (('random rows', z), ('identical rows, full-size tensor (same HBM traffic)', zc), ('all zeros', torch.zeros_like(z))):
    ms = timeit(lambda: proj(a), reps=7)
    print(f'gemm_as qkv {name:52s} {ms:7.3f} ms', flush=True)
