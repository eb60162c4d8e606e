"""abx_assemble_pair_bias against abx_assemble_pair + the 192 -> 32 pair-bias projection.   python tools/probes/kb_assemble_bias.py [Bc] [L]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 352
LL = L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
ps, temb, prev, ga, be = r(1, L, L, 128), r(Bc, 32), r(Bc, L, L, 192), r(192), r(192)
ppos, ptab = torch.randint(0, 15, (Bc, L, L), device=DEV), r(15, 192)
W, bi = r(192, 32) / 14, r(32)
W3, cs = ops.split_weights(W), W.sum(0).contiguous()
z0, bT = torch.empty(Bc, L, L, 192, device=DEV), torch.empty(Bc, 32, LL, device=DEV)
def old():
    ops.assemble_pair(ps, temb, prev, ga, be, ppos, ptab, z0, Bc, L, 128, 32)
    ops.gemm(z0.view(Bc, LL, 192), W, bT.transpose(1, 2), bias=bi, ln=(None, cs), B3=W3, exact=2)
for rep in range(3):
    a = timeit(old, reps=5)
    b = timeit(lambda: ops.assemble_pair_bias(ps, temb, prev, ga, be, ppos, ptab, z0, W3, cs, bi, bT, Bc, L), reps=5)
    print(f'Bc={Bc} L={L}: assemble_pair + projection {a:7.3f} ms | fused {b:7.3f} ms ({4.0 * Bc * LL * (192 * 2 + 34) / b / 1e6:6.0f} GB/s)', flush=True)
