"""What the L2-miss re-reads of z cost the fused kernels: the same launch with its A rows (and residual rows) collapsed onto ONE row
(row stride 0: every DMA / load still issues, but hits the same 768 bytes in the L2), and with the output rows collapsed too.
    python tools/probes/kb_rereads.py [Bc]
fused transition (gemm3_mlp_kernel: z read once per hidden chunk: 7 x on the counters), gated attention tail (gemm3_gtail_kernel: 3 x)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 352
M2 = Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
z1 = r(1, 192).expand(M2, 192)
o = r(M2, 192)
o1 = r(1, 192).expand(M2, 192)
out = torch.empty(M2, 192, device=DEV)
out1 = torch.empty(1, 192, device=DEV).expand(M2, 192)
W1, W2 = r(192, 768) / 14, r(768, 192) / 28
b1, cs1, b2 = r(768), r(768), r(192)
W13, W23p = ops.split_weights(W1), ops.split_weights(ops.permute_k16(W2))
Wg, Wo = r(192, 192) / 14, r(192, 192) / 14
bg, csg, bo = r(192), r(192), r(192)
Wg3, Wo3p = ops.split_weights(Wg), ops.split_weights(ops.permute_k16(Wo))
def mlp(a, res, c):
    ops.gemm(a, W1, c, bias=b1, ln=(None, cs1), B3=W13, act=1, resid=res, exact=2, mlp=(W23p, b2))
def tail(a, g, res, c):
    ops.gemm(a, Wg, c, bias=bg, ln=(None, csg), B3=Wg3, act=2, gate=g, resid=res, exact=2, mlp=(Wo3p, bo))
for rep in range(2):
    for name, fn in (('transition  normal', lambda: mlp(z, z, out)), ('transition  A + residual rows -> 1 row (no HBM / L2-miss reads)', lambda: mlp(z1, z1, out)),
                     ('transition  and the output rows -> 1 row', lambda: mlp(z1, z1, out1)),
                     ('tail        normal', lambda: tail(z, o, z, out)), ('tail        z rows (A + residual) -> 1 row', lambda: tail(z1, o, z1, out)),
                     ('tail        z and o rows -> 1 row', lambda: tail(z1, o1, z1, out)), ('tail        all rows -> 1 row', lambda: tail(z1, o1, z1, out1))):
        ms = timeit(fn, reps=5)
        print(f'{name:70s} Bc={Bc} {ms:8.3f} ms', flush=True)
