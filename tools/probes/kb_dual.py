"""The tri-mul tail (dual GEMM: proj_out(LN product) x sigmoid(final_gate(LN z)) + z) on one block per row tile (round 6, z walked once)
(tune bit 7) against the default two-tile kernel, same box, alternating.
    python tools/probes/kb_dual.py [Bc] [L]"""
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
tt, z, out = r(Bc, 128, LL), r(Bc, LL, 192), torch.empty(Bc, LL, 192, device=DEV)
Wo, Wg = r(128, 192) / 11, r(192, 192) / 14
Wo3, cso, bo, Wg3, csg, bg = ops.split_weights(Wo), Wo.sum(0).contiguous(), r(192), ops.split_weights(Wg), Wg.sum(0).contiguous(), r(192)
def dual(tune):
    ops.gemm(tt.transpose(1, 2), Wo, out, bias=bo, ln=(None, cso), B3=Wo3, resid=z, dual=(z, Wg3, csg, bg), exact=2, tune=tune)
fl = 2.0 * Bc * LL * 192 * 320
for rep in range(3):
    for name, tune in (('dual, one block per row tile (tune 128)', 128), ('dual, two column tiles (default)', 0)):
        ms = timeit(lambda: dual(tune), reps=7)
        print(f'{name:44s} Bc={Bc} L={L} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
