"""OuterProductMean's output projection: the fused kernel (abx_opm_out_fwd, round 6) against opm_features + the K = 128 GEMM.
    python tools/probes/kb_opm_out.py [Bc] [L]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 352
M1, M2 = Bc * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
lr, Wt, bias, z, feat = r(M1, 128), r(128, 192) / 11, r(192), r(M2, 192), torch.empty(M2, 128, device=DEV)
W3 = ops.split_weights(Wt)
def old():
    ops.opm_features(lr, feat, Bc, L, 64)
    ops.gemm(feat, Wt, z, bias=bias, B3=W3, resid=z, exact=2)
for rep in range(3):
    a = timeit(old, reps=5)
    b = timeit(lambda: ops.opm_out(lr, Wt, bias, z, Bc, L), reps=5)
    print(f'Bc={Bc} L={L}: opm_features + GEMM {a:7.3f} ms | fused opm_out {b:7.3f} ms ({4.0 * M2 * 384 / b / 1e6:6.0f} GB/s of z read + written)', flush=True)
    z.normal_()
