"""Would the outer-product-mean pay as a plane x plane contraction (no 128-wide feature tensor)?  Times the existing pair
(opm_features + the 128 -> 192 output projection with residual) against ONE two-level-batched plane GEMM of the shape the restructured
form would launch: batch = (b, i), A = planes of [left_j | 1 | 0] (L x K', shared by all i of a sample), B = planes of the per-(b, i)
weights diag(right_i) W1 + W2 (192 x K'), C = z[b, i] (+ residual, in place).   python tools/probes/kb_opm.py [Bc] [L]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 352
M1, M2 = Bc * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
lr = r(M1, 128)
feat = torch.empty(M2, 128, device=DEV)
W, b = r(128, 192) / 11, r(192)
W3 = ops.split_weights(W)
def old():
    ops.opm_features(lr, feat, Bc, L, 64)
    ops.gemm(feat, W, z, bias=b, B3=W3, resid=z, exact=2)
for KT in (5, 6):
    planes = lambda *s: (r(*s) * 0.5).half().view(torch.int16)
    A = planes(Bc, 1, KT, 2, L, 16).expand(Bc, L, KT, 2, L, 16)
    B = planes(Bc, L, KT, 2, 192, 16)
    zc = z.view(Bc * L, L, 192)
    def new():
        ops.gemm(A, B, zc, resid=zc, exact=2)
    try:
        new(); torch.cuda.synchronize()
        print(f'plane x plane, K = {KT * 16}: {timeit(new, reps=7):7.3f} ms   (operand planes of the weights: {B.numel() * 2 / 1e9:.2f} GB)', flush=True)
    except Exception as e:
        print('plane form failed:', repr(e)[:300])
    del A, B
print(f'opm_features + output projection: {timeit(old, reps=7):7.3f} ms')
ms = timeit(lambda: ops.opm_features(lr, feat, Bc, L, 64), reps=7)
print(f'   opm_features alone: {ms:7.3f} ms')
