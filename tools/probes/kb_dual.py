"""The TriangleMultiplication tail (dual GEMM: LN(product) @ Wo * sigmoid(LN(z) @ Wg + bg) + z) at the bench geometry: three blocks per CU
(the default) against the four-block build (tune bit 10: one B sub-tile in flight, 32-column store groups, one accumulator set parked in scratch); outputs compared bit for bit."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 352
LL = L * L
ops.RANGE_CHECK = False
z = torch.randn(Bc, LL, 192, device=DEV)
tt = torch.randn(Bc, 128, LL, device=DEV)
wo, wg = torch.randn(128, 192, device=DEV) / 11, torch.randn(192, 192, device=DEV) / 14
bio, big, cso, csg = torch.randn(192, device=DEV), torch.randn(192, device=DEV), wo.sum(0).contiguous(), wg.sum(0).contiguous()
w3o, w3g = ops.split_weights(wo), ops.split_weights(wg)
outs = []
for tune in (0, 1024, 0, 1024):
    out = torch.empty_like(z)
    ms = timeit(lambda: ops.gemm(tt.transpose(1, 2), wo, out, bias=bio, ln=(None, cso), B3=w3o, resid=z, dual=(z, w3g, csg, big), exact=2, tune=tune), reps=11)
    outs.append(out)
    print(f'dual Bc={Bc} tune={tune:4d}: {ms:7.3f} ms', flush=True)
print('bit-identical:', torch.equal(outs[0], outs[1]))
