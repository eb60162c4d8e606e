"""The N = 192 residual-update GEMMs of the pair stack (triangle attention proj_out: K = 192; OPM out_proj: K = 128) on 128 x 192 tiles (3 blocks
per CU) against 128 x 96 tiles (tune bit 9: 4 blocks per CU, two walks of the A panel): ms per launch, outputs compared bit for bit."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M2 = Bc * 352 * 352
ops.RANGE_CHECK = False
for K in (192, 128):
    A = torch.randn(M2, K, device=DEV)
    W = torch.randn(K, 192, device=DEV) / K ** 0.5
    W3, bias = ops.split_weights(W), torch.randn(192, device=DEV)
    z = torch.randn(M2, 192, device=DEV)
    outs = []
    for tune in (0, 512, 0, 512):
        out = torch.empty(M2, 192, device=DEV)
        ms = timeit(lambda: ops.gemm(A, W, out, bias=bias, B3=W3, resid=z, exact=2, tune=tune), reps=15)
        outs.append(out)
        print(f'K={K} tune={tune:3d}: {ms:7.3f} ms', flush=True)
    print('  bit-identical:', torch.equal(outs[0], outs[1]))
    del A, z, outs
