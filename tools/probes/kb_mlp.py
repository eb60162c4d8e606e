import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV='cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 352
M2 = Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
W1, W2 = r(192, 768) / 14, r(768, 192) / 28
b1, cs1, b2 = r(768), r(768), r(192)
W13, W23, W23p = ops.split_weights(W1), ops.split_weights(W2), ops.split_weights(ops.permute_k16(W2))
hid = torch.empty(M2, 768, device=DEV)
out = torch.empty(M2, 192, device=DEV)
def two():
    ops.gemm(z, W1, hid, bias=b1, ln=(None, cs1), B3=W13, act=1, exact=2)
    ops.gemm(hid, W2, out, bias=b2, B3=W23, resid=z, exact=2)
def fused():
    ops.gemm(z, W1, out, bias=b1, ln=(None, cs1), B3=W13, act=1, resid=z, exact=2, mlp=(W23p, b2))
def fused3():
    ops.gemm(z, W1, out, bias=b1, ln=(None, cs1), B3=W13, act=1, resid=z, exact=2, mlp=(W23p, b2), tune=32)
def fused1():
    ops.gemm(z, W1, out, bias=b1, ln=(None, cs1), B3=W13, act=1, resid=z, exact=2, mlp=(W23p, b2), tune=16)
fl = 2.0 * M2 * 768 * 384
ref = None
fused(); ref = out.clone(); fused3(); print('3-stage ring identical:', bool(torch.equal(ref, out)))
for name, fn in (('two launches', two), ('fused mlp (2 blocks/CU)', fused), ('fused mlp, 3-stage GEMM 1 ring', fused3), ('fused mlp (2 blocks/CU)', fused), ('fused mlp, 3-stage GEMM 1 ring', fused3)):
    ms = timeit(fn, reps=7)
    print(f'{name:34s} Bc={Bc} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
if len(sys.argv) > 2 and sys.argv[2] == 'occ1':     # the one-block-per-CU instantiation (tune bit 4): how much the second resident block is worth
    for name, fn in (('fused mlp (1 block/CU)', fused1), ('fused mlp (2 blocks/CU)', fused)):
        ms = timeit(fn, reps=7)
        print(f'{name:34s} Bc={Bc} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
