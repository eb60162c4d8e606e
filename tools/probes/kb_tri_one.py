"""tri_attn8 alone at the bench geometry (library under test: ABX_HIP_LIB / tools/ab_lib.py).  python tools/probes/kb_tri_one.py [Bc] [L]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops, _lib
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 352
LL, M2 = L * L, Bc * L * L
x, bT, mask, o = torch.randn(M2, 576, device=DEV), torch.randn(Bc, 4, LL, device=DEV), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
for per_row in (True, False):
    for rep in range(2):
        ms = timeit(lambda: ops.tri_attn(x, bT, mask, o, Bc, L, per_row, bias_is_qk=True, bias_log2=True), reps=7)
        print(f'{_lib.LIB_PATH.split("/")[-1]:22s} per_row={per_row}: {ms:7.3f} ms  {4.0 * Bc * L * 4 * LL * 48 / ms / 1e9:6.1f} TFLOP/s', flush=True)
