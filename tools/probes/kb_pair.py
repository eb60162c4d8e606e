"""The IPA pair-slab kernel alone (abx_ipa_pair: out[b,i,h,:] = sum_j attn[b,i,j,h] z[b,i,j,:]) at several batch sizes: ms per launch and
the HBM rate of its one read of z (A / B of library variants: tools/ab_lib.py)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
L = 352
for Bc in [int(a) for a in sys.argv[1:]] or [1, 13, 100]:
    z = torch.randn(Bc * L * L, 128, device=DEV)
    attn = torch.rand(Bc * L * L, 12, device=DEV)
    feat = torch.zeros(Bc * L, 2112, device=DEV)
    ms = timeit(lambda: ops.ipa_pair(attn, z, feat, Bc, L), reps=15)
    print(f'ipa_pair B={Bc:3d}: {ms * 1e3:8.1f} us  {Bc * L * L * (512 + 48) / ms / 1e6:7.1f} GB/s  checksum {float(feat.double().sum()):.6e}', flush=True)
    del z, attn, feat
