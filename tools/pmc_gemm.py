#!/usr/bin/env python
"""Launch the dominant kernels a few times (for rocprofv3 --pmc passes)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abx_amd import ops
DEV = 'cuda:0'
Bc, L = 10, 352
M2 = Bc * L * L
which = sys.argv[1] if len(sys.argv) > 1 else 'gemm'
r = lambda *s: torch.randn(*s, device=DEV)
if which == 'gemm':
    z = r(M2, 192); stats = ops.row_stats(z)
    W, C, bias, csum = r(192, 768) / 14, torch.empty(M2, 768, device=DEV), r(768), r(768)
    for _ in range(3):
        ops.gemm(z, W, C, bias=bias, ln=(stats, csum))
elif which == 'tri':
    x, bT, mask, o = r(M2, 768), r(Bc, 4, L * L), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
    for _ in range(3):
        ops.tri_attn(x, bT, mask, o, Bc, L, True)
torch.cuda.synchronize()
