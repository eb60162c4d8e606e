"""Is the VALU split of the A operand what holds the weight GEMMs back?  The q|k|v|gate projection (N = 768, K = 192, LayerNorm folded)
with A as fp32 rows (split + row statistics in the main loop, the product path) against the SAME GEMM with A handed over as pre-split
k-tiled bf16 planes + precomputed row statistics (AMODE 2: pure DMA, no VALU in front of the MFMAs)."""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 352
M2 = Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
z, W = r(M2, 192), r(192, 768) / 14
C, bias, csum, W3 = torch.empty(M2, 768, device=DEV), r(768), W.sum(0).contiguous(), ops.split_weights(W)
stats = ops.row_stats(z)
# k-tiled planes of z: (1, K/16, 3, M, 16)
p0 = z.bfloat16(); r1 = z - p0.float(); p1 = r1.bfloat16(); p2 = (r1 - p1.float()).bfloat16()
pl = torch.stack([p0.view(torch.int16), p1.view(torch.int16), p2.view(torch.int16)], 0)          # (3, M, K)
zp = pl.view(3, M2, 12, 16).permute(2, 0, 1, 3).contiguous().unsqueeze(0)                       # (1, 12, 3, M, 16)
del p0, p1, p2, r1, pl
fl = 2.0 * M2 * 192 * 768
C2 = torch.empty(1, M2, 768, device=DEV)
ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2)
ops.gemm(zp, W3.unsqueeze(0), C2, bias=bias, ln=(stats, csum), exact=2, tune=2)
print('max |fp32-A path - planes-A path|', float((C - C2[0]).abs().max()))
for name, fn in (('A fp32 rows, inline split + stats (product)', lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2)),
                 ('A fp32 rows, given stats', lambda: ops.gemm(z, W, C, bias=bias, ln=(stats, csum), B3=W3, exact=2)),
                 ('A pre-split planes + stats, 128x128 tiles', lambda: ops.gemm(zp, W3.unsqueeze(0), C2, bias=bias, ln=(stats, csum), exact=2, tune=2)),
                 ('A pre-split planes + stats, 128x192 tiles', lambda: ops.gemm(zp, W3.unsqueeze(0), C2, bias=bias, ln=(stats, csum), exact=2, tune=4))):
    ms = timeit(fn, reps=7)
    print(f'{name:48s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
