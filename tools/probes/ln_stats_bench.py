"""A/B: LayerNorm-folded 192 -> 768 GEMM with the row statistics computed inline from the A stream (every n-tile recomputes them)
vs read from a precomputed (mean, rstd) buffer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abx_amd import ops
DEV = 'cuda'
M = 100 * 352 * 352
x = torch.randn(M, 192, device=DEV)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
stats = ops.row_stats(x)
print('row_stats        %.3f ms' % t(lambda: ops.row_stats(x, out=stats)))
for N in (768, 512):
    W = torch.randn(192, N, device=DEV) / 14
    W3 = ops.split_weights(W)
    b, cs = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    print('N=%d inline stats  %.3f ms' % (N, t(lambda: ops.gemm(x, W, out, bias=b, ln=(None, cs), B3=W3, exact=2))))
    print('N=%d given stats   %.3f ms' % (N, t(lambda: ops.gemm(x, W, out, bias=b, ln=(stats, cs), B3=W3, exact=2))))
    print('N=%d no LN         %.3f ms' % (N, t(lambda: ops.gemm(x, W, out, bias=b, B3=W3, exact=2))))
