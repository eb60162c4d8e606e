"""A/B: Linear(192 -> 128) + LayerNorm as two kernels vs the fused out_ln GEMM (B = 100, L = 352 pair rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abx_amd import ops
DEV = 'cuda'
M = 100 * 352 * 352
x = torch.randn(M, 192, device=DEV)
W = torch.randn(192, 128, device=DEV) / 12
W3 = ops.split_weights(W)
b = torch.randn(128, device=DEV)
ga, be = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
out = torch.empty(M, 128, device=DEV)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('gemm            %.3f ms' % t(lambda: ops.gemm(x, W, out, bias=b, B3=W3, exact=2)))
print('layernorm128    %.3f ms' % t(lambda: ops.layernorm(out, ga, be, out=out)))
print('gemm + out_ln   %.3f ms' % t(lambda: ops.gemm(x, W, out, bias=b, B3=W3, exact=2, out_ln=(ga, be))))
