"""Bit pattern + fp64 error of one LayerNorm -> Linear split-f16 GEMM (A / B of library variants: ABX_HIP_LIB or tools/ab_lib.py)."""
import sys, hashlib, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
DEV = 'cuda:0'
g = torch.Generator().manual_seed(5)
M, K, N = 128 * 300, 192, 768
z = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2)).to(DEV)      # rows of very different scales
W = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
bias = torch.randn(N, generator=g).to(DEV)
W3, csum = ops.split_weights(W), W.sum(0).contiguous()
C = torch.empty(M, N, device=DEV)
ops.RANGE_CHECK = False
ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2)
torch.cuda.synchronize()
x = z.double()
ln = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
ref = ln @ W.double() + bias.double()
err = (C.double() - ref).abs()
print('sha', hashlib.sha256(C.cpu().numpy().tobytes()).hexdigest()[:16], 'max err %.3e mean err %.3e' % (err.max().item(), err.mean().item()))
