"""T(K) = a + b K of the LayerNorm -> 768-wide GEMM (128 x 128 tiles) at 20 samples of L = 352: the per-tile fixed cost a (prologue DMA,
epilogue through LDS, stores) against the k-loop slope b."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
M2 = 20 * 352 * 352
r = lambda *s: torch.randn(*s, device=DEV)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
C, bias = torch.empty(M2, N, device=DEV), r(N)
pts = []
for K in (64, 128, 192, 256, 384):
    z, W = r(M2, K), r(K, N) / K ** 0.5
    W3, csum = ops.split_weights(W), W.sum(0).contiguous()
    ms = timeit(lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2), reps=9)
    pts.append((K, ms))
    print(f'K {K:4d}: {ms:7.3f} ms', flush=True)
    del z
(k0, t0), (k1, t1) = pts[0], pts[-1]
b = (t1 - t0) / (k1 - k0)
print(f'slope {b * 1e3:.2f} us per k, intercept {t0 - b * k0:.3f} ms ({(t0 - b * k0) / pts[2][1] * 100:.0f} % of the K = 192 time)')
