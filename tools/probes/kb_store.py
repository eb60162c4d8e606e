"""What the K-independent half of the 768-wide LayerNorm -> Linear GEMM (q | k | v | gate, 128 x 128 tiles) is made of: the same launch
with its output rows / its input rows collapsed onto ONE row (row stride 0: every store / every A load still issues, but hits the same
3 KB / 768 bytes in the L2 instead of HBM), at 20 samples of L = 352 and K = 192.
    normal            : HBM reads of z + HBM writes of the 768-wide rows
    C rows -> 1 row   : no HBM writes (same store instructions)
    A rows -> 1 row   : no HBM reads (same DMA instructions)
    both              : the instruction streams alone
If "C rows -> 1 row" is as slow as "normal", the K-independent cost is instruction issue / latency, not HBM write bandwidth."""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 768
K = 192
M2 = Bc * 352 * 352
r = lambda *s: torch.randn(*s, device=DEV)
z, W, bias = r(M2, K), r(K, N) / K ** 0.5, r(N)
W3, csum = ops.split_weights(W), W.sum(0).contiguous()
C = torch.empty(M2, N, device=DEV)
C1 = torch.empty(1, N, device=DEV).expand(M2, N)
z1 = r(1, K).expand(M2, K)
ops.RANGE_CHECK = False
TUNE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cases = (('normal', z, C), ('C rows -> 1 row (no HBM writes)', z, C1), ('A rows -> 1 row (no HBM reads)', z1, C),
         ('both (instruction streams alone)', z1, C1), ('normal', z, C))
if len(sys.argv) > 3 and sys.argv[3] == 'ab':          # A / B runs of library variants: the normal launch and the HBM-free one, more repeats
    cases = (('normal', z, C), ('both (instruction streams alone)', z1, C1), ('normal', z, C), ('both (instruction streams alone)', z1, C1))
for name, a, c in cases:
    ms = timeit(lambda: ops.gemm(a, W, c, bias=bias, ln=(None, csum), B3=W3, exact=2, tune=TUNE), reps=21)
    print(f'{name:36s} Bc={Bc} N={N} K={K}: {ms:7.3f} ms', flush=True)
