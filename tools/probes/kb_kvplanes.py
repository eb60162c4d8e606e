"""q | k | v projection + triangle attention with k | v as fp32 columns (producer wave loads, splits, writes) against k | v as the operand images
the projection writes (AbxGemm.c_planes_from; the producer wave issues DMA only).   python tools/probes/kb_kvplanes.py [Bc] [L]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 352
LL, M2 = L * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
W, Wp = r(192, 576) / 14, r(192, 4) / 14
C = {False: torch.empty(M2, 576, device=DEV), True: torch.empty(M2, 576, device=DEV)}
bT = torch.empty(Bc, 4, LL, device=DEV)
bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
mask, o = torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
def proj(pl):
    ops.gemm_side(ops.gemm(z, W, C[pl], bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True, c_plane_cols=(192, 48) if pl else None),
                  ops.gemm(z.view(Bc, LL, 192), Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True, alpha=ops.TRI_BIAS_LOG2))
def attn(pl, per_row):
    ops.tri_attn(C[pl], bT.view(Bc, 4, L, L), mask, o, Bc, L, per_row, bias_is_qk=True, bias_log2=True, kv_planes=pl)
for pl in (False, True): proj(pl)
for rep in range(2):
    for pl in (False, True):
        tag = 'k | v as operand images' if pl else 'k | v as fp32          '
        a = timeit(lambda: proj(pl), reps=7)
        b = timeit(lambda: attn(pl, True), reps=7)
        c = timeit(lambda: attn(pl, False), reps=7)
        print(f'{tag} Bc={Bc} L={L}: projection {a:7.3f} ms | attention starting {b:7.3f} ms, ending {c:7.3f} ms | projection + attention {a + (b + c) / 2:7.3f} ms', flush=True)
