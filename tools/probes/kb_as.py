"""A-stationary GEMM (csrc/gemm_as.hip) against the tile kernels of gemm3.hip (tune bit 11 = 2048 keeps the latter): bit identity and
time for the q | k | v (+ pair bias side) projection, the 768-wide projection and the glu projection of the triangle multiplication.
    python tools/probes/kb_as.py [Bc] [L] [check|time|all]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit

DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = int(sys.argv[2]) if len(sys.argv) > 2 else 352
what = sys.argv[3] if len(sys.argv) > 3 else 'all'
OLD = 2048
LL, M2 = L * L, Bc * L * L
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192) * 1.7 + 0.3
z3 = z.view(Bc, LL, 192)


_out = {}


def qkv_side(N, tune, with_side=True):
    W, b, cs = _w[N]
    key = (N, tune)
    if key not in _out:
        _out[key] = (torch.empty(M2, N, device=DEV), torch.empty(Bc, 4, LL, device=DEV))
    out, bT = _out[key]
    if not with_side:
        ops.gemm(z, W, out, bias=b, ln=(None, cs), B3=_w3[N], exact=2, tune=tune)
        return out, None
    g1 = ops.gemm(z, W, out, bias=b, ln=(None, cs), B3=_w3[N], exact=2, tune=tune, defer=True)
    g2 = ops.gemm(z3, Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, tune=tune, defer=True)
    ops.gemm_side(g1, g2)
    return out, bT


_w, _w3 = {}, {}
for N in (576, 768, 512):
    W = r(192, N) / 14
    _w[N] = (W, r(N), W.sum(0).contiguous())
    _w3[N] = ops.split_weights(W)
Wp = r(192, 4) / 14
bp, csp = r(4), Wp.sum(0).contiguous()
Wp3 = ops.split_weights(Wp)

Lp = (L + 3) // 4 * 4
KT = (Lp + 15) // 16
pm = (torch.rand(Bc * L * Lp, device=DEV) > 0.1).float()


def glu(tune, outgoing):
    W, b, cs = _w[512]
    key = ('glu', tune, outgoing)
    if key not in _out:
        _out[key] = torch.zeros(Bc, 256, KT, 2, L, 16, device=DEV, dtype=torch.int16)
    lrp = _out[key]
    ops.gemm(z3, W, lrp, bias=b, ln=(None, cs), B3=_w3[512], exact=2, tune=tune, rowscale=pm, glu=True, c_split_nA=128, c_split_tile=True,
             a_pair_transpose=0 if outgoing else L, pair=(L, Lp), a_pair=True)
    return lrp


def same(a, b):
    if a.dtype == torch.int16:
        return bool((a == b).all()), int((a != b).sum())
    return bool((a.view(torch.int32) == b.view(torch.int32)).all()), float((a - b).abs().max())


if what in ('check', 'all'):
    for N, side in ((576, True), (768, False), (576, False)):
        o1, s1 = qkv_side(N, 0, side)
        o0, s0 = qkv_side(N, OLD, side)
        print(f'plain N={N} side={side}: main identical={same(o1, o0)}', ('side identical=%s' % (same(s1, s0),)) if side else '',
              'finite', bool(torch.isfinite(o1).all()), flush=True)
    for outgoing in (True, False):
        a, b = glu(0, outgoing), glu(OLD, outgoing)
        print(f'glu outgoing={outgoing}: identical={same(a, b)} nonzero={int((a != 0).sum())}', flush=True)
    torch.cuda.synchronize()

if what in ('time', 'all'):
    for name, fn in (('qkv 576 + side  AS', lambda: qkv_side(576, 0)), ('qkv 576 + side  tiles', lambda: qkv_side(576, OLD)),
                     ('qkvg 768        AS', lambda: qkv_side(768, 0, False)), ('qkvg 768        tiles', lambda: qkv_side(768, OLD, False)),
                     ('glu 512 outgoing AS', lambda: glu(0, True)), ('glu 512 outgoing tiles', lambda: glu(OLD, True)),
                     ('glu 512 incoming AS', lambda: glu(0, False)), ('glu 512 incoming tiles', lambda: glu(OLD, False)),
                     ('qkv 576 + side  AS', lambda: qkv_side(576, 0)), ('qkv 576 + side  tiles', lambda: qkv_side(576, OLD))):
        ms = timeit(fn, reps=7)
        print(f'{name:26s} Bc={Bc} L={L} {ms:8.3f} ms', flush=True)
