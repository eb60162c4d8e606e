"""Upper bound of what fusing a row-local op-group TAIL into the next group's row-local HEAD could return: every head
of the pair stack re-reads from HBM the z rows the previous tail has just written.  A fused kernel would hand them over in LDS; the most
it can save on the head's side is what the head gains when its A rows cost no HBM traffic at all - measured here by collapsing the A
operand onto ONE row (row stride 0: every DMA still issues and every byte still crosses L2 -> LDS, but all of them hit one 768-byte line).
The same for the tails' own z reads (gate operand + residual) and for the closing chain (transition -> proj_init_pair_act + LN -> IPA
pair-bias projection).  What a fused kernel would ADD (the tail's arithmetic inside the head's block, un-overlapped) is not in these
numbers: they are the ceiling of the gain, per launch, at the bench geometry.
    python tools/probes/kb_fuse.py [Bc] [L]"""
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
zb = r(1, 192)
z1 = zb.expand(M2, 192)
z3, z31 = z.view(Bc, LL, 192), zb.view(1, 1, 192).expand(Bc, LL, 192)
rows = []


def rec(name, normal, collapsed, note=''):
    a, b = timeit(normal, reps=7), timeit(collapsed, reps=7)
    rows.append((name, a, b))
    print(f'{name:58s} normal {a:8.3f} ms | A rows -> 1 row {b:8.3f} ms | ceiling of the gain {a - b:6.3f} ms {note}', flush=True)


# ---- heads: q | k | v + pair bias (gemm_as PLAIN + SIDE)
W, Wp = r(192, 576) / 14, r(192, 4) / 14
C, bT = torch.empty(M2, 576, device=DEV), torch.empty(Bc, 4, LL, device=DEV)
bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
def qkv(a2, a3):
    ops.gemm_side(ops.gemm(a2, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True),
                  ops.gemm(a3, Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True))
rec('head: q | k | v + bias (gemm_as<0, true>)', lambda: qkv(z, z3), lambda: qkv(z1, z31), '(x 6 per step)')
del C
# ---- heads: glu projections -> operand images (gemm_as GLU)
Wv, Wg = r(192, 256) / 14, r(192, 256) / 14
Wglu, bglu = ops.pack_glu_weights(Wv, Wg, r(256), r(256))
Wglu3, csglu = ops.split_weights(Wglu), Wglu.sum(0).contiguous()
lrp = torch.zeros(Bc, 256, (L + 15) // 16, 2, L, 16, dtype=torch.int16, device=DEV)
pm = torch.ones(Bc * LL, device=DEV)
def glu(a3, tr):
    ops.gemm(a3, Wglu, lrp, bias=bglu, ln=(None, csglu), B3=Wglu3, rowscale=pm, glu=True, exact=2, c_split_nA=128, c_split_tile=True, pair=(L, L), a_pair=True,
             a_pair_transpose=L if tr else 0)
rec('head: glu outgoing (gemm_as<1>)', lambda: glu(z3, False), lambda: glu(z31, False), '(x 3 per step)')
rec('head: glu incoming, pair-transposed rows (gemm_as<1>)', lambda: glu(z3, True), lambda: glu(z31, True), '(x 3 per step)')
del lrp
# ---- tail: tri-mul output projection x final gate + residual (gemm3_dual)
tt, Wo, Wfg = r(Bc, 128, LL), r(128, 192) / 11, r(192, 192) / 14
Wo3, cso, bo, Wfg3, csfg, bfg = ops.split_weights(Wo), Wo.sum(0).contiguous(), r(192), ops.split_weights(Wfg), Wfg.sum(0).contiguous(), r(192)
zout = torch.empty(Bc, LL, 192, device=DEV)
def dual(zin):
    ops.gemm(tt.transpose(1, 2), Wo, zout, bias=bo, ln=(None, cso), B3=Wo3, resid=zin, dual=(zin, Wfg3, csfg, bfg), exact=2)
rec('tail: tri-mul proj_out x gate + z (gemm3_dual), z rows', lambda: dual(z3), lambda: dual(z31), '(x 6 per step; its own z read: stays in a fused kernel)')
del tt, zout
# ---- closing chain: transition -> proj_init_pair_act + LN (gemm3_oln) -> IPA pair-bias projection (128 -> 12)
Wi, bi = r(192, 128) / 14, r(128)
Wi3 = ops.split_weights(Wi)
lnw, lnb = 1 + 0.1 * r(128), 0.1 * r(128)
zi = torch.empty(M2, 128, device=DEV)
def oln(a2):
    ops.gemm(a2, Wi, zi, bias=bi, B3=Wi3, out_ln=(lnw, lnb), exact=2)
rec('closing: proj_init_pair_act + LN (gemm3_oln), z rows', lambda: oln(z), lambda: oln(z1), '(x 3 per step)')
Wb, bb = r(128, 12) / 11, r(12)
Wb3 = ops.split_weights(Wb)
zi.normal_()
zi1 = r(1, 128).expand(M2, 128)
b12 = torch.empty(M2, 12, device=DEV)
def ipab(a2):
    ops.gemm(a2, Wb, b12, bias=bb, B3=Wb3, exact=2)
rec('closing: IPA pair-bias projection 128 -> 12, rows', lambda: ipab(zi), lambda: ipab(zi1), '(x 3 per step)')
# ---- front: seq-attention pair bias 192 -> 32 (transposed store)
Ws, bs = r(192, 32) / 14, r(32)
Ws3, css = ops.split_weights(Ws), Ws.sum(0).contiguous()
bT32 = torch.empty(Bc, 32, LL, device=DEV)
def sab(a3):
    ops.gemm(a3, Ws, bT32.transpose(1, 2), bias=bs, ln=(None, css), B3=Ws3, exact=2)
rec('front: seq-attention pair bias 192 -> 32, z rows', lambda: sab(z3), lambda: sab(z31), '(x 3 per step)')
# ---- the fused kernels' own re-reads (r05b_kb_rereads.txt, repeated on this box for one table)
W1, W2 = r(192, 768) / 14, r(768, 192) / 28
b1, cs1, b2 = r(768), r(768), r(192)
W13, W23p = ops.split_weights(W1), ops.split_weights(ops.permute_k16(W2))
out = torch.empty(M2, 192, device=DEV)
def mlp(a, res):
    ops.gemm(a, W1, out, bias=b1, ln=(None, cs1), B3=W13, act=1, resid=res, exact=2, mlp=(W23p, b2))
rec('transition (gemm3_mlp<2>): z rows, A + residual', lambda: mlp(z, z), lambda: mlp(z1, z1), '(x 3 per step; 7 reads of z -> 0)')
o = r(M2, 192)
Wg2, Wo2 = r(192, 192) / 14, r(192, 192) / 14
bg, csg, bo2 = r(192), r(192), r(192)
Wg23, Wo23p = ops.split_weights(Wg2), ops.split_weights(ops.permute_k16(Wo2))
def tail(a, res):
    ops.gemm(a, Wg2, out, bias=bg, ln=(None, csg), B3=Wg23, act=2, gate=o, resid=res, exact=2, mlp=(Wo23p, bo2))
rec('attention tail (gemm3_gtail): z rows, A + residual', lambda: tail(z, z), lambda: tail(z1, z1), '(x 6 per step; 3 reads of z -> 0)')
per_step = {'head: q | k | v + bias (gemm_as<0, true>)': 6, 'head: glu outgoing (gemm_as<1>)': 3, 'head: glu incoming, pair-transposed rows (gemm_as<1>)': 3,
            'closing: proj_init_pair_act + LN (gemm3_oln), z rows': 3, 'closing: IPA pair-bias projection 128 -> 12, rows': 3}
tot = sum((a - b) * per_step[n] for n, a, b in rows if n in per_step)
print(f'\nceiling for ALL head-side reads of the step (6 q|k|v + 6 glu + 3 oln + 3 IPA-bias launches): {tot:.1f} ms per step at Bc = {Bc}, L = {L}')
tot2 = sum((a - b) * k for (n, a, b), k in zip(rows[-2:], (3, 6)))
print(f'ceiling for the re-reads inside the fused transition and the attention tail (A-stationary forms): {tot2:.1f} ms per step')
