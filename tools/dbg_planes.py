import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abx_amd import ops
DEV = 'cuda:0'
torch.manual_seed(0)
Bc, L = 5, 120
LL = L * L
z3 = torch.randn(Bc, LL, 192, device=DEV)
W = torch.randn(192, 128, device=DEV) / 14
Wg = torch.randn(192, 256, device=DEV) / 14
w3, wg3 = ops.split_weights(W), ops.split_weights(Wg)
pm = (torch.rand(Bc, L, L, device=DEV) > 0.2).float()
pm = pm * pm.transpose(1, 2)
pmask = pm.reshape(-1).contiguous()
KT = (L + 15) // 16
for outgoing in (True, False):
    pt = 0 if outgoing else L
    GT = torch.empty(Bc, 256, LL, device=DEV)
    ops.gemm(z3, Wg, GT.transpose(1, 2), act=2, B3=wg3, a_pair_transpose=pt)
    GTr = torch.empty(Bc, 256, LL, device=DEV)
    ops.gemm(z3, Wg, GTr.transpose(1, 2), act=2, exact=True)
    if not outgoing:
        GTr = GTr.view(Bc, 256, L, L).transpose(2, 3).reshape(Bc, 256, LL)
    print('gates', outgoing, (GT - GTr).abs().max().item())
    lp = torch.zeros(Bc, 128, KT, 3, L, 16, dtype=torch.int16, device=DEV)
    rp = torch.zeros(Bc, 128, KT, 3, L, 16, dtype=torch.int16, device=DEV)
    ops.gemm(z3, W, lp, rowscale=pmask, gate=GT[:, :128].transpose(1, 2), gate_sigmoid=False, B3=w3, a_pair_transpose=pt)
    ops.gemm(z3, W, rp, rowscale=pmask, gate=GT[:, 128:].transpose(1, 2), gate_sigmoid=False, B3=w3, a_pair_transpose=pt)
    tz = torch.empty(Bc * 128, L, L, device=DEV)
    ops.gemm(lp.view(Bc * 128, KT, 3, L, 16), rp.view(Bc * 128, KT, 3, L, 16), tz)
    # reference with torch
    zz = z3.view(Bc, L, L, 192).double()
    g = torch.sigmoid(zz @ Wg.double())
    left = (zz @ W.double()) * pm.double()[..., None] * g[..., :128]
    right = (zz @ W.double()) * pm.double()[..., None] * g[..., 128:]
    ref = torch.einsum('bikc,bjkc->bijc', left, right) if outgoing else torch.einsum('bkic,bkjc->bijc', left, right)
    got = tz.view(Bc, 128, L, L).permute(0, 2, 3, 1).double()
    print('contraction', outgoing, (got - ref).abs().max().item(), ref.abs().max().item())
    lf = ops.planes_to_float(lp, dim=3)     # (Bc,128,KT,L,16)
    lf = lf.permute(0, 1, 3, 2, 4).reshape(Bc, 128, L, KT * 16)[..., :L]     # [b,c,i,k]
    lref = left.permute(0, 3, 1, 2) if outgoing else left.permute(0, 3, 2, 1)
    print('left planes', outgoing, (lf.double() - lref).abs().max().item())
