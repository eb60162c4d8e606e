"""Where a block of the A-stationary GEMM spends its time (probe library built with -DABX_AS_STAMP): shader ticks of wave 0 in the walk's
rendezvous, in the walk, before it (A burst + split) and after it (last epilogue), averaged over the blocks; and the in-kernel clock.
    python tools/ab_lib.py tools/probes/bin/libabx_hip_stamp.so tools/probes/kb_as_stamp.py [Bc]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit

DEV = 'cuda:0'
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
z = r(M2, 192)
z3 = z.view(Bc, LL, 192)
W = r(192, 576) / 14
b, cs, W3 = r(576), W.sum(0).contiguous(), ops.split_weights(W)
Wp = r(192, 4) / 14
bp, csp, Wp3 = r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
out = torch.empty(M2, 576, device=DEV)
bT = torch.empty(Bc, 4, LL, device=DEV)
Wg = r(192, 512) / 14
bg, csg, Wg3 = r(512), Wg.sum(0).contiguous(), ops.split_weights(Wg)
lrp = torch.zeros(Bc, 256, L // 16, 2, L, 16, device=DEV, dtype=torch.int16)
pm = torch.ones(Bc * L * L, device=DEV)
NBLK = (M2 + 63) // 64 + Bc * 4
acc = torch.zeros(16 + 16 * NBLK, dtype=torch.int64, device=DEV)


def side(tune):
    g1 = ops.gemm(z, W, out, bias=b, ln=(None, cs), B3=W3, exact=2, tune=tune, defer=True, clock_probe=acc)
    g2 = ops.gemm(z3, Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, tune=tune, defer=True)
    ops.gemm_side(g1, g2)


def glu(tune):
    ops.gemm(z3, Wg, lrp, bias=bg, ln=(None, csg), B3=Wg3, exact=2, tune=tune, rowscale=pm, glu=True, c_split_nA=128, c_split_tile=True,
             a_pair_transpose=0, pair=(L, L), a_pair=True, clock_probe=acc)


for name, fn in (('side', side), ('glu', glu)):
    for abl, tag in ((0, 'full'), (1, 'no slice stores'), (2, 'no MFMA'), (3, 'no weight DMA in the walk')):
        fn(abl << 12)
        torch.cuda.synchronize()
        acc.zero_()
        ms = timeit(lambda: fn(abl << 12), reps=3)
        nb_launch = (M2 + 63) // 64 if name == 'side' else Bc * ((L + 7) // 8 * 8) * ((L + 15) // 16 * 16) // 64
        allrec = acc[16:].view(-1, 8).double()
        rec2 = allrec[nb_launch:2 * nb_launch]
        rec2 = rec2[rec2[:, 7] > 0]
        if rec2.shape[0] and float(rec2[:, 0].sum()) > 0:
            print(f'      streaming wave (0, 1), per block: vmcnt wait {float(rec2[:, 0].mean()):8.0f}  barrier wait {float(rec2[:, 1].mean()):8.0f} ticks')
        rec = allrec[:nb_launch]
        rec = rec[rec[:, 7] > 0]
        m = rec.mean(0).tolist()
        print(f'{name:5s} {tag:26s} {ms:7.3f} ms  per block ({rec.shape[0]} blocks): total {m[6]:8.0f} = issue {m[4]:6.0f} | burst landed {m[5]:6.0f} | '
              f'stats + split {m[2] - m[4] - m[5]:6.0f} | walk {m[1]:8.0f} (rendezvous {m[0]:8.0f}) | last epilogue {m[3]:7.0f} ticks', flush=True)
