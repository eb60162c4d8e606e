"""Shader-clock stamps of tri_attn8_kernel (diagnostic build -DTRI8_STAMP, tools/probes/build_stamp.sh): wave 0 and the producer wave of
the first 4096 workgroups stamp s_memtime at the prologue barrier, around every chunk barrier and at the end.
    python tools/ab_lib.py tools/probes/bin/libabx_stamp.so tools/probes/tri_stamps.py [Bc] [L] [tune]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 352
tune = int(sys.argv[3]) if len(sys.argv) > 3 else 0
DEV = 'cuda:0'
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
x, bT, mask, o = r(M2, 768), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True, tune=tune)
acc = torch.zeros(16 + 4096 * 32, dtype=torch.int64, device=DEV)
ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True, clock_probe=acc, tune=tune)
torch.cuda.synchronize()
st = acc[16:].view(4096, 32).cpu().double()
w0, pr = st[:, :16], st[:, 16:]
n0 = int((w0[0] > 0).sum()); npr = int((pr[0] > 0).sum())
d0 = (w0[:, 1:n0] - w0[:, :n0 - 1]).mean(0)
print(f'tri_attn8 stamps Bc={Bc} L={L} tune={tune}: mean shader-clock ticks between consecutive stamps, over the LAST 4096 workgroups that wrote a slot')
names0 = ['start -> prologue done (Q, chunk 0 staged, first bias issued)', 'first barrier wait']
nch = (n0 - 4) // 2
for c in range(nch):
    names0 += [f'chunk {c} compute', f'chunk {c} barrier wait']
names0 += ['epilogue (gate, store)']
print(' wave 0:')
for nm, v in zip(names0, d0.tolist()):
    print(f'   {nm:64s} {v:9.0f}')
print(f'   {"total":64s} {float((w0[:, n0 - 1] - w0[:, 0]).mean()):9.0f}')
print(' producer wave (relative to wave 0 start):')
for i in range(npr):
    print(f'   stamp {i}: {float((pr[:, i] - w0[:, 0]).mean()):9.0f}')
