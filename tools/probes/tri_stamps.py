"""Shader-clock stamps of the persistent tri_attn8_kernel (diagnostic build -DTRI8_STAMP, tools/probes/build_stamp.sh): wave 0 and the
producer wave of every workgroup stamp s_memtime while the workgroup is on its fourth row.
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
W = 768 if (len(sys.argv) > 4 and sys.argv[4] == "gate") else 576          # (round 5: q | k | v only, the gated tail applies the gate)
x, bT, mask, o = r(M2, W), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True, tune=tune)
NWG = 256
acc = torch.zeros(16 + NWG * 32, dtype=torch.int64, device=DEV)
ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True, clock_probe=acc, tune=tune)
torch.cuda.synchronize()
st = acc[16:].view(NWG, 32).cpu().double()
w0, pr = st[:, :16], st[:, 16:]
n0 = int((w0[0] > 0).sum()); npr = int((pr[0] > 0).sum())
d0 = (w0[:, 1:n0] - w0[:, :n0 - 1])
print(f'tri_attn8 stamps Bc={Bc} L={L} tune={tune}: shader-clock ticks between consecutive stamps of wave 0 on the 4th row of each workgroup (mean / min / max over {NWG} workgroups)')
names0 = ['row start -> Q split, first bias issued']
nch = (n0 - 3) // 2
for c in range(nch):
    names0 += [f'chunk {c} compute', f'chunk {c} barrier wait']
names0 += ['epilogue (normalise, store)']
for i, nm in enumerate(names0):
    print(f'   {nm:48s} {float(d0[:, i].mean()):9.0f} {float(d0[:, i].min()):9.0f} {float(d0[:, i].max()):9.0f}')
print(f'   {"row total":48s} {float((w0[:, n0 - 1] - w0[:, 0]).mean()):9.0f}')
print(' producer wave, relative to wave 0 row start (stage begin / end per chunk of the row, then row end):')
print('   ' + '  '.join(f'{float((pr[:, i] - w0[:, 0]).mean()):8.0f}' for i in range(npr)))
