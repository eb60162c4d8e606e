"""Phase stamps of ipa_weights_kernel (probe build: tools/probes/build_probe.sh ipastamp ipa.hip -DIPA_STAMP):
    python tools/ab_lib.py tools/probes/bin/libabx_hip_ipastamp.so tools/probes/kb_ipa_stamps.py [Bc] [L]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops, _lib
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 352
M1, M2 = Bc * L, Bc * L * L
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV)
qp, kp, vp = r(ops.ipa_qpack_numel(Bc, L)), r(M1 * 12 * 28), r(M1 * 12 * 40)
bias2d = r(M2, 12)
mask = torch.ones(Bc, L, device=DEV)
R = torch.linalg.qr(r(M1, 3, 3))[0].reshape(M1, 9).contiguous()
t, pw = r(M1, 3), torch.rand(12, device=DEV)
attn, feat = torch.empty(M2, 12, device=DEV), torch.zeros(M1, 2112, device=DEV)
for _ in range(3):
    ops.ipa_weights(qp, kp, vp, bias2d, mask, R, t, pw, attn, feat, Bc, L)
torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64)
lib = _lib.load()
lib.abx_ipa_stamps.argtypes = [C.c_void_p]
assert lib.abx_ipa_stamps(buf.ctypes.data) == 0
st = buf.reshape(1024, 8).astype(np.float64)
d = st[:, 1:] - st[:, :-1]
names = ['phase A (logits: K loads, Q through the scalar cache, packed FMAs)', 'barrier after phase A', 'bias + mask, V prefetch, barrier', 'softmax + weights to HBM, barrier',
         'phase B1 (scalar / point outputs)', 'fold of the key groups (6 barriers)', 'tail (frames, norms, feature stores)']
print(f'ipa_weights stamps Bc={Bc} L={L}: shader-clock ticks of thread 0 per phase, mean / min / max over 1024 workgroups in the middle of the grid')
for i, nm in enumerate(names):
    print(f'   {nm:72s} {d[:, i].mean():9.0f} {d[:, i].min():9.0f} {d[:, i].max():9.0f}')
print(f'   {"workgroup total":72s} {(st[:, 7] - st[:, 0]).mean():9.0f}')
