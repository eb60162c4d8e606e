"""The two launches of the IPA core (abx_ipa_weights, abx_ipa_pair) timed separately at the bench geometry.
    python tools/probes/kb_ipa.py [Bc] [L]        (under tools/ab_lib.py <variant library> for A / B runs)"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 352
M1, M2 = Bc * L, Bc * L * L
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV)
qp, kp, vp = r(ops.ipa_qpack_numel(Bc, L)), r(M1 * 12 * 28), r(M1 * 12 * 40)
bias2d, zz = r(M2, 12), r(M2, 128)
mask = (torch.rand(Bc, L, device=DEV) > 0.05).float()
mask[:, 0] = 1
R = torch.linalg.qr(r(M1, 3, 3))[0].reshape(M1, 9).contiguous()
t, pw = r(M1, 3), torch.rand(12, device=DEV)
attn, feat = torch.empty(M2, 12, device=DEV), torch.zeros(M1, 2112, device=DEV)
for rep in range(2):
    ms = timeit(lambda: ops.ipa_weights(qp, kp, vp, bias2d, mask, R, t, pw, attn, feat, Bc, L), reps=9)
    print(f'ipa_weights Bc={Bc} L={L}: {ms:7.3f} ms', flush=True)
    ms = timeit(lambda: ops.ipa_pair(attn, zz, feat, Bc, L), reps=9)
    print(f'ipa_pair    Bc={Bc} L={L}: {ms:7.3f} ms   {4.0 * M2 * 128 / ms / 1e9:6.2f} TB/s of slab', flush=True)
print('checksum', float(feat.double().abs().sum()), float(attn.double().sum()))
