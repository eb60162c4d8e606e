"""Triangle attention variants (AbxTriAttn.tune): 0 = library choice (tri_attn8: 11 computing waves walk two query tiles together + 1
producer wave), 2 = the same with the other key-chunk size (128 <-> 192), 4 = the round-3 kernel tri_attn4 (one query tile at a time; producer
wave), 5 = tri_attn4 with 12 computing waves that share the staging.  Usage: kb_tri.py [Bc] [L] [m = mask the last 7 keys | -] [log2]"""
import sys
import torch
sys.path.insert(0, '/root/repo')
from abx_amd import ops
from tools.kbench import timeit
DEV = 'cuda:0'
Bc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
x, bT, mask = r(M2, 576), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV)          # (round 5: q | k | v only: no gate)
if len(sys.argv) > 3 and sys.argv[3] == 'm': mask[:, L - 7:] = 0
BL2 = len(sys.argv) > 4 and sys.argv[4] == 'log2'      # AbxTriAttn.bias_log2: the bias arrives in accumulator units
outs = {}
for per_row in (True, False):
    for tune in (0, 2, 4, 5):
        o = torch.empty(M2, 192, device=DEV)
        ops.tri_attn(x, bT, mask, o, Bc, L, per_row, bias_is_qk=True, tune=tune, bias_log2=BL2)
        outs[(per_row, tune)] = o
        ms = timeit(lambda: ops.tri_attn(x, bT, mask, o, Bc, L, per_row, bias_is_qk=True, tune=tune, bias_log2=BL2), reps=7)
        print(f'per_row={per_row} tune={tune}: {ms:7.3f} ms  {4.0 * Bc * L * 4 * LL * 48 / ms / 1e9:6.1f} TFLOP/s', flush=True)
    print('   bit-identical 0 vs 2:', torch.equal(outs[(per_row, 0)], outs[(per_row, 2)]), ' 4 vs 5:', torch.equal(outs[(per_row, 4)], outs[(per_row, 5)]),
          ' max |tri_attn8 - tri_attn4|', float((outs[(per_row, 0)] - outs[(per_row, 4)]).abs().max()))
if L <= 389:
    oe = torch.empty(M2, 192, device=DEV)
    ops.tri_attn(x, bT, mask, oe, Bc, L, True, bias_is_qk=True, exact=True)
    print('max |split - exact|', float((outs[(True, 0)] - oe).abs().max()), 'max |exact|', float(oe.abs().max()))
