"""Golden vectors for the token reverse RATES: the UNMODIFIED reference's
DiscreteDiffuser.reverse (diffuser/discrete_diffuser.py:130-190) run on CPU with the ARGUMENT of its Poisson draw recorded.

Run in the build container only:   python tests/golden/make_golden_rates.py        -> tests/golden/rates_tiny.npz

The reference draws `torch.distributions.Poisson(reverse_rates * dt).sample()`, i.e. `torch.poisson(reverse_rates * dt)`; the
recorder below captures that tensor (`lam`), the draw (`jumps`) and the returned tokens.  Cases: t in {1.0, 0.5, 0.02} x dt in
{0.01, 0.1}; tokens that include 0, 19 and out-of-range values (clamped by the reference); logits = N(0, 3^2) rows plus peaked rows
(one-hot * 30: rates * dt up to ~5 at t = 0.02, dt = 0.1) and rows peaked ON the current token.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.makedirs('/tmp/abx_golden_scratch', exist_ok=True)
os.chdir('/tmp/abx_golden_scratch')

import torch  # noqa: E402

torch.set_num_threads(4)
from ref_shims import ConfigDict  # noqa: E402

cfg_json = json.load(open('/root/reference/config/config_model.json'))
cfg = ConfigDict(cfg_json)
from diffuser import discrete_diffuser  # noqa: E402

assert discrete_diffuser.__file__.startswith(ref_shims.REF)
dd = discrete_diffuser.DiscreteDiffuser(cfg.diffuser.seq)

B, L, K = 3, 40, 20
g = torch.Generator().manual_seed(2024)
x_t = torch.randint(0, K, (B, L), generator=g)
x_t[0, 0], x_t[0, 1], x_t[0, 2], x_t[0, 3] = 0, 19, -1, 25          # edge tokens; -1 / 25 are clamped by the reference
x_t[1, 0], x_t[1, 1] = 19, 0
logits = 3.0 * torch.randn(B, L, K, generator=g)
for b in range(B):
    for l in range(4, 16):                                              # peaked on another token
        s = int((int(x_t[b, l]) + 1 + 3 * l) % K)
        logits[b, l] = 0.0
        logits[b, l, s] = 30.0
    for l in range(16, 20):                                             # peaked on the current token
        logits[b, l] = 0.0
        logits[b, l, int(torch.clamp(x_t[b, l], 0, K - 1))] = 30.0

_orig = torch.poisson
rec = []


def rec_poisson(rate, *a, **k):
    z = _orig(rate, *a, **k)
    rec.append((rate.clone(), z.clone()))
    return z


out = dict(x_t=x_t.numpy(), logits=logits.numpy(), rate_const=np.float32(dd.rate_const))
cases = []
torch.manual_seed(77)
torch.poisson = rec_poisson
try:
    for ti, t in enumerate((1.0, 0.5, 0.02)):
        for di, dt in enumerate((0.01, 0.1)):
            rec.clear()
            # the loop's dtypes: t a float64 0-dim tensor tiled over the batch (inference.py:216), dt torch.tensor(1/num_t) fp32
            t_ = torch.tile(torch.tensor(np.float64(t)), (B,))
            x_new = dd.reverse(x_t=x_t, logits_t=logits, t=t_, dt=torch.tensor(dt))
            assert len(rec) == 1
            key = f'c{ti}{di}'
            out[key + '.t'] = np.float64(t)
            out[key + '.dt'] = torch.tensor(dt).numpy()
            out[key + '.lam'] = rec[0][0].numpy()
            out[key + '.jumps'] = rec[0][1].numpy()
            out[key + '.x_new'] = x_new.numpy()
            cases.append(key)
            print(key, 't', t, 'dt', dt, 'lam max', float(rec[0][0].max()), 'dtype', rec[0][0].dtype, 'changed', int((x_new != torch.clamp(x_t, 0, 19)).sum()))
finally:
    torch.poisson = _orig
out['cases'] = np.array(cases)
path = os.path.join(HERE, 'rates_tiny.npz')
np.savez_compressed(path, **out)
print('wrote rates_tiny.npz', os.path.getsize(path) // 1024, 'KiB')
