import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from abx_amd import ops, synthetic, features, sampler
from abx_amd.config import default_config
from abx_amd.model.abx import ScoreNetwork
from abx_amd.diffuser.full_diffuser import FullDiffuser
DEV = 'cuda:0'
cfg = default_config()
D = FullDiffuser(cfg.diffuser).to(DEV)
import json
from collections import OrderedDict
keys = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'sd_keys.json')))
shapes = OrderedDict((k, tuple(s)) for k, s in keys)
sd = synthetic.random_state_dict(shapes, seed=7)
model = ScoreNetwork(cfg.model, D)
model.load_state_dict(sd, strict=True)
model = model.to(DEV).eval()
w = dict(L_heavy=50, L_light=44, L_antigen=26, cdr=(30, 39))
B = 5
cx = synthetic.make_complex(seed=3, n_masked_tail=3, **w)
raw = {k: v.to(DEV) for k, v in synthetic.replicate(cx, B).items()}
torch.manual_seed(11)
b = features.build_features(raw, D)
t_ = torch.full((B,), 0.4040404040404041, dtype=torch.float64, device=DEV)
b = sampler.set_t_feats(b, D, t_, torch.ones(B, device=DEV))

def run(exact, planes_ok=True):
    bb = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
    ops.GEMM_EXACT = exact
    saved = ops.gemm_split_eligible
    if not planes_ok:
        ops.gemm_split_eligible = lambda *a, **k: False
    model.invalidate_static()
    r = model(bb)
    torch.cuda.synchronize()
    ops.GEMM_EXACT = False
    ops.gemm_split_eligible = saved
    return r['representations']['pair'].clone(), r['heads']['folding']['rigids'].clone()

pe, re_ = run(True)
ps, rs = run(False)
pn, rn = run(False, planes_ok=False)
print('split+planes vs exact: pair', (ps - pe).abs().max().item(), 'rigids', (rs - re_).abs().max().item())
print('split no planes vs exact: pair', (pn - pe).abs().max().item(), 'rigids', (rn - re_).abs().max().item())
