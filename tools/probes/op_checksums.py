"""Per-op checksums of one model call (n = 2, 6qd7): run solo and concurrently with a second process on the same GPU; the first
op whose checksum differs points at the kernel with a scheduling-dependent result."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from collections import OrderedDict
from abx_amd import features, sampler, synthetic, ops
import abx_amd.model.forward as F
from abx_amd.config import default_config
from abx_amd.diffuser.full_diffuser import FullDiffuser
from abx_amd.model.abx import ScoreNetwork
from abx_amd.data.antibody import load_complex
dev = torch.device('cuda:0')
cfg = default_config()
D = FullDiffuser.get(cfg.diffuser).to(dev)
model = ScoreNetwork(cfg.model, D)
sd = synthetic.random_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items()), seed=0)
model.load_state_dict(sd, strict=True)
model = model.to(dev).eval()
cb = load_complex(os.path.join(ROOT, 'tests/golden/pdb', '6qd7_X_Z_F|E.pdb'), seed=0)
one = {k: v.to(dev) for k, v in cb.items() if torch.is_tensor(v)}
L = one['seq'].shape[1]
ids = [0, 1]
raw = {k: v.expand(2, *v.shape[1:]).contiguous() for k, v in one.items()}
b = features.build_features(raw, D, generate_area='H3', noise=features.per_sample_init_noise(ids, L, 0, dev))
b['_shared_context'] = True
b = sampler.set_t_feats(b, D, torch.full((2,), 0.5, dtype=torch.float64, device=dev), torch.ones(2, device=dev))
log = []
def cks(t):
    if t.dtype in (torch.float32, torch.int32):
        v = t.contiguous().view(torch.int32) if t.is_contiguous() else t.clone().contiguous().view(torch.int32)
    elif t.dtype in (torch.float64, torch.int64):
        v = (t.contiguous() if t.is_contiguous() else t.clone()).view(torch.int64)
    elif t.dtype == torch.int16:
        v = t.contiguous().to(torch.int32) if t.numel() < (1 << 28) else t.contiguous()[: 1 << 28].to(torch.int32)
    else:
        v = t.to(torch.int32)
    return int(v.to(torch.int64).sum().item()) & 0xffffffffffff
skip = ('gemm_kernel_name', 'gemm_split_eligible', 'tri_attn_kernel_name', 'gemm_mode', 'ipa_qpack_numel', 'split_weights', 'vdw_radius_table')
REPS = int(os.environ.get('REPS', 1))
for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not name.startswith('_') and name not in skip and getattr(fn, '__module__', '') == ops.__name__:
        def wrap(fn=fn, name=name):
            def inner(*a, **k):
                r = fn(*a, **k)
                ts = [x for x in list(a) + list(k.values()) if torch.is_tensor(x) and x.is_cuda]
                for x in list(k.values()):
                    if isinstance(x, tuple): ts += [y for y in x if torch.is_tensor(y) and y.is_cuda]
                log.append((name, tuple(cks(x) for x in ts)))
                return r
            return inner
        setattr(ops, name, wrap())
        if hasattr(F.ops, name): setattr(F.ops, name, getattr(ops, name))
for rep in range(REPS):
    del log[:]
    bb = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
    r = model(bb)
    torch.cuda.synchronize()
    with open(os.environ.get('OUT', '/tmp/race.txt') + (f'.{rep}' if REPS > 1 else ''), 'w') as f:
        for i, (n, c) in enumerate(log):
            f.write(f'{i} {n} {" ".join(map(str, c))}\n')
print('ops', len(log))
