import json
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def cfg():
    from abx_amd.config import default_config
    return default_config()


@pytest.fixture(scope='session')
def sd_shapes():
    keys = json.load(open(os.path.join(GOLDEN, 'sd_keys.json')))
    return OrderedDict((k, tuple(s)) for k, s in keys)


@pytest.fixture(scope='session')
def params(sd_shapes):
    from abx_amd import synthetic
    return synthetic.random_state_dict(sd_shapes, seed=7)


@pytest.fixture(scope='session')
def oracle_diffuser(cfg):
    """Full 1000x1000 IGSO(3) tables on CPU (~10 s), cached in /tmp for the session."""
    from oracle import abx_oracle as O
    cache = '/tmp/abx_oracle_igso3_tables.npz'
    tables = None
    if os.path.exists(cache):
        z = np.load(cache)
        tables = {k: torch.from_numpy(z[k]) for k in ('pdf', 'cdf', 'score_norms')}
    d = O.OracleDiffuser(cfg.diffuser, tables)
    if tables is None:
        np.savez(cache, pdf=d.so3._pdf.numpy(), cdf=d.so3._cdf.numpy(), score_norms=d.so3._score_norms.numpy())
    return d


def tt(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def feat_batch_from_golden(g):
    """Rebuild the featurised tiny batch (reference FeatureBuilder output) from feat_tiny.npz."""
    b = {k[4:]: tt(v) for k, v in g.items() if k.startswith('raw.')}
    for k, v in g.items():
        if k.startswith('feat.') and not k.startswith('feat.rigidgroups_gt_frames'):
            b[k[5:]] = tt(v)
    b['rigidgroups_gt_frames'] = (tt(g['feat.rigidgroups_gt_frames.0']), tt(g['feat.rigidgroups_gt_frames.1']))
    return b


def digest_batch(g, name, diffuser, build_features, device=None):
    """Rebuild the featurised batch of an L256 / L352 digest fixture: the bench's synthetic complex (regenerated from its seed),
    the feature pipeline run with the recorded init noise, then the reference's own diffusion features on top (so that the network
    sees bit-identical inputs), and the t features of the in-loop call."""
    from abx_amd import synthetic
    cx = synthetic.make_complex(seed=int(g['seed']), **synthetic.WORKLOADS[name])
    raw = synthetic.collate([cx])
    if device is not None:
        raw = {k: v.to(device) for k, v in raw.items()}
    noise = {k[6:]: tt(v) for k, v in g.items() if k.startswith('noise.')}
    if device is not None:
        noise = {k: v.to(device) for k, v in noise.items()}
    b = build_features(raw, diffuser, generate_area='H3', noise=noise)
    mine = {k: b[k].detach().cpu().clone() for k in ('rigids_t', 'seq_t', 'fixed_mask', 'torsion_angles_sin_cos', 'rigids_0')}
    for k in ('rigids_t', 'seq_t', 'fixed_mask', 'torsion_angles_sin_cos', 'rigids_0'):
        v = tt(g['feat.' + k])
        b[k] = v.to(device) if device is not None else v
    for k in ('t', 'rot_score_scaling', 'trans_score_scaling'):
        v = tt(g['in.' + k])
        b[k] = v.to(device) if device is not None else v
    return b, mine


def esm_tensor(g, B, Lab):
    """The seeded stand-in for the ESM2 per-layer representations used by tests/golden/make_golden_esm.py."""
    gen = torch.Generator().manual_seed(int(g['esm_seed']))
    return float(g['esm_scale']) * torch.randn(B, Lab, 2560, 37, generator=gen)


@pytest.fixture(scope='session')
def esm_setup():
    """Config with esm.enabled and the 197-tensor parameter set (190 + layer-mix weights + projection MLP)."""
    from abx_amd import synthetic
    from abx_amd.config import default_config
    cfg = default_config()
    cfg.model.embeddings_and_seqformer.esm.enabled = True
    keys = json.load(open(os.path.join(GOLDEN, 'sd_keys_esm.json')))
    shapes = OrderedDict((k, tuple(s)) for k, s in keys)
    return cfg, synthetic.random_state_dict(shapes, seed=7)
