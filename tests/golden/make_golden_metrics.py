"""Golden vectors for the design-quality metrics, produced by the UNMODIFIED reference:

    python tests/golden/make_golden_metrics.py          -> tests/golden/metrics_6qd7.npz

`abx.common.ab_utils.calc_ab_metrics` (abx/common/ab_utils.py:124-167: Kabsch-aligned C-alpha RMSD and amino-acid recovery per CDR,
plus the CDR-H3 loop variants) on the antibody of the shipped 6qd7 complex (pdb_6qd7.npz: C-alpha coordinates, IMGT region codes,
sequence) against seeded perturbed / rigidly moved copies with mutated CDR residues.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from abx.common import ab_utils  # noqa: E402

assert ab_utils.__file__.startswith(ref_shims.REF)
z = np.load(os.path.join(HERE, 'pdb_6qd7.npz'))
Lab = z['batch.anchor_flag'].shape[1]
ca = z['batch.atom14_gt_positions'][0, :Lab, 1].astype(np.float64)
cdr_def = z['batch.cdr_def'][0, :Lab]
seq = str(z['batch.str_heavy_seq']) + str(z['batch.str_light_seq'])
rng = np.random.default_rng(2025)
out = dict(gt_coord=ca, cdr_def=cdr_def, gt_str_seq=np.array(seq))
cases = []
for ci, (sigma, mut_rate) in enumerate(((0.0, 0.0), (0.3, 0.2), (1.5, 0.5), (4.0, 1.0))):
    # a random rigid motion on top of the noise: the metric must be invariant to it
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])
    pred = (ca + sigma * rng.normal(size=ca.shape)) @ R.T + rng.normal(size=3) * 20
    aas = 'ACDEFGHIKLMNPQRSTVWY'
    ps = ''.join((aas[rng.integers(20)] if (cdr_def[i] in (1, 3, 5, 8, 10, 12) and rng.random() < mut_rate) else ch) for i, ch in enumerate(seq))
    m = ab_utils.calc_ab_metrics(ca, pred, cdr_def, seq, ps)
    key = f'c{ci}'
    out[key + '.pred_coord'] = pred
    out[key + '.pred_str_seq'] = np.array(ps)
    out[key + '.names'] = np.array(list(m.keys()))
    out[key + '.values'] = np.array([float(v) for v in m.values()], dtype=np.float64)
    cases.append(key)
    print(key, {k: round(float(v), 4) for k, v in m.items()})
out['cases'] = np.array(cases)
path = os.path.join(HERE, 'metrics_6qd7.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
