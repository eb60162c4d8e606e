"""Golden vectors for the peptide-geometry violation terms, produced by the UNMODIFIED reference:

    python tests/golden/make_golden_vio.py          -> tests/golden/vio_pdb.npz

`eval/metric_scripts/cal_vio.py::between_residue_bond_loss` (:29-110) is the only violation code the reference ships.  It returns
just the C-N violation mask; its other results (the three mean losses, the three masks, the per-residue flat-bottom losses and the
two angle violation masks) are read from the function's frame when it returns (sys.setprofile: nothing in the reference is edited).
Inputs: the two shipped example complexes as featurised by tests/golden/make_golden_pdb.py (pdb_6ct7.npz / pdb_6qd7.npz: atom14
coordinates, masks, chain ids, aatype) plus seeded Gaussian perturbations of the coordinates (sigma 0.05 / 0.15 / 0.4 A) so that
every flat bottom is left on many residues.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location('ref_cal_vio', os.path.join(ref_shims.REF, 'eval', 'metric_scripts', 'cal_vio.py'))
cal_vio = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cal_vio)
assert cal_vio.__file__.startswith(ref_shims.REF)

KEEP = ('c_n_loss_per_residue', 'ca_c_n_loss_per_residue', 'c_n_ca_loss_per_residue', 'c_n_loss', 'ca_c_n_loss', 'c_n_ca_loss',
        'c_n_violation_mask', 'ca_c_n_violation_mask', 'c_n_ca_violation_mask', 'has_no_gap_mask')


def run(pos, mask, chain, aatype):
    grabbed = {}

    def prof(frame, event, arg):
        if event == 'return' and frame.f_code.co_name == 'between_residue_bond_loss':
            for k in KEEP:
                grabbed[k] = frame.f_locals[k].detach().clone()

    sys.setprofile(prof)
    try:
        ret = cal_vio.between_residue_bond_loss(pos, mask, chain, aatype)
    finally:
        sys.setprofile(None)
    assert torch.equal(ret, grabbed['c_n_violation_mask'])
    return grabbed


out = {}
cases = []
for code in ('6ct7', '6qd7'):
    z = np.load(os.path.join(HERE, f'pdb_{code}.npz'))
    x0 = torch.from_numpy(z['batch.atom14_gt_positions'])
    m = torch.from_numpy(z['batch.atom14_gt_exists']).float()
    ch = torch.from_numpy(z['batch.chain_id'])
    aa = torch.from_numpy(z['batch.seq'])
    g = torch.Generator().manual_seed(17)
    for si, sigma in enumerate((0.0, 0.05, 0.15, 0.4)):
        x = x0 + sigma * torch.randn(x0.shape, generator=g)
        r = run(x, m, ch, aa)
        key = f'{code}.s{si}'
        out[key + '.pos'] = x.numpy()
        for k, v in r.items():
            out[f'{key}.{k}'] = v.numpy()
        cases.append(key)
        print(key, 'sigma', sigma, 'violations C-N', int(r['c_n_violation_mask'].sum()), 'CA-C-N', int(r['ca_c_n_violation_mask'].sum()),
              'C-N-CA', int(r['c_n_ca_violation_mask'].sum()), 'of', int(r['has_no_gap_mask'].sum()), 'pairs; losses',
              float(r['c_n_loss']), float(r['ca_c_n_loss']), float(r['c_n_ca_loss']))
out['cases'] = np.array(cases)
path = os.path.join(HERE, 'vio_pdb.npz')
np.savez_compressed(path, **out)
print('wrote vio_pdb.npz', os.path.getsize(path) // 1024, 'KiB')
