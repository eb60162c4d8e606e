"""Golden vectors for the raw-PDB featurisation, produced by the UNMODIFIED reference in the build container:

    python tests/golden/make_golden_pdb.py

Biopython and ANARCI are not available here, so the reference's own parser / numbering cannot run; what CAN run is everything
after them.  The chain features and IMGT region labels produced by abx_amd.io.pdb_reader / abx_amd.data.antibody (the `struc`
dict of the reference's make_pdb_npz) are fed through the reference's IgStructureData.get_structure_label_npz (centring,
Patch_Around_Anchor), the antigen window crop, collate_fn and the seven FeatureBuilder transforms; the outputs are committed as
tests/golden/pdb_<code>.npz.  The two PDB files under tests/golden/pdb/ are the reference's own example inputs (test_data/).
"""
import copy
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.makedirs('/tmp/abx_golden_scratch', exist_ok=True)
os.chdir('/tmp/abx_golden_scratch')

import torch  # noqa: E402
from ref_shims import ConfigDict  # noqa: E402

cfg_json = json.load(open('/root/reference/config/config_model.json'))
cfg_json['diffuser']['so3']['use_cached_score'] = True
cfg = ConfigDict(cfg_json)
from diffuser.full_diffuser import FullDiffuser  # noqa: E402

diffuser = FullDiffuser.get(cfg.diffuser)
from abx.data import dataset as ref_dataset  # noqa: E402
from abx.model.features import FeatureBuilder  # noqa: E402

sys.path.insert(0, ROOT)
from abx_amd.data import antibody as A  # noqa: E402
assert ref_dataset.__file__.startswith(ref_shims.REF)

feat_conf = json.load(open('/root/reference/config/config_data_feature.json'))
feats = []
for fn, opts in feat_conf:
    opts = dict(opts)
    if 'device' in opts:
        opts['device'] = torch.device('cpu')
    if 'diffuse' in fn:
        opts['diff_conf'] = cfg_json['diffuser']
        opts.pop('optimize_steps', None)
    feats.append((fn, opts))

SEED = 0
for fname in ('6ct7_H_L_S.pdb', '6qd7_X_Z_F|E.pdb'):
    path = os.path.join(HERE, 'pdb', fname)
    name, code, heavy, light, antigens = A.parse_pdb_name(path)
    struc = A.make_pdb_features(path, heavy, light, antigens)
    cdrs = struc.pop('cdrs')
    # the same chain features in the schema of the reference's preprocessed data set (make_pdb_npz): the inputs of its --name_idx /
    # --data_dir path (abx/data/dataset.py:90-214), used by tests/test_pdb_features.py::test_npz_entry_*
    os.makedirs(os.path.join(HERE, 'npz'), exist_ok=True)
    A.save_struc_npz(struc, os.path.join(HERE, 'npz', name + '.npz'))
    ds = object.__new__(ref_dataset.IgStructureData)               # the reference class without its Biopython-based __init__
    ds.ret, ds.pdb_name, ds.is_training, ds.max_antigen_seq_len = copy.deepcopy(struc), name, False, 32
    random.seed(SEED)                                              # the reference's window crop uses the global `random`
    ret = next(iter(ds))
    batch = ds.collate_fn([ret])
    torch.manual_seed(99)
    out = FeatureBuilder(feats, is_training=False)(copy.deepcopy(batch))
    torch.manual_seed(99)
    B, L = batch['seq'].shape
    noise = dict(rot_axis=torch.randn(B, L, 3), rot_u=torch.rand(B, L), trans_z=torch.randn(B, L, 3), seq=torch.randint(low=0, high=20, size=(B, L)))
    g = {('struc.' + k): np.asarray(v) for k, v in struc.items()}
    g.update({('batch.' + k): v.numpy() for k, v in batch.items() if torch.is_tensor(v)})
    g['batch.str_heavy_seq'] = np.array(batch['str_heavy_seq'][0])
    g['batch.str_light_seq'] = np.array(batch['str_light_seq'][0])
    g['batch.antigen_origin_str_seq'] = np.array(batch['antigen_origin_str_seq'][0])
    g['batch.antigen_origin_residx'] = np.asarray(batch['antigen_origin_residx'][0])
    g['batch.antigen_origin_chain_ids'] = np.asarray(batch['antigen_origin_chain_ids'][0])
    g.update({('noise.' + k): v.numpy() for k, v in noise.items()})
    for k in ('rigids_t', 'seq_t', 't', 'fixed_mask', 'rigids_0', 'torsion_angles_sin_cos', 'atom37_gt_positions', 'atom37_gt_exists', 'pseudo_beta'):
        g['feat.' + k] = out[k].numpy()
    g['cdr_h3'] = np.array(cdrs[0]['cdr3'])
    g['seed'] = np.int64(SEED)
    outp = os.path.join(HERE, f'pdb_{code}.npz')
    np.savez_compressed(outp, **g)
    print('wrote', outp, os.path.getsize(outp) // 1024, 'KiB; L =', L, 'Lab =', batch['anchor_flag'].shape[1], 'diffused', int((1 - out['fixed_mask']).sum()))
print('done')
