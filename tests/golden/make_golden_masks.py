"""Golden vectors for the diffusion masks of the reference's make_diffuser_features (abx/model/features.py:130-212, eval branch)
on hand-built anchor layouts that the two shipped complexes do not reach: a closing anchor on the LAST antibody
residue with and without an antigen behind it (the structure-loss window is clipped with the TOTAL length, features.py:167), a row
with an odd number of anchors of the chosen CDR (the reference pairs a FLAT row-major list of anchor positions, features.py:159-161:
an unpaired anchor of a single row is ignored, and in a batch the pairing runs across rows), and generate_area = 'cdr' with several
CDRs.  Run in the build container only:   python tests/golden/make_golden_masks.py   ->  tests/golden/masks_cases.npz"""
import json
import os
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

cfg_json = json.load(open('/root/reference/config/config_model.json'))
cfg_json['diffuser']['so3']['use_cached_score'] = True
from ref_shims import ConfigDict  # noqa: E402
from diffuser.full_diffuser import FullDiffuser  # noqa: E402

FullDiffuser.get(ConfigDict(cfg_json['diffuser']))          # the singleton must be created from the attribute dict first
from abx.model import features as ref_features  # noqa: E402



def case(anchor_rows, Ltot, area):
    """anchor_rows: (B, Lab) int anchor flags.  Identity frames, alanine everywhere: only the masks are recorded."""
    af = torch.tensor(anchor_rows, dtype=torch.int32)
    B, Lab = af.shape
    rots = torch.eye(3)[None, None, None].expand(B, Ltot, 8, 3, 3).contiguous()
    trans = torch.zeros(B, Ltot, 8, 3)
    batch = dict(seq=torch.zeros(B, Ltot, dtype=torch.int64), mask=torch.ones(B, Ltot, dtype=torch.bool), anchor_flag=af,
                 rigidgroups_gt_frames=(rots, trans))
    torch.manual_seed(0)
    # (@take1st: the decorated function takes everything but the batch and returns the transform)
    out = ref_features.make_diffuser_features(generate_area=area, diff_conf=cfg_json['diffuser'], is_training=False)(batch)
    return dict(anchor_flag=af.numpy(), Ltot=np.int64(Ltot), fixed_mask=out['fixed_mask'].numpy(), struc_loss_mask=out['struc_loss_mask'].numpy())


H3 = 5          # residue_constants.cdr_str_to_enum['H3']
row = lambda Lab, marks: [marks.get(i, 0) for i in range(Lab)]
cases = {
    # closing anchor on the last antibody residue, antigen behind it / no antigen
    'last_anchor_antigen': (case([row(12, {4: H3, 11: H3})], 16, 'H3'), 'H3'),
    'last_anchor_no_antigen': (case([row(12, {4: H3, 11: H3})], 12, 'H3'), 'H3'),
    # odd anchor count: one row / three identical rows (flat pairing across rows)
    'odd_single_row': (case([row(14, {2: H3, 6: H3, 10: H3})], 18, 'H3'), 'H3'),
    'odd_three_rows': (case([row(14, {2: H3, 6: H3, 10: H3})] * 3, 18, 'H3'), 'H3'),
    # two complexes with different anchor columns, two CDRs each, generate_area = 'cdr'
    'cdr_all': (case([row(16, {1: 4, 4: 4, 8: 5, 13: 5}), row(16, {2: 4, 6: 4, 9: 5, 12: 5})], 20, 'cdr'), 'cdr'),
}
flat = {}
for name, (d, area) in cases.items():
    for k, v in d.items():
        flat[f'{name}.{k}'] = v
    flat[f'{name}.area'] = np.array(area)
path = os.path.join(HERE, 'masks_cases.npz')
np.savez_compressed(path, **flat)
print('wrote', path, os.path.getsize(path), 'bytes;', {n: d['fixed_mask'].tolist() for n, (d, _) in cases.items()})
