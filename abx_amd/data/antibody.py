"""Raw-PDB featurisation of an antibody-antigen complex without Biopython / ANARCI: the input side of
`design.py --pdb_file` (reference design.py:318-322 -> abx/data/dataset.py:286-464 `IgStructureData`).

Pipeline, each step citing what it restates:
  parse_pdb_name            dataset.py:290-293          '6ct7_H_L_S.pdb' -> code, heavy / light chain ids, antigen chain ids ('F|E')
  locate_variable_domain    preprocess/numbering.py:45-109 (ANARCI + IMGT region table) -> motif-based IMGT locator, see below
  make_pdb_features         preprocess/make_ab_data_from_mmcif.py:107-191 (make_pdb_npz, merge_chains)
  structure_labels          dataset.py:311-381 (get_structure_label_npz: tensors, centring on the antibody CA centre)
  patch_around_anchor       dataset.py:32-42,497-551 (antigen residues within 16 A of a CDR anchor, +-5 residues)
  crop_antigen              dataset.py:300-309,469-495 (at most 32 antigen residues; the reference draws the window with
                            `random`, here from a seeded generator)
  collate_single            dataset.py:383-464 (collate_fn for one complex)

IMGT locator.  ANARCI (HMM alignment) is not available; the variable domain and its CDRs are located from the conserved
landmarks of the IMGT numbering instead: Cys23, Trp41, Cys104 and the J-region motif [WF]-G-x-G at 118-121.
  CDR1 = 27..38  = (Cys23 + 4) .. (Trp41 - 3)          FR2 = 39..55 = 17 residues, no gaps in practice
  CDR2 = 56..65  = (Trp41 + 15) .. (Cys104 - n_FR3)    FR3 = 66..104 holds 38 residues in heavy chains (gap at 73) and 36 in
                                                       kappa / lambda chains (gaps at 73, 81, 82), Cys104 included
  CDR3 = 105..117 = (Cys104 + 1) .. (J motif - 1)      FR4 = 118..128: 11 residues (heavy) / 10 (light: 118..127)
Region codes as numbering.py:66-79: fr1 0, cdr1 1, fr2 2, cdr2 3, fr3 4, cdr3 5, fr4 6 (+7 for the light chain).
PARITY: pinned on the two complexes the reference ships (CDR-H3 of 6ct7 = TSAH, of 6qd7 = AKSVRLSRPSPFDL, SURVEY.md §0 fact 3,
all six CDRs checked by eye against the IMGT definitions); an unusual germline can differ from ANARCI by a residue at the
CDR1 / CDR2 edges.  Everything downstream of the locator is pinned against the reference's own functions
(tests/golden/make_golden_pdb.py feeds these arrays through the reference's Patch_Around_Anchor / collate_fn / FeatureBuilder).
"""
import os
import random
import re

import numpy as np
import torch

from .. import residue_constants as rc
from ..io.pdb_reader import chain_feature, read_pdb

_J_MOTIF = re.compile(r'(?=([WF]G.G))')


def parse_pdb_name(pdb_file):
    """'.../6qd7_X_Z_F|E.pdb' -> ('6qd7_X_Z_F|E', '6qd7', 'X', 'Z', ['F', 'E'])   (dataset.py:290-293, data/utils.py:54-58)."""
    name = os.path.basename(pdb_file).split('.')[0]
    parts = name.split('_')
    code, chain_ids = parts[0], parts[1:]
    if len(chain_ids) != 3:
        raise ValueError(f'{pdb_file}: expected <code>_<heavy>_<light>_<antigen chains joined by |>.pdb')
    heavy, light, antigen = chain_ids
    if heavy.islower() and heavy.upper() == light:
        heavy = heavy.upper()
    elif light.islower() and light.upper() == heavy:
        light = light.upper()
    return name, code, heavy, light, [s.replace(' ', '') for s in antigen.split('|')]


def locate_variable_domain(seq, chain_type):
    """Landmark-based IMGT regions of one antibody chain.  chain_type 'H' | 'L'.
    -> dict(start, end (exclusive), cdr_def (end - start,) int, landmarks)."""
    heavy = chain_type == 'H'
    j = c104 = None
    for m in _J_MOTIF.finditer(seq):
        cand = [i for i in range(max(0, m.start() - 35), m.start() - 2) if seq[i] == 'C']
        if cand and m.start() >= 60:
            j, c104 = m.start(), cand[-1]
            break
    if j is None:
        raise ValueError(f'no Cys104 ... [WF]GxG landmark in the {chain_type} chain: not an antibody variable domain?')
    c23 = [i for i in range(max(0, c104 - 90), max(0, c104 - 54)) if seq[i] == 'C']
    if not c23:
        raise ValueError(f'no Cys23 landmark in the {chain_type} chain')
    c23 = min(c23, key=lambda i: abs((c104 - i) - (75 if heavy else 65)))
    w41 = [i for i in range(c23 + 10, min(c23 + 21, c104)) if seq[i] == 'W']
    if not w41:
        raise ValueError(f'no Trp41 landmark in the {chain_type} chain')
    # the conserved Trp is followed by x-[RKQ]-Q / x-Q-[QH] (WVRQ, WYQQ, WYQH ...): prefer such a one, else the last in the window
    good = [i for i in w41 if i + 3 < len(seq) and (seq[i + 3] in 'QH' or seq[i + 2] in 'RQ')]
    w41 = (good or w41)[-1]
    start = max(0, c23 - 22)
    end = min(len(seq), j + (11 if heavy else 10))
    cdr1 = (c23 + 4, w41 - 3)
    cdr2_first = w41 + 15
    cdr2_last = c104 - (38 if heavy else 36)
    cdr2_last = min(max(cdr2_last, cdr2_first), cdr2_first + 11)
    cdr3 = (c104 + 1, j - 1)
    region = np.full(len(seq), -1, dtype=np.int64)
    region[start:cdr1[0]] = 0
    region[cdr1[0]:cdr1[1] + 1] = 1
    region[cdr1[1] + 1:cdr2_first] = 2
    region[cdr2_first:cdr2_last + 1] = 3
    region[cdr2_last + 1:c104 + 1] = 4
    region[cdr3[0]:cdr3[1] + 1] = 5
    region[j:end] = 6
    region = region[start:end] + (0 if heavy else 7)
    return dict(start=start, end=end, cdr_def=region, landmarks=dict(C23=c23, W41=w41, C104=c104, J=j),
                cdrs=dict(cdr1=seq[cdr1[0]:cdr1[1] + 1], cdr2=seq[cdr2_first:cdr2_last + 1], cdr3=seq[cdr3[0]:cdr3[1] + 1]))


def _merge_chains(features, antibody):
    """merge_chains (make_ab_data_from_mmcif.py:107-143)."""
    prefix = 'antibody' if antibody else 'antigen'
    chain_ids, residx, cdr_def = [], [], []
    for i, d in enumerate(features):
        n = len(d['str_seq'])
        chain_ids.append(np.full(n, i + (0 if antibody else 2)))
        r = np.arange(0, n)
        if antibody and i > 0:
            r = r + rc.residue_chain_index_offset
        residx.append(r)
        cdr_def.append(d['cdr_def'] if antibody else np.full(n, 14))
    out = dict(str_seq=''.join(d['str_seq'] for d in features),
               coords=np.concatenate([d['coords'] for d in features], axis=0),
               coord_mask=np.concatenate([d['coord_mask'] for d in features], axis=0),
               chain_ids=np.concatenate(chain_ids), residx=np.concatenate(residx), cdr_def=np.concatenate(cdr_def))
    return {f'{prefix}_{k}': v for k, v in out.items()}


def make_pdb_features(pdb_file, heavy_chain_id, light_chain_id, antigen_chain_ids):
    """make_pdb_npz (make_ab_data_from_mmcif.py:145-191): antibody chains cut to their variable domains with IMGT region labels,
    antigen chains whole.  -> dict of numpy arrays antibody_* / antigen_*."""
    chains = read_pdb(pdb_file)
    feats = {}
    ab = []
    for cid, ctype in ((heavy_chain_id, 'H'), (light_chain_id, 'L')):
        if not cid:
            continue
        if cid not in chains:
            raise ValueError(f'{pdb_file}: chain {cid} not found (chains: {list(chains)})')
        f = chain_feature(chains[cid])
        dom = locate_variable_domain(f['str_seq'], ctype)
        s, e = dom['start'], dom['end']
        ab.append(dict(str_seq=f['str_seq'][s:e], coords=f['coords'][s:e], coord_mask=f['coord_mask'][s:e], cdr_def=dom['cdr_def'],
                       cdrs=dom['cdrs']))
    feats.update(_merge_chains(ab, antibody=True))
    feats['cdrs'] = [a['cdrs'] for a in ab]
    ag = [chain_feature(chains[c]) for c in antigen_chain_ids if c in chains]
    if ag:
        feats.update(_merge_chains(ag, antibody=False))
    return feats


def _str_seq_to_index(s):
    return [rc.restype_order.get(a, rc.unk_restype_index) for a in s]


def structure_labels(struc, name):
    """get_structure_label_npz (dataset.py:311-381), scale_factor 1: torch tensors, coordinates centred on the antibody CA centre."""
    t = torch.from_numpy
    ab_coords = t(struc['antibody_coords'].astype(np.float32))
    ab_mask = t(struc['antibody_coord_mask'])
    ab_chain = t(struc['antibody_chain_ids'])
    heavy_len = int((ab_chain == 0).sum())
    ab_str = str(struc['antibody_str_seq'])
    ag_coords = t(struc.get('antigen_coords', np.zeros((0, 14, 3), dtype=np.float32)).astype(np.float32))
    ag_mask = t(struc.get('antigen_coord_mask', np.zeros((0, 14), dtype=bool)))
    ag_str = str(struc.get('antigen_str_seq', ''))
    ca = rc.atom_order['CA']
    centre = torch.sum(ab_coords[:, ca], dim=0) / (torch.sum(ab_mask[:, ca], dim=0, keepdim=True) + 1e-5)
    ab_coords = (ab_coords - centre[None, None, :]) * ab_mask[..., None]
    ag_coords = (ag_coords - centre[None, None, :]) * ag_mask[..., None]
    return dict(
        name=name,
        antibody_seq=torch.tensor(_str_seq_to_index(ab_str), dtype=torch.int64), antibody_residx=t(struc['antibody_residx']),
        antibody_mask=torch.ones_like(ab_chain, dtype=torch.bool), str_heavy_seq=ab_str[:heavy_len], str_light_seq=ab_str[heavy_len:],
        antibody_atom14_gt_positions=ab_coords, antibody_atom14_gt_exists=ab_mask, antibody_cdr_def=t(struc['antibody_cdr_def']),
        antibody_chain_ids=ab_chain,
        antigen_atom14_gt_positions=ag_coords, antigen_atom14_gt_exists=ag_mask, antigen_str_seq=ag_str,
        antigen_seq=torch.tensor(_str_seq_to_index(ag_str), dtype=torch.int64), antigen_mask=torch.ones(len(ag_str), dtype=torch.bool),
        antigen_chain_ids=t(struc.get('antigen_chain_ids', np.zeros((0,), dtype=np.int64))),
        antigen_residx=t(struc.get('antigen_residx', np.zeros((0,), dtype=np.int64))),
        antigen_cdr_def=t(struc.get('antigen_cdr_def', np.zeros((0,), dtype=np.int64))))


def _patch_idx(a, b, mask_a, mask_b, threshold):
    """dataset.py:32-42: residues of a with an atom closer than `threshold` to an atom of b, widened by -5 .. +4 residues."""
    diff = a[:, None, :, None, :] - b[None, :, None, :, :]
    mask = mask_a[:, None, :, None] * mask_b[None, :, None, :]
    dist = torch.where(mask, torch.norm(diff, dim=-1), torch.tensor(1e10))
    dist = dist.reshape(a.shape[0], b.shape[0], -1).min(dim=2)[0]
    near = torch.nonzero(dist.min(dim=1)[0] < threshold).reshape(-1).tolist()
    return sorted({i for j in near for i in range(j - 5, j + 5)})


def patch_around_anchor(data, distance_threshold=16.0):
    """Patch_Around_Anchor (dataset.py:497-551), inference branch: anchors = the residues flanking each CDR; the antigen keeps
    the residues near an anchor.  Quirk kept: the 'has a CA' filter is `torch.nonzero` of the (N, 3) CA coordinates flattened,
    i.e. the set of residues with a non-zero CA coordinate UNION the column indices {0, 1, 2} that occur."""
    cdr_def = data['antibody_cdr_def']
    anchor_flag = torch.zeros_like(cdr_def)
    idx = []
    n_ab = data['antibody_seq'].shape[0]
    for sele in ('H1', 'H2', 'H3', 'L1', 'L2', 'L3'):
        flag = cdr_def == rc.cdr_str_to_enum[sele]
        if not bool(flag.any()):
            continue
        pos = torch.arange(flag.shape[0])[flag]
        left, right = max(0, int(pos.min()) - 1), min(int(pos.max()) + 1, n_ab - 1)
        anchor_flag[left] = rc.cdr_str_to_enum[sele]
        anchor_flag[right] = rc.cdr_str_to_enum[sele]
        idx.extend(_patch_idx(data['antigen_atom14_gt_positions'], data['antibody_atom14_gt_positions'][[left, right]],
                              data['antigen_atom14_gt_exists'], data['antibody_atom14_gt_exists'][[left, right]], distance_threshold))
    ca = data['antigen_atom14_gt_positions'][:, rc.atom_order['CA']]
    mask_idx = set(torch.nonzero(ca).reshape(-1).tolist())
    keep = sorted(set(idx).intersection(mask_idx))
    origin = {f'antigen_origin_{k}': data[f'antigen_{k}'] for k in ('atom14_gt_positions', 'atom14_gt_exists', 'str_seq', 'residx', 'chain_ids')}
    out = dict(data)
    out['anchor_flag'] = anchor_flag
    for k in ('atom14_gt_positions', 'atom14_gt_exists', 'residx', 'chain_ids', 'seq', 'cdr_def', 'mask'):
        out['antigen_' + k] = data['antigen_' + k][keep]
    out['antigen_str_seq'] = ''.join(data['antigen_str_seq'][i] for i in keep)
    # (the reference fills antigen_origin_* from the ALREADY patched entries, dataset.py:541-547; the whole antigen is kept in
    # antigen_full_* for the output writer)
    out.update({f'antigen_origin_{k}': out[f'antigen_{k}'] for k in ('atom14_gt_positions', 'atom14_gt_exists', 'str_seq', 'residx', 'chain_ids')})
    out.update({k.replace('origin', 'full'): v for k, v in origin.items()})
    return out if keep else None


def crop_antigen(ret, max_antigen_seq_len=32, rng=None):
    """IgStructureData.__iter__ + sample_with_struc (dataset.py:300-309,469-495): a window of at most 32 antigen residues."""
    n = len(ret.get('antigen_str_seq', ''))
    if n <= max_antigen_seq_len:
        return ret
    rng = rng or random.Random(0)
    struc_mask = ret['antigen_atom14_gt_exists'][:, 1]
    num = int(struc_mask.sum())
    if 0 < num < n:
        s0, s1 = 0, n
        while s0 < n and not bool(struc_mask[s0]):
            s0 += 1
        while s1 > 0 and not bool(struc_mask[s1 - 1]):
            s1 -= 1
        if s1 - s0 > max_antigen_seq_len:
            start = rng.randint(s0, s1 - max_antigen_seq_len)
        else:
            extra = max_antigen_seq_len - (s1 - s0)
            start = rng.randint(s0 - extra // 2 - 10, s1 + extra // 2 + 10)
            start = max(0, start)
            if start + max_antigen_seq_len > n:
                start = n - max_antigen_seq_len
    else:
        start = rng.randint(0, n - max_antigen_seq_len)
    end = start + max_antigen_seq_len
    out = dict(ret)
    for k, v in ret.items():
        if 'antigen' in k and 'origin' not in k and 'full' not in k:
            out[k] = v[start:end]
    return out


def collate_single(ret):
    """collate_fn (dataset.py:383-464) for one complex: antibody ++ antigen, batch dimension 1."""
    cat = lambda a, b: torch.cat([ret['antibody_' + a], ret['antigen_' + b]], dim=0)[None]
    out = dict(
        name=(ret['name'],), str_heavy_seq=(ret['str_heavy_seq'],), str_light_seq=(ret['str_light_seq'],),
        seq=cat('seq', 'seq'), mask=cat('mask', 'mask'), atom14_gt_positions=cat('atom14_gt_positions', 'atom14_gt_positions'),
        atom14_gt_exists=cat('atom14_gt_exists', 'atom14_gt_exists'), cdr_def=cat('cdr_def', 'cdr_def'),
        chain_id=cat('chain_ids', 'chain_ids'), residx=cat('residx', 'residx'), anchor_flag=ret['anchor_flag'][None])
    for k in ('str_seq', 'atom14_gt_positions', 'atom14_gt_exists', 'chain_ids', 'residx'):
        v = ret['antigen_origin_' + k]
        out['antigen_origin_' + k] = (v if isinstance(v, str) else v.cpu().numpy(),)
    return out


NPZ_KEYS = ('antibody_str_seq', 'antibody_coords', 'antibody_coord_mask', 'antibody_chain_ids', 'antibody_residx', 'antibody_cdr_def',
            'antigen_str_seq', 'antigen_coords', 'antigen_coord_mask', 'antigen_chain_ids', 'antigen_residx', 'antigen_cdr_def')


def save_struc_npz(struc, path):
    """Write the chain features in the schema of the reference's preprocessed data set (`make_pdb_npz`,
    abx/preprocess/make_ab_data_from_mmcif.py:145-191: one <name>.npz per complex under --data_dir)."""
    np.savez(path, **{k: np.asarray(struc[k]) for k in NPZ_KEYS if k in struc})


def _finish(struc, name, max_antigen_seq_len, seed, source):
    ret = patch_around_anchor(structure_labels(struc, name))
    if ret is None:
        raise ValueError(f'{source}: no antigen residue within 16 A of a CDR anchor')
    ret = crop_antigen(ret, max_antigen_seq_len, random.Random(seed))
    return collate_single(ret)


def load_complex_npz(data_dir, name, max_antigen_seq_len=32, seed=0):
    """`dataset.load` for ONE entry of a --name_idx list (abx/data/dataset.py:90-214,554-571): <data_dir>/<name>.npz in the
    reference's `make_pdb_npz` schema -> the collated batch (B = 1), through the same centring / antigen patch / window crop /
    collate as load_complex.  BASELINE configs 3 / 4 (diffab_test.idx, the RAbD list) are lists of such entries."""
    path = os.path.join(data_dir, name + '.npz')
    with np.load(path) as z:
        struc = {k: z[k] for k in z.files}
    missing = [k for k in ('antibody_str_seq', 'antibody_coords', 'antibody_coord_mask', 'antibody_chain_ids', 'antibody_residx',
                           'antibody_cdr_def') if k not in struc]
    if missing:
        raise ValueError(f'{path}: not a make_pdb_npz file (missing {missing})')
    return _finish(struc, name, max_antigen_seq_len, seed, path)


def load_complex(pdb_file, max_antigen_seq_len=32, seed=0):
    """`dataset.load_single` for one PDB file named <code>_<H>_<L>_<antigen chains>.pdb -> the collated batch (B = 1) that
    abx_amd.features.build_features consumes, plus `meta` for the PDB writer."""
    name, code, heavy, light, antigens = parse_pdb_name(pdb_file)
    struc = make_pdb_features(pdb_file, heavy, light, antigens)
    batch = _finish(struc, name, max_antigen_seq_len, seed, pdb_file)
    batch['cdrs'] = struc['cdrs']
    return batch
