"""Seeded synthetic inputs for tests and bench: a collated antibody-antigen batch in the
layout the reference's `IgStructureData.collate_fn` produces (abx/data/dataset.py:206-283) and a seeded
re-randomisation of the 190 ScoreNetwork parameters (SURVEY.md §8c "mandatory oracle hygiene": the default
init zero-fills every 'final'/'gate' layer, so an untouched model is an identity frame update).

No external data: there are no trained checkpoints / npz datasets in this environment.
"""
from collections import OrderedDict

import numpy as np
import torch

from abx_amd import residue_constants as rc

_EMBED_NAMES = ('proj_aa_type', 'aatype_embed', 'cdr_embed', 'aa_pair_embed', 'relpos_embed',
                'aapair_to_distcoef', 'dgram_embed', 'proj_rel_pos', 'proj_prev_pos')


def random_state_dict(shapes, seed=0, dtype=torch.float32):
    """shapes: ordered mapping name -> shape.  One CPU generator per tensor (seed + index) so that the
    values do not depend on which other tensors exist."""
    out = OrderedDict()
    for idx, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(int(seed) * 100003 + idx)
        shape = tuple(shape)
        z = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = name.rsplit('.', 1)[-1]
        if name.endswith('trainable_point_weights'):
            v = float(np.log(np.e - 1.0)) + 0.3 * z          # softplus^-1(1) + noise
        elif len(shape) == 2 and any(e in name for e in _EMBED_NAMES):
            v = 0.5 * z
        elif len(shape) == 2:
            v = z / float(np.sqrt(shape[1]))
            if 'affine_update' in name:
                v = 0.2 * v
        elif leaf == 'weight':                               # LayerNorm gamma
            v = 1.0 + 0.1 * z
        else:                                                # biases, LayerNorm beta
            v = 0.1 * z
            if 'affine_update' in name:
                v = 0.2 * v
        out[name] = v.to(dtype)
    return out


def make_complex(L_heavy=120, L_light=108, L_antigen=28, cdr=(97, 109), seed=1, n_masked_tail=0):
    """One synthetic complex (un-batched tensors, CPU).

    cdr=(first,last): inclusive index range of the CDR-H3-like segment inside the heavy chain; anchors sit at
    first-1 / last+1 (reference Patch_Around_Anchor, dataset.py:505-508).
    n_masked_tail: number of trailing antigen residues that are padding (mask False, no atoms).
    """
    g = torch.Generator().manual_seed(int(seed))
    Lab = L_heavy + L_light
    L = Lab + L_antigen
    seq = torch.randint(0, 20, (L,), generator=g, dtype=torch.int64)
    # C-alpha random walk per chain, step N(0, 2.2^2) per coordinate, atoms = CA + N(0, 1.5^2)
    steps = 2.2 * torch.randn((L, 3), generator=g)
    ca = torch.cumsum(steps, dim=0)
    ca = ca - ca[:Lab].mean(dim=0, keepdim=True)
    atom14 = ca[:, None, :] + 1.5 * torch.randn((L, 14, 3), generator=g)
    atom14[:, 1] = ca
    exists = torch.from_numpy(rc.restype_atom14_mask)[seq].clone()          # (L,14) bool
    chain_id = torch.cat([torch.zeros(L_heavy), torch.ones(L_light), 2 * torch.ones(L_antigen)]).to(torch.int32)
    residx = torch.cat([torch.arange(L_heavy), torch.arange(L_light) + rc.residue_chain_index_offset,
                        torch.arange(L_antigen)]).to(torch.int32)
    cdr_def = torch.zeros(L, dtype=torch.int32)
    first, last = cdr
    cdr_def[:first] = 4
    cdr_def[first:last + 1] = rc.cdr_str_to_enum['H3']
    cdr_def[last + 1:L_heavy] = 6
    cdr_def[L_heavy:Lab] = 7
    cdr_def[Lab:] = rc.num_ab_regions
    anchor_flag = torch.zeros(Lab, dtype=torch.int32)
    anchor_flag[max(0, first - 1)] = rc.cdr_str_to_enum['H3']
    anchor_flag[min(last + 1, Lab - 1)] = rc.cdr_str_to_enum['H3']
    mask = torch.ones(L, dtype=torch.bool)
    if n_masked_tail > 0:
        mask[L - n_masked_tail:] = False
        exists[L - n_masked_tail:] = False
        seq[L - n_masked_tail:] = rc.unk_restype_index
        cdr_def[L - n_masked_tail:] = 0
        chain_id[L - n_masked_tail:] = 0
        residx[L - n_masked_tail:] = 0
    atom14 = atom14 * exists[..., None]
    return dict(seq=seq, mask=mask, atom14_gt_positions=atom14.float(), atom14_gt_exists=exists,
                cdr_def=cdr_def, chain_id=chain_id, residx=residx, anchor_flag=anchor_flag)


def collate(complexes):
    """Stack equally-shaped complexes into the collated batch dict (dataset.py:258-270)."""
    keys = ('seq', 'mask', 'atom14_gt_positions', 'atom14_gt_exists', 'cdr_def', 'chain_id', 'residx', 'anchor_flag')
    return {k: torch.stack([c[k] for c in complexes], dim=0) for k in keys}


def replicate(complex_, n):
    """Batch of n identical copies of one complex (n samples of one design task, inference.py:369-373)."""
    return {k: v[None].expand(n, *v.shape).clone() for k, v in complex_.items()}


# Named workloads.  Lab = 228 = 120 H + 108 L.
WORKLOADS = {
    'L352': dict(L_heavy=120, L_light=108, L_antigen=124, cdr=(97, 109)),   # BASELINE nominal ~350 residues
    'L256': dict(L_heavy=120, L_light=108, L_antigen=28, cdr=(97, 109)),    # realistic cropped complex
    '6ct7like': dict(L_heavy=113, L_light=107, L_antigen=10, cdr=(96, 99)),  # L = 230, 3 diffused residues (BASELINE configs 1-2)
    '6qd7like': dict(L_heavy=120, L_light=109, L_antigen=32, cdr=(96, 109)),  # L = 261, 13 diffused residues (BASELINE config 5)
    'tiny': dict(L_heavy=10, L_light=6, L_antigen=4, cdr=(4, 8)),
}
