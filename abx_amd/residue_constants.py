"""Numeric residue tables used on the hot path (data dumped by tests/golden/make_tables.py from the
reference's abx/common/residue_constants.py:215-377; file `data/residue_tables.npz`)."""
import os

import numpy as np

_T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'residue_tables.npz'))

restypes = ['A', 'R', 'N', 'D', 'C', 'Q', 'E', 'G', 'H', 'I', 'L', 'K', 'M', 'F', 'P', 'S', 'T', 'W', 'Y', 'V']
restypes_with_x = restypes + ['X']
restype_order = {r: i for i, r in enumerate(restypes)}
restype_num = 20
unk_restype_index = 20
num_ab_regions = 14                 # reference residue_constants.py:11
residue_chain_index_offset = 512    # reference residue_constants.py:12
cdr_str_to_enum = {'H1': 1, 'H2': 3, 'H3': 5, 'L1': 8, 'L2': 10, 'L3': 12}   # residue_constants.py:14-21
atom_types = ['N', 'CA', 'C', 'CB', 'O', 'CG', 'CG1', 'CG2', 'OG', 'OG1', 'SG', 'CD', 'CD1', 'CD2', 'ND1', 'ND2',
              'OD1', 'OD2', 'SD', 'CE', 'CE1', 'CE2', 'CE3', 'NE', 'NE1', 'NE2', 'OE1', 'OE2', 'CH2', 'NH1', 'NH2',
              'OH', 'CZ', 'CZ2', 'CZ3', 'NZ', 'OXT']
atom_order = {a: i for i, a in enumerate(atom_types)}

restype_atom14_to_atom37 = _T['restype_atom14_to_atom37']
restype_atom37_to_atom14 = _T['restype_atom37_to_atom14']
restype_atom14_mask = _T['restype_atom14_mask']
restype_atom37_mask = _T['restype_atom37_mask']
restype_atom14_to_rigid_group = _T['restype_atom14_to_rigid_group']
restype_atom14_rigid_group_positions = _T['restype_atom14_rigid_group_positions']
restype_rigid_group_default_frame = _T['restype_rigid_group_default_frame']
restype_rigidgroup_mask = _T['restype_rigidgroup_mask']
restype_rigidgroup_base_atom37_idx = _T['restype_rigidgroup_base_atom37_idx']
restype_rigidgroup_is_ambiguous = _T['restype_rigidgroup_is_ambiguous']
restype_rigidgroup_rots = _T['restype_rigidgroup_rots']
restype_ambiguous_atoms_swap_index = _T['restype_ambiguous_atoms_swap_index']
restype_atom14_is_ambiguous = _T['restype_atom14_is_ambiguous']
chi_angles_atom_indices = _T['chi_angles_atom_indices']
chi_angles_mask = _T['chi_angles_mask']
chi_pi_periodic = _T['chi_pi_periodic']

# ---- names (output writer only; abx/common/residue_constants.py: restype_1to3, restype_name_to_atom14_names) -------------
atom_type_num = len(atom_types)
restype_1to3 = {'A': 'ALA', 'R': 'ARG', 'N': 'ASN', 'D': 'ASP', 'C': 'CYS', 'Q': 'GLN', 'E': 'GLU', 'G': 'GLY', 'H': 'HIS',
                'I': 'ILE', 'L': 'LEU', 'K': 'LYS', 'M': 'MET', 'F': 'PHE', 'P': 'PRO', 'S': 'SER', 'T': 'THR', 'W': 'TRP',
                'Y': 'TYR', 'V': 'VAL'}
restype_name_to_atom14_names = {
    restype_1to3[r]: [atom_types[int(restype_atom14_to_atom37[i, j])] if restype_atom14_mask[i, j] else '' for j in range(14)]
    for i, r in enumerate(restypes)}
restype_name_to_atom14_names['UNK'] = [''] * 14
