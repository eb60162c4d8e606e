"""Dump the numeric residue tables the hot path reads from the reference's
`abx/common/residue_constants.py` into `abx_amd/data/residue_tables.npz`.

These are DATA (AlphaFold-style literature atom positions / index maps), not source: the build consumes them
as constant memory for the torsion->frames->atom14 kernel and the feature transforms.
Run in this container only:  python tests/golden/make_tables.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shims  # noqa: E402

ref_shims.install()
from abx.common import residue_constants as rc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'abx_amd', 'data', 'residue_tables.npz')

tables = dict(
    restype_atom14_to_atom37=rc.restype_atom14_to_atom37,                    # (21,14) i32
    restype_atom37_to_atom14=rc.restype_atom37_to_atom14,                    # (21,37) i32
    restype_atom14_mask=rc.restype_atom14_mask,                              # (21,14) bool
    restype_atom37_mask=rc.restype_atom37_mask,                              # (21,37) bool
    restype_atom14_to_rigid_group=rc.restype_atom14_to_rigid_group,          # (21,14) i32
    restype_atom14_rigid_group_positions=rc.restype_atom14_rigid_group_positions,  # (21,14,3) f32
    restype_rigid_group_default_frame=rc.restype_rigid_group_default_frame,  # (21,8,4,4) f32
    restype_rigidgroup_mask=rc.restype_rigidgroup_mask,                      # (21,8) bool
    restype_rigidgroup_base_atom37_idx=rc.restype_rigidgroup_base_atom37_idx,  # (21,8,3)
    restype_rigidgroup_is_ambiguous=rc.restype_rigidgroup_is_ambiguous,
    restype_rigidgroup_rots=rc.restype_rigidgroup_rots,
    restype_ambiguous_atoms_swap_index=rc.restype_ambiguous_atoms_swap_index,
    restype_atom14_is_ambiguous=rc.restype_atom14_is_ambiguous,
    chi_angles_atom_indices=np.asarray(rc.chi_angles_atom_indices, dtype=np.int64),  # (21,4,4)
    chi_angles_mask=np.asarray(rc.chi_angles_mask, dtype=np.float32),        # (21,4)
    chi_pi_periodic=np.asarray(rc.chi_pi_periodic, dtype=np.float32),        # (21,4)
)
for k, v in tables.items():
    print(k, np.asarray(v).shape, np.asarray(v).dtype)
np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in tables.items()})
print('wrote', os.path.abspath(OUT), os.path.getsize(OUT), 'bytes')
print('restypes', rc.restypes, 'atom_types', rc.atom_types)
print('cdr_str_to_enum', rc.cdr_str_to_enum, rc.num_ab_regions, rc.residue_chain_index_offset)
