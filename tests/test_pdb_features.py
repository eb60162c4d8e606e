"""Raw-PDB featurisation on the two complexes the reference ships: the plain-text PDB reader, the landmark
IMGT locator and the antigen patch / crop / collate steps, against vectors produced by the reference's own
get_structure_label_npz / Patch_Around_Anchor / collate_fn / FeatureBuilder (tests/golden/make_golden_pdb.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, tt

PDB = {'6ct7': os.path.join(GOLDEN, 'pdb', '6ct7_H_L_S.pdb'), '6qd7': os.path.join(GOLDEN, 'pdb', '6qd7_X_Z_F|E.pdb')}


def test_pdb_reader_and_imgt_locator():
    from abx_amd.data import antibody as A
    from abx_amd.io.pdb_reader import read_pdb, chain_feature
    assert A.parse_pdb_name(PDB['6qd7']) == ('6qd7_X_Z_F|E', '6qd7', 'X', 'Z', ['F', 'E'])
    ch = read_pdb(PDB['6ct7'])
    assert list(ch) == ['L', 'H', 'S']                      # order of appearance in the file
    f = chain_feature(ch['H'])
    assert len(f['str_seq']) == 214 and f['coords'].shape == (214, 14, 3) and f['str_seq'].startswith('EVQLVESGGG')
    assert int(f['coord_mask'].sum()) == 1582               # heavy atoms of the standard residues, hydrogens / waters dropped
    # CDR-H3 strings pinned by SURVEY.md §0 fact 3; the other CDRs follow the IMGT definitions
    expect = {'6ct7': (dict(cdr1='GFDFEKAW', cdr2='IKSTADGGTT', cdr3='TSAH'), dict(cdr1='ALPMQF', cdr2='KDS', cdr3='QSPDSTNTYEV'), 113, 108),
              '6qd7': (dict(cdr1='GFPLRDYA', cdr2='IGGNDNAA', cdr3='AKSVRLSRPSPFDL'), dict(cdr1='QSVSTY', cdr2='EAS', cdr3='QQRASWPLT'), 120, 107)}
    for code, (h, l, nh, nl) in expect.items():
        name, _, hc, lc, ag = A.parse_pdb_name(PDB[code])
        st = A.make_pdb_features(PDB[code], hc, lc, ag)
        assert st['cdrs'] == [h, l], (code, st['cdrs'])
        assert int((st['antibody_chain_ids'] == 0).sum()) == nh and int((st['antibody_chain_ids'] == 1).sum()) == nl
        cd = st['antibody_cdr_def']
        assert cd.min() == 0 and cd.max() == 13 and np.all(np.diff(cd) >= 0)          # fr1 .. fr4 of H then of L, in order
        assert int((cd == 5).sum()) == len(h['cdr3']) and int((cd == 12).sum()) == len(l['cdr3'])
        assert st['antibody_residx'][nh] == 512 and st['antibody_residx'][nh - 1] == nh - 1
    with pytest.raises(ValueError):
        A.locate_variable_domain('MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQAPILSRVGDGTQDNLSGAEKAVQVKVKALPDAQFEVV', 'H')


@pytest.mark.parametrize('code', ['6ct7', '6qd7'])
def test_complex_pipeline_matches_reference(code, oracle_diffuser):
    """load_complex (centre, antigen patch within 16 A of the CDR anchors, 32-residue window, collate) and the feature pipeline
    against the reference's own functions run on the same parsed arrays."""
    from abx_amd import features
    from abx_amd.data import antibody as A
    g = load_npz(f'pdb_{code}.npz')
    b = A.load_complex(PDB[code], seed=int(g['seed']))
    for k in ('seq', 'mask', 'atom14_gt_exists', 'cdr_def', 'chain_id', 'residx', 'anchor_flag'):
        assert np.array_equal(b[k].numpy(), g['batch.' + k]), k
    assert np.array_equal(b['atom14_gt_positions'].numpy(), g['batch.atom14_gt_positions'])
    assert b['str_heavy_seq'][0] == str(g['batch.str_heavy_seq']) and b['str_light_seq'][0] == str(g['batch.str_light_seq'])
    assert b['antigen_origin_str_seq'][0] == str(g['batch.antigen_origin_str_seq'])
    assert np.array_equal(b['antigen_origin_residx'][0], g['batch.antigen_origin_residx'])
    L = b['seq'].shape[1]
    assert L == {'6ct7': 231, '6qd7': 259}[code] and (L - b['anchor_flag'].shape[1]) == {'6ct7': 10, '6qd7': 32}[code]
    noise = {k[6:]: tt(v) for k, v in g.items() if k.startswith('noise.')}
    raw = {k: v for k, v in b.items() if torch.is_tensor(v)}
    out = features.build_features(raw, oracle_diffuser, generate_area='H3', noise=noise)
    assert np.array_equal(out['fixed_mask'].numpy(), g['feat.fixed_mask']) and np.array_equal(out['seq_t'].numpy(), g['feat.seq_t'])
    assert int((1 - out['fixed_mask']).sum()) == {'6ct7': 3, '6qd7': 13}[code]
    h3 = str(g['cdr_h3'])
    dif = (1 - out['fixed_mask'][0]).nonzero().reshape(-1).tolist()
    assert b['str_heavy_seq'][0][dif[0]:dif[-1] + 2] == h3                          # the last CDR residue stays fixed (features.py:166)
    for k, tol in (('rigids_0', 2e-6), ('torsion_angles_sin_cos', 2e-6), ('atom37_gt_positions', 0), ('pseudo_beta', 0)):
        assert np.abs(out[k].numpy() - g['feat.' + k]).max() <= tol, k
    d = np.abs(out['rigids_t'].numpy() - g['feat.rigids_t'])
    assert d.max() <= 1e-5 + 1e-6 * np.abs(g['feat.rigids_t']).max(), d.max()


def test_writer_output_round_trips_through_the_reader(tmp_path):
    """PDB writer (8f-2) -> PDB reader (8f-1): sequences, atom14 coordinates (3 decimals) and the antigen chains of a complex
    survive the round trip; multi-antigen names ('F|E') are split on '|' as design.py:152 does."""
    from abx_amd.data import antibody as A
    from abx_amd.io import postprocess_trajectory
    from abx_amd.io.pdb_reader import read_pdb, chain_feature
    b = A.load_complex(PDB['6qd7'], seed=0)
    Lab = b['anchor_flag'].shape[1]
    rec = {'seq': b['seq'][:, :Lab], 'atom14_results': b['atom14_gt_positions'][:, :Lab], 'pLDDT': torch.full((1, Lab), 77.0), 'time': 0.01}
    meta = {k: list(b[k]) for k in ('name', 'str_heavy_seq', 'str_light_seq', 'antigen_origin_str_seq', 'antigen_origin_atom14_gt_positions',
                                    'antigen_origin_atom14_gt_exists', 'antigen_origin_chain_ids')}
    files = postprocess_trajectory(meta, [rec], str(tmp_path))
    ch = read_pdb(files[0])
    assert list(ch) == ['X', 'Z', 'F', 'E']
    fx = chain_feature(ch['X'])
    assert fx['str_seq'] == b['str_heavy_seq'][0] and chain_feature(ch['Z'])['str_seq'] == b['str_light_seq'][0]
    nh = len(fx['str_seq'])
    m = b['atom14_gt_exists'][0, :nh].numpy()
    # atoms the input lacks are written at the origin by the writer (the reference writes every atom14 slot of the residue type)
    assert np.abs(fx['coords'][m] - b['atom14_gt_positions'][0, :nh].numpy()[m]).max() < 6e-4
    ag = chain_feature(ch['F'])['str_seq'] + chain_feature(ch['E'])['str_seq']
    assert ag == b['antigen_origin_str_seq'][0]


def test_ab_metrics_rmsd_and_aar():
    """calc_ab_metrics (ab_utils.py:124-167): a rigidly moved copy of the 6qd7 antibody has zero RMSD everywhere and AAR 1; a
    perturbed CDR-H3 with two mutations shows up in heavy_cdr3 only."""
    from abx_amd.data import antibody as A
    from abx_amd import metrics
    b = A.load_complex(PDB['6qd7'], seed=0)
    Lab = b['anchor_flag'].shape[1]
    ca = b['atom14_gt_positions'][0, :Lab, 1].numpy().astype(np.float64)
    cdr = b['cdr_def'][0, :Lab].numpy()
    seq = b['str_heavy_seq'][0] + b['str_light_seq'][0]
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    moved = ca @ R.T + np.array([5.0, -3.0, 11.0])
    m = metrics.calc_ab_metrics(ca, moved, cdr, seq, seq)
    assert list(m)[:4] == ['heavy_cdr1_AAR', 'heavy_cdr1_RMSD', 'heavy_cdr2_AAR', 'heavy_cdr2_RMSD']
    assert all(abs(v) < 1e-9 for k, v in m.items() if k.endswith('RMSD')) and all(v == 1.0 for k, v in m.items() if k.endswith('AAR'))
    h3 = np.nonzero(cdr == 5)[0]
    pert = moved.copy()
    pert[h3[5]] += np.array([3.0, 0.0, 0.0])
    mut = list(seq)
    mut[h3[5]] = 'W' if seq[h3[5]] != 'W' else 'A'
    mut[h3[0]] = 'W' if seq[h3[0]] != 'W' else 'A'
    m2 = metrics.calc_ab_metrics(ca, pert, cdr, seq, ''.join(mut))
    assert m2['heavy_cdr3_AAR'] == pytest.approx(1 - 2 / len(h3)) and m2['heavy_cdr3_Loop_AAR'] == pytest.approx(1 - 1 / (len(h3) - 6))
    assert 0.5 < m2['heavy_cdr3_RMSD'] < 1.0 and m2['light_cdr3_RMSD'] < 0.1 and m2['heavy_cdr1_AAR'] == 1.0


@pytest.mark.parametrize('code', ['6ct7', '6qd7'])
def test_npz_entry_of_a_name_idx_list_matches_reference(code):
    """The --name_idx / --data_dir path of the reference's inference.py (abx/data/dataset.py:90-214): <data_dir>/<name>.npz in the
    make_pdb_npz schema -> the collated batch.  Against the reference's own get_structure_label_npz / Patch_Around_Anchor / collate_fn
    outputs on the same arrays (pdb_<code>.npz 'batch.*') and against the raw-PDB route."""
    from abx_amd.data import antibody as A
    g = load_npz(f'pdb_{code}.npz')
    name = os.path.basename(PDB[code])[:-4]
    data_dir = os.path.join(GOLDEN, 'npz')
    with np.load(os.path.join(data_dir, name + '.npz')) as z:
        assert set(z.files) == set(A.NPZ_KEYS)                                     # the schema of make_pdb_npz
        for k in A.NPZ_KEYS:
            assert np.array_equal(z[k], g['struc.' + k]), k
    b = A.load_complex_npz(data_dir, name, seed=int(g['seed']))
    for k in ('seq', 'mask', 'atom14_gt_exists', 'cdr_def', 'chain_id', 'residx', 'anchor_flag', 'atom14_gt_positions'):
        assert np.array_equal(b[k].numpy(), g['batch.' + k]), k
    assert b['name'] == (name,) and b['str_heavy_seq'][0] == str(g['batch.str_heavy_seq'])
    assert b['antigen_origin_str_seq'][0] == str(g['batch.antigen_origin_str_seq'])
    p = A.load_complex(PDB[code], seed=int(g['seed']))
    for k, v in b.items():
        if torch.is_tensor(v):
            assert torch.equal(v, p[k]), k
    with pytest.raises(ValueError, match='make_pdb_npz'):
        np.savez(os.path.join('/tmp', 'abx_bad_entry.npz'), foo=np.zeros(3))
        A.load_complex_npz('/tmp', 'abx_bad_entry')


def test_model_features_json():
    """read_model_features: generate_area / optimize_steps of the reference's feature-pipeline JSON (config_data_feature.json layout)."""
    import json
    from abx_amd import design
    path = '/tmp/abx_feats.json'
    json.dump([["make_to_device", {"fields": ["seq"], "device": "%(device)s"}], ["make_gt_frames", {}],
               ["make_diffuser_features", {"generate_area": "H3", "optimize_steps": [4, 8]}]], open(path, 'w'))
    assert design.read_model_features(path) == ('H3', [4, 8])


def test_pdb_reader_error_context_and_resname_disorder(tmp_path):
    """malformed records name the file and line; two residue names at one position (point-mutation disorder) are
    not merged: like Biopython's DisorderedResidue (every atom line selects the child of its residue name) the name of the LAST atom
    line of the position is the one that is kept, whatever the occupancies."""
    from abx_amd.io.pdb_reader import read_pdb, chain_feature, PdbFormatError
    fmt = lambda i, name, resn, alt, x, occ: f"ATOM  {i:5d}  {name:<3s}{alt}{resn} A   5    {x:8.3f}{0.0:8.3f}{0.0:8.3f}{occ:6.2f}{10.0:6.2f}           {name[0]:>2s}  \n"
    p = tmp_path / 'dis.pdb'
    p.write_text(fmt(1, 'N', 'SER', 'A', 1.0, 0.3) + fmt(2, 'CA', 'SER', 'A', 2.0, 0.3) + fmt(3, 'OG', 'SER', 'A', 3.0, 0.3) +
                 fmt(4, 'N', 'ALA', 'B', 1.1, 0.7) + fmt(5, 'CA', 'ALA', 'B', 2.1, 0.7) + fmt(6, 'CB', 'ALA', 'B', 3.1, 0.7))
    ch = read_pdb(str(p))
    f = chain_feature(ch['A'])
    assert f['str_seq'] == 'A' and abs(float(f['coords'][0, 1, 0]) - 2.1) < 1e-6 and int(f['coord_mask'].sum()) == 3
    # the last name read wins also when it has the LOWER occupancy, and when the two names alternate line by line
    p2 = tmp_path / 'dis2.pdb'
    p2.write_text(fmt(1, 'N', 'ALA', 'A', 1.1, 0.7) + fmt(2, 'CA', 'ALA', 'A', 2.1, 0.7) + fmt(3, 'CB', 'ALA', 'A', 3.1, 0.7) +
                  fmt(4, 'N', 'SER', 'B', 1.0, 0.3) + fmt(5, 'CA', 'SER', 'B', 2.0, 0.3) + fmt(6, 'OG', 'SER', 'B', 3.0, 0.3))
    f2 = chain_feature(read_pdb(str(p2))['A'])
    assert f2['str_seq'] == 'S' and abs(float(f2['coords'][0, 1, 0]) - 2.0) < 1e-6 and int(f2['coord_mask'].sum()) == 3
    p3 = tmp_path / 'dis3.pdb'
    p3.write_text(fmt(1, 'N', 'SER', 'A', 1.0, 0.5) + fmt(2, 'N', 'ALA', 'B', 1.1, 0.5) + fmt(3, 'CA', 'SER', 'A', 2.0, 0.5) + fmt(4, 'CA', 'ALA', 'B', 2.1, 0.5))
    assert chain_feature(read_pdb(str(p3))['A'])['str_seq'] == 'A'
    bad = tmp_path / 'bad.pdb'
    bad.write_text(fmt(1, 'N', 'SER', ' ', 1.0, 1.0) + 'ATOM      2  CA  SER A   x       1.000\n')
    with pytest.raises(PdbFormatError, match=r'bad\.pdb:2'):
        read_pdb(str(bad))
