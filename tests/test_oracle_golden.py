"""Pins the oracle (oracle/abx_oracle.py) and the host feature pipeline against vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_npz, tt, feat_batch_from_golden
from oracle import abx_oracle as O


def close(a, b, atol, rtol=0.0, name=''):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b.astype(np.float64))
    assert np.all(err <= tol), f'{name}: max err {err.max():.3e} (tol {atol}+{rtol}*|x|), max|x| {np.abs(b).max():.3e}'


def test_features_match_reference(oracle_diffuser):
    from abx_amd import features, synthetic
    g = load_npz('feat_tiny.npz')
    raw = {k[4:]: tt(v) for k, v in g.items() if k.startswith('raw.')}
    # the synthetic builder is deterministic: the committed raw batch is what it produces today
    w = synthetic.WORKLOADS['tiny']
    again = synthetic.collate([synthetic.make_complex(seed=11, **w), synthetic.make_complex(seed=12, n_masked_tail=1, **w)])
    for k in raw:
        assert torch.equal(raw[k], again[k]), k
    noise = {k[6:]: tt(v) for k, v in g.items() if k.startswith('noise.')}
    out = features.build_features(dict(raw), oracle_diffuser, generate_area='H3', noise=noise)
    for k in ('atom14_atom_exists', 'residx_atom37_to_atom14', 'atom37_atom_exists', 'atom37_gt_exists',
              'rigidgroups_gt_exists', 'torsion_angles_mask', 'fixed_mask', 'struc_loss_mask', 'seq_t'):
        assert np.array_equal(out[k].numpy(), g['feat.' + k]), k
    close(out['atom37_gt_positions'], g['feat.atom37_gt_positions'], 0, name='atom37')
    close(out['rigidgroups_gt_frames'][0], g['feat.rigidgroups_gt_frames.0'], 2e-6, name='frames R')
    close(out['rigidgroups_gt_frames'][1], g['feat.rigidgroups_gt_frames.1'], 0, name='frames t')
    close(out['torsion_angles_sin_cos'], g['feat.torsion_angles_sin_cos'], 2e-6, name='torsions')
    close(out['pseudo_beta'], g['feat.pseudo_beta'], 0, name='pseudo_beta')
    close(out['rigids_0'], g['feat.rigids_0'], 2e-6, name='rigids_0')
    close(out['rigids_t'], g['feat.rigids_t'], 1e-5, rtol=1e-6, name='rigids_t')
    close(out['t'], g['feat.t'], 0, name='t')
    assert out['rigids_t'].dtype == torch.float32 and out['seq_t'].dtype == torch.int64


def test_igso3_tables(oracle_diffuser):
    g = load_npz('igso3_small.npz')
    so3 = oracle_diffuser.so3
    close(so3.discrete_sigma, g['big_sigma'], 0, name='sigma grid')
    close(so3.discrete_omega, g['big_omega'], 0, name='omega grid')
    i, j = g['spot_i'], g['spot_j']
    close(so3._pdf[i, j], g['spot_pdf'], 1e-5, 1e-4, 'pdf spots')
    close(so3._cdf[i, j], g['spot_cdf'], 1e-5, 1e-4, 'cdf spots')
    rows = g['rows']
    # score norms are a quotient of two cancelling 1000-term series: compare with an absolute floor
    close(so3._score_norms[rows], g['rows_score_norms'], 2e-2, 1e-3, 'score_norm rows')
    close(so3._score_scaling, g['big_score_scaling'], 1e-4, 1e-4, 'score scaling')
    small = O.OracleSO3(dict(min_sigma=0.1, max_sigma=1.5, num_sigma=40, num_omega=40))
    close(small._pdf, g['small_pdf'], 1e-5, 1e-4, 'small pdf')
    close(small._cdf, g['small_cdf'], 1e-5, 1e-4, 'small cdf')
    close(small._score_norms, g['small_score_norms'], 2e-2, 1e-3, 'small score norms')


@pytest.fixture(scope='module')
def pinned_diffuser(oracle_diffuser):
    """Oracle diffuser whose score-norm/cdf rows used by the tiny tests are the REFERENCE's rows, so that the
    piecewise-constant lookup (SURVEY §7 hard part 1) is compared on identical tables."""
    g = load_npz('igso3_small.npz')
    so3 = oracle_diffuser.so3
    so3._score_norms = so3._score_norms.clone()
    so3._cdf = so3._cdf.clone()
    so3._score_norms[g['rows']] = tt(g['rows_score_norms'])
    so3._cdf[g['rows']] = tt(g['rows_cdf'])
    return oracle_diffuser


def _state(m, feat):
    b = dict(feat)
    for k in ('seq_t', 'rigids_t', 't', 'prev_pos', 'prev_seq', 'prev_pair', 'rot_score_scaling', 'trans_score_scaling'):
        b[k] = tt(m['in.' + k])
    return b


def test_modules_match_reference(params, cfg, pinned_diffuser):
    m = load_npz('modules_tiny.npz')
    feat = feat_batch_from_golden(load_npz('feat_tiny.npz'))
    p = params
    # final-pass inputs as the reference had them (after two recycles)
    b = _state(m, feat)
    b.update(seq_t=tt(m['final.seq_t_after']), prev_pos=tt(m['final.prev_pos_in']), prev_seq=tt(m['final.prev_seq_in']),
             prev_pair=tt(m['final.prev_pair_in']))
    mask = b['mask']
    with torch.no_grad():
        close(O.residue_embedding(p, b), m['enc_residue.out'], 2e-5, 1e-5, 'residue embedding')
        close(O.pair_embedding(p, b, dict(cfg.model.embeddings_and_seqformer.prev_pos)), m['enc_pair.out'], 2e-5, 1e-5,
              'pair embedding')
        seq, pair = tt(m['block.seq_in']), tt(m['block.pair_in'])
        s2, p2 = O.embed_and_seqformer(p, b, cfg)
        d = O.seq_attention(p, seq, pair, mask)
        close(d, m['seq_attn.out'], 2e-5, 1e-5, 'seq_attn')
        seq = seq + tt(m['seq_attn.out'])
        close(O.transition(p, O.P_BLK + 'seq_transition', seq), m['seq_transition.out'], 2e-5, 1e-5, 'seq_transition')
        seq = seq + tt(m['seq_transition.out'])
        close(O.outer_product_mean(p, seq, mask), m['opm.out'], 2e-5, 1e-5, 'opm')
        pair = pair + tt(m['opm.out'])
        close(O.triangle_multiplication(p, 'triangle_multiplication_outgoing', pair, mask, True), m['trimul_out.out'],
              5e-5, 1e-5, 'trimul out')
        pair = pair + tt(m['trimul_out.out'])
        close(O.triangle_multiplication(p, 'triangle_multiplication_incoming', pair, mask, False), m['trimul_in.out'],
              5e-5, 1e-5, 'trimul in')
        pair = pair + tt(m['trimul_in.out'])
        close(O.triangle_attention(p, 'triangle_attention_starting_node', pair, mask, True), m['triattn_start.out'],
              2e-5, 1e-5, 'triattn start')
        pair = pair + tt(m['triattn_start.out'])
        close(O.triangle_attention(p, 'triangle_attention_ending_node', pair, mask, False), m['triattn_end.out'],
              2e-5, 1e-5, 'triattn end')
        pair = pair + tt(m['triattn_end.out'])
        close(O.transition(p, O.P_BLK + 'pair_transition', pair), m['pair_transition.out'], 2e-5, 1e-5, 'pair_transition')
        close(s2, m['out.seq'], 1e-4, 1e-5, 'trunk seq')
        close(p2, m['out.pair'], 2e-4, 1e-5, 'trunk pair')
        ipa = O.ipa_attention(p, tt(m['ipa0.in_1d']), tt(m['ipa0.in_2d']), mask.float(), tt(m['ipa0.rots']),
                              tt(m['ipa0.trans']), cfg.model.heads.diffusion_module.IPA)
        close(ipa, m['ipa0.out'], 5e-5, 1e-5, 'ipa layer 0')


def test_full_call_matches_reference(params, cfg, pinned_diffuser):
    """One in-loop ScoreNetwork call (2 recycles + final, fp64 t) incl. in-place batch mutation and get_prev."""
    m = load_npz('modules_tiny.npz')
    b = _state(m, feat_batch_from_golden(load_npz('feat_tiny.npz')))
    assert b['t'].dtype == torch.float64
    ret = O.score_network(params, b, cfg, pinned_diffuser)
    f = ret['heads']['folding']
    assert torch.equal(b['seq_t'], tt(m['final.seq_t_after']))
    assert torch.equal(ret['heads']['sequence_module']['seq_0'], tt(m['out.seq_0']))
    close(f['rigids'], m['out.rigids'], 2e-4, 1e-5, 'rigids')
    close(f['representations']['structure_module'], m['out.structure_module'], 2e-4, 1e-5, 'structure_module')
    close(f['angles_sin_cos'], m['out.angles'], 2e-4, 0, 'angles')
    close(f['final_atom14_positions'], m['out.atom14'], 5e-4, 1e-5, 'atom14')
    close(f['final_atom_positions'], m['out.atom37'], 5e-4, 1e-5, 'atom37')
    close(ret['heads']['sequence_module']['logits'], m['out.logits'], 2e-4, 1e-5, 'logits')
    close(ret['heads']['predicted_lddt']['pLDDT'], m['out.pLDDT'], 2e-3, 1e-5, 'pLDDT')
    assert f['trans_score'].dtype == torch.float64 and f['rot_score'].dtype == torch.float32
    close(f['trans_score'], m['out.trans_score'], 2e-4, 1e-5, 'trans_score')
    # rot_score is a piecewise-constant table lookup: allow the (rare) one-bucket disagreement
    rs, rs_ref = f['rot_score'].numpy(), m['out.rot_score']
    bad = np.abs(rs - rs_ref) > (2e-4 + 1e-4 * np.abs(rs_ref))
    assert bad.reshape(-1, 3).any(axis=1).mean() <= 0.02, f'rot_score mismatches {bad.mean()}'
    prev = O.get_prev(b, ret, cfg)
    assert (prev['prev_pos'].numpy() != m['out.prev_pos']).mean() < 1e-3


def test_warmup_call_fp32_t(params, cfg, pinned_diffuser):
    """The self-conditioning warm-up call sees fp32 t and no prev_* (inference.py:209-211)."""
    m = load_npz('modules_tiny.npz')
    b = feat_batch_from_golden(load_npz('feat_tiny.npz'))
    ones = torch.ones(b['seq'].shape[0])
    b = O.set_t_feats(b, pinned_diffuser, np.linspace(0.01, 1.0, 100)[::-1][0], ones)
    assert b['t'].dtype == torch.float32
    ret = O.score_network(params, b, cfg, pinned_diffuser)
    assert ret['heads']['folding']['trans_score'].dtype == torch.float32
    close(ret['heads']['folding']['rigids'], m['warm.rigids'], 2e-4, 1e-5, 'warm rigids')
    close(ret['heads']['sequence_module']['logits'], m['warm.logits'], 2e-4, 1e-5, 'warm logits')
    assert torch.equal(ret['heads']['sequence_module']['seq_0'], tt(m['warm.seq_0']))
    close(ret['heads']['folding']['trans_score'], m['warm.trans_score'], 2e-4, 1e-4, 'warm trans_score')
    b.update(O.get_prev(b, ret, cfg))
    assert (b['prev_pos'].numpy() != m['warm.prev_pos']).mean() < 1e-3


def test_reverse_step_matches_reference(cfg, pinned_diffuser):
    s = load_npz('step_tiny.npz')
    dm = tt(s['diffuse_mask'])
    for i in range(3):
        g = lambda k: tt(s[f's{i}.{k}'])
        noise = dict(z_rot=g('z_rot'), z_trans=g('z_trans'), jumps=g('jumps'))
        rig, seq = pinned_diffuser.reverse(rigid_t=g('rigid_in'), seq_t=g('seq_in'), rot_score=g('rot_score'),
                                           trans_score=g('trans_score'), logits_t=g('logits'), t=g('t'), dt=tt(s['dt']),
                                           diffuse_mask=dm, noise=noise)
        assert rig.dtype == torch.float64 and seq.dtype == torch.int64
        assert torch.equal(seq, g('seq_out')), f'step {i}: tokens differ'
        close(rig, s[f's{i}.rigid_out'], 1e-9, 1e-9, f'step {i} rigids (fp64)')
        rs, ts = pinned_diffuser.score_scaling(g('t'))
        close(rs, s[f's{i}.rot_score_scaling'], 1e-4, 1e-4, 'rot score scaling')
        close(ts, s[f's{i}.trans_score_scaling'], 1e-12, 1e-12, 'trans score scaling')
        # Poisson rates: the closed-form transition matrix vs the reference's fp32 eigen-decomposition
        rates, _ = pinned_diffuser.seq.reverse_rates(g('seq_in'), g('logits'), g('t'))
        assert torch.all(rates >= 0)


def test_short_trajectory_matches_reference(params, cfg, pinned_diffuser):
    """The reference's own sample_fn (num_t=4, trajectory mode) vs the oracle driver under injected noise."""
    tj = load_npz('traj_tiny.npz')
    b = feat_batch_from_golden(load_npz('feat_tiny.npz'))

    def noise_fn(k):
        return dict(z_rot=tt(tj[f'n{k}.z_rot']), z_trans=tt(tj[f'n{k}.z_trans']), jumps=tt(tj[f'n{k}.jumps']))

    traj = O.sample_fn(params, b, cfg, pinned_diffuser, mode='trajectory', num_t=4, noise_fn=noise_fn)
    assert len(traj) == 4
    for k, d in enumerate(traj):
        assert float(d['time']) == float(tj[f'k{k}.time'])
        same = (d['seq'].numpy() == tj[f'k{k}.seq']).mean()
        assert same == 1.0, f'step {k}: token agreement {same}'
        close(d['atom14_results'], tj[f'k{k}.atom14'], 5e-3, 1e-4, f'step {k} atom14')
        close(d['pLDDT'], tj[f'k{k}.pLDDT'], 1e-2, 1e-4, f'step {k} pLDDT')
    close(traj[-1]['rigids_t'], tj['final.rigids_t'], 2e-3, 1e-4, 'final rigids')


def test_optimize_mode_matches_reference(params, cfg, pinned_diffuser):
    """BASELINE config 4 shape of the path: forward_marginal noising at t = opt_step/100 (incl. the x_tilde token jump) and the
    shortened reverse loop of the reference's sample_fn(mode='optimize'), under the recorded draws."""
    from abx_amd import features
    g = load_npz('optimize_tiny.npz')
    f = load_npz('feat_tiny.npz')
    raw = {k[4:]: tt(v) for k, v in f.items() if k.startswith('raw.')}
    noise = {k[6:]: tt(v) for k, v in g.items() if k.startswith('noise.')}
    b = features.build_features(dict(raw), pinned_diffuser, generate_area='H3', opt_step=4, noise=noise)
    assert torch.equal(b['seq_t'], tt(g['feat.seq_t'])) and torch.equal(b['fixed_mask'], tt(g['feat.fixed_mask']))
    close(b['t'], g['feat.t'], 0, name='t')
    close(b['rigids_t'], g['feat.rigids_t'], 2e-5, 1e-6, 'optimize rigids_t')
    close(b['trans_score'], g['feat.trans_score'], 2e-5, 1e-5, 'optimize trans_score')
    close(b['trans_score_scaling'], g['feat.trans_score_scaling'], 1e-6, 1e-6, 'trans score scaling')

    def noise_fn(k):
        return dict(z_rot=tt(g[f'n{k}.z_rot']), z_trans=tt(g[f'n{k}.z_trans']), jumps=tt(g[f'n{k}.jumps']))

    traj = O.sample_fn(params, b, cfg, pinned_diffuser, mode='optimize', num_t=100, noise_fn=noise_fn)
    assert len(traj) == 4 and float(traj[-1]['time']) == float(g['last.time'])
    assert np.array_equal(traj[-1]['seq'].numpy(), g['last.seq'])
    close(traj[-1]['atom14_results'], g['last.atom14'], 5e-3, 1e-4, 'optimize atom14')
    close(traj[-1]['rigids_t'], g['final.rigids_t'], 2e-3, 1e-4, 'optimize final rigids')


# ---------------------------------------------------------------------------------------------------------------------
# Beyond the tiny complex: L = 48 with a padded tail, and the bench's L = 256 / 352 complexes
# ---------------------------------------------------------------------------------------------------------------------
def test_features_match_reference_L48(oracle_diffuser):
    from abx_amd import features, synthetic
    g = load_npz('feat_L48.npz')
    raw = {k[4:]: tt(v) for k, v in g.items() if k.startswith('raw.')}
    w = dict(L_heavy=20, L_light=16, L_antigen=12, cdr=(10, 16))
    again = synthetic.collate([synthetic.make_complex(seed=21, **w), synthetic.make_complex(seed=22, n_masked_tail=3, **w)])
    for k in raw:
        assert torch.equal(raw[k], again[k]), k
    noise = {k[6:]: tt(v) for k, v in g.items() if k.startswith('noise.')}
    out = features.build_features(dict(raw), oracle_diffuser, generate_area='H3', noise=noise)
    for k in ('atom14_atom_exists', 'residx_atom37_to_atom14', 'atom37_atom_exists', 'atom37_gt_exists',
              'rigidgroups_gt_exists', 'torsion_angles_mask', 'fixed_mask', 'struc_loss_mask', 'seq_t'):
        assert np.array_equal(out[k].numpy(), g['feat.' + k]), k
    close(out['atom37_gt_positions'], g['feat.atom37_gt_positions'], 0, name='atom37')
    close(out['rigidgroups_gt_frames'][0], g['feat.rigidgroups_gt_frames.0'], 2e-6, name='frames R')
    close(out['torsion_angles_sin_cos'], g['feat.torsion_angles_sin_cos'], 2e-6, name='torsions')
    close(out['rigids_0'], g['feat.rigids_0'], 2e-6, name='rigids_0')
    close(out['rigids_t'], g['feat.rigids_t'], 1e-5, rtol=1e-6, name='rigids_t')
    assert int((1 - out['fixed_mask']).sum()) == 2 * 6           # [anchor+1, anchor_r-1): the last CDR residue stays fixed


def _l48_state(m):
    b = feat_batch_from_golden(m)
    for k in ('seq_t', 'rigids_t', 't', 'rot_score_scaling', 'trans_score_scaling'):
        b[k] = tt(m['in.' + k])
    return b


def test_modules_match_reference_L48(params, cfg, pinned_diffuser):
    """Per-module parity at L = 48 with a 3-residue padded antigen tail (mask False): the final pass of one call.  Pair-shaped
    outputs are compared on the stored [::3, ::3] sub-grid and through their full-tensor sums."""
    m = load_npz('modules_L48.npz')
    S = int(m['sub'])
    p = params
    b = _l48_state(m)
    b.update(seq_t=tt(m['pass.seq_t']), prev_pos=tt(m['pass.prev_pos']).long(), prev_seq=tt(m['pass.prev_seq']),
             prev_pair=tt(m['pass.prev_pair']))
    mask = b['mask']
    assert int((~mask).sum()) == 3

    def pclose(x, key, atol, rtol, name):
        close(x[:, ::S, ::S], m[key], atol, rtol, name)
        assert abs(float(x.double().sum()) - float(m[key + '.sum'])) <= 2e-5 * float(m[key + '.abssum']) + 1e-3, name + ' sum'

    with torch.no_grad():
        close(O.residue_embedding(p, b), m['enc_residue.out'], 2e-5, 1e-5, 'residue embedding')
        pclose(O.pair_embedding(p, b, dict(cfg.model.embeddings_and_seqformer.prev_pos)), 'enc_pair.out', 2e-5, 1e-5, 'pair embedding')
        seq, pair = tt(m['block.seq_in']), tt(m['block.pair_in'])
        d = O.seq_attention(p, seq, pair, mask)
        close(d, m['seq_attn.out'], 2e-5, 1e-5, 'seq_attn')
        seq = seq + tt(m['seq_attn.out'])
        close(O.transition(p, O.P_BLK + 'seq_transition', seq), m['seq_transition.out'], 2e-5, 1e-5, 'seq_transition')
        seq = seq + tt(m['seq_transition.out'])
        for name, fn in (('opm', lambda z: O.outer_product_mean(p, seq, mask)),
                         ('trimul_out', lambda z: O.triangle_multiplication(p, 'triangle_multiplication_outgoing', z, mask, True)),
                         ('trimul_in', lambda z: O.triangle_multiplication(p, 'triangle_multiplication_incoming', z, mask, False)),
                         ('triattn_start', lambda z: O.triangle_attention(p, 'triangle_attention_starting_node', z, mask, True)),
                         ('triattn_end', lambda z: O.triangle_attention(p, 'triangle_attention_ending_node', z, mask, False)),
                         ('pair_transition', lambda z: O.transition(p, O.P_BLK + 'pair_transition', z))):
            upd = fn(pair)                   # the running pair tensor is the oracle's own (only sub-grids are stored)
            pclose(upd, name + '.out', 1e-4, 2e-5, name)
            pair = pair + upd
        pclose(pair, 'out.pair', 3e-4, 2e-5, 'trunk pair')
        ipa = O.ipa_attention(p, tt(m['ipa0.in_1d']), tt(m['ipa0.in_2d']), mask.float(), tt(m['ipa0.rots']),
                              tt(m['ipa0.trans']), cfg.model.heads.diffusion_module.IPA)
        close(ipa, m['ipa0.out'], 5e-5, 1e-5, 'ipa layer 0')


def _check_call(ret, m, fixed, pair_key, S, tol_scale=1.0):
    f = ret['heads']['folding']
    assert torch.equal(ret['heads']['sequence_module']['seq_0'].cpu(), tt(m['out.seq_0']))
    g = lambda x: x.detach().cpu()
    close(g(f['rigids']), m['out.rigids'], 2e-4 * tol_scale, 1e-5, 'rigids')
    close(g(f['final_atom14_positions']), m['out.atom14'], 5e-4 * tol_scale, 1e-5, 'atom14')
    close(g(ret['heads']['sequence_module']['logits']), m['out.logits'], 2e-4 * tol_scale, 1e-5, 'logits')
    close(g(ret['heads']['predicted_lddt']['pLDDT']), m['out.pLDDT'], 2e-3 * tol_scale, 1e-5, 'pLDDT')
    close(g(f['trans_score']), m['out.trans_score'], 2e-4 * tol_scale, 1e-5, 'trans_score')
    close(g(ret['representations']['seq']), m['out.seq'], 2e-4 * tol_scale, 1e-5, 'trunk seq')
    pr = g(ret['representations']['pair'])
    close(pr[:, ::S, ::S], m[pair_key], 3e-4 * tol_scale, 2e-5, 'trunk pair (sub-grid)')
    assert abs(float(pr.double().sum()) - float(m['out.pair.sum'])) <= 2e-5 * float(m['out.pair.abssum']), 'trunk pair sum'
    rs, ref = g(f['rot_score']).numpy(), m['out.rot_score']
    dif = (np.asarray(fixed) == 0).reshape(-1)
    bad = (np.abs(rs - ref) > 2e-4 + 1e-4 * np.abs(ref)).reshape(-1, 3).any(axis=1)[dif].mean()
    assert bad <= 0.1, f'rot_score bucket mismatches {bad}'


def test_full_call_matches_reference_L48(params, cfg, pinned_diffuser):
    """One in-loop call from a zero self-conditioning state at L = 48, padded tail: oracle vs the reference's outputs."""
    m = load_npz('modules_L48.npz')
    b = _l48_state(m)
    ret = O.score_network(params, b, cfg, pinned_diffuser)
    assert torch.equal(b['seq_t'], tt(m['final.seq_t_after']))
    _check_call(ret, m, b['fixed_mask'], 'out.pair', int(m['sub']))
    prev = O.get_prev(b, ret, cfg)
    assert (prev['prev_pos'].numpy() != m['out.prev_pos']).mean() < 1e-3


@pytest.mark.parametrize('name', ['L256', 'L352'])
def test_large_shape_digest(params, cfg, pinned_diffuser, name):
    """The bench's synthetic complexes at L = 256 / 352 (B = 1): host feature pipeline vs the reference's, then one in-loop call of
    the oracle vs the reference's recorded outputs — the oracle is pinned at the benchmark's own size, not only at L = 20."""
    from abx_amd import features
    from conftest import digest_batch
    g = load_npz(f'{name}_digest.npz')
    b, mine = digest_batch(g, name, pinned_diffuser, features.build_features)
    assert torch.equal(mine['seq_t'], tt(g['feat.seq_t'])) and torch.equal(mine['fixed_mask'], tt(g['feat.fixed_mask']))
    close(mine['rigids_t'], g['feat.rigids_t'], 2e-5, 1e-6, 'rigids_t')
    close(mine['torsion_angles_sin_cos'], g['feat.torsion_angles_sin_cos'], 2e-6, 0, 'torsions')
    close(mine['rigids_0'], g['feat.rigids_0'], 2e-6, 1e-6, 'rigids_0')
    ret = O.score_network(params, b, cfg, pinned_diffuser)
    assert torch.equal(b['seq_t'], tt(g['final.seq_t_after']))
    _check_call(ret, g, b['fixed_mask'], 'out.pair_sub', int(g['pair_sub']))
    prev = O.get_prev(b, ret, cfg)
    assert (prev['prev_pos'].numpy() != g['out.prev_pos']).mean() < 1e-4


def test_esm_hook_matches_reference(esm_setup, pinned_diffuser):
    """ESM2 embedding hook (8f-3): the reference run with esm.enabled and its ESM module replaced by a seeded stand-in tensor;
    the oracle mixes the 37 layers with softmax(esm_embed_weights) and projects as seqformer.py:185-191."""
    from conftest import esm_tensor
    cfg, params = esm_setup
    g = load_npz('esm_tiny.npz')
    b = feat_batch_from_golden(load_npz('feat_tiny.npz'))
    for k in ('seq_t', 'rigids_t', 't', 'rot_score_scaling', 'trans_score_scaling'):
        b[k] = tt(g['in.' + k])
    b['esm_embed'] = esm_tensor(g, b['seq'].shape[0], b['anchor_flag'].shape[1])
    ret = O.score_network(params, b, cfg, pinned_diffuser)
    f = ret['heads']['folding']
    assert torch.equal(ret['heads']['sequence_module']['seq_0'], tt(g['out.seq_0'])) and torch.equal(b['seq_t'], tt(g['final.seq_t_after']))
    close(ret['representations']['seq'], g['out.seq'], 2e-4, 1e-5, 'trunk seq')
    close(ret['representations']['pair'], g['out.pair'], 2e-4, 1e-5, 'trunk pair')
    close(f['rigids'], g['out.rigids'], 2e-4, 1e-5, 'rigids')
    close(ret['heads']['sequence_module']['logits'], g['out.logits'], 2e-4, 1e-5, 'logits')
    close(f['final_atom14_positions'], g['out.atom14'], 5e-4, 1e-5, 'atom14')
    # and the hook matters: without the ESM term the trunk output is different
    cfg0 = type(cfg)(__import__('copy').deepcopy(dict(cfg)))
    cfg0.model.embeddings_and_seqformer.esm.enabled = False
    b2 = feat_batch_from_golden(load_npz('feat_tiny.npz'))
    for k in ('seq_t', 'rigids_t', 't', 'rot_score_scaling', 'trans_score_scaling'):
        b2[k] = tt(g['in.' + k])
    r0 = O.score_network(params, b2, cfg0, pinned_diffuser)
    assert float((r0['representations']['seq'] - tt(g['out.seq'])).abs().max()) > 1e-2


# ---------------------------------------------------------------------------------------------------------------------------
# token reverse rates and the inverse-cdf Poisson sampler (discrete_diffuser.py:130-190)
# ---------------------------------------------------------------------------------------------------------------------------
def test_token_reverse_rates_match_reference_recorded_poisson_argument(cfg):
    """rates_tiny.npz holds the ARGUMENT of the reference's torch.poisson call (reverse_rates * dt) at t in {1, 0.5, 0.02},
    dt in {0.01, 0.1}, tokens 0 / 19 / out of range, peaked and flat logits.  The literal (fp32 eigh) route reproduces it to
    2e-6; the closed-form route (what the HIP kernel evaluates) differs only through the reference's own eigh rounding."""
    z = load_npz('rates_tiny.npz')
    x_t, lg = tt(z['x_t']), tt(z['logits'])
    lit = O.OracleSeq({'rate_const': float(z['rate_const'])}, eigh=True)
    closed = O.OracleSeq({'rate_const': float(z['rate_const'])})
    for c in z['cases']:
        t, dt, ref = torch.tensor(float(z[c + '.t'])), tt(z[c + '.dt']), tt(z[c + '.lam'])
        assert ref.dtype == torch.float32
        for seq, tol in ((lit, 2e-6), (closed, 1e-4)):
            r, xc = seq.reverse_rates(x_t, lg, t)
            lam = r * dt
            assert torch.equal(lam == 0, ref == 0)                                  # zero exactly at the current token
            assert torch.equal(xc, torch.clamp(x_t, 0, 19))
            rel = ((lam - ref).abs() / ref.abs().clamp_min(1e-30))[ref > 0]
            assert float(rel.max()) < tol, (c, float(rel.max()))
        # and the recorded draw applied to the tokens gives the reference's x_new on both routes
        for seq in (lit, closed):
            assert torch.equal(seq.reverse(x_t, lg, t, dt, jumps=tt(z[c + '.jumps'])), tt(z[c + '.x_new']))
    assert float(tt(z['c21.lam']).max()) > 4.0                                      # the vectors do reach rate * dt ~ 5


def test_closed_form_transition_is_closer_to_fp64_than_the_reference_route():
    lit, closed = O.OracleSeq({'rate_const': 0.3}, eigh=True), O.OracleSeq({'rate_const': 0.3})
    for t in (1.0, 0.5, 0.1, 0.02):
        e = np.exp(-6.0 * t)
        truth = e * np.eye(20) + (1 - e) / 20
        el = np.abs(lit.transition(torch.tensor([t])).double().numpy()[0] - truth) / truth
        ec = np.abs(closed.transition(torch.tensor([t])).double().numpy()[0] - truth) / truth
        assert ec.max() < 3e-7 and ec.max() <= el.max() and el.max() < 1e-4


def test_poisson_icdf_is_the_poisson_inverse_cdf():
    """Exact against a float64 cdf table away from the fp32 rounding of the cdf edges; mean / variance over 4e5 draws."""
    from scipy.stats import poisson
    rng = np.random.default_rng(5)
    lam = np.concatenate([np.zeros(10), rng.uniform(0, 0.05, 100000), rng.uniform(0, 5.0, 100000), [5.0, 4.736, 1e-8, 20.0]]).astype(np.float32)
    u = ((rng.integers(0, 1 << 24, lam.shape).astype(np.float32) + 0.5) / np.float32(1 << 24)).astype(np.float32)
    k = O.poisson_icdf(lam, u)
    assert k.dtype == np.float32 and k.min() >= 0 and k.max() <= 64
    lo = poisson.cdf(k - 1, lam.astype(np.float64))          # u must lie in (cdf(k-1), cdf(k)]
    hi = poisson.cdf(k, lam.astype(np.float64))
    tol = 4e-7
    assert np.all(u.astype(np.float64) > lo - tol) and np.all(u.astype(np.float64) <= hi + tol)
    assert np.all(k[:10] == 0)
    # saturated tail: a uniform above the largest fp32 cdf value must not run to the cap
    top = np.float32(1.0) - np.float32(2.0 ** -25)
    kt = O.poisson_icdf(np.array([0.5, 3.0, 0.01], np.float32), np.array([top, top, top], np.float32))
    assert np.all(kt < 40), kt
    sel = slice(100010, 200010)
    draws = np.stack([O.poisson_icdf(lam[sel], ((rng.integers(0, 1 << 24, 100000).astype(np.float32) + 0.5) / np.float32(1 << 24)))
                      for _ in range(4)])
    lm = lam[sel].astype(np.float64)
    assert abs(draws.mean() - lm.mean()) < 4 * np.sqrt(lm.mean() / draws.size)
    resid = (draws - lm[None]) ** 2
    assert abs(resid.mean() - lm.mean()) < 0.02 * lm.mean()                        # Var = lam


# ---------------------------------------------------------------------------------------------------------------------------
# peptide-geometry violation terms (guidance): pinned on the reference's eval/metric_scripts/cal_vio.py
# ---------------------------------------------------------------------------------------------------------------------------
def test_peptide_violation_terms_match_reference_cal_vio():
    z = load_npz('vio_pdb.npz')
    n_viol = 0
    for c in z['cases']:
        code = str(c).split('.')[0]
        p = load_npz(f'pdb_{code}.npz')
        m, ch, aa = tt(p['batch.atom14_gt_exists']), tt(p['batch.chain_id']), tt(p['batch.seq'])
        got = O.peptide_violation_terms(tt(z[f'{c}.pos']), m, aa, ch)               # residx=None: the reference's chain-only rule
        for k in ('c_n_loss_per_residue', 'ca_c_n_loss_per_residue', 'c_n_ca_loss_per_residue'):
            assert float((got[k] - tt(z[f'{c}.{k}'])).abs().max()) < 2e-6, (c, k)
        for k in ('c_n_violation_mask', 'ca_c_n_violation_mask', 'c_n_ca_violation_mask', 'has_no_gap_mask'):
            assert torch.equal(got[k].bool(), tt(z[f'{c}.{k}']).bool()), (c, k)
        # the reference's mean losses from the restated per-residue terms and masks
        for loss, per, mk in (('c_n_loss', 'c_n_loss_per_residue', 'c_n_mask'), ('ca_c_n_loss', 'ca_c_n_loss_per_residue', 'ca_c_n_mask'),
                              ('c_n_ca_loss', 'c_n_ca_loss_per_residue', 'c_n_ca_mask')):
            mine = (got[mk] * got[per]).sum() / (got[mk].sum() + 1e-6)
            assert abs(float(mine) - float(z[f'{c}.{loss}'])) < 1e-6 + 1e-5 * abs(float(z[f'{c}.{loss}'])), (c, loss)
        n_viol += int(got['c_n_violation_mask'].sum() + got['ca_c_n_violation_mask'].sum() + got['c_n_ca_violation_mask'].sum())
        # violation_energy is the un-normalised sum of exactly these terms
        _, eb, ea = O.violation_energy(tt(z[f'{c}.pos']), m, aa, ch, w_clash=0.0)
        assert abs(float(eb) - float((tt(z[f'{c}.c_n_loss_per_residue']) * got['c_n_mask']).sum())) < 1e-4
        want_a = (tt(z[f'{c}.ca_c_n_loss_per_residue']) * got['ca_c_n_mask']).sum() + (tt(z[f'{c}.c_n_ca_loss_per_residue']) * got['c_n_ca_mask']).sum()
        assert abs(float(ea) - float(want_a)) < 1e-4
    assert n_viol > 500
    # 6qd7 as shipped: the reference's chain-only rule sees a "broken bond" across the gap of the cropped antigen patch; linking by
    # residue number (the guidance default) does not
    p = load_npz('pdb_6qd7.npz')
    args = (tt(z['6qd7.s0.pos']), tt(p['batch.atom14_gt_exists']), tt(p['batch.seq']), tt(p['batch.chain_id']))
    assert int(O.peptide_violation_terms(*args)['c_n_violation_mask'].sum()) == 1
    assert int(O.peptide_violation_terms(*args, residx=tt(p['batch.residx']))['c_n_violation_mask'].sum()) == 0


def test_diffusion_masks_match_reference_on_edge_layouts():
    """abx_amd.features.make_diffuser_features against the reference's own masks (tests/golden/make_golden_masks.py ran the unmodified
    abx/model/features.py:130-212) on anchor layouts the shipped complexes do not reach: closing anchor on the last antibody residue
    with / without an antigen (loss window clipped with the total length), an odd anchor count in a single row and in a batch of three
    (flat row-major pairing), generate_area = 'cdr'."""
    import numpy as np
    from conftest import load_npz
    from abx_amd import features
    g = load_npz('masks_cases.npz')
    names = sorted({k.split('.')[0] for k in g})
    assert len(names) == 5

    class NoDiffuser:                       # the masks do not depend on the noising: record what it is asked to diffuse
        def sample_ref(self, n_samples, impute_rigids, impute_seq, diffuse_mask, noise=None):
            return {'rigids_t': impute_rigids, 'seq_t': impute_seq}

    for nm in names:
        af = torch.as_tensor(g[nm + '.anchor_flag'])
        B, Lab = af.shape
        Ltot = int(g[nm + '.Ltot'])
        rots = torch.eye(3)[None, None, None].expand(B, Ltot, 8, 3, 3).contiguous()
        batch = dict(seq=torch.zeros(B, Ltot, dtype=torch.int64), mask=torch.ones(B, Ltot, dtype=torch.bool), anchor_flag=af,
                     rigidgroups_gt_frames=(rots, torch.zeros(B, Ltot, 8, 3)))
        out = features.make_diffuser_features(batch, str(g[nm + '.area']), NoDiffuser())
        assert np.array_equal(out['fixed_mask'].numpy(), g[nm + '.fixed_mask']), nm
        assert np.array_equal(out['struc_loss_mask'].numpy(), g[nm + '.struc_loss_mask']), nm

