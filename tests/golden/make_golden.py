"""Generate the committed golden vectors by running the UNMODIFIED reference (/root/reference) on CPU.

Run in the build container only (the GPU box has no reference):   python tests/golden/make_golden.py
Writes tests/golden/*.npz / *.json.  Fixtures are DATA (inputs, recorded noise, expected outputs); the
weights are not stored: they are regenerated from `abx_amd.synthetic.random_state_dict(shapes, seed)`.

Fixture list:
  sd_keys.json        ordered state_dict keys + shapes (checkpoint contract)
  feat_tiny.npz       raw collated batch, recorded init noise, outputs of the 7 feature transforms
  modules_tiny.npz    one in-loop ScoreNetwork call (t=0.5, fp64 t): per-module outputs of the final pass
  step_tiny.npz       FullDiffuser.reverse with recorded noise (fp64 state) at t in {1.0, 0.5, 0.02}
  traj_tiny.npz       the reference sample_fn (inference.py:180-273) with num_t=4, design mode
  igso3_small.npz     IGSO(3) tables for num_sigma=num_omega=40 + spot values / checksums of the 1000x1000 tables
"""
import copy
import hashlib
import json
import os
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.makedirs('/tmp/abx_golden_scratch', exist_ok=True)
os.chdir('/tmp/abx_golden_scratch')          # IGSO3 cache is CWD-relative (so3_diffuser.py:131-142)

import torch  # noqa: E402

torch.set_num_threads(8)
from ref_shims import ConfigDict  # noqa: E402

cfg_json = json.load(open('/root/reference/config/config_model.json'))
cfg_json['model']['embeddings_and_seqformer']['esm']['enabled'] = False
cfg_json['diffuser']['so3']['use_cached_score'] = True
cfg = ConfigDict(cfg_json)

from diffuser.full_diffuser import FullDiffuser  # noqa: E402
from diffuser import so3_diffuser  # noqa: E402

diffuser = FullDiffuser.get(cfg.diffuser)
from abx.model.abx import ScoreNetwork, get_prev  # noqa: E402
from abx.model.features import FeatureBuilder  # noqa: E402
import inference as ref_inference  # noqa: E402

# the repository root goes on the path only AFTER the reference's packages are in sys.modules: the repo ships alias packages
# named `abx` and `diffuser` (drop-in import paths) that must not shadow the reference here
sys.path.insert(0, ROOT)
from abx_amd import synthetic  # noqa: E402
assert ref_inference.__file__.startswith(ref_shims.REF) and sys.modules['diffuser.full_diffuser'].__file__.startswith(ref_shims.REF)

SEED_W = 7
model = ScoreNetwork(cfg.model, diffuser).eval()
shapes = OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
json.dump([[k, list(s)] for k, s in shapes.items()], open(os.path.join(HERE, 'sd_keys.json'), 'w'), indent=0)
model.load_state_dict(synthetic.random_state_dict(shapes, seed=SEED_W), strict=True)


def npy(x):
    if isinstance(x, (tuple, list)):
        return [npy(v) for v in x]
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def save(name, d):
    flat = {}
    for k, v in d.items():
        if isinstance(v, (tuple, list)) and len(v) and torch.is_tensor(v[0]):
            for i, vi in enumerate(v):
                flat[f'{k}.{i}'] = npy(vi)
        else:
            flat[k] = npy(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **flat)
    print('wrote', name, os.path.getsize(path) // 1024, 'KiB', len(flat), 'arrays')


# ---------------------------------------------------------------------------------------------------
# 1. features
# ---------------------------------------------------------------------------------------------------
def tiny_raw():
    w = synthetic.WORKLOADS['tiny']
    c0 = synthetic.make_complex(seed=11, **w)
    c1 = synthetic.make_complex(seed=12, n_masked_tail=1, **w)
    return synthetic.collate([c0, c1])


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

raw = tiny_raw()
torch.manual_seed(1234)
batch0 = FeatureBuilder(feats, is_training=False)(copy.deepcopy(raw))
# the init noise in the reference's draw order (SURVEY.md §7 hard part 2)
torch.manual_seed(1234)
B, L = raw['seq'].shape
init_noise = dict(rot_axis=torch.randn(B, L, 3), rot_u=torch.rand(B, L), trans_z=torch.randn(B, L, 3),
                  seq=torch.randint(low=0, high=20, size=(B, L)))
out = {('raw.' + k): v for k, v in raw.items()}
out.update({('noise.' + k): v for k, v in init_noise.items()})
for k in ('atom14_atom_exists', 'residx_atom37_to_atom14', 'atom37_atom_exists', 'atom37_gt_positions',
          'atom37_gt_exists', 'rigidgroups_gt_frames', 'rigidgroups_gt_exists', 'torsion_angles_sin_cos',
          'torsion_angles_mask', 'pseudo_beta', 'pseudo_beta_mask', 'rigids_t', 'seq_t', 't', 'fixed_mask',
          'rigids_0', 'struc_loss_mask'):
    out['feat.' + k] = batch0[k]
save('feat_tiny.npz', out)

# ---------------------------------------------------------------------------------------------------
# 2. one in-loop ScoreNetwork call with hooks on the final pass
# ---------------------------------------------------------------------------------------------------
captured = {}
blk = model.impl.seqformer.seqformer.blocks[0]
hook_targets = {
    'enc_residue': model.impl.seqformer.encode_residue_emb,
    'enc_pair': model.impl.seqformer.encode_pair_emb,
    'seq_attn': blk.seq_attn, 'seq_transition': blk.seq_transition, 'opm': blk.outer_product_mean,
    'trimul_out': blk.triangle_multiplication_outgoing, 'trimul_in': blk.triangle_multiplication_incoming,
    'triattn_start': blk.triangle_attention_starting_node, 'triattn_end': blk.triangle_attention_ending_node,
    'pair_transition': blk.pair_transition,
    'ipa0': model.impl.diffusion_module.ScoreNetwork.attention_module,
}
handles = []
for name, mod in hook_targets.items():
    def mk(name):
        def hook(m, args, kwargs, output):
            if name == 'ipa0':
                if 'ipa0.out' in captured and captured.get('_pass') == captured.get('_ipa_pass'):
                    return
                captured['_ipa_pass'] = captured.get('_pass')
                captured['ipa0.in_1d'] = kwargs['inputs_1d'].clone()
                captured['ipa0.in_2d'] = kwargs['inputs_2d'].clone()
                captured['ipa0.rots'] = kwargs['in_rigids'][0].clone()
                captured['ipa0.trans'] = kwargs['in_rigids'][1].clone()
            captured[name + '.out'] = output.clone()
        return hook
    handles.append(mod.register_forward_hook(mk(name), with_kwargs=True))


def blk_pre(m, args, kwargs):
    captured['block.seq_in'] = args[0].clone()
    captured['block.pair_in'] = args[1].clone()
    captured['_pass'] = captured.get('_pass', 0) + 1


handles.append(blk.register_forward_pre_hook(blk_pre, with_kwargs=True))

batch = copy.deepcopy(batch0)
ones = torch.ones(B, dtype=torch.float32)
# warm-up self-conditioning at t=1 (inference.py:209-211) so that prev_* are non-trivial
batch = ref_inference._set_t_feats(batch, diffuser, np.linspace(0.01, 1.0, 100)[::-1][0], ones)
with torch.no_grad():
    warm = model(batch)
    batch.update(get_prev(batch, warm, cfg.model))
warm_out = dict(rigids=warm['heads']['folding']['rigids'], seq_0=warm['heads']['sequence_module']['seq_0'],
                prev_pos=batch['prev_pos'], logits=warm['heads']['sequence_module']['logits'],
                rot_score=warm['heads']['folding']['rot_score'], trans_score=warm['heads']['folding']['trans_score'])
t_np = np.linspace(0.01, 1.0, 100)[::-1][50]          # ~0.5, np.float64
t_ = torch.tile(torch.tensor(t_np), (B,))
batch = ref_inference._set_t_feats(batch, diffuser, t_, ones)
state_in = {k: batch[k].clone() for k in ('seq_t', 'rigids_t', 't', 'prev_pos', 'prev_seq', 'prev_pair',
                                            'rot_score_scaling', 'trans_score_scaling')}
captured.clear()
with torch.no_grad():
    ret = model(batch)
for h in handles:
    h.remove()
f = ret['heads']['folding']
mods = {('in.' + k): v for k, v in state_in.items()}
mods.update({('warm.' + k): v for k, v in warm_out.items()})
mods.update({k: v for k, v in captured.items() if not k.startswith('_')})
mods.update({
    'final.seq_t_after': batch['seq_t'], 'final.prev_pos_in': batch['prev_pos'],
    'final.prev_seq_in': batch['prev_seq'], 'final.prev_pair_in': batch['prev_pair'],
    'out.seq': ret['representations']['seq'], 'out.pair': ret['representations']['pair'],
    'out.rot_score': f['rot_score'], 'out.trans_score': f['trans_score'], 'out.rigids': f['rigids'],
    'out.structure_module': f['representations']['structure_module'],
    'out.angles': f['sidechains'][-1]['angles_sin_cos'],
    'out.atom14': f['final_atom14_positions'], 'out.atom37': f['final_atom_positions'],
    'out.logits': ret['heads']['sequence_module']['logits'], 'out.seq_0': ret['heads']['sequence_module']['seq_0'],
    'out.pLDDT': ret['heads']['predicted_lddt']['pLDDT'],
})
prev = get_prev(batch, ret, cfg.model)
mods['out.prev_pos'] = prev['prev_pos']
save('modules_tiny.npz', mods)

# ---------------------------------------------------------------------------------------------------
# 3. reverse step with recorded noise
# ---------------------------------------------------------------------------------------------------
_orig_randn = torch.randn
_orig_poisson = torch.poisson
rec = []


def rec_randn(*a, **k):
    z = _orig_randn(*a, **k)
    rec.append(('randn', z.clone()))
    return z


def rec_poisson(*a, **k):
    z = _orig_poisson(*a, **k)
    rec.append(('poisson', z.clone()))
    return z


step = {}
diffuse_mask = (1 - batch0['fixed_mask']) * batch0['atom14_gt_exists'][..., 0]
step['diffuse_mask'] = diffuse_mask
dt = torch.tensor(1 / 100)
rig = batch0['rigids_t'].clone()
seq_t = batch0['seq_t'].clone()
g = torch.Generator().manual_seed(99)
for i, tv in enumerate([1.0, np.linspace(0.01, 1.0, 100)[::-1][50], 0.02]):
    t_ = torch.tile(torch.tensor(np.float64(tv)), (B,))
    rot_score = _orig_randn(B, L, 3, generator=g) * 2.0
    trans_score = (_orig_randn(B, L, 3, generator=g) * 3.0).double()       # fp64 in the loop
    logits = _orig_randn(B, L, 20, generator=g) * 2.0
    torch.randn, torch.poisson = rec_randn, rec_poisson
    rec.clear()
    torch.manual_seed(500 + i)
    try:
        rig1, seq1 = diffuser.reverse(rigid_t=rig, seq_t=seq_t, rot_score=rot_score, trans_score=trans_score,
                                      logits_t=logits, diffuse_mask=diffuse_mask, t=t_, dt=dt, center=True,
                                      noise_scale=1.0)
    finally:
        torch.randn, torch.poisson = _orig_randn, _orig_poisson
    kinds = [k for k, _ in rec]
    assert kinds == ['randn', 'randn', 'poisson'], kinds
    step.update({f's{i}.t': t_, f's{i}.rigid_in': rig, f's{i}.seq_in': seq_t, f's{i}.rot_score': rot_score,
                 f's{i}.trans_score': trans_score, f's{i}.logits': logits, f's{i}.z_rot': rec[0][1],
                 f's{i}.z_trans': rec[1][1], f's{i}.jumps': rec[2][1], f's{i}.rigid_out': rig1,
                 f's{i}.seq_out': seq1})
    rs, ts = diffuser.score_scaling(t_)
    step.update({f's{i}.rot_score_scaling': rs, f's{i}.trans_score_scaling': ts})
    rig, seq_t = rig1, seq1           # chain: second/third steps see fp64 rigids, int64 tokens
step['dt'] = dt
save('step_tiny.npz', step)

# ---------------------------------------------------------------------------------------------------
# 4. short trajectory through the reference's own sample_fn
# ---------------------------------------------------------------------------------------------------
traj_cap = {}


def fake_post(batch_, traj, args):
    traj_cap['traj'] = traj
    traj_cap['batch'] = batch_


ref_inference.postprocess_trajectory = fake_post


class Args:
    mode = 'trajectory'
    output_dir = '/tmp/abx_golden_scratch'


noise_log = []


def log_randn(*a, **k):
    z = _orig_randn(*a, **k)
    noise_log.append(('randn', z.clone()))
    return z


def log_poisson(*a, **k):
    z = _orig_poisson(*a, **k)
    noise_log.append(('poisson', z.clone()))
    return z


torch.randn, torch.poisson = log_randn, log_poisson
torch.manual_seed(4321)
try:
    ref_inference.sample_fn(copy.deepcopy(batch0), cfg, diffuser, model, Args(), num_t=4)
finally:
    torch.randn, torch.poisson = _orig_randn, _orig_poisson
tj = {}
for k, d in enumerate(traj_cap['traj']):
    tj[f'k{k}.seq'] = d['seq']
    tj[f'k{k}.atom14'] = d['atom14_results']
    tj[f'k{k}.pLDDT'] = d['pLDDT']
    tj[f'k{k}.time'] = np.float64(d['time'])
kinds = [k for k, _ in noise_log]
assert kinds == ['randn', 'randn', 'poisson'] * 3, kinds
for s in range(3):
    tj[f'n{s}.z_rot'] = noise_log[3 * s][1]
    tj[f'n{s}.z_trans'] = noise_log[3 * s + 1][1]
    tj[f'n{s}.jumps'] = noise_log[3 * s + 2][1]
tj['final.rigids_t'] = traj_cap['batch']['rigids_t']
tj['final.seq_t'] = traj_cap['batch']['seq_t']
tj['final.t'] = traj_cap['batch']['t']
save('traj_tiny.npz', tj)

# ---------------------------------------------------------------------------------------------------
# 4b. optimize mode (BASELINE config 4): forward_marginal noising at t = opt_step/100 and the shortened reverse loop
# ---------------------------------------------------------------------------------------------------
OPT_STEP = 4
feats_opt = []
for fn, opts in feats:
    opts = dict(opts)
    if 'diffuse' in fn:
        opts['diff_conf'] = dict(cfg_json['diffuser'], opt_step=OPT_STEP)
    feats_opt.append((fn, opts))
_orig_rand, _orig_normal = torch.rand, torch.normal
_orig_cat_sample = torch.distributions.categorical.Categorical.sample
fm_log = []


def _rec(kind, fn):
    def inner(*a, **k):
        z = fn(*a, **k)
        fm_log.append((kind, z.clone(), [x.clone() for x in a if torch.is_tensor(x)], {kk: vv.clone() for kk, vv in k.items() if torch.is_tensor(vv)}))
        return z
    return inner


torch.randn, torch.rand, torch.normal = _rec('randn', _orig_randn), _rec('rand', _orig_rand), _rec('normal', _orig_normal)
torch.distributions.categorical.Categorical.sample = _rec('cat', _orig_cat_sample)
torch.manual_seed(777)
try:
    batch_opt = FeatureBuilder(feats_opt, is_training=False)(copy.deepcopy(raw))
finally:
    torch.randn, torch.rand, torch.normal = _orig_randn, _orig_rand, _orig_normal
    torch.distributions.categorical.Categorical.sample = _orig_cat_sample
kinds = [k for k, *_ in fm_log]
assert kinds == ['randn', 'rand', 'normal', 'cat', 'cat', 'cat'], kinds
nz = fm_log[2]
mean, std = nz[3]['mean'], nz[3]['std']
opt = {'noise.rot_axis': fm_log[0][1], 'noise.rot_u': fm_log[1][1], 'noise.trans_z': (nz[1] - mean) / std,
       'noise.trans_xt': nz[1], 'noise.seq_xt': fm_log[3][1].view(B, L), 'noise.seq_dim': fm_log[4][1], 'noise.seq_new': fm_log[5][1]}
for k in ('rigids_t', 'seq_t', 't', 'fixed_mask', 'rigids_0', 'rot_score', 'trans_score', 'rot_score_scaling', 'trans_score_scaling'):
    opt['feat.' + k] = batch_opt[k]


class ArgsOpt:
    mode = 'optimize'
    output_dir = '/tmp/abx_golden_scratch'


noise_log.clear()
traj_cap.clear()
torch.randn, torch.poisson = log_randn, log_poisson
torch.manual_seed(888)
try:
    ref_inference.sample_fn(copy.deepcopy(batch_opt), cfg, diffuser, model, ArgsOpt(), num_t=100)
finally:
    torch.randn, torch.poisson = _orig_randn, _orig_poisson
nsteps = len(noise_log) // 3
assert [k for k, _ in noise_log] == ['randn', 'randn', 'poisson'] * nsteps and nsteps == OPT_STEP - 1, (len(noise_log), nsteps)
for s_ in range(nsteps):
    opt[f'n{s_}.z_rot'] = noise_log[3 * s_][1]
    opt[f'n{s_}.z_trans'] = noise_log[3 * s_ + 1][1]
    opt[f'n{s_}.jumps'] = noise_log[3 * s_ + 2][1]
assert len(traj_cap['traj']) == 1
d = traj_cap['traj'][0]
opt.update({'last.seq': d['seq'], 'last.atom14': d['atom14_results'], 'last.pLDDT': d['pLDDT'], 'last.time': np.float64(d['time']),
            'final.rigids_t': traj_cap['batch']['rigids_t'], 'final.seq_t': traj_cap['batch']['seq_t']})
save('optimize_tiny.npz', opt)

# ---------------------------------------------------------------------------------------------------
# 5. IGSO(3) tables
# ---------------------------------------------------------------------------------------------------
small_conf = dict(cfg_json['diffuser']['so3'], num_sigma=40, num_omega=40, cache_dir='/tmp/abx_golden_scratch/.cache_small/')
small = so3_diffuser.SO3Diffuser(small_conf)
big = diffuser._so3_diffuser
ig = dict(small_pdf=small._pdf, small_cdf=small._cdf, small_score_norms=small._score_norms,
          small_score_scaling=small._score_scaling, small_sigma=small.discrete_sigma, small_omega=small.discrete_omega,
          big_sigma=big.discrete_sigma, big_omega=big.discrete_omega, big_score_scaling=big._score_scaling)
rs = np.random.RandomState(5)
ii, jj = rs.randint(0, 1000, 256), rs.randint(0, 1000, 256)
ig.update(spot_i=ii, spot_j=jj, spot_pdf=big._pdf[ii, jj], spot_cdf=big._cdf[ii, jj],
          spot_score_norms=big._score_norms[ii, jj])
for k in ('_pdf', '_cdf', '_score_norms'):
    ig['sha256' + k] = np.frombuffer(hashlib.sha256(getattr(big, k).numpy().tobytes()).digest(), dtype=np.uint8)
# rows of the big tables used by the tiny tests (t = 1.0, ~0.5, 0.02, 0.67, 0.34): full rows, so that the score
# lookup can be checked without regenerating the 1000x1000 tables on CPU
ts = torch.tensor([1.0, float(t_np), 0.02, 0.67, 0.34, 0.01, 0.04, 0.03], dtype=torch.float64)
rows = sorted(set(big.t_to_idx(ts)))
ig['rows'] = np.asarray(rows)
ig['rows_score_norms'] = big._score_norms[rows]
ig['rows_cdf'] = big._cdf[rows]
save('igso3_small.npz', ig)
print('done')
