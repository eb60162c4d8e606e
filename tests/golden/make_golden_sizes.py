"""Golden vectors beyond the tiny complex (feat_L48, module_* at L = 48, L256 / L352 digests), produced
by running the UNMODIFIED reference (/root/reference) on CPU in the build container:

    python tests/golden/make_golden_sizes.py

Fixtures are DATA (inputs, recorded noise, expected outputs); weights are regenerated from
`abx_amd.synthetic.random_state_dict(shapes, seed=7)`, complexes from `abx_amd.synthetic.make_complex`.

  feat_L48.npz      collated batch of two L = 48 complexes (the second with a 3-residue padded antigen tail), recorded init noise,
                    outputs of the 7 feature transforms
  modules_L48.npz   ONE in-loop ScoreNetwork call (t ~ 0.5 fp64, from zero self-conditioning state) on the masked-tail complex:
                    inputs of the final pass, per-module outputs of the final pass (pair-shaped outputs sub-sampled [::3, ::3] to
                    keep the fixture small; their full-tensor sums are stored too), call outputs
  L256_digest.npz / L352_digest.npz
                    the bench's synthetic complexes (seed 1), B = 1: recorded init noise, the reference's diffusion features, and
                    the outputs of one in-loop call (rigids, rot/trans score, logits, seq_0, pLDDT, atom14, sub-sampled pair)
"""
import copy
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

diffuser = FullDiffuser.get(cfg.diffuser)
from abx.model.abx import ScoreNetwork, get_prev  # noqa: E402
from abx.model.features import FeatureBuilder  # noqa: E402
import inference as ref_inference  # noqa: E402

sys.path.insert(0, ROOT)                      # after the reference's `abx` / `diffuser` are in sys.modules (alias packages)
from abx_amd import synthetic  # noqa: E402
assert ref_inference.__file__.startswith(ref_shims.REF) and sys.modules['diffuser.full_diffuser'].__file__.startswith(ref_shims.REF)

model = ScoreNetwork(cfg.model, diffuser).eval()
shapes = OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
model.load_state_dict(synthetic.random_state_dict(shapes, seed=7), strict=True)

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

FEAT_KEYS = ('atom14_atom_exists', 'residx_atom37_to_atom14', 'atom37_atom_exists', 'atom37_gt_positions',
             'atom37_gt_exists', 'rigidgroups_gt_frames', 'rigidgroups_gt_exists', 'torsion_angles_sin_cos',
             'torsion_angles_mask', 'pseudo_beta', 'pseudo_beta_mask', 'rigids_t', 'seq_t', 't', 'fixed_mask',
             'rigids_0', 'struc_loss_mask')


def npy(x):
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


def featurise(raw, seed):
    """Reference FeatureBuilder + the init noise it drew (same seed, the reference's draw order)."""
    torch.manual_seed(seed)
    batch = FeatureBuilder(feats, is_training=False)(copy.deepcopy(raw))
    torch.manual_seed(seed)
    B, L = raw['seq'].shape
    noise = dict(rot_axis=torch.randn(B, L, 3), rot_u=torch.rand(B, L), trans_z=torch.randn(B, L, 3),
                 seq=torch.randint(low=0, high=20, size=(B, L)))
    return batch, noise


SUB = 3       # pair-shaped module outputs are stored as x[:, ::SUB, ::SUB] plus their full sum


def pair_digest(prefix, x, out):
    out[prefix] = x[:, ::SUB, ::SUB].contiguous()
    out[prefix + '.sum'] = x.double().sum()
    out[prefix + '.abssum'] = x.double().abs().sum()


# ---------------------------------------------------------------------------------------------------
# 1. L = 48 features (B = 2) and modules (the masked-tail complex, B = 1)
# ---------------------------------------------------------------------------------------------------
W48 = dict(L_heavy=20, L_light=16, L_antigen=12, cdr=(10, 16))
c0 = synthetic.make_complex(seed=21, **W48)
c1 = synthetic.make_complex(seed=22, n_masked_tail=3, **W48)
raw = synthetic.collate([c0, c1])
batch0, init_noise = featurise(raw, 2345)
out = {('raw.' + k): v for k, v in raw.items()}
out.update({('noise.' + k): v for k, v in init_noise.items()})
for k in FEAT_KEYS:
    out['feat.' + k] = batch0[k]
save('feat_L48.npz', out)

raw1 = synthetic.collate([c1])
batch1, noise1 = featurise(raw1, 3456)
B = 1
captured = {}
blk = model.impl.seqformer.seqformer.blocks[0]
hook_targets = {
    'enc_residue': model.impl.seqformer.encode_residue_emb, 'enc_pair': model.impl.seqformer.encode_pair_emb,
    'seq_attn': blk.seq_attn, 'seq_transition': blk.seq_transition, 'opm': blk.outer_product_mean,
    'trimul_out': blk.triangle_multiplication_outgoing, 'trimul_in': blk.triangle_multiplication_incoming,
    'triattn_start': blk.triangle_attention_starting_node, 'triattn_end': blk.triangle_attention_ending_node,
    'pair_transition': blk.pair_transition, 'ipa0': model.impl.diffusion_module.ScoreNetwork.attention_module,
}
handles = []
for name, mod in hook_targets.items():
    def mk(name):
        def hook(m, args, kwargs, output):
            if name == 'ipa0':
                if captured.get('_ipa_pass') == captured.get('_pass'):
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
    # the self-conditioning inputs of THIS pass (the final pass overwrites the earlier captures)
    captured['pass.seq_t'] = cur_batch['seq_t'].clone()
    for k in ('prev_pos', 'prev_seq', 'prev_pair'):
        captured['pass.' + k] = cur_batch[k].clone()


handles.append(blk.register_forward_pre_hook(blk_pre, with_kwargs=True))
ones = torch.ones(B, dtype=torch.float32)
t_np = np.linspace(0.01, 1.0, 100)[::-1][50]          # ~0.5, np.float64
cur_batch = copy.deepcopy(batch1)
cur_batch = ref_inference._set_t_feats(cur_batch, diffuser, torch.tile(torch.tensor(t_np), (B,)), ones)
state_in = {k: cur_batch[k].clone() for k in ('seq_t', 'rigids_t', 't', 'rot_score_scaling', 'trans_score_scaling')}
with torch.no_grad():
    ret = model(cur_batch)
for h in handles:
    h.remove()
f = ret['heads']['folding']
mods = {('raw.' + k): v for k, v in raw1.items()}
mods.update({('noise.' + k): v for k, v in noise1.items()})
mods.update({('feat.' + k): batch1[k] for k in FEAT_KEYS})
mods.update({('in.' + k): v for k, v in state_in.items()})
for k, v in captured.items():
    if k.startswith('_'):
        continue
    if k.endswith('.out') and v.dim() == 4 and v.shape[1] == v.shape[2] and k != 'ipa0.out':
        pair_digest(k, v, mods)
    elif k == 'pass.prev_pos':
        mods[k] = v.to(torch.int8)
    else:
        mods[k] = v
pair_digest('out.pair', ret['representations']['pair'], mods)
mods.update({
    'final.seq_t_after': cur_batch['seq_t'], 'out.seq': ret['representations']['seq'],
    'out.rot_score': f['rot_score'], 'out.trans_score': f['trans_score'], 'out.rigids': f['rigids'],
    'out.structure_module': f['representations']['structure_module'], 'out.angles': f['sidechains'][-1]['angles_sin_cos'],
    'out.atom14': f['final_atom14_positions'], 'out.atom37': f['final_atom_positions'],
    'out.logits': ret['heads']['sequence_module']['logits'], 'out.seq_0': ret['heads']['sequence_module']['seq_0'],
    'out.pLDDT': ret['heads']['predicted_lddt']['pLDDT'],
    'out.prev_pos': get_prev(cur_batch, ret, cfg.model)['prev_pos'].to(torch.int8),
    'sub': np.int64(SUB),
})
save('modules_L48.npz', mods)

# ---------------------------------------------------------------------------------------------------
# 2. large-shape digests on the bench's complexes
# ---------------------------------------------------------------------------------------------------
for wname in ('L256', 'L352'):
    cx = synthetic.make_complex(seed=1, **synthetic.WORKLOADS[wname])
    rawL = synthetic.collate([cx])
    bL, nL = featurise(rawL, 4567)
    L = rawL['seq'].shape[1]
    b = copy.deepcopy(bL)
    b = ref_inference._set_t_feats(b, diffuser, torch.tile(torch.tensor(t_np), (1,)), torch.ones(1))
    with torch.no_grad():
        r = model(b)
    fl = r['heads']['folding']
    dg = {('noise.' + k): v for k, v in nL.items()}
    dg.update({'feat.rigids_t': bL['rigids_t'], 'feat.seq_t': bL['seq_t'], 'feat.fixed_mask': bL['fixed_mask'], 'feat.t': bL['t'],
               'feat.torsion_angles_sin_cos': bL['torsion_angles_sin_cos'], 'feat.rigids_0': bL['rigids_0'],
               'in.t': b['t'], 'in.rot_score_scaling': b['rot_score_scaling'], 'in.trans_score_scaling': b['trans_score_scaling'],
               'out.rigids': fl['rigids'], 'out.rot_score': fl['rot_score'], 'out.trans_score': fl['trans_score'],
               'out.logits': r['heads']['sequence_module']['logits'], 'out.seq_0': r['heads']['sequence_module']['seq_0'],
               'out.pLDDT': r['heads']['predicted_lddt']['pLDDT'], 'out.atom14': fl['final_atom14_positions'],
               'out.seq': r['representations']['seq'], 'final.seq_t_after': b['seq_t'],
               'out.prev_pos': get_prev(b, r, cfg.model)['prev_pos'].to(torch.int8), 'workload': np.array(wname), 'seed': np.int64(1)})
    sub = 16
    dg['out.pair_sub'] = r['representations']['pair'][:, ::sub, ::sub].contiguous()
    dg['out.pair.sum'] = r['representations']['pair'].double().sum()
    dg['out.pair.abssum'] = r['representations']['pair'].double().abs().sum()
    dg['pair_sub'] = np.int64(sub)
    save(f'{wname}_digest.npz', dg)
print('done')
