"""End-to-end driver in the shape of the reference's design.py / inference.py main (design.py:277-375): build the diffuser and
the score network, featurise one complex, run the reverse diffusion for `num_samples` samples and write the PDB files (per step
in trajectory mode, asynchronously).

    python -m abx_amd.design --pdb_file 6ct7_H_L_S.pdb --num_samples 100 --mode design --output_dir out/      (raw PDB, 8f-1)
    python -m abx_amd.design --workload L256 --num_samples 4 --mode trajectory --num_t 10 --output_dir out/   (synthetic complex)

--pdb_file follows the reference's naming contract <code>_<heavy>_<light>_<antigen chains joined by |>.pdb (dataset.py:290-293);
it is read by abx_amd.data.antibody (plain-text parser, landmark IMGT locator, 16 A antigen patch, 32-residue window).
Weights: a checkpoint with the reference's `model_state_dict`, or seeded random weights (no checkpoint ships with the reference)."""
import argparse
from collections import OrderedDict

import torch

from . import features, sampler, synthetic
from .config import default_config, load_config
from .diffuser.full_diffuser import FullDiffuser
from .io import TrajectoryWriter, index_to_str_seq
from .model.abx import ScoreNetwork


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--pdb_file', default=None, help='antibody-antigen complex, <code>_<H>_<L>_<antigen chains>.pdb')
    ap.add_argument('--workload', default='L256', choices=sorted(synthetic.WORKLOADS), help='synthetic complex when no --pdb_file')
    ap.add_argument('--num_samples', type=int, default=4)
    ap.add_argument('--mode', default='design', choices=['design', 'trajectory', 'optimize'])
    ap.add_argument('--optimize_steps', type=int, default=10, help='optimize mode: start the reverse process at t = steps / 100')
    ap.add_argument('--num_t', type=int, default=100)
    ap.add_argument('--generate_area', default='H3')
    ap.add_argument('--model_config', default=None, help='the reference config/config_model.json (default: built-in copy)')
    ap.add_argument('--ckpt', '--model', dest='ckpt', default=None, help='checkpoint with model_state_dict (default: seeded random weights)')
    ap.add_argument('--output_dir', default='design_out')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--device', default='cuda:0')
    a = ap.parse_args(argv)

    cfg = load_config(a.model_config) if a.model_config else default_config()
    dev = torch.device(a.device)
    diffuser = FullDiffuser.get(cfg.diffuser).to(dev)
    model = ScoreNetwork(cfg.model, diffuser)
    if a.ckpt:
        sd = torch.load(a.ckpt, map_location='cpu')['model_state_dict']
    else:
        sd = synthetic.random_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items()), seed=a.seed)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()

    B = a.num_samples
    if a.pdb_file:
        from .data.antibody import load_complex
        cb = load_complex(a.pdb_file, seed=a.seed)
        raw = {k: v.to(dev).expand(B, *v.shape[1:]).contiguous() for k, v in cb.items() if torch.is_tensor(v)}
        meta = {k: list(cb[k]) * B for k in ('name', 'str_heavy_seq', 'str_light_seq', 'antigen_origin_str_seq',
                                            'antigen_origin_atom14_gt_positions', 'antigen_origin_atom14_gt_exists',
                                            'antigen_origin_chain_ids')}
        meta['name'] = [f'{n}' if B == 1 else f'{n.split("_")[0]}-{i:03d}_' + '_'.join(n.split('_')[1:]) for i, n in enumerate(meta['name'])]
        L = raw['seq'].shape[1]
    else:
        w = synthetic.WORKLOADS[a.workload]
        cx = synthetic.make_complex(seed=a.seed + 1, **w)
        raw = {k: v.to(dev) for k, v in synthetic.replicate(cx, B).items()}
        nh, nl = w['L_heavy'], w['L_light']
        seq = cx['seq'].tolist()
        meta = dict(name=[f'{a.workload}-{i:03d}_H_L_A' for i in range(B)], str_heavy_seq=[index_to_str_seq(seq[:nh])] * B,
                    str_light_seq=[index_to_str_seq(seq[nh:nh + nl])] * B)
        L = raw['seq'].shape[1]
    ids = list(range(B))
    batch = features.build_features(raw, diffuser, generate_area=a.generate_area,
                                    opt_step=a.optimize_steps if a.mode == 'optimize' else None,
                                    noise=features.per_sample_init_noise(ids, L, a.seed, dev))
    batch['_shared_context'] = True
    diffuser.seed = a.seed
    writer = TrajectoryWriter(meta, a.output_dir, multi=a.mode == 'trajectory')
    sampler.sample_fn(batch, cfg, diffuser, model, mode=a.mode, num_t=a.num_t, sample_ids=torch.arange(B, device=dev), on_record=writer.submit)
    torch.cuda.synchronize()
    files = writer.close()
    print(f'{len(files)} PDB files in {a.output_dir}')
    return files


if __name__ == '__main__':
    main()
