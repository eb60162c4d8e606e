"""Minimal end-to-end driver in the shape of the reference's inference.py / design.py main (inference.py:276-331): build the
diffuser and the score network, featurise one complex, run the reverse diffusion for `num_samples` samples and write the
PDB files (per step in trajectory mode, asynchronously).  Weights: a checkpoint with the reference's `model_state_dict`, or
seeded random weights (no checkpoint ships with the reference); complex: one of the synthetic workloads of
`abx_amd.synthetic` (raw-PDB featurisation is SURVEY 8f-1, not built).

    python -m abx_amd.design --workload L256 --num_samples 4 --mode trajectory --num_t 10 --output_dir out/"""
import argparse
from collections import OrderedDict

import torch

from . import features, sampler, synthetic
from .config import default_config, load_config
from .diffuser.full_diffuser import FullDiffuser
from .io import TrajectoryWriter
from .model.abx import ScoreNetwork


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='L256', choices=sorted(synthetic.WORKLOADS))
    ap.add_argument('--num_samples', type=int, default=4)
    ap.add_argument('--mode', default='design', choices=['design', 'trajectory'])
    ap.add_argument('--num_t', type=int, default=100)
    ap.add_argument('--generate_area', default='H3')
    ap.add_argument('--model_config', default=None, help='the reference config/config_model.json (default: built-in copy)')
    ap.add_argument('--ckpt', default=None, help='checkpoint with model_state_dict (default: seeded random weights)')
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

    w = synthetic.WORKLOADS[a.workload]
    cx = synthetic.make_complex(seed=a.seed + 1, **w)
    raw = {k: v.to(dev) for k, v in synthetic.replicate(cx, a.num_samples).items()}
    torch.manual_seed(a.seed)
    batch = features.build_features(raw, diffuser, generate_area=a.generate_area)
    batch['_shared_context'] = True
    B, nh, nl = a.num_samples, w['L_heavy'], w['L_light']
    seq = cx['seq'].tolist()
    from .io import index_to_str_seq
    meta = dict(name=[f'{a.workload}-{i:03d}_H_L_A' for i in range(B)], str_heavy_seq=[index_to_str_seq(seq[:nh])] * B,
                str_light_seq=[index_to_str_seq(seq[nh:nh + nl])] * B)
    writer = TrajectoryWriter(meta, a.output_dir, multi=a.mode == 'trajectory')
    sampler.sample_fn(batch, cfg, diffuser, model, mode=a.mode, num_t=a.num_t, sample_ids=torch.arange(B, device=dev), on_record=writer.submit)
    torch.cuda.synchronize()
    files = writer.close()
    print(f'{len(files)} PDB files in {a.output_dir}')
    return files


if __name__ == '__main__':
    main()
