"""End-to-end driver in the shape of the reference's design.py / inference.py main (design.py:277-375, inference.py:59-82,
275-392): build the diffuser and the score network, featurise each complex, run the reverse diffusion for `num_samples` samples
and write the PDB files (per step in trajectory mode, asynchronously).

    python -m abx_amd.design --pdb_file 6ct7_H_L_S.pdb --num_samples 100 --mode design --output_dir out/      (raw PDB, 8f-1)
    python -m abx_amd.design --workload L256 --num_samples 4 --mode trajectory --num_t 10 --output_dir out/   (synthetic complex)
    python -m abx_amd.design --pdb_file 6ct7_H_L_S.pdb --mode optimize --optimize_steps 10 --guidance --num_samples 100     (config 4)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m abx_amd.design \
        --pdb_list diffab_test.txt --pdb_dir pdbs/ --num_samples 100 --output_dir out/                        (a test set on 8 GPUs)

--pdb_file follows the reference's naming contract <code>_<heavy>_<light>_<antigen chains joined by |>.pdb (dataset.py:290-293);
it is read by abx_amd.data.antibody (plain-text parser, landmark IMGT locator, 16 A antigen patch, 32-residue window).
Several files (or --pdb_list, one name per line, relative to --pdb_dir) are processed one after the other.
Multi-GPU (one process per GPU under torch.distributed.run): the SAMPLES of every complex are sharded over the ranks
(sampler.shard_sample_ids: contiguous blocks; per-sample noise keys, so a sample's trajectory does not depend on where it runs),
every rank writes the PDB files of its own samples, and the designed sequences / pLDDT are gathered with one RCCL all_gather per
field (sampler.gather_results) for `<output_dir>/<complex>_designs.tsv`, written by rank 0.
Weights: a checkpoint with the reference's `model_state_dict`, or seeded random weights (no checkpoint ships with the reference)."""
import argparse
import os
from collections import OrderedDict

import torch

from . import features, sampler, synthetic
from .config import default_config, load_config
from .diffuser.full_diffuser import FullDiffuser
from .io import TrajectoryWriter, index_to_str_seq
from .model.abx import ScoreNetwork


def complex_list(pdb_files, pdb_list, pdb_dir):
    """The complexes of a run, in order: --pdb_file entries, then the lines of --pdb_list (blank lines and # comments skipped;
    '.pdb' appended when missing), both relative to --pdb_dir when given."""
    names = list(pdb_files or [])
    if pdb_list:
        with open(pdb_list) as f:
            for line in f:
                line = line.split('#')[0].strip()
                if line:
                    names.append(line if line.endswith('.pdb') else line + '.pdb')
    return [os.path.join(pdb_dir, n) if pdb_dir and not os.path.isabs(n) else n for n in names]


def sample_names(name, ids, num_samples):
    """Output stem of every sample: the complex name for a single sample, <code>-<sample id>_<chains> otherwise (global ids)."""
    if num_samples == 1:
        return [name for _ in ids]
    head, tail = name.split('_')[0], '_'.join(name.split('_')[1:])
    return [f'{head}-{i:03d}_{tail}' for i in ids]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--pdb_file', nargs='*', default=None, help='antibody-antigen complex(es), <code>_<H>_<L>_<antigen chains>.pdb')
    ap.add_argument('--pdb_list', default=None, help='text file with one complex name per line (a test-set index)')
    ap.add_argument('--pdb_dir', default=None, help='directory the names of --pdb_file / --pdb_list are relative to')
    ap.add_argument('--workload', default='L256', choices=sorted(synthetic.WORKLOADS), help='synthetic complex when no PDB is given')
    ap.add_argument('--num_samples', type=int, default=4)
    ap.add_argument('--mode', default='design', choices=['design', 'trajectory', 'optimize'])
    ap.add_argument('--optimize_steps', type=int, default=10, help='optimize mode: start the reverse process at t = steps / 100')
    ap.add_argument('--num_t', type=int, default=100)
    ap.add_argument('--generate_area', default='H3')
    ap.add_argument('--model_config', default=None, help='the reference config/config_model.json (default: built-in copy)')
    ap.add_argument('--ckpt', '--model', dest='ckpt', default=None, help='checkpoint with model_state_dict (default: seeded random weights)')
    ap.add_argument('--output_dir', default='design_out')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--device', default=None, help='default: cuda:<LOCAL_RANK>')
    ap.add_argument('--guidance', action='store_true', help='structural-violation guidance (clash + C-N bond terms, abx_clash_grad) on')
    ap.add_argument('--guidance_scale', type=float, nargs=2, default=[1.0, 1.0], metavar=('TRANS', 'ROT'),
                    help='step scales of the guidance gradients on the translation / rotation scores')
    ap.add_argument('--debug_one_gpu', action='store_true', help='debugging on a 1-GPU box: every rank uses cuda:0 and the gloo backend')
    a = ap.parse_args(argv)

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device(a.device if a.device else ('cuda:0' if a.debug_one_gpu else f'cuda:{local_rank}'))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group('gloo' if a.debug_one_gpu else 'nccl', rank=rank, world_size=world)      # "nccl" is RCCL on ROCm

    cfg = load_config(a.model_config) if a.model_config else default_config()
    diffuser = FullDiffuser.get(cfg.diffuser).to(dev)
    model = ScoreNetwork(cfg.model, diffuser)
    if a.ckpt:
        sd = torch.load(a.ckpt, map_location='cpu')['model_state_dict']
    else:
        sd = synthetic.random_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items()), seed=a.seed)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()

    guide = None
    if a.guidance:
        from .guidance import ViolationGuidance
        guide = ViolationGuidance(scale_trans=a.guidance_scale[0], scale_rot=a.guidance_scale[1])
    N = a.num_samples
    ids = sampler.shard_sample_ids(N, rank, world)             # the global sample ids this rank runs, for every complex
    n = len(ids)
    complexes = complex_list(a.pdb_file, a.pdb_list, a.pdb_dir) or [None]
    os.makedirs(a.output_dir, exist_ok=True)
    files = []
    for path in complexes:
        if path is not None:
            from .data.antibody import load_complex
            cb = load_complex(path, seed=a.seed)
            one = {k: v.to(dev) for k, v in cb.items() if torch.is_tensor(v)}
            cname = cb['name'][0]
            meta = {k: list(cb[k]) * n for k in ('str_heavy_seq', 'str_light_seq', 'antigen_origin_str_seq',
                                                 'antigen_origin_atom14_gt_positions', 'antigen_origin_atom14_gt_exists',
                                                 'antigen_origin_chain_ids')}
        else:
            w = synthetic.WORKLOADS[a.workload]
            cx = synthetic.make_complex(seed=a.seed + 1, **w)
            one = {k: v[None].to(dev) for k, v in cx.items()}
            nh, nl = w['L_heavy'], w['L_light']
            seq = cx['seq'].tolist()
            cname = f'{a.workload}_H_L_A'
            meta = dict(str_heavy_seq=[index_to_str_seq(seq[:nh])] * n, str_light_seq=[index_to_str_seq(seq[nh:nh + nl])] * n)
        meta['name'] = sample_names(cname, ids, N)
        L, Lab = one['seq'].shape[1], one['anchor_flag'].shape[1]
        if n > 0:
            raw = {k: v.expand(n, *v.shape[1:]).contiguous() for k, v in one.items()}
            batch = features.build_features(raw, diffuser, generate_area=a.generate_area,
                                            opt_step=a.optimize_steps if a.mode == 'optimize' else None,
                                            noise=features.per_sample_init_noise(ids, L, a.seed, dev))
            batch['_shared_context'] = True
            diffuser.seed = a.seed
            writer = TrajectoryWriter(meta, a.output_dir, multi=a.mode == 'trajectory')
            traj = sampler.sample_fn(batch, cfg, diffuser, model, mode=a.mode, num_t=a.num_t,
                                     sample_ids=torch.tensor(ids, device=dev, dtype=torch.int64), on_record=writer.submit, guidance=guide)
            torch.cuda.synchronize()
            files += writer.close()
            local = {'seq': traj[-1]['seq'], 'pLDDT': traj[-1]['pLDDT']}
        else:                                                   # more ranks than samples: join the gather with zero-row blocks
            local = {'seq': torch.zeros(0, Lab, dtype=torch.int64, device=dev), 'pLDDT': torch.zeros(0, Lab, device=dev)}
        if a.debug_one_gpu and world > 1:                       # gloo moves host tensors
            local = {k: v.cpu() for k, v in local.items()}
        res = sampler.gather_results(local, N, rank, world, group)
        if rank == 0:
            tsv = os.path.join(a.output_dir, f'{cname}_designs.tsv')
            with open(tsv, 'w') as f:
                f.write('sample\tmean_pLDDT\tantibody_sequence\n')
                for i in range(N):
                    f.write(f'{i}\t{float(res["pLDDT"][i].float().mean()):.3f}\t{index_to_str_seq(res["seq"][i].tolist())}\n')
            files.append(tsv)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    print(f'rank {rank}/{world}: {len(files)} files in {a.output_dir}')
    return files


if __name__ == '__main__':
    main()
