"""End-to-end driver in the shape of the reference's design.py / inference.py main (design.py:277-375, inference.py:59-82,
275-392): build the diffuser and the score network, featurise each complex, run the reverse diffusion for `num_samples` samples
and write the PDB files (per step in trajectory mode, asynchronously).

    python -m abx_amd.design --pdb_file 6ct7_H_L_S.pdb --num_samples 100 --mode design --output_dir out/      (raw PDB, 8f-1)
    python -m abx_amd.design --workload L256 --num_samples 4 --mode trajectory --num_t 10 --output_dir out/   (synthetic complex)
    python -m abx_amd.design --pdb_file 6ct7_H_L_S.pdb --mode optimize --optimize_steps 10 --guidance --num_samples 100     (config 4)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m abx_amd.design \
        --pdb_list diffab_test.txt --pdb_dir pdbs/ --num_samples 100 --output_dir out/                        (a test set on 8 GPUs)

    python -m abx_amd.design --name_idx diffab_test.idx --data_dir npz/ --model_features config_data_feature.json \
        --model abx_diffab.ckpt --model_config config_model.json --gpu_list 0 1 2 3 4 5 6 7 --num_samples 100 --output_dir out/
                                                                     (inference.py's own command line: BASELINE configs 3 / 4)

--name_idx / --data_dir / --model_features / --gpu_list / --batch_size are the arguments of the reference's inference.py
(inference.py:398-416): one complex name per line, <data_dir>/<name>.npz in the `make_pdb_npz` schema (abx_amd.data.antibody.
load_complex_npz), the feature-pipeline JSON (its make_diffuser_features entry supplies generate_area and, in optimize mode, the
list of optimize_steps that is looped over like inference.py:310-345), and the output layout of inference.py:
<output_dir>/<mode>[/OPT-<step>]/reference/<name>.pdb (the ground truth) and .../<k:04d>/<name>.pdb for sample k.  --gpu_list with
more than one entry starts one rank per listed GPU through torch.distributed.run (the reference spawns one worker per GPU that all
repeat the same work, inference.py:376-381; here the samples are sharded).  --batch_size (complexes per DataLoader batch upstream) is
accepted and ignored: complexes run one after the other with all local samples of a complex in one batch.
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


TIMINGS = []      # one dict per complex of the last main() call: seconds spent reading / featurising, sampling, waiting for the writer


def read_model_features(path):
    """The make_diffuser_features entry of a feature-pipeline JSON (config/config_data_feature.json): (generate_area, optimize_steps)."""
    import json
    with open(path, encoding='utf-8') as f:
        feats = json.load(f)
    for name, opts in feats:
        if 'diffuse' in name:
            return opts.get('generate_area', 'H3'), list(opts.get('optimize_steps', []))
    return 'H3', []


def _relaunch_on_gpus(gpu_list, argv):
    """--gpu_list a b c ... outside torch.distributed.run: one rank per listed GPU on 127.0.0.1."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES=','.join(str(g) for g in gpu_list), HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={len(gpu_list)}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), '-m', 'abx_amd.design'] + list(argv)
    return subprocess.call(cmd, env=env)


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
    # the reference's inference.py arguments (inference.py:398-416)
    ap.add_argument('--name_idx', default=None, help='text file with one complex name per line (entries of --data_dir)')
    ap.add_argument('--data_dir', default=None, help='directory of <name>.npz files in the make_pdb_npz schema')
    ap.add_argument('--model_features', default=None, help='feature-pipeline JSON (config/config_data_feature.json): generate_area, optimize_steps')
    ap.add_argument('--gpu_list', type=int, nargs='+', default=None, help='GPUs to use, one rank each (default: the launcher\'s ranks / GPU 0)')
    ap.add_argument('--batch_size', type=int, default=1, help='accepted for compatibility (complexes per batch upstream); ignored')
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--exact_gemm', action='store_true', help='exact fp32-MFMA kernels instead of the split-f16 ones (slower; the remedy when '
                    'the sampler reports non-finite frames: an activation beyond the split kernels\' range)')
    a = ap.parse_args(argv)
    if a.exact_gemm:
        from abx_amd import ops
        ops.GEMM_EXACT = True
    if a.gpu_list and len(a.gpu_list) > 1 and 'WORLD_SIZE' not in os.environ:
        import sys
        rc_ = _relaunch_on_gpus(a.gpu_list, sys.argv[1:] if argv is None else argv)
        if rc_ != 0:
            raise SystemExit(rc_)
        return []
    if a.name_idx and not a.data_dir:
        ap.error('--name_idx needs --data_dir')
    opt_steps = [a.optimize_steps]
    if a.model_features:
        a.generate_area, steps = read_model_features(a.model_features)
        if a.mode == 'optimize' and steps:
            opt_steps = steps

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpu_list and len(a.gpu_list) == 1 and not a.device and 'WORLD_SIZE' not in os.environ:
        a.device = f'cuda:{a.gpu_list[0]}'
    if a.device in ('gpu', 'cpu'):                      # the reference's --device choices
        if a.device == 'cpu':
            raise SystemExit('abx_amd runs on an MI355X only: there is no CPU path (the oracle under oracle/ is test infrastructure)')
        a.device = None
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
    # jobs: (kind, reference to the complex, output directory, optimize step, inference.py layout?)
    jobs = []
    if a.name_idx:
        with open(a.name_idx) as f:
            names = [x.strip() for x in f if x.strip()]
        root = os.path.join(a.output_dir, a.mode)
        for step in (opt_steps if a.mode == 'optimize' else [None]):
            out = os.path.join(root, f'OPT-{step}') if a.mode == 'optimize' else root
            jobs += [('npz', nm, out, step, True) for nm in names]
    else:
        for step in (opt_steps if a.mode == 'optimize' else [None]):
            out = os.path.join(a.output_dir, f'OPT-{step}') if (a.mode == 'optimize' and len(opt_steps) > 1) else a.output_dir
            jobs += [('pdb' if path is not None else 'synthetic', path, out, step, False)
                     for path in (complex_list(a.pdb_file, a.pdb_list, a.pdb_dir) or [None])]
    os.makedirs(a.output_dir, exist_ok=True)
    files = []
    import time
    del TIMINGS[:]
    for kind, path, out_dir, opt_step, ref_layout in jobs:
        t_job = time.perf_counter()
        os.makedirs(out_dir, exist_ok=True)
        if kind in ('pdb', 'npz'):
            from .data.antibody import load_complex, load_complex_npz
            cb = load_complex(path, seed=a.seed) if kind == 'pdb' else load_complex_npz(a.data_dir, path, seed=a.seed)
            one = {k: v.to(dev) for k, v in cb.items() if torch.is_tensor(v)}
            cname = cb['name'][0]
            meta = {k: list(cb[k]) * n for k in ('str_heavy_seq', 'str_light_seq', 'antigen_origin_str_seq',
                                                 'antigen_origin_atom14_gt_positions', 'antigen_origin_atom14_gt_exists',
                                                 'antigen_origin_chain_ids')}
            if ref_layout and rank == 0:
                # inference.py:346-361: the "reference batch" = the ground-truth antibody with pLDDT 100, written once per complex
                from .io import postprocess_trajectory
                Lab0 = one['anchor_flag'].shape[1]
                ref_meta = {k: list(cb[k]) for k in meta}
                ref_meta['name'] = [cname]
                files += postprocess_trajectory(ref_meta, [{'seq': one['seq'][:, :Lab0], 'atom14_results': one['atom14_gt_positions'][:, :Lab0],
                                                            'pLDDT': torch.full((1, Lab0), 100.0), 'time': 0.0}],
                                                os.path.join(out_dir, 'reference'))
        else:
            w = synthetic.WORKLOADS[a.workload]
            cx = synthetic.make_complex(seed=a.seed + 1, **w)
            one = {k: v[None].to(dev) for k, v in cx.items()}
            nh, nl = w['L_heavy'], w['L_light']
            seq = cx['seq'].tolist()
            cname = f'{a.workload}_H_L_A'
            meta = dict(str_heavy_seq=[index_to_str_seq(seq[:nh])] * n, str_light_seq=[index_to_str_seq(seq[nh:nh + nl])] * n)
        if ref_layout:                                      # inference.py:363-367: <k:04d>/<name>.pdb
            meta['name'] = [cname] * n
            meta['subdir'] = [f'{i:04d}' for i in ids]
        else:
            meta['name'] = sample_names(cname, ids, N)
        L, Lab = one['seq'].shape[1], one['anchor_flag'].shape[1]
        if n > 0:
            raw = {k: v.expand(n, *v.shape[1:]).contiguous() for k, v in one.items()}
            batch = features.build_features(raw, diffuser, generate_area=a.generate_area,
                                            opt_step=opt_step if a.mode == 'optimize' else None,
                                            noise=features.per_sample_init_noise(ids, L, a.seed, dev))
            batch['_shared_context'] = True
            diffuser.seed = a.seed
            writer = TrajectoryWriter(meta, out_dir, multi=a.mode == 'trajectory')
            torch.cuda.synchronize()
            t_feat = time.perf_counter()
            traj = sampler.sample_fn(batch, cfg, diffuser, model, mode=a.mode, num_t=a.num_t,
                                     sample_ids=torch.tensor(ids, device=dev, dtype=torch.int64), on_record=writer.submit, guidance=guide)
            torch.cuda.synchronize()
            t_samp = time.perf_counter()
            new_files = writer.close()
            files += new_files
            t_done = time.perf_counter()
            TIMINGS.append(dict(complex=cname, L=int(L), samples=n, mode=a.mode, opt_step=opt_step, read_and_featurise_s=t_feat - t_job,
                                sampling_s=t_samp - t_feat, writer_tail_s=t_done - t_samp, files=len(new_files)))
            local = {'seq': traj[-1]['seq'], 'pLDDT': traj[-1]['pLDDT']}
        else:                                                   # more ranks than samples: join the gather with zero-row blocks
            local = {'seq': torch.zeros(0, Lab, dtype=torch.int64, device=dev), 'pLDDT': torch.zeros(0, Lab, device=dev)}
        if a.debug_one_gpu and world > 1:                       # gloo moves host tensors
            local = {k: v.cpu() for k, v in local.items()}
        res = sampler.gather_results(local, N, rank, world, group)
        if rank == 0:
            tsv = os.path.join(out_dir, f'{cname}_designs.tsv')
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
