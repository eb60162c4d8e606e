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


def _write_designs(out_dir, cname, rows):
    """<out_dir>/<complex>_designs.tsv: (sample id, mean pLDDT, designed antibody sequence) per sample."""
    tsv = os.path.join(out_dir, f'{cname}_designs.tsv')
    with open(tsv, 'w') as f:
        f.write('sample\tmean_pLDDT\tantibody_sequence\n')
        for i, pl, toks in rows:
            f.write(f'{i}\t{pl:.3f}\t{index_to_str_seq(toks)}\n')
    return tsv


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
    ap.add_argument('--min_block', type=int, default=50, help='set-level schedule: samples per work unit (a complex is split into '
                    'num_samples // min_block blocks)')
    ap.add_argument('--shard_samples', action='store_true', help='several complexes on several ranks: shard the samples of EVERY complex over '
                    'the ranks (one gather per complex) instead of dealing (complex, sample block) units to the ranks')
    ap.add_argument('--force_collective', action='store_true', help='single rank: still initialise RCCL and run the final gather through a '
                    '1-rank all_gather (exercises the collective path on a 1-GPU box; same results)')
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
        area, steps = read_model_features(a.model_features)
        if area != a.generate_area and a.generate_area != ap.get_default('generate_area'):
            import warnings
            warnings.warn(f'--generate_area {a.generate_area} is overridden by the make_diffuser_features entry of --model_features ({area})')
        a.generate_area = area
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
    if world > 1 or a.force_collective:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29577')
            if a.debug_one_gpu:
                dist.init_process_group('gloo', rank=rank, world_size=world)
            else:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)      # "nccl" is RCCL on ROCm

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
    # jobs: (kind, reference to the complex, output directory, optimize step, inference.py layout?, directory of the ground-truth copy)
    jobs = []
    if a.name_idx:
        with open(a.name_idx) as f:
            names = [x.strip() for x in f if x.strip()]
        root = os.path.join(a.output_dir, a.mode)
        for step in (opt_steps if a.mode == 'optimize' else [None]):
            out = os.path.join(root, f'OPT-{step}') if a.mode == 'optimize' else root
            # inference.py:321-322, 354-355: the ground truth goes to <output_dir>/<mode>/reference/ in every mode (in optimize mode it
            # is re-written, unchanged, for every step: once is enough)
            jobs += [('npz', nm, out, step, True, os.path.join(root, 'reference') if step == opt_steps[0] or a.mode != 'optimize' else None)
                     for nm in names]
    else:
        for step in (opt_steps if a.mode == 'optimize' else [None]):
            out = os.path.join(a.output_dir, f'OPT-{step}') if (a.mode == 'optimize' and len(opt_steps) > 1) else a.output_dir
            jobs += [('pdb' if path is not None else 'synthetic', path, out, step, False, None)
                     for path in (complex_list(a.pdb_file, a.pdb_list, a.pdb_dir) or [None])]
    os.makedirs(a.output_dir, exist_ok=True)
    files = []
    import time
    del TIMINGS[:]

    loaded = {}

    def load_job(ji):
        """The complex of job ji (host tensors, read once per complex): dict(cb, cname, L, Lab, kind)."""
        kind, path = jobs[ji][0], jobs[ji][1]
        key = (kind, path)
        if key not in loaded:
            if kind in ('pdb', 'npz'):
                from .data.antibody import load_complex, load_complex_npz
                cb = load_complex(path, seed=a.seed) if kind == 'pdb' else load_complex_npz(a.data_dir, path, seed=a.seed)
                one = {k: v for k, v in cb.items() if torch.is_tensor(v)}
                loaded[key] = dict(cb=cb, one=one, cname=cb['name'][0], L=one['seq'].shape[1], Lab=one['anchor_flag'].shape[1])
            else:
                w = synthetic.WORKLOADS[a.workload]
                cx = synthetic.make_complex(seed=a.seed + 1, **w)
                loaded[key] = dict(cb=None, one={k: v[None] for k, v in cx.items()}, cname=f'{a.workload}_H_L_A', L=cx['seq'].shape[0],
                                   Lab=cx['anchor_flag'].shape[0], w=w, seq=cx['seq'].tolist())
        return loaded[key]

    # ---- who runs what.  One complex, or fewer (complex, 50-sample block) units than ranks: the samples of every complex are sharded
    # over the ranks and gathered per complex.  A set of complexes (BASELINE configs 3 / 4): whole units are dealt to the ranks
    # longest-first (cost ~ L^3 x samples), each GPU runs batches of >= 50 samples (0.98 of the 100-sample rate per GPU instead of the
    # 0.91 of 12-13-sample shards), and ONE gather of the designs table closes the set.  Per-sample noise keys make
    # a sample's trajectory independent of where and with whom it runs, so both schemes write the same files.
    plan = None
    if (world > 1 or a.force_collective) and len(jobs) >= 2 and not a.shard_samples:
        if a.min_block < 1:
            raise SystemExit('--min_block must be >= 1')
        plan = sampler.plan_work_units([float(load_job(ji)['L']) ** 3 for ji in range(len(jobs))], N, world, min_block=a.min_block, force=a.force_collective)
    if plan is None:
        work = [(ji, sampler.shard_sample_ids(N, rank, world)) for ji in range(len(jobs))]
    else:
        work = plan[rank]
        if a.verbose or rank == 0:
            print(f'set-level schedule: {sum(len(p) for p in plan)} units of >= {min(a.min_block, N)} samples over {world} ranks; '
                  f'rank {rank} runs {[(jobs[ji][1], len(ids_)) for ji, ids_ in work]}')
    set_rows = []                                               # set-level mode: (job, sample id, mean pLDDT, Lab, tokens...) rows of this rank
    maxLab = max([load_job(ji)['Lab'] for ji in range(len(jobs))]) if plan is not None else 0

    ref_written = set()
    for ji, ids in work:
        kind, path, out_dir, opt_step, ref_layout, ref_dir = jobs[ji]
        n = len(ids)
        t_job = time.perf_counter()
        os.makedirs(out_dir, exist_ok=True)
        J = load_job(ji)
        cname, L, Lab = J['cname'], J['L'], J['Lab']
        one = {k: v.to(dev) for k, v in J['one'].items()}
        if kind in ('pdb', 'npz'):
            cb = J['cb']
            meta = {k: list(cb[k]) * n for k in ('str_heavy_seq', 'str_light_seq', 'antigen_origin_str_seq',
                                                 'antigen_origin_atom14_gt_positions', 'antigen_origin_atom14_gt_exists',
                                                 'antigen_origin_chain_ids')}
            # the "reference batch" = the ground-truth antibody with pLDDT 100, once per complex: by rank 0 when the samples of the
            # complex are sharded, by the rank that runs the complex's first block under the set-level schedule
            first_block = bool(ids) and ids[0] == 0
            if ref_layout and ref_dir is not None and (rank == 0 if plan is None else first_block) and (cname, ref_dir) not in ref_written:
                from .io import postprocess_trajectory
                ref_written.add((cname, ref_dir))
                ref_meta = {k: list(cb[k]) for k in meta}
                ref_meta['name'] = [cname]
                files += postprocess_trajectory(ref_meta, [{'seq': one['seq'][:, :Lab], 'atom14_results': one['atom14_gt_positions'][:, :Lab],
                                                            'pLDDT': torch.full((1, Lab), 100.0), 'time': 0.0}], ref_dir)
        else:
            w = J['w']
            nh, nl = w['L_heavy'], w['L_light']
            meta = dict(str_heavy_seq=[index_to_str_seq(J['seq'][:nh])] * n, str_light_seq=[index_to_str_seq(J['seq'][nh:nh + nl])] * n)
        if ref_layout:                                      # inference.py:363-367: <k:04d>/<name>.pdb
            meta['name'] = [cname] * n
            meta['subdir'] = [f'{i:04d}' for i in ids]
        else:
            meta['name'] = sample_names(cname, ids, N)
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
                                sampling_s=t_samp - t_feat, writer_tail_s=t_done - t_samp, files=len(new_files),
                                range_fallbacks=len(traj[-1].get('range_fallbacks', [])),
                                range_sticky_ops=list(traj[-1].get('range_sticky_ops', []))))
            local = {'seq': traj[-1]['seq'], 'pLDDT': traj[-1]['pLDDT']}
        else:                                                   # more ranks than samples: join the gather with zero-row blocks
            local = {'seq': torch.zeros(0, Lab, dtype=torch.int64, device=dev), 'pLDDT': torch.zeros(0, Lab, device=dev)}
        if plan is not None:
            row = torch.zeros(n, 4 + maxLab, dtype=torch.float64)
            row[:, 0], row[:, 1], row[:, 3] = ji, torch.tensor(ids, dtype=torch.float64), Lab
            row[:, 2] = local['pLDDT'].float().mean(1).double().cpu()       # (the float32 mean of the sample-sharded path: same TSV digits)
            row[:, 4:4 + Lab] = local['seq'].double().cpu()
            set_rows.append(row)
            continue
        if a.debug_one_gpu and world > 1:                       # gloo moves host tensors
            local = {k: v.cpu() for k, v in local.items()}
        res = sampler.gather_results(local, N, rank, world, group, force=a.force_collective)
        if rank == 0:
            files.append(_write_designs(out_dir, cname, [(i, float(res['pLDDT'][i].float().mean()), res['seq'][i].tolist()) for i in range(N)]))
    if plan is not None:
        # ---- the one collective of the set: every rank's rows of the designs table (counts known from the common plan)
        table = torch.cat(set_rows, 0) if set_rows else torch.zeros(0, 4 + maxLab, dtype=torch.float64)
        if not (a.debug_one_gpu and world > 1):
            table = table.to(dev)
        counts = [sum(len(ids_) for _, ids_ in p) for p in plan]
        full = sampler.gather_rows(table, counts, rank, world, group, force=a.force_collective).cpu()
        if rank == 0:
            for ji in range(len(jobs)):
                rows = full[full[:, 0] == ji]
                rows = rows[torch.argsort(rows[:, 1])]
                assert rows.shape[0] == N, (jobs[ji][1], rows.shape)
                files.append(_write_designs(jobs[ji][2], load_job(ji)['cname'],
                                            [(int(r[1]), float(r[2]), r[4:4 + int(r[3])].long().tolist()) for r in rows]))
    if world > 1 or a.force_collective:
        import torch.distributed as dist
        dist.barrier()
        if a.force_collective and world == 1:
            dist.destroy_process_group()
    print(f'rank {rank}/{world}: {len(files)} files in {a.output_dir}')
    return files


if __name__ == '__main__':
    main()
