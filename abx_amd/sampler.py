"""Reverse-time sampling driver — the build's counterpart of the reference's `sample_fn` (inference.py:166-273, identical in
design.py:182-275) plus the multi-GPU sample sharding the reference only stubs (inference.py:59-82).

Loop semantics reproduced: self-conditioning warm-up call at the first grid point with fp32 `t`;
100 grid points / 99 reverse steps, `t` float64 inside the loop; `get_prev` after every model call; the model mutates
`batch['seq_t']` before `diffuser.reverse` reads it; the last grid point takes rigids / seq_0 from the model with the stale
`batch['t']`; optimize mode keeps grid points <= opt_step/num_t + 1e-8; trajectory mode keeps every step.
Differences: schedule scalars never leave the device (no `.tolist()` syncs), pLDDT stays on the device until the end.
"""
import numpy as np
import torch

from abx_amd.model.abx import get_prev


def set_t_feats(feats, diffuser, t, ones):
    """inference.py:166-171."""
    feats['t'] = t * ones
    rs, ts = diffuser.score_scaling(feats['t'])
    feats['rot_score_scaling'] = rs * ones
    feats['trans_score_scaling'] = ts * ones
    return feats


def sample_fn(data_init, config, diffuser, model, mode='design', num_t=100, min_t=0.01, center=True, self_condition=True,
              noise_scale=1.0, eps=1e-8, noise_fn=None, sample_ids=None, on_step=None, on_record=None, guidance=None, use_graph=False):
    """Returns the trajectory: list of dicts {seq (B,Lab) i64, atom14_results (B,Lab,14,3), pLDDT (B,Lab), time,
    rigids_t, seq_t}; only the last element unless mode == 'trajectory'.  All tensors stay on the device.
    on_record(rec): called for every element that enters the trajectory, e.g. `abx_amd.io.TrajectoryWriter.submit` to dump the
    per-step PDB files asynchronously (device->host copy on a side stream, formatting and disk I/O on a worker thread).
    guidance: None (the reference's un-guided sampler, bit-identical code path) or an abx_amd.guidance.ViolationGuidance whose
    clash / bond gradients on the predicted structure are subtracted from the scores before the reverse step.
    use_graph: record the step into two hipGraphs (abx_amd.graph.GraphedSteps) after one eager step and replay them; needs the
    device noise generator (noise_fn None) and gives the same results as the eager loop."""
    model_conf = config.model
    sc_conf = model_conf.heads.diffusion_module
    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_init.items()}
    device = batch['rigids_t'].device
    bb_mask = batch['atom14_gt_exists'][..., 0]
    diffuse_mask = ((1 - batch['fixed_mask']) * bb_mask).to(torch.int32)
    Lab = batch['anchor_flag'].shape[1]
    B = batch['rigids_t'].shape[0]
    ones = torch.ones(B, device=device, dtype=torch.float32)
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = float(np.float32(1 / num_t))                 # value of torch.tensor(1/num_t) (fp32)
    if mode == 'optimize':
        opt_step = float(batch['t'][0])
        if opt_step < 1.0:
            steps = steps[steps <= opt_step + eps]
    batch.pop('_static', None)          # trajectory-invariant embeddings: rebuilt once per trajectory from THIS batch
    log_start = len(getattr(model, 'range_log', None) or [])     # (the model's range log is cumulative: report this trajectory's part)
    traj = []
    with torch.no_grad():
        if sc_conf.embed.embed_self_conditioning and self_condition and len(steps) > 0:
            batch = set_t_feats(batch, diffuser, float(steps[0]), ones)
            out = model(batch)
            batch.update(get_prev(batch, out, model_conf))
        dm_f = diffuse_mask.to(torch.float32)
        graphed = None
        if use_graph:
            assert noise_fn is None, 'graph replay uses the device Philox generator (no injected noise)'
            from abx_amd.graph import GraphedSteps
            graphed = GraphedSteps(batch, config, diffuser, model, diffuse_mask, dt, sample_ids, center, noise_scale, guidance)
        for k, t in enumerate(steps):
            if t > min_t and graphed is not None:
                out = graphed.run(k, t)
                rigids_t, seq_t = batch['rigids_t'], batch['seq_t']
            elif t > min_t:
                t_ = torch.full((B,), float(t), device=device, dtype=torch.float64)
                batch = set_t_feats(batch, diffuser, t_, ones)
                out = model(batch)
                f = out['heads']['folding']
                if sc_conf.embed.embed_self_conditioning:
                    batch.update(get_prev(batch, out, model_conf))
                rot_score, trans_score = f['rot_score'], f['trans_score']
                if guidance is not None:
                    rot_score, trans_score = guidance(batch, out, rot_score, trans_score, diffuse_mask)
                rigids_t, seq_t = diffuser.reverse(
                    rigid_t=batch['rigids_t'], seq_t=batch['seq_t'], rot_score=rot_score, trans_score=trans_score,
                    logits_t=out['heads']['sequence_module']['logits'], diffuse_mask=diffuse_mask, t=t_, dt=dt,
                    center=center, noise_scale=noise_scale, noise=noise_fn(k) if noise_fn else None,
                    sample_ids=sample_ids, step=k)
            else:
                out = model(batch)
                rigids_t = out['heads']['folding']['rigids']
                seq_t = out['heads']['sequence_module']['seq_0']
            batch['rigids_t'] = rigids_t
            batch['seq_t'] = seq_t
            pl = out['heads']['predicted_lddt']['pLDDT']
            pl = torch.sum(pl * dm_f, dim=1) / torch.sum(dm_f, dim=1)
            rec = {'seq': torch.clamp(seq_t[:, :Lab], min=0, max=19).long(),
                   'atom14_results': out['heads']['folding']['final_atom14_positions'][:, :Lab],
                   'pLDDT': torch.tile(pl[:, None], (1, Lab)), 'time': float(t), 'rigids_t': rigids_t, 'seq_t': seq_t}
            if mode == 'trajectory' or k == len(steps) - 1:
                traj.append({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in rec.items()})
                if on_record is not None:
                    # finiteness before a file is written: on the first record, every 10th and the last one (a host synchronisation each;
                    # an out-of-range activation never gets here: ScoreNetwork repeats that pass on the exact kernels)
                    if len(traj) == 1 or len(traj) % 10 == 0 or k == len(steps) - 1:
                        check_finite(traj[-1]['rigids_t'], traj[-1]['atom14_results'], what=f'step {k} (t = {float(t):.4f})')
                    on_record(traj[-1])
            if on_step is not None:
                on_step(k, t, batch, out)
    if not traj:
        raise ValueError(f'no grid point to sample: optimize step t = {float(batch["t"][0]):.4f} lies below min_t = {min_t} '
                         f'(grid of {num_t} points on [{min_t}, 1])')
    check_finite(batch['rigids_t'], traj[-1]['atom14_results'])
    log = getattr(model, 'range_log', None)
    if log and len(log) > log_start:
        traj[-1]['range_fallbacks'] = list(log[log_start:])     # calls of THIS trajectory repeated with an op class on the exact kernels
        # the classes that ended up sticky-exact for this complex: a digest that differs between two placements of the same samples
        # (different batch mates in a flagged call, a different flag history per rank) is explained by these two entries
        traj[-1]['range_sticky_ops'] = list(getattr(model, 'range_sticky_ops', []))
    return traj


def check_finite(*tensors, what='the end of the trajectory'):
    """One reduction + host sync.  The split-f16 contractions turn an operand beyond their range (include/abx_hip.h, "Split-f16
    operands") into NaN rows instead of wrong numbers, the network repeats such a pass on the exact kernels by itself
    (model/abx.py, AbxGemm.range_flag); what is still not finite here comes from the inputs or the weights."""
    ok = torch.stack([torch.isfinite(t).all() for t in tensors if t is not None and t.numel() > 0] or [torch.tensor(True)]).all()
    if not bool(ok):
        raise FloatingPointError(f'non-finite frames / coordinates at {what}: the inputs or the weights are not finite (an activation '
                                 'beyond the range of the split-f16 kernels is handled inside ScoreNetwork: the pass is repeated on the '
                                 'exact fp32-MFMA kernels, see ScoreNetwork.range_log)')


# -------------------------------------------------------------------------------------------------------------------
# multi-GPU: independent samples shard over ranks; ONE gather of the final results
# -------------------------------------------------------------------------------------------------------------------
def shard_sample_ids(num_samples, rank, world_size):
    """Contiguous blocks, sizes differing by at most one; returns the global sample ids of this rank."""
    base, rem = divmod(num_samples, world_size)
    start = rank * base + min(rank, rem)
    n = base + (1 if rank < rem else 0)
    return list(range(start, start + n))


def gather_results(local, num_samples, rank, world_size, group=None, force=False):
    """All-gather of per-sample results {name: tensor (n_local, ...)} into tensors (num_samples, ...) ordered by sample id.
    One collective per field on padded equal-size blocks (RCCL all_gather over xGMI; gloo in the CPU tests).
    force: run the collective on a single rank too (exercises the RCCL path on a 1-GPU box; same values)."""
    import torch.distributed as dist
    if world_size == 1 and not force:
        return local
    nmax = -(-num_samples // world_size)
    out = {}
    for name, t in local.items():
        pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world_size)]
        dist.all_gather(parts, pad, group=group)
        rows = []
        for r in range(world_size):
            n = len(shard_sample_ids(num_samples, r, world_size))
            rows.append(parts[r][:n])
        out[name] = torch.cat(rows, dim=0)
    return out


def plan_work_units(costs, num_samples, world_size, min_block=50, force=False):
    """Set-level scheduling (BASELINE configs 3 / 4: a list of complexes x num_samples on several GPUs; the reference walks the
    complexes one after the other, inference.py:296-373).  Work units = (complex, block of >= min_block samples; all samples when
    there are fewer than 2 * min_block), cost = costs[complex] * block size (costs ~ L^3), dealt to the ranks longest-first onto the
    least loaded rank.  Every rank computes the same plan from the same inputs: no communication.  Returns a list per rank of
    (complex index, [global sample ids]) in complex order, or None when there are fewer units than ranks (then sharding the samples
    of each complex keeps every GPU busy)."""
    units = []
    for j, c in enumerate(costs):
        nb = max(1, num_samples // max(1, int(min_block)))
        for r in range(nb):
            ids = shard_sample_ids(num_samples, r, nb)
            if ids:
                units.append((float(c) * len(ids), j, ids))
    if (world_size <= 1 and not force) or len(units) < world_size:      # (force: the plan of a single rank, to exercise the path)
        return None
    load = [0.0] * world_size
    plan = [[] for _ in range(world_size)]
    for cost, j, ids in sorted(units, key=lambda u: (-u[0], u[1], u[2][0])):
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += cost
        plan[r].append((j, ids))
    return [sorted(p, key=lambda u: (u[0], u[1][0])) for p in plan]


def gather_rows(table, rows_per_rank, rank, world_size, group=None, force=False):
    """ONE all-gather of a per-rank table (n_rank, W) whose row counts every rank already knows (rows_per_rank, from the common
    plan): blocks padded to the largest count, the padding dropped on arrival.  Returns the (sum rows, W) table in rank order."""
    import torch.distributed as dist
    assert table.shape[0] == rows_per_rank[rank], (table.shape, rows_per_rank, rank)
    if world_size == 1 and not force:
        return table
    nmax = max(max(rows_per_rank), 1)
    pad = torch.zeros((nmax,) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
    pad[:table.shape[0]] = table
    parts = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][:rows_per_rank[r]] for r in range(world_size)], dim=0)


def design_samples(complex_feats, config, diffuser, model, num_samples, rank=0, world_size=1, mode='design', num_t=100,
                   seed=0, group=None, features_fn=None, generate_area='H3', opt_step=None):
    """num_samples independent designs of ONE complex, sharded over ranks, gathered at the end.
    complex_feats: un-batched raw tensors of the complex on the device (abx_amd.synthetic.make_complex layout).
    features_fn(batch, sample_ids) builds the per-sample diffusion features; the default is the inference feature pipeline with
    init noise keyed by (seed, sample id) (features.per_sample_init_noise), so that — together with the per-sample Philox keys of
    the reverse step — a sample's whole trajectory is independent of the rank and of the batch it runs in.
    Ranks without samples (world_size > num_samples) still take part in the gather, with zero-row blocks."""
    from abx_amd import features
    ids = shard_sample_ids(num_samples, rank, world_size)
    n = len(ids)
    device = next(iter(v for v in complex_feats.values() if torch.is_tensor(v))).device
    L, Lab = complex_feats['seq'].shape[0], complex_feats['anchor_flag'].shape[0]
    if n > 0:
        batch = {k: v[None].expand(n, *v.shape).contiguous() for k, v in complex_feats.items()}
        sid = torch.tensor(ids, device=device, dtype=torch.int64)
        if features_fn is None:
            batch = features.build_features(batch, diffuser, generate_area=generate_area, opt_step=opt_step,
                                            noise=features.per_sample_init_noise(ids, L, seed, device))
        else:
            batch = features_fn(batch, sid)
        batch['_shared_context'] = True
        diffuser.seed = seed
        traj = sample_fn(batch, config, diffuser, model, mode=mode, num_t=num_t, sample_ids=sid)
        last = traj[-1]
        local = {'rigids': last['rigids_t'].double(), 'seq': last['seq'], 'atom14': last['atom14_results'], 'pLDDT': last['pLDDT']}
    else:
        local = {'rigids': torch.zeros(0, L, 7, dtype=torch.float64, device=device),
                 'seq': torch.zeros(0, Lab, dtype=torch.int64, device=device),
                 'atom14': torch.zeros(0, Lab, 14, 3, device=device), 'pLDDT': torch.zeros(0, Lab, device=device)}
    return gather_results(local, num_samples, rank, world_size, group)
