#!/usr/bin/env python
"""Benchmark of the AbX reverse-diffusion sampling hot path on MI355X (BASELINE.json metric: diffusion-steps/sec).

  python bench.py --gpus N --steps K --warmup W [--scaling strong|weak] [--workload L352|6ct7like|6qd7like] [--samples S]

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or — when WORLD_SIZE is
not set — this script starts them itself through torch.distributed.run on 127.0.0.1 and relays rank 0's JSON line.

A "step" = one iteration of the reference's reverse loop (inference.py:213-268) for the samples of ONE complex that a rank
holds: ScoreNetwork call (2 recycles + final pass = 3 network passes) + get_prev + FullDiffuser.reverse.  All inputs are
resident in HBM when the timed region starts.  value = (samples on all ranks) * K / max-over-ranks(time)  [sample-steps / s].

Scaling.  The metric is "100 samples of one ~350-residue complex at 1/2/4/8 GPUs": the S = 100 samples are sharded over the
ranks in contiguous blocks (sampler.shard_sample_ids; 12-13 samples per GPU at N = 8) -> "scaling": "strong" (default).  There
is no collective inside a step; ONE all-gather of the final results (sampler.gather_results, RCCL over xGMI) runs after the
timed loop and is reported as `gather_ms`.  `--scaling weak` gives every rank its own S samples instead; at N > 1 the default
run also times that variant for a few steps and reports it under "weak" on the same JSON line.

Workloads (abx_amd.synthetic.WORKLOADS): 'L352' (Lab 228 + antigen 124, BASELINE's nominal "~350-res complex", the headline),
'6ct7like' (L = 230, the cropped 6ct7 complex of BASELINE configs 1-2), '6qd7like' (L = 261, config 5: use --samples 32).
Seeded random weights (no trained checkpoint exists offline), ESM disabled, device Philox noise, fp32 compute with float64
diffuser state as the reference.

Extra objects on the JSON line: "roofline" for the dominant kernel (measured live with HIP events on the launch stream) and
"cpu_baseline": the oracle (CPU port of the same step) on sample 0 of the same complex, 1 warm-up + 3 timed steps on the
physical cores of this box, with the HIP-vs-oracle parity of that very call ("parity").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TF = 2516.6      # dense f16 MFMA peak (v_mfma_f32_32x32x16_f16: 1024 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz)
MFMA_SPLIT_PEAK_TF = MFMA_F16_PEAK_TF / 3.0    # one fp32 product = 3 f16 MFMA products (split-f16, csrc/gemm3.hip)
WEIGHT_SEED = 7


def algorithmic_bytes_per_sample_step(L):
    """3 passes x 9893 channel-passes over the L^2 pair grid x 4 B."""
    return 3 * 9893 * 4.0 * L * L


def algorithmic_flops_per_sample_step(L):
    return 3 * 2 * (1.064e6 * L * L + 1024.0 * L ** 3 + 12.4e6 * L + 1088.0 * L * L)


def library_stamp():
    """Size and sha256 of the libabx_hip.so this process runs on (what profiles/pmc_traffic.json is stamped with)."""
    import hashlib
    from abx_amd import _lib
    path = _lib.library_path()
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    from tools.pmc_traffic import source_sha256
    return {'lib_bytes': os.path.getsize(path), 'lib_sha256': h.hexdigest(), 'src_sha256': source_sha256(ROOT)}


def model_parameter_shapes(cfg):
    """Names and shapes of the 190 ScoreNetwork parameters, from the module tree itself (the checkpoint contract)."""
    from abx_amd.model.abx import ScoreNetwork
    m = ScoreNetwork(cfg.model, None)
    return OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())


class OpTimer:
    """Per-op HIP-event timing on the current (launch) stream, used for ONE instrumented step after the timed region."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []
        self.saved = {}

    def _sig(self, name, args, kw):
        if name == 'gemm':
            A, B, C = args[0], args[1], args[2]
            if A.dtype == torch.int16:          # k-tiled f16 operand images (b, K/16, 2, rows, 16): the tri-mul contraction
                if A.dim() == 6:        # two-level batch (channel slices)
                    nb, M, K, N = A.shape[0] * A.shape[1], A.shape[4], A.shape[2] * 16, B.shape[4]
                else:
                    nb, M, K, N = A.shape[0], A.shape[3], A.shape[1] * 16, B.shape[3]
                kern = self.ops.gemm_kernel_name(M, N, K, nb, split=True, a_split=True, exact=kw.get('exact'))
                return kern, 2.0 * nb * M * N * K, nb * (4.0 * (M + N) * K + 4.0 * M * N)
            nb = A.shape[0] if A.dim() == 3 else 1
            M, K = A.shape[-2], A.shape[-1]
            N = B.shape[-1]
            c_planes = C.dtype == torch.int16
            if kw.get('mlp') is not None:       # fused transition / gated attention tail: both layers' flops, rows in + rows out of HBM
                N2 = kw['mlp'][0].shape[2]
                if kw.get('gate') is not None:  # (z rows for the gate and the residual once, the attention output, the rows out)
                    return 'gemm3_gtail_kernel', 2.0 * nb * M * N * (K + N2), 4.0 * nb * M * (K + N + N2) + 4.0 * N * (K + N2)
                return 'gemm3_mlp_kernel<2>', 2.0 * nb * M * N * (K + N2), 4.0 * nb * M * (K + N2) + 4.0 * N * (K + N2)
            desc = self.saved['gemm'](*args, **dict(kw, defer=True))            # the filled descriptor, nothing launched
            kas = self.ops.gemm_as_kernel_name(desc)
            if kas is not None:                 # A-stationary kernel (glu: N / 2 output channels as 4-byte operand-image elements)
                return kas, 2.0 * nb * M * N * K, 4.0 * nb * M * K + 4.0 * nb * M * (N // 2 if kw.get('glu') else N) + 4.0 * K * N
            kern = self.ops.gemm_kernel_name(M, N, K, nb, A.stride(-1) == 1, B.stride(-1) == 1, c_planes or C.stride(-1) != 1,
                                             split=kw.get('B3') is not None, exact=kw.get('exact'), dual=kw.get('dual') is not None,
                                             out_ln=kw.get('out_ln') is not None)
            K2 = kw['dual'][0].shape[-1] if kw.get('dual') is not None else 0
            return kern, 2.0 * nb * M * N * (K + K2), 4.0 * nb * (M * (K + K2)) + 4.0 * nb * M * N + 4.0 * (K + K2) * N
        if name == 'gemm_side':             # (main, side) descriptors in one launch: the rows are read once for both products
            fl = sum(2.0 * d.batch * d.M * d.N * d.K for d in args[:2])
            by = 4.0 * args[0].batch * args[0].M * args[0].K + sum(4.0 * d.batch * d.M * d.N for d in args[:2])
            return (self.ops.gemm_as_kernel_name(args[0], args[1]) or 'gemm3_side_kernel'), fl, by
        if name == 'gemm_splitk':
            S, M, N = args[2].shape
            return 'gemm3 split-K (K slices as batch)', 2.0 * M * N * args[0].shape[1], 4.0 * M * (args[0].shape[1] + S * N)
        if name == 'tri_attn':
            Bc, L = args[4], args[5]
            return self.ops.tri_attn_kernel_name(L, kw.get('exact')), 4.0 * Bc * L * 4 * L * L * 48, 4.0 * Bc * L * L * (args[0].shape[1] + 192 + 4)
        if name == 'ipa_tail':
            M, K1 = args[0].shape
            return 'ipa_tail_kernel', 2.0 * M * 256 * (K1 + 3 * 256), 4.0 * M * (K1 + 2 * 256) + 4.0 * 256 * (K1 + 3 * 256)
        if name == 'ipa_attn':
            Bc, L = args[10], args[11]
            return 'ipa_attn (weights + pair slab kernels)', 2.0 * Bc * L * L * 12 * (28 + 40 + 128), 4.0 * Bc * L * L * (128 + 12 + 24)
        if name == 'ipa_pair':
            Bc, L = args[3], args[4]
            # the HBM stream of the IPA layer: the pair slab read once + the 12 weights per pair
            return 'ipa_pair_kernel', 2.0 * Bc * L * L * 12 * 128, 4.0 * Bc * L * L * (128 + 12)
        if name == 'ipa_weights':
            Bc, L = args[10], args[11]
            return 'ipa_weights_kernel', 2.0 * Bc * L * L * 12 * (28 + 40), 4.0 * Bc * L * L * (12 + 12)
        if name == 'opm_out':           # the reference's feature form: 2 x 128 x 192 flops per pair row; z read + written
            Bc, L = args[4], args[5]
            return 'opm_out_kernel', 2.0 * Bc * L * L * 128 * 192, 4.0 * Bc * L * L * (192 + 192)
        if name == 'assemble_pair_bias':    # the assembly's bytes + the bias written (the projection: 2 x 192 x 32 flops per pair row)
            Bc, L = args[12], args[13]
            return 'assemble_pair_bias_kernel', 2.0 * Bc * L * L * 192 * 32, 4.0 * Bc * L * L * (192 + 192 + 2 + 32) + 4.0 * L * L * 128
        if name == 'assemble_pair':     # 320 channel reads (prev_pair 192 + static 128, shared over the samples) + 192 written per pair
            Bc, L = args[8], args[9]
            return 'assemble_pair192_kernel', 0.0, 4.0 * Bc * L * L * (192 + 192 + 2) + 4.0 * L * L * 128
        return name, 0.0, 0.0

    def __enter__(self):
        skip = ('gemm_kernel_name', 'gemm_as_kernel_name', 'gemm_split_eligible', 'tri_attn_kernel_name', 'gemm_mode', 'ipa_qpack_numel', 'permute_k16', 'atom14_mask_table', 'vdw_radius_table',
                'range_word', 'range_names', 'range_ptr', 'range_words', 'first_range_tag', 'pad_planes_128', 'weights_to_float',
                'planes_to_float')       # (host-side helpers: no launch to time)
        for name in dir(self.ops):
            fn = getattr(self.ops, name)
            if callable(fn) and not isinstance(fn, type) and not name.startswith('_') and name not in skip and getattr(fn, '__module__', '') == self.ops.__name__:
                self.saved[name] = fn

                def wrap(fn=fn, name=name):
                    def inner(*a, **k):
                        if k.get('defer'):            # a descriptor for gemm_side: nothing is launched
                            return fn(*a, **k)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        r = fn(*a, **k)
                        e1.record()
                        self.records.append((self._sig(name, a, k), e0, e1))
                        return r
                    return inner
                setattr(self.ops, name, wrap())
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ops, name, fn)
        torch.cuda.synchronize()

    def summary(self):
        """Aggregate per KERNEL (like `rocprofv3 --stats`): total ms, launches, total algorithmic flops / bytes."""
        agg = defaultdict(lambda: [0.0, 0, 0.0, 0.0])
        for (sig, fl, by), e0, e1 in self.records:
            a = agg[sig]
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
            a[2] += fl
            a[3] += by
        return sorted(((k, v[0], v[1], v[2], v[3]) for k, v in agg.items()), key=lambda x: -x[1])


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle) + parity of the same call
# ---------------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """(physical cores, logical cpus, model name) of this host."""
    logical = os.cpu_count() or 1
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name') and model == 'unknown':
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                phys = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                core = line.split(':', 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    physical = min(len(cores), logical) if cores else logical
    return max(1, physical), logical, model


def cpu_baseline(model, D, batch, cfg, L, t_value, timed_steps=3):
    """Sample 0 of the complex, one step (ScoreNetwork 3 passes + get_prev + reverse) with the CURRENT self-conditioning state:
    run on the HIP path here and on the oracle in a subprocess (bounded threads, hard timeout); returns the timing of the
    oracle (1 warm-up + `timed_steps` timed calls) and the difference between the two results."""
    import tempfile
    physical, logical, cpu_model = host_cpu_info()
    # SURVEY §8d asks for the physical cores; on a 2-socket 128-core host the torch CPU path is faster on fewer threads, so the
    # warm-up step is run at each candidate count and the timed steps use the fastest one (all of it is reported)
    candidates = sorted({physical, min(physical, 64), min(physical, 32)}, reverse=True)
    threads = physical
    unit = 'sample-steps/s'
    one = {}
    for k, v in batch.items():
        if k.startswith('_'):
            continue
        if torch.is_tensor(v):
            one[k] = (v[:1] if v.dim() > 0 else v).clone()
        elif isinstance(v, tuple):
            one[k] = tuple(x[:1].clone() for x in v)
    dev = one['seq'].device
    t1 = torch.full((1,), t_value, dtype=torch.float64, device=dev)
    from abx_amd import sampler
    sampler.set_t_feats(one, D, t1, torch.ones(1, device=dev))
    blob = {k: (v.cpu() if torch.is_tensor(v) else tuple(x.cpu() for x in v)) for k, v in one.items()}
    with torch.no_grad():
        out = model(one)
        torch.cuda.synchronize()
    hip = {'rigids': out['heads']['folding']['rigids'].cpu(), 'trans_score': out['heads']['folding']['trans_score'].cpu(),
           'logits': out['heads']['sequence_module']['logits'].cpu(), 'seq_0': out['heads']['sequence_module']['seq_0'].cpu(),
           'pLDDT': out['heads']['predicted_lddt']['pLDDT'].cpu(), 'prev_pos': out['_prev_pos'].cpu()}
    blob['_tables'] = dict(pdf=D._pdf.cpu(), cdf=D._cdf.cpu(), score_norms=D.score_norms.cpu())
    blob['_t'] = t_value
    blob['_hip'] = hip
    blob['_timed_steps'] = timed_steps
    blob['_thread_candidates'] = candidates
    base = {'value': None, 'unit': unit, 'cores': threads, 'kind': 'port'}
    with tempfile.TemporaryDirectory() as d:
        torch.save(blob, os.path.join(d, 'in.pt'))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', d], env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [x for x in res.stdout.splitlines() if x.startswith('CPU_BASELINE ')]
            if res.returncode != 0 or not line:
                return dict(base, sample='worker failed: ' + res.stderr[-300:])
            r = json.loads(line[0][len('CPU_BASELINE '):])
        except subprocess.TimeoutExpired:
            return dict(base, sample='timed out after 600 s')
    ctime = float(np.mean(r['times']))
    warm = ', '.join(f'{n} threads {t_:.1f} s' for n, t_ in r['warmups'])
    return dict(base, value=1.0 / ctime, cores=r['threads'],
                sample=f'1 diffusion step (ScoreNetwork 3 passes + get_prev + reverse) of sample 0 at L={L}, same complex, weights and '
                       f'self-conditioning state as the HIP run; warm-up steps at each candidate thread count ({warm}), then '
                       f'{len(r["times"])} timed steps ({", ".join("%.1f" % x for x in r["times"])} s) on the fastest = {r["threads"]} threads; host: '
                       f'{physical} physical cores / {logical} logical CPUs, {cpu_model}; trajectory-invariant embeddings cached as in '
                       'the HIP path; CPU batching does not help (BASELINE.md section 2)',
                cpu_model=cpu_model, physical_cores=physical, logical_cpus=logical, step_times_s=r['times'], parity=r['parity'])


def cpu_baseline_worker(d):
    from abx_amd import synthetic
    from abx_amd.config import default_config
    from oracle import abx_oracle as O
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
    blob = torch.load(os.path.join(d, 'in.pt'))
    cfg = default_config()
    params = synthetic.random_state_dict(model_parameter_shapes(cfg), seed=WEIGHT_SEED)
    od = O.OracleDiffuser(cfg.diffuser, blob.pop('_tables'))
    tc = torch.full((1,), blob.pop('_t'), dtype=torch.float64)
    hip = blob.pop('_hip')
    nt = int(blob.pop('_timed_steps'))
    cands = [int(c) for c in blob.pop('_thread_candidates')]

    def one_step():
        cpu = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in blob.items()}
        cpu = O.set_t_feats(cpu, od, tc, torch.ones(1))
        c0 = time.perf_counter()
        ro = O.score_network(params, cpu, cfg, od, static)
        prev = O.get_prev(cpu, ro, cfg)
        dm = (1 - cpu['fixed_mask']) * cpu['atom14_gt_exists'][..., 0]
        od.reverse(cpu['rigids_t'], cpu['seq_t'], ro['heads']['folding']['rot_score'], ro['heads']['folding']['trans_score'],
                   ro['heads']['sequence_module']['logits'], tc, torch.tensor(0.01), dm)
        return time.perf_counter() - c0, ro, prev

    with torch.no_grad():
        static = O.static_embeddings(params, blob, cfg)
        warmups = []
        for n in cands:
            torch.set_num_threads(n)
            warmups.append((n, one_step()[0]))
        best = min(warmups, key=lambda x: x[1])[0]
        torch.set_num_threads(best)
        times = []
        for _ in range(nt):
            dt_, ro, prev = one_step()
            times.append(dt_)
    f = ro['heads']['folding']
    mx = lambda a, b: float((a.double() - b.double()).abs().max())
    parity = {
        'max_abs_rigids': mx(hip['rigids'], f['rigids']), 'max_abs_trans_score': mx(hip['trans_score'], f['trans_score']),
        'max_abs_logits': mx(hip['logits'], ro['heads']['sequence_module']['logits']),
        'max_abs_pLDDT': mx(hip['pLDDT'], ro['heads']['predicted_lddt']['pLDDT']),
        'tokens_equal': bool(torch.equal(hip['seq_0'], ro['heads']['sequence_module']['seq_0'])),
        'distogram_bins_differing': int((hip['prev_pos'] != prev['prev_pos']).sum()),
        'what': 'HIP vs oracle, final pass of one call on sample 0 (L as benchmarked), absolute differences; rigids in Angstrom / unit quaternions',
    }
    print('CPU_BASELINE ' + json.dumps({'warmups': warmups, 'threads': best, 'times': times, 'parity': parity}))


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun starts its own ranks
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n, argv, port=None):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """Start n ranks of this script and relay rank 0's JSON line (the only line this process prints on stdout)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    res = subprocess.run(launcher_command(n, argv), env=env, capture_output=True, text=True)
    lines = [x for x in res.stdout.splitlines() if x.startswith('{')]
    sys.stderr.write(res.stderr[-4000:])
    if res.returncode != 0 or not lines:
        sys.stderr.write(f'\nbench.py: the {n}-rank launch failed (rc {res.returncode})\n')
        sys.exit(res.returncode or 1)
    print(lines[-1])


def launcher_selftest(args):
    """CPU check of the launch + rendezvous logic (gloo): every rank contributes its rank to an all_reduce and its shard of the
    sample ids to an all_gather; rank 0 prints what it saw."""
    import torch.distributed as dist
    from abx_amd import sampler
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    x = torch.tensor([float(rank)])
    dist.all_reduce(x)
    ids = sampler.shard_sample_ids(args.samples, rank, world)
    local = {'ids': torch.tensor(ids, dtype=torch.int64).reshape(-1, 1)}
    got = sampler.gather_results(local, args.samples, rank, world)
    if rank == 0:
        print(json.dumps({'launcher_selftest': True, 'world': world, 'rank_sum': float(x), 'gathered_ids': got['ids'].reshape(-1).tolist(),
                          'samples_per_rank': [len(sampler.shard_sample_ids(args.samples, r, world)) for r in range(world)]}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='L352')
    ap.add_argument('--samples', type=int, default=100, help='samples of the complex: in total (strong scaling) or per GPU (weak)')
    ap.add_argument('--scaling', choices=['strong', 'weak'], default='strong')
    ap.add_argument('--chunk', type=int, default=0, help='samples per pair-stack launch (0 = as many as fit: the whole per-GPU batch)')
    ap.add_argument('--exact-class', default='', help='comma-separated op classes (abx_amd.ops.RANGE_TAGS: tri_attn, pair_transition, '
                    'plane_projection, ...) pinned to the exact fp32-MFMA kernels: the price of a checkpoint whose activations leave the '
                    'split-f16 operand range in that class (the sticky state of ScoreNetwork)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-op-profile', action='store_true')
    ap.add_argument('--no-weak', action='store_true', help='N > 1: skip the additional weak-scaling measurement')
    ap.add_argument('--launcher-selftest', action='store_true', help='CPU/gloo check of the multi-rank launch path, no GPU work')
    ap.add_argument('--force-collective', action='store_true', help='--gpus 1: still initialise the RCCL process group and run the barrier, the '
                    'MAX all_reduce and the final gather through 1-rank collectives on device tensors (exercises the multi-GPU code '
                    'path on a 1-GPU box; same results)')
    ap.add_argument('--debug-one-gpu', action='store_true', help='debugging on a 1-GPU box: every rank uses cuda:0 and the gloo backend '
                                                                 '(exercises the whole multi-rank code path; the numbers mean nothing)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])
    if args.launcher_selftest:
        return launcher_selftest(args)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = 0 if args.debug_one_gpu else int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    coll = world > 1 or args.force_collective        # collectives run (a 1-rank group under --force-collective)
    if coll:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29578')
        if args.debug_one_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from abx_amd import synthetic, features, sampler, ops, _lib
    from abx_amd.config import default_config
    from abx_amd.model.abx import ScoreNetwork, get_prev
    from abx_amd.diffuser.full_diffuser import FullDiffuser

    lib = _lib.load()
    rc = lib.abx_init(local_rank)
    assert rc == 0, lib.abx_last_error_string()
    cfg = default_config()
    cfg.diffuser.so3.cache_dir = f'/tmp/abx_bench_cache_{rank}/'
    params = synthetic.random_state_dict(model_parameter_shapes(cfg), seed=WEIGHT_SEED)
    D = FullDiffuser(cfg.diffuser).to(dev)                     # IGSO(3) tables by the HIP kernel
    model = ScoreNetwork(cfg.model, D)
    model.load_state_dict(params, strict=True)
    model = model.to(dev).eval()
    model.max_chunk = args.chunk or None
    model.forced_exact_ops = tuple(c for c in args.exact_class.split(',') if c)

    w = synthetic.WORKLOADS[args.workload]
    cx = synthetic.make_complex(seed=1, **w)
    L = cx['seq'].shape[0]
    grid = np.linspace(0.01, 1.0, 100)[::-1]
    dt = float(np.float32(0.01))
    D.seed = 2024

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, steps, warmup):
        """Builds this rank's samples, runs warm-up + `steps` timed steps; returns (elapsed max over ranks, state)."""
        total = args.samples if scaling == 'strong' else args.samples * world
        ids = sampler.shard_sample_ids(total, rank, world)
        B = len(ids)
        st = {'ids': ids, 'B': B, 'total': total}
        if B > 0:
            raw = {k: v.to(dev) for k, v in synthetic.replicate(cx, B).items()}
            batch = features.build_features(raw, D, noise=features.per_sample_init_noise(ids, L, seed=1234, device=dev))
            batch['_shared_context'] = True
            diffuse_mask = ((1 - batch['fixed_mask']) * batch['atom14_gt_exists'][..., 0]).to(torch.int32)
            ones = torch.ones(B, device=dev)
            sid = torch.tensor(ids, device=dev, dtype=torch.int64)

            def one_step(k):
                t_ = torch.full((B,), float(grid[k % 99]), device=dev, dtype=torch.float64)
                sampler.set_t_feats(batch, D, t_, ones)
                out = model(batch)
                f = out['heads']['folding']
                batch.update(get_prev(batch, out, cfg.model))
                rig, seq = D.reverse(rigid_t=batch['rigids_t'], seq_t=batch['seq_t'], rot_score=f['rot_score'],
                                     trans_score=f['trans_score'], logits_t=out['heads']['sequence_module']['logits'],
                                     diffuse_mask=diffuse_mask, t=t_, dt=dt, sample_ids=sid, step=k)
                batch['rigids_t'], batch['seq_t'] = rig, seq
                return out
        else:
            batch, one_step = None, (lambda k: None)
        with torch.no_grad():
            if B > 0:
                # self-conditioning warm-up call of the sampler (inference.py:209-211: one extra network call per trajectory).  It is
                # outside the K timed steps (the contract times exactly K steps) and timed on its own for the T = 100 figure below;
                # the first call also pays the one-off set-up (static embeddings, weight packing), so it is run twice
                sampler.set_t_feats(batch, D, float(grid[0]), ones)
                out = model(batch)
                torch.cuda.synchronize()
                w0 = time.perf_counter()
                out = model(batch)
                torch.cuda.synchronize()
                st['selfcond_call_ms'] = 1000.0 * (time.perf_counter() - w0)
                batch.update(get_prev(batch, out, cfg.model))
            for k in range(warmup):
                out = one_step(k)
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                out = one_step(warmup + k)
            barrier()
            elapsed = time.perf_counter() - t0
        if coll:
            tt_ = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            elapsed = float(tt_)
        st.update(batch=batch, one_step=one_step, out=out, next_k=warmup + steps,
                  finite=bool(torch.isfinite(batch['rigids_t']).all()) if B > 0 else True)
        return elapsed, st

    elapsed, st = measure(args.scaling, args.steps, args.warmup)
    total = st['total']
    value = total * args.steps / elapsed
    per_rank = [len(sampler.shard_sample_ids(total, r, world)) for r in range(world)]

    # ---- the one collective of the path: final results of all samples to every rank (after the timed loop)
    gather_ms, rccl_ranks = None, 1
    Lab = cx['anchor_flag'].shape[0]
    if st['B'] > 0:
        o, b_ = st['out'], st['batch']
        local = {'rigids': b_['rigids_t'].double(), 'seq': torch.clamp(b_['seq_t'][:, :Lab], 0, 19).long(),
                 'atom14': o['heads']['folding']['final_atom14_positions'][:, :Lab].contiguous(),
                 'pLDDT': o['heads']['predicted_lddt']['pLDDT'][:, :Lab].contiguous()}
    else:
        local = {'rigids': torch.zeros(0, L, 7, dtype=torch.float64, device=dev), 'seq': torch.zeros(0, Lab, dtype=torch.int64, device=dev),
                 'atom14': torch.zeros(0, Lab, 14, 3, device=dev), 'pLDDT': torch.zeros(0, Lab, device=dev)}
    got = local
    if coll:
        sampler.gather_results(local, total, rank, world, force=args.force_collective)        # first call: communicator set-up
        barrier()
        g0 = time.perf_counter()
        got = sampler.gather_results(local, total, rank, world, force=args.force_collective)
        barrier()
        gather_ms = 1000.0 * (time.perf_counter() - g0)
        assert got['rigids'].shape[0] == total
        cnt = torch.ones(1, device=dev)
        dist.all_reduce(cnt)
        rccl_ranks = int(cnt)
    # sha256 over the final state of all samples in sample order (what the gather returns): neither the placement of the samples nor
    # the collective may change a bit of it
    import hashlib
    hd = hashlib.sha256()
    for kk in ('rigids', 'seq', 'atom14', 'pLDDT'):
        hd.update(got[kk].contiguous().cpu().numpy().tobytes())

    B0 = st['B']
    result = {
        'metric': 'diffusion-steps/sec (100 samples, ~350-res complex)', 'value': value, 'unit': 'sample-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'dtype_note': 'fp32 storage and accumulation everywhere; the large pair-stack GEMMs evaluate each fp32 product from '
                      'split float16 operands (3 exact MFMA products, 23-bit operand images, fp32 accumulate: measured as accurate as the native fp32 MFMA, '
                      'tests/test_gpu_kernels.py), everything else is native fp32 / fp64',
        'config': {'workload': f'{args.workload}: L={L} (Lab {w["L_heavy"] + w["L_light"]} + antigen {w["L_antigen"]}), '
                               f'{total} samples of one complex over {world} GPU(s), 1 step = ScoreNetwork (3 passes) + get_prev '
                               '+ reverse, seeded random weights, ESM off', 'L': L, 'samples_total': total,
                   'samples_per_rank': per_rank, 'chunk': args.chunk or 'auto', 'parallelism': f'sample-shard x{world}, no collective in the step'},
        'exact_classes': list(model.forced_exact_ops), 'range_fallbacks': len(model.range_log),
        'finite': st['finite'], 'rccl_ranks': rccl_ranks, 'gather_ms': gather_ms, 'result_digest': hd.hexdigest(),
        # a T = 100 trajectory is 100 steps + the self-conditioning warm-up call; this is the rate with that call inside the wall
        'selfcond_warmup_call_ms': st.get('selfcond_call_ms'),
        'value_T100_trajectory_incl_warmup_call': (total * 100.0 / ((100.0 * elapsed / args.steps) + st['selfcond_call_ms'] / 1e3)
                                                   if st.get('selfcond_call_ms') is not None else None),
        'step_hbm_frac': value / world * algorithmic_bytes_per_sample_step(L) / 1e9 / HBM_PEAK_GBS,
        'step_mfma_f32_frac': value / world * algorithmic_flops_per_sample_step(L) / 1e12 / MFMA_F32_PEAK_TF,
    }

    if rank == 0 and not args.no_op_profile and B0 > 0:
        # per-KERNEL view of one more step: the three pair-stack op groups run through one C call each in the timed region
        # (abx_tri_mul_fwd / abx_tri_attn_block_fwd / abx_transition_fwd, csrc/blocks.hip); for this instrumented step the engine
        # issues the same kernels in the same order one descriptor at a time, so that every launch gets its own pair of HIP events
        eng = model._get_engine(dev)
        blk_saved, eng.block_api = eng.block_api, False
        with torch.no_grad():
            st['one_step'](st['next_k'])            # (un-timed: the descriptor-level path allocates its own scratch once)
        st['next_k'] += 1
        with torch.no_grad(), OpTimer(ops) as tm:
            st['one_step'](st['next_k'])
        eng.block_api = blk_saved
        summ = tm.summary()
        tot_ms = sum(s[1] for s in summ)
        name, ms, calls, fl, by = summ[0]
        dur = ms / 1e3                      # all launches of the kernel in one step; fl / by are summed over them too
        # which roof binds is decided by the kernel's ALGORITHMIC intensity against the ridge of its arithmetic:
        # split-f16 kernels price an fp32 flop at three f16 MFMA products (ceiling = dense f16 peak / 3), everything else at the native
        # fp32 MFMA / VALU peak; both fractions are reported, `frac` is the binding one
        split = name.startswith(('gemm3_', 'gemm_as_', 'tri_attn4', 'tri_attn8', 'opm_out'))
        peak_tf = MFMA_SPLIT_PEAK_TF if split else MFMA_F32_PEAK_TF
        ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        intensity = fl / by if by else float('inf')
        frac_mfma, frac_hbm = fl / dur / 1e12 / peak_tf, by / dur / 1e9 / HBM_PEAK_GBS
        if intensity < ridge:
            roof = {'bound': 'hbm', 'achieved': by / dur / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': frac_hbm}
        else:
            roof = {'bound': 'mfma', 'achieved': fl / dur / 1e12, 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': frac_mfma}
        roof.update(intensity_flop_per_byte=intensity, ridge_flop_per_byte=ridge, frac_mfma=frac_mfma, frac_hbm=frac_hbm,
                    achieved_tflops=fl / dur / 1e12, achieved_gbs=by / dur / 1e9,
                    peak_note=('matrix peak for this arithmetic: dense f16 MFMA 2516.6 TF / 3 products per fp32 product (split-f16) = 838.9 TF'
                               if split else 'matrix peak: native fp32 MFMA 157.3 TF'))
        # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/README.md; tools/pmc_traffic.py).  The table is stamped
        # with the library it was collected on and its geometry (`_meta`): a table that belongs to another build or another batch is
        # reported as stale, and the whole-step sum is only formed at the table's own geometry (launch bytes scale with the rows)
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        pmc_tab = json.load(open(pmc)) if os.path.exists(pmc) else {}
        meta = pmc_tab.pop('_meta', None) if pmc_tab else None
        lib_now = library_stamp()
        # (the same sources built in another directory give a library of the same size with another hash: hipcc embeds the build path)
        same_build = bool(meta) and (meta.get('lib_sha256') == lib_now['lib_sha256'] or
                                     (meta.get('src_sha256') == lib_now['src_sha256'] and meta.get('lib_bytes') == lib_now['lib_bytes']))
        same_geometry = bool(meta) and int(meta.get('samples', -1)) == B0 and int(meta.get('L', -1)) == L
        traffic_stale = bool(pmc_tab) and not (same_build and same_geometry)
        result['traffic_table'] = {'file': 'profiles/pmc_traffic.json' if pmc_tab else None, 'meta': meta, 'running_library': lib_now,
                                   'same_build': same_build, 'same_geometry': same_geometry}
        result['traffic_stale'] = traffic_stale
        per_launch = (lambda nm: pmc_tab.get(nm, {}).get('hbm_bytes_per_launch')) if (pmc_tab and (same_geometry or not meta)) else (lambda nm: None)
        traffic = per_launch(name)
        roof.update(traffic=traffic, kernel=name, calls_per_step=calls,
                    avg_launch_ms=ms / calls, share_of_step=ms / tot_ms,
                    traffic_source=('profiles/pmc_traffic.json: HBM bytes per launch of this kernel from separate rocprofv3 --pmc passes at the '
                                    'bench geometry (FETCH_SIZE x 2 + WRITE_SIZE); NOT collected in this run'
                                    + ('; STALE: collected on another build of the library' if not same_build else '')) if traffic is not None else None)
        # whole-step counter traffic: launches of this step x the per-launch HBM bytes of the same PMC table (kernels of the op profile
        # that the table knows; the share of the step time they cover is reported next to it)
        if pmc_tab and (same_geometry or not meta):
            tb, covered = 0.0, 0.0
            for nm, ms_, calls_, _, _ in summ:
                t_ = per_launch(nm)
                if t_ is not None:
                    tb += t_ * calls_
                    covered += ms_
            result['step_traffic_bytes'] = tb
            result['step_traffic_note'] = (f'sum over the kernels of one step of launches x HBM bytes per launch (profiles/pmc_traffic.json, rocprofv3 --pmc '
                                           f'FETCH_SIZE x 2 + WRITE_SIZE at the bench geometry); covers {covered / tot_ms:.3f} of the step time; algorithmic '
                                           f'bytes of the step: {B0 * algorithmic_bytes_per_sample_step(L):.4g}')
            result['step_traffic_over_algorithmic'] = tb / (B0 * algorithmic_bytes_per_sample_step(L))
        elif pmc_tab:
            result['step_traffic_note'] = (f'not formed: profiles/pmc_traffic.json was collected at {meta.get("samples")} samples of L = {meta.get("L")}, '
                                           f'this run has {B0} of L = {L} (bytes per launch scale with the rows)')
        # the whole roofline picture: every kernel that takes >= 2 % of the step, priced like the dominant one
        kernels = []
        for nm, ms_, calls_, fl_, by_ in summ:
            if ms_ < 0.02 * tot_ms:
                continue
            sp = nm.startswith(('gemm3_', 'gemm_as_', 'tri_attn4', 'tri_attn8', 'opm_out'))
            pk = MFMA_SPLIT_PEAK_TF if sp else MFMA_F32_PEAK_TF
            d_ = ms_ / 1e3
            ent = {'kernel': nm, 'calls_per_step': calls_, 'avg_launch_ms': ms_ / calls_, 'share_of_step': ms_ / tot_ms,
                   'arithmetic': 'split-f16 (peak 838.9 TF)' if sp else 'fp32 (peak 157.3 TF)'}
            if by_:
                inten, rdg = (fl_ / by_), pk * 1e12 / (HBM_PEAK_GBS * 1e9)
                ent.update(bound='hbm' if inten < rdg else 'mfma', intensity_flop_per_byte=inten,
                           frac_hbm=by_ / d_ / 1e9 / HBM_PEAK_GBS, frac_mfma=(fl_ / d_ / 1e12 / pk) if fl_ else None,
                           algorithmic_bytes_per_launch=by_ / calls_)
                ent['frac'] = ent['frac_hbm'] if ent['bound'] == 'hbm' else ent['frac_mfma']
                t_ = per_launch(nm)
                ent['traffic_bytes_per_launch'] = t_
                ent['traffic_over_algorithmic'] = (t_ / (by_ / calls_)) if t_ else None
            kernels.append(ent)
        result['roofline_kernels'] = kernels
        priced = [k for k in kernels if k.get('frac') is not None]
        if priced:
            worst = min(priced, key=lambda k: k['frac'])
            result['roofline_furthest_below_roof'] = {'kernel': worst['kernel'], 'bound': worst['bound'], 'frac': worst['frac']}
            rat = [k for k in priced if k.get('traffic_over_algorithmic')]
            if rat:
                wr = max(rat, key=lambda k: k['traffic_over_algorithmic'])
                result['roofline_worst_traffic_ratio'] = {'kernel': wr['kernel'], 'traffic_over_algorithmic': wr['traffic_over_algorithmic']}
        result['roofline'] = roof
        # the kernel north_star calls HBM-bound (the IPA pair-slab stream), priced the same way next to the dominant one
        for nm, ms2, calls2, fl2, by2 in summ:
            if nm == 'ipa_pair_kernel':
                tr2 = per_launch(nm)
                result['roofline_ipa_pair_slab'] = {
                    'bound': 'hbm', 'achieved': by2 / (ms2 / 1e3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': by2 / (ms2 / 1e3) / 1e9 / HBM_PEAK_GBS, 'traffic': tr2, 'kernel': nm, 'calls_per_step': calls2,
                    'avg_launch_ms': ms2 / calls2, 'share_of_step': ms2 / tot_ms}
        result['op_profile_ms'] = [{'op': s[0], 'ms': round(s[1], 3), 'calls': s[2], 'avg_ms': round(s[1] / s[2], 4),
                                    'tflops': (s[3] / (s[1] / 1e3) / 1e12 if s[3] else None)} for s in summ[:40]]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(model, D, st['batch'], cfg, L, float(grid[(st['next_k'] + 1) % 99]))

    if world > 1 and args.scaling == 'strong' and not args.no_weak:
        st = None
        torch.cuda.empty_cache()
        wsteps = max(1, min(args.steps, 3))
        welapsed, wst = measure('weak', wsteps, 1)
        result['weak'] = {'scaling': 'weak', 'samples_per_gpu': args.samples, 'steps': wsteps, 'warmup': 1,
                          'value': wst['total'] * wsteps / welapsed, 'unit': 'sample-steps/s', 'ms_per_step': 1000.0 * welapsed / wsteps}

    if rank == 0:
        print(json.dumps(result))
    if coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-baseline-worker':
        cpu_baseline_worker(sys.argv[2])
    else:
        main()
