#!/usr/bin/env python
"""Benchmark of the AbX reverse-diffusion sampling hot path on MI355X (BASELINE.json metric: diffusion-steps/sec).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one iteration of the reference's reverse loop (inference.py:213-268) for a batch of B samples of ONE complex:
ScoreNetwork call (2 recycles + final pass = 3 network passes) + get_prev + FullDiffuser.reverse.  All inputs are resident
in HBM when the timed region starts.  value = (samples on all ranks) * K / max-over-ranks(time)  [sample-steps / s].
Workload (config.workload): synthetic complex 'L352' (Lab 228 + antigen 124 = 352 residues, BASELINE's "~350-res complex"),
100 samples per GPU (weak scaling: every rank designs its own 100 samples), seeded random weights (no trained checkpoint
exists offline), ESM disabled, device Philox noise.  fp32 compute with float64 diffuser state, as the reference.

Extra objects on the JSON line: "roofline" for the dominant kernel (measured live with HIP events on the launch stream)
and "cpu_baseline" (the oracle = CPU port of the same step, B = 1, timed on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2516.6     # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 1024 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz)
MFMA_SPLIT_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0   # fp32-accurate product = 6 bf16 MFMA products (csrc/gemm3.hip)


def algorithmic_bytes_per_sample_step(L):
    """3 passes x 9893 channel-passes over the L^2 pair grid x 4 B."""
    return 3 * 9893 * 4.0 * L * L


def algorithmic_flops_per_sample_step(L):
    return 3 * 2 * (1.064e6 * L * L + 1024.0 * L ** 3 + 12.4e6 * L + 1088.0 * L * L)


class OpTimer:
    """Per-op HIP-event timing on the current (launch) stream, used for ONE instrumented step after the timed region."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []
        self.saved = {}

    def _sig(self, name, args, kw):
        if name == 'gemm':
            A, B, C = args[0], args[1], args[2]
            if A.dtype == torch.int16:          # k-tiled bf16 planes (b, K/16, 3, rows, 16): the tri-mul contraction
                if A.dim() == 6:        # two-level batch (channel slices)
                    nb, M, K, N = A.shape[0] * A.shape[1], A.shape[4], A.shape[2] * 16, B.shape[4]
                else:
                    nb, M, K, N = A.shape[0], A.shape[3], A.shape[1] * 16, B.shape[3]
                kern = self.ops.gemm_kernel_name(M, N, K, nb, split=True, a_split=True)
                return kern, 2.0 * nb * M * N * K, nb * (6.0 * (M + N) * K + 4.0 * M * N)
            nb = A.shape[0] if A.dim() == 3 else 1
            M, K = A.shape[-2], A.shape[-1]
            N = B.shape[-1]
            c_planes = C.dtype == torch.int16
            kern = self.ops.gemm_kernel_name(M, N, K, nb, A.stride(-1) == 1, B.stride(-1) == 1, c_planes or C.stride(-1) != 1,
                                             split=kw.get('B3') is not None)
            return kern, 2.0 * nb * M * N * K, 4.0 * nb * (M * K) + (6.0 if c_planes else 4.0) * nb * M * N + 4.0 * K * N
        if name == 'tri_attn':
            Bc, L = args[4], args[5]
            exact = self.ops.GEMM_EXACT if kw.get('exact') is None else kw['exact']
            slots = ((L + 15) // 16 + 11) // 12
            kern = 'tri_attn_kernel' if exact else f'tri_attn3_kernel<{2 if slots <= 2 else 4 if slots <= 4 else 8}>'
            return kern, 4.0 * Bc * L * 4 * L * L * 48, 4.0 * Bc * L * L * (4 * 192 + 192 + 4)
        if name == 'ipa_attn':
            Bc, L = args[-2], args[-1]
            return 'ipa_attn_kernel', 2.0 * Bc * L * L * 12 * (28 + 40 + 128), 4.0 * Bc * L * L * (128 + 12)
        return name, 0.0, 0.0

    def __enter__(self):
        for name in dir(self.ops):
            fn = getattr(self.ops, name)
            if callable(fn) and not name.startswith('_') and name not in ('gemm_kernel_name', 'gemm_split_eligible') and \
                    getattr(fn, '__module__', '') == self.ops.__name__:
                self.saved[name] = fn

                def wrap(fn=fn, name=name):
                    def inner(*a, **k):
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


def cpu_baseline(args, D, batch, B, L, t_value):
    """The oracle (CPU port of the same step definition) for ONE sample of the same complex, in a subprocess with a
    bounded thread count and a hard timeout so that the default bench run stays within minutes."""
    import subprocess
    import tempfile
    threads = min(32, os.cpu_count() or 1)
    with tempfile.TemporaryDirectory() as d:
        blob = {}
        for k, v in batch.items():
            if torch.is_tensor(v) and not k.startswith('prev_'):
                blob[k] = (v[:1] if v.dim() > 0 and v.shape[0] == B else v).cpu()
        blob['rigids_t'] = blob['rigids_t'].double()
        blob['_tables'] = dict(pdf=D._pdf.cpu(), cdf=D._cdf.cpu(), score_norms=D.score_norms.cpu())
        blob['_t'] = t_value
        torch.save(blob, os.path.join(d, 'in.pt'))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', d], env=env,
                                 capture_output=True, text=True, timeout=420)
            line = [x for x in out.stdout.splitlines() if x.startswith('CPU_BASELINE ')]
            if out.returncode != 0 or not line:
                return {'value': None, 'unit': 'sample-steps/s', 'cores': threads, 'kind': 'port',
                        'sample': 'worker failed: ' + out.stderr[-300:]}
            ctime = float(line[0].split()[1])
        except subprocess.TimeoutExpired:
            return {'value': None, 'unit': 'sample-steps/s', 'cores': threads, 'kind': 'port', 'sample': 'timed out after 420 s'}
    return {'value': 1.0 / ctime, 'unit': 'sample-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'1 diffusion step of 1 sample at L={L} (same complex, weights, step definition; trajectory-invariant '
                      f'embeddings cached as in the HIP path; CPU batching does not help, BASELINE.md section 2), {ctime:.1f} s '
                      f'on {threads} threads of {os.cpu_count()} logical CPUs'}


def cpu_baseline_worker(d):
    from collections import OrderedDict as OD
    from abx_amd import synthetic
    from abx_amd.config import default_config
    from oracle import abx_oracle as O
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
    blob = torch.load(os.path.join(d, 'in.pt'))
    cfg = default_config()
    keys = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'sd_keys.json')))
    params = synthetic.random_state_dict(OD((k, tuple(s)) for k, s in keys), seed=7)
    od = O.OracleDiffuser(cfg.diffuser, blob.pop('_tables'))
    tc = torch.full((1,), blob.pop('_t'), dtype=torch.float64)
    cpu = blob
    with torch.no_grad():
        cpu = O.set_t_feats(cpu, od, tc, torch.ones(1))
        static = O.static_embeddings(params, cpu, cfg)
        c0 = time.perf_counter()
        ro = O.score_network(params, cpu, cfg, od, static)
        cpu.update(O.get_prev(cpu, ro, cfg))
        dm = (1 - cpu['fixed_mask']) * cpu['atom14_gt_exists'][..., 0]
        od.reverse(cpu['rigids_t'], cpu['seq_t'], ro['heads']['folding']['rot_score'], ro['heads']['folding']['trans_score'],
                   ro['heads']['sequence_module']['logits'], tc, torch.tensor(0.01), dm)
        print('CPU_BASELINE', time.perf_counter() - c0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='L352')
    ap.add_argument('--samples', type=int, default=100, help='samples per GPU')
    ap.add_argument('--chunk', type=int, default=100, help='samples per pair-stack launch (workspace ~0.8 GB per sample at L = 352)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-op-profile', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus or world == 1 and args.gpus == 1, f'--gpus {args.gpus} but WORLD_SIZE {world}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
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
    keys = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'sd_keys.json')))
    params = synthetic.random_state_dict(OrderedDict((k, tuple(s)) for k, s in keys), seed=7)
    D = FullDiffuser(cfg.diffuser).to(dev)                     # IGSO(3) tables by the HIP kernel
    model = ScoreNetwork(cfg.model, D)
    model.load_state_dict(params, strict=True)
    model = model.to(dev).eval()
    model.max_chunk = args.chunk

    w = synthetic.WORKLOADS[args.workload]
    B = args.samples
    cx = synthetic.make_complex(seed=1, **w)
    L = cx['seq'].shape[0]
    raw = {k: v.to(dev) for k, v in synthetic.replicate(cx, B).items()}
    torch.manual_seed(1234 + rank)
    batch = features.build_features(raw, D)
    batch['_shared_context'] = True
    diffuse_mask = ((1 - batch['fixed_mask']) * batch['atom14_gt_exists'][..., 0]).to(torch.int32)
    ones = torch.ones(B, device=dev)
    sid = torch.arange(B, device=dev) + rank * B
    grid = np.linspace(0.01, 1.0, 100)[::-1]
    dt = float(np.float32(0.01))
    D.seed = 2024

    def one_step(k):
        t_ = torch.full((B,), float(grid[k % 99]), device=dev, dtype=torch.float64)
        sampler.set_t_feats(batch, D, t_, ones)
        out = model(batch)
        f = out['heads']['folding']
        batch.update(get_prev(batch, out, cfg.model))
        rig, seq = D.reverse(rigid_t=batch['rigids_t'], seq_t=batch['seq_t'], rot_score=f['rot_score'], trans_score=f['trans_score'],
                             logits_t=out['heads']['sequence_module']['logits'], diffuse_mask=diffuse_mask, t=t_, dt=dt,
                             sample_ids=sid, step=k)
        batch['rigids_t'], batch['seq_t'] = rig, seq
        return out

    with torch.no_grad():
        # self-conditioning warm-up call of the sampler (untimed set-up, like the IGSO(3) table build)
        sampler.set_t_feats(batch, D, float(grid[0]), ones)
        out = model(batch)
        batch.update(get_prev(batch, out, cfg.model))
        for k in range(args.warmup):
            one_step(k)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k)
        barrier()
        elapsed = time.perf_counter() - t0
        finite = bool(torch.isfinite(batch['rigids_t']).all())
    if world > 1:
        tt_ = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        elapsed = float(tt_)
    value = world * B * args.steps / elapsed

    result = {
        'metric': 'diffusion-steps/sec (100 samples, ~350-res complex)', 'value': value, 'unit': 'sample-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'dtype_note': 'fp32 storage and accumulation everywhere; the large pair-stack GEMMs evaluate each fp32 product from '
                      '3-way bf16 operand splits (6 MFMA products, fp32 accumulate: as accurate as the native fp32 MFMA, '
                      'tests/test_gpu_kernels.py), everything else is native fp32 / fp64',
        'config': {'workload': f'{args.workload}: L={L} (Lab {w["L_heavy"] + w["L_light"]} + antigen {w["L_antigen"]}), '
                               f'{B} samples/GPU of one complex, 1 step = ScoreNetwork (3 passes) + get_prev + reverse, '
                               'seeded random weights, ESM off', 'L': L, 'samples_per_gpu': B, 'chunk': args.chunk},
        'finite': finite,
        'step_hbm_frac': value / world * algorithmic_bytes_per_sample_step(L) / 1e9 / HBM_PEAK_GBS,
        'step_mfma_f32_frac': value / world * algorithmic_flops_per_sample_step(L) / 1e12 / MFMA_F32_PEAK_TF,
    }

    if rank == 0 and not args.no_op_profile:
        with torch.no_grad(), OpTimer(ops) as tm:
            one_step(args.warmup + args.steps)
        summ = tm.summary()
        total = sum(s[1] for s in summ)
        top = summ[0]
        name, ms, calls, fl, by = top
        dur = ms / 1e3                      # all launches of the kernel in one step; fl / by are summed over them too
        if name.startswith('ipa_attn'):
            roof = {'bound': 'hbm', 'achieved': by / dur / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s'}
        elif name.startswith('gemm3_kernel') or name.startswith('tri_attn3_kernel'):
            # algorithmic fp32 flops against the matrix-core peak for this arithmetic: every fp32 product costs six bf16 MFMA
            # products, so the ceiling is the dense bf16 peak / 6
            roof = {'bound': 'mfma', 'achieved': fl / dur / 1e12, 'peak': MFMA_SPLIT_PEAK_TF, 'unit': 'TFLOP/s',
                    'peak_note': 'dense bf16 MFMA peak 2516.6 TF / 6 products per fp32-accurate product; '
                                 f'{fl / dur / 1e12 / MFMA_F32_PEAK_TF:.2f} of the native fp32 MFMA peak 157.3 TF'}
        else:
            roof = {'bound': 'mfma', 'achieved': fl / dur / 1e12, 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s'}
        traffic = None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):             # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/README.md)
            traffic = json.load(open(pmc)).get(name, {}).get('hbm_bytes_per_launch')
        roof.update(frac=roof['achieved'] / roof['peak'], traffic=traffic, kernel=name, calls_per_step=calls,
                    avg_launch_ms=ms / calls, share_of_step=ms / total)
        result['roofline'] = roof
        result['op_profile_ms'] = [{'op': s[0], 'ms': round(s[1], 3), 'calls': s[2],
                                    'avg_ms': round(s[1] / s[2], 4),
                                    'tflops': (s[3] / (s[1] / 1e3) / 1e12 if s[3] else None)} for s in summ[:40]]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args, D, batch, B, L, float(grid[1]))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-baseline-worker':
        cpu_baseline_worker(sys.argv[2])
    else:
        main()
