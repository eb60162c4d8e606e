#!/usr/bin/env python
"""End-to-end wall clock of the BASELINE single-GPU configurations on the reference's own example complexes:
from reading the PDB file to the last output file closed, T = 100 (101 network calls, 99 reverse steps: the self-conditioning warm-up
call is INSIDE the wall), PDB writer on.

    python tools/e2e_bench.py config2      6ct7 (L = 231), mode design, 100 samples                       -> profiles/r03_e2e_config2.json
    python tools/e2e_bench.py config5      6qd7 (L = 259), mode trajectory, 32 samples, 100 x 32 PDB files -> profiles/r03_e2e_config5.json
    python tools/e2e_bench.py config4      6ct7, mode optimize, optimize_steps 10, guidance on, 100 samples (the per-complex unit of config 4)

Seeded random weights (no checkpoint ships with the reference).  Prints one JSON line."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from abx_amd import design  # noqa: E402

PDB = os.path.join(ROOT, 'tests', 'golden', 'pdb')
CONFIGS = {
    'config2': dict(pdb='6ct7_H_L_S.pdb', args=['--mode', 'design', '--num_samples', '100']),
    'config5': dict(pdb='6qd7_X_Z_F|E.pdb', args=['--mode', 'trajectory', '--num_samples', '32']),
    'config4': dict(pdb='6ct7_H_L_S.pdb', args=['--mode', 'optimize', '--optimize_steps', '10', '--guidance', '--num_samples', '100']),
}
which = sys.argv[1] if len(sys.argv) > 1 else 'config2'
extra = sys.argv[2:]
c = CONFIGS[which]
out = tempfile.mkdtemp(prefix='abx_e2e_')
# process start-up (library load, IGSO(3) tables from the .npy cache or built by the kernel, weights) is reported separately: it is
# paid once per process, not per complex
t0 = time.perf_counter()
design.main(['--workload', 'tiny', '--num_samples', '1', '--num_t', '2', '--output_dir', os.path.join(out, 'warm')])
torch.cuda.synchronize()
t_start = time.perf_counter() - t0
t0 = time.perf_counter()
files = design.main(['--pdb_file', os.path.join(PDB, c['pdb']), '--output_dir', os.path.join(out, 'run')] + c['args'] + extra)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
tm = design.TIMINGS[0]
n, mode = tm['samples'], tm['mode']
calls = 101 if mode != 'optimize' else 11            # optimize, t0 = 0.10: warm-up + 9 reverse steps + the final call
nbytes = sum(os.path.getsize(f) for f in files)
print(json.dumps({
    'what': which, 'complex': tm['complex'], 'L': tm['L'], 'samples': n, 'mode': mode, 'T': 100, 'network_calls_per_sample': calls,
    'wall_s': round(wall, 3), 'process_startup_s': round(t_start, 3),
    'read_and_featurise_s': round(tm['read_and_featurise_s'], 3), 'sampling_s': round(tm['sampling_s'], 3),
    'writer_tail_s': round(tm['writer_tail_s'], 3), 'files_written': len(files), 'bytes_written': nbytes,
    'sample_steps_per_s_end_to_end': round(n * (calls - 1) / wall, 2),
    'sample_network_calls_per_s_sampling_only': round(n * calls / tm['sampling_s'], 2),
    'note': 'wall = PDB read -> last file closed; 1 step = 1 network call (3 passes) + get_prev + reverse; the warm-up call is inside the wall',
}))
shutil.rmtree(out, ignore_errors=True)
