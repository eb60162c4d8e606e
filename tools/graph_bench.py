#!/usr/bin/env python
"""Step time of the eager loop vs hipGraph replay at small per-GPU batches:
    python tools/graph_bench.py [--L352] -> one line per (B, mode)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from abx_amd import features, synthetic  # noqa: E402
from abx_amd.config import default_config  # noqa: E402
from abx_amd.diffuser.full_diffuser import FullDiffuser  # noqa: E402
from abx_amd.graph import GraphedSteps  # noqa: E402
from abx_amd.model.abx import ScoreNetwork  # noqa: E402

dev = torch.device('cuda:0')
cfg = default_config()
cfg.diffuser.so3.cache_dir = '/tmp/abx_bench_cache_0/'
D = FullDiffuser(cfg.diffuser).to(dev)
model = ScoreNetwork(cfg.model, D)
model.load_state_dict(synthetic.random_state_dict(bench.model_parameter_shapes(cfg), seed=7), strict=True)
model = model.to(dev).eval()
wl = sys.argv[1] if len(sys.argv) > 1 else 'L352'
cx = synthetic.make_complex(seed=1, **synthetic.WORKLOADS[wl])
L = cx['seq'].shape[0]
grid = np.linspace(0.01, 1.0, 100)[::-1]
for B in (1, 4, 12, 25, 100):
    for mode in ('eager', 'graph'):
        raw = {k: v.to(dev) for k, v in synthetic.replicate(cx, B).items()}
        batch = features.build_features(raw, D, noise=features.per_sample_init_noise(list(range(B)), L, 1234, dev))
        batch['_shared_context'] = True
        dm = ((1 - batch['fixed_mask']) * batch['atom14_gt_exists'][..., 0]).to(torch.int32)
        sid = torch.arange(B, device=dev)
        with torch.no_grad():
            gs = GraphedSteps(batch, cfg, D, model, dm, float(np.float32(0.01)), sid)
            if mode == 'eager':
                gs.graphs = None
                run = lambda k: gs._body() if (gs.t.fill_(float(grid[k % 99])), gs.step.fill_(k)) else None
            else:
                run = lambda k: gs.run(k, grid[k % 99])
            for k in range(3):
                run(k)
            torch.cuda.synchronize()
            n = 6 if B >= 25 else 12
            t0 = time.perf_counter()
            for k in range(3, 3 + n):
                run(k)
            torch.cuda.synchronize()
            ms = 1000 * (time.perf_counter() - t0) / n
        print(f'{wl} B={B:4d} {mode:6s} {ms:9.2f} ms/step  {B / ms * 1000:8.1f} sample-steps/s', flush=True)
        del gs, batch, raw
        torch.cuda.empty_cache()
