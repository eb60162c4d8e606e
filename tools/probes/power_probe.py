#!/usr/bin/env python
"""Board power and shader clock while one kernel runs in a loop (rocm-smi sampled from a thread).
    python tools/probes/power_probe.py <which of tools/pmc_kernels.py> [seconds]"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from abx_amd import ops  # noqa: E402

which = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
DEV = 'cuda:0'
Bc, L = 10, 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)
if which == 'qkvg':
    z, W = r(M2, 192), r(192, 768) / 14
    C, bias, csum, W3 = torch.empty(M2, 768, device=DEV), r(768), r(768), ops.split_weights(W)
    fn = lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2)
    flops = 2.0 * M2 * 192 * 768
elif which == 'qkvg_exact':
    z, W = r(M2, 192), r(192, 768) / 14
    C, bias, csum = torch.empty(M2, 768, device=DEV), r(768), r(768)
    fn = lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), exact=1)
    flops = 2.0 * M2 * 192 * 768
elif which == 'contract':
    KT = (L + 15) // 16
    lrp = (torch.randn(Bc, 256, KT, 2, L, 16, device=DEV) * 100).to(torch.int16)
    tz = torch.empty(Bc * 128, L, L, device=DEV)
    fn = lambda: ops.gemm(lrp[:, 0:128], lrp[:, 128:256], tz, exact=2)
    flops = 2.0 * Bc * 128 * L * L * L
elif which == 'tri':
    x, bT, mask, o = r(M2, 768), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
    fn = lambda: ops.tri_attn(x, bT, mask, o, Bc, L, True, bias_is_qk=True)
    flops = 4.0 * Bc * L * 4 * L * L * 48
elif which == 'copy':
    a, b = r(M2, 768), torch.empty(M2, 768, device=DEV)
    fn = lambda: b.copy_(a)
    flops = 0.0
else:
    raise SystemExit(which)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out).get('card0', {})
            pw = next((float(v) for k, v in d.items() if 'Power' in k and 'W' in k), None)
            sclk = next((v for k, v in d.items() if k.lower().startswith('sclk clock level')), None)
            samples.append((pw, sclk))
        except Exception as e:  # noqa: BLE001
            samples.append((None, str(e)[:60]))
        time.sleep(0.05)


fn(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(20):
        fn()
    n += 20
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / n
pws = [p for p, _ in samples if p is not None]
print(f'{which:12s} {ms:8.3f} ms/launch  {flops / ms / 1e9 if flops else 0:7.1f} TFLOP/s  power samples {len(pws)}: mean {sum(pws) / max(len(pws), 1):.0f} W max {max(pws) if pws else 0:.0f} W; sclk samples: {sorted(set(s for _, s in samples if s))[-4:]}')
