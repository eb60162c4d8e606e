#!/usr/bin/env python
"""Shader clock a kernel ACTUALLY runs at, measured inside the kernel.

Every workgroup of the probed launch adds its elapsed s_memtime ticks (shader clock) and s_memrealtime ticks (constant 100 MHz)
to two device counters (AbxGemm.clock_probe / AbxTriAttn.clock_probe):   f_shader = 100 MHz * sum(ticks) / sum(real ticks).
The kernel loops for `seconds`; rocm-smi / amd-smi are sampled alongside for the board power and the driver's own sclk reading.
Also runs a bare busy-loop kernel (torch elementwise chain: VALU only, low power) to show what the same ratio reads un-throttled.

    python tools/probes/clock_probe.py [seconds]            (writes nothing; redirect into profiles/r03_clock.txt)
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from abx_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
DEV = 'cuda:0'
Bc, L = 10, 352
LL, M2 = L * L, Bc * L * L
r = lambda *s: torch.randn(*s, device=DEV)


def smi_sample():
    out = {}
    try:
        d = json.loads(subprocess.run(['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True,
                                      timeout=5).stdout).get('card0', {})
        for k, v in d.items():
            if 'Power' in k and 'W' in k:
                out['power_w'] = float(v)
            if k.lower().startswith('sclk clock speed') or k.lower().startswith('sclk clock level'):
                m = re.search(r'(\d+)\s*Mhz', str(v), flags=re.I)
                out['sclk_' + ('mhz' if m else 'raw')] = int(m.group(1)) if m else v
    except Exception as e:  # noqa: BLE001
        out['err'] = str(e)[:80]
    return out


def amd_smi_clock():
    try:
        txt = subprocess.run(['/opt/rocm/bin/amd-smi', 'metric', '--clock', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(txt)
        d = d[0] if isinstance(d, list) else d
        d = d.get('gpu_data', [d])[0] if isinstance(d, dict) and 'gpu_data' in d else d
        clk = d.get('clock', {})
        vals = []
        for k, v in clk.items():
            if k.lower().startswith('gfx'):
                c = v.get('clk', v.get('cur_clk')) if isinstance(v, dict) else None
                if isinstance(c, dict):
                    c = c.get('value')
                if c not in (None, 'N/A'):
                    vals.append(float(c))
        return vals
    except Exception as e:  # noqa: BLE001
        return str(e)[:80]


def run(name, fn, acc, flops):
    fn(); torch.cuda.synchronize()
    acc.zero_()
    samples, amd, stop = [], [], [False]

    def sampler():
        while not stop[0]:
            samples.append(smi_sample())
            time.sleep(0.05)

    th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
        if n % 200 == 20:
            amd.append(amd_smi_clock())
    e1.record(); torch.cuda.synchronize()
    stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / n
    ticks, real = int(acc[0]), int(acc[1])
    mhz = 100.0 * ticks / max(real, 1)
    pw = [s['power_w'] for s in samples if 'power_w' in s]
    sclk = sorted({s.get('sclk_mhz', s.get('sclk_raw')) for s in samples if ('sclk_mhz' in s or 'sclk_raw' in s)}, key=str)
    print(f'{name:34s} {ms:8.3f} ms/launch {flops / ms / 1e9 if flops else 0:7.1f} TFLOP/s | in-kernel shader clock {mhz:7.1f} MHz '
          f'(s_memtime {ticks} / s_memrealtime {real}) | power mean {sum(pw) / max(len(pw), 1):5.0f} W max {max(pw) if pw else 0:5.0f} W ({len(pw)} samples) '
          f'| rocm-smi sclk {sclk[:6]} | amd-smi gfx clk {amd[:2]}', flush=True)
    return mhz


acc = torch.zeros(2, dtype=torch.int64, device=DEV)
# (1) calibration of the tick ratio on a light kernel: a skinny exact-class GEMM that leaves the board far below its power cap
z, W = r(M2, 192), r(192, 32) / 14
C32, W3n = torch.empty(M2, 32, device=DEV), ops.split_weights(W)
run('narrow 192->32 projection (HBM)', lambda: ops.gemm(z, W, C32, B3=W3n, exact=2, clock_probe=acc), acc, 2.0 * M2 * 192 * 32)
# (2) the dominant kernels
W = r(192, 768) / 14
C, bias, csum, W3 = torch.empty(M2, 768, device=DEV), r(768), r(768), ops.split_weights(W)
run('gemm3 128x128 N=768 K=192 (qkvg)', lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2, clock_probe=acc), acc, 2.0 * M2 * 192 * 768)
W2 = r(768, 192) / 28
C2, W23 = torch.empty(M2, 192, device=DEV), ops.split_weights(W2)
run('gemm3 128x192 N=192 K=768 (trans2)', lambda: ops.gemm(C, W2, C2, bias=None, resid=z, B3=W23, exact=2, clock_probe=acc), acc, 2.0 * M2 * 192 * 768)
KT = (L + 15) // 16
lrp = (torch.randn(Bc, 256, KT, 2, L, 16, device=DEV) * 100).to(torch.int16)
tz = torch.empty(Bc * 128, L, L, device=DEV)
run('gemm3 contraction (planes)', lambda: ops.gemm(lrp[:, 0:128], lrp[:, 128:256], tz, exact=2, clock_probe=acc), acc, 2.0 * Bc * 128 * L * L * L)
del lrp, tz
bT, mask, o = r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
run('tri_attn4', lambda: ops.tri_attn(C, bT, mask, o, Bc, L, True, bias_is_qk=True, clock_probe=acc), acc, 4.0 * Bc * L * 4 * L * L * 48)
