#!/usr/bin/env python
"""Does the board's power limit hold the clock of the matrix kernels?  (one decisive measurement.)

One kernel loops on the full chip and on a chip of which 64 / 128 / 192 of the 256 CUs are taken away by a sleeping hog kernel
(tools/probes/probe_hog.hip: one 160 KB-LDS workgroup per CU, s_sleep until a deadline - no issue slots, next to no power).  Per
configuration: ms per launch (HIP events), the clock INSIDE the kernel (AbxGemm / AbxTriAttn.clock_probe: s_memtime against the
100 MHz s_memrealtime), board power / sclk from rocm-smi at >= 10 Hz, one raw amd-smi metric dump (power, clocks, throttle status).
Reading: if the clock and the per-CU throughput RISE when CUs are taken away, the cap on the whole board is what binds the full-grid
launch (and only bytes / products per result can make the step faster); if they do not, something in the schedule does.

    python tools/probes/power_limit.py [seconds per configuration] [kernels: qkv,mlp,tri]
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from abx_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
kernels = (sys.argv[2] if len(sys.argv) > 2 else 'qkv,mlp,tri').split(',')
DEV = 'cuda:0'
Bc, L = 10, 352
LL, M2 = L * L, Bc * L * L
ops.RANGE_CHECK = False
r = lambda *s: torch.randn(*s, device=DEV)
hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libprobe_hog.so'))
hog.probe_hog.restype = C.c_int
hog.probe_hog.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
side = torch.cuda.Stream()


def smi():
    out = {}
    try:
        d = json.loads(subprocess.run(['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True,
                                      timeout=5).stdout).get('card0', {})
        for k, v in d.items():
            if 'Power' in k and 'W' in k:
                out['power_w'] = float(v)
            if k.lower().startswith('sclk clock'):
                m = re.search(r'(\d+)\s*Mhz', str(v), flags=re.I)
                if m:
                    out['sclk'] = int(m.group(1))
    except Exception as e:  # noqa: BLE001
        out['err'] = str(e)[:80]
    return out


def raw_dump():
    """Raw text of the tools' power / clock / throttle views (whatever this amd-smi build offers), trimmed."""
    txt = []
    for cmd in (['/opt/rocm/bin/amd-smi', 'metric', '-g', '0', '--power', '--clock', '--throttle'],
                ['/opt/rocm/bin/amd-smi', 'metric', '-g', '0', '--power', '--clock'],
                ['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks', '--showperflevel', '--showmaxpower']):
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if res.returncode == 0 and res.stdout.strip():
                keep = [ln.strip() for ln in res.stdout.splitlines()
                        if re.search(r'SOCKET_POWER|PPT_|PROCHOT_VIOLATION_STATUS|THERMAL_VIOLATION_STATUS|sclk clock|Max Graphics Package Power|Current Socket', ln)
                        and not re.search(r'HOST_LIMIT', ln)]
                txt.append('$ ' + ' '.join(cmd) + '\n      ' + ' | '.join(keep[:14]))
        except Exception as e:  # noqa: BLE001
            txt.append(f'$ {" ".join(cmd)}: {str(e)[:80]}')
    return '\n'.join(txt)


def run(name, fn, acc, flops, nhog, dump):
    fn(); torch.cuda.synchronize()
    where = torch.zeros(2 * max(nhog, 1), dtype=torch.int32, device=DEV)
    if nhog:
        rc = hog.probe_hog(nhog, int((secs + 1.5) * 1e8), where.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
        time.sleep(0.3)                                   # the hog is resident before the kernel under test starts
    acc.zero_()
    samples, stop, dumped = [], [False], []

    def sampler():
        t_ = time.time()
        while not stop[0]:
            samples.append(smi())
            if dump and not dumped and time.time() - t_ > secs / 2:
                dumped.append(raw_dump())
            time.sleep(0.02)

    th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.current_stream().synchronize()
    e1.record(); torch.cuda.current_stream().synchronize()
    stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / n
    torch.cuda.synchronize()                               # (the hog's deadline)
    ticks, real = int(acc[0]), int(acc[1])
    mhz = 100.0 * ticks / max(real, 1)
    pw = [s['power_w'] for s in samples if 'power_w' in s]
    sclk = [s['sclk'] for s in samples if 'sclk' in s]
    cus = 256 - nhog
    place = ''
    if nhog:
        w = where.cpu().view(-1, 2)
        cu_set = {(int(x[1]) & 15, (int(x[0]) >> 13) & 7, (int(x[0]) >> 12) & 1, (int(x[0]) >> 8) & 15) for x in w}
        per_xcc = {}
        for c in cu_set:
            per_xcc[c[0]] = per_xcc.get(c[0], 0) + 1
        place = f' | hog on {len(cu_set)} distinct CUs, per XCC {sorted(per_xcc.items())}'
    tf = flops / ms / 1e9
    print(f'{name:26s} CUs {cus:3d} | {ms:8.3f} ms/launch {tf:7.1f} TFLOP/s = {1e3 * tf / cus:6.1f} GFLOP/s per CU | in-kernel clock {mhz:7.1f} MHz | '
          f'power mean {sum(pw) / max(len(pw), 1):5.0f} W max {max(pw) if pw else 0:5.0f} W ({len(pw)} samples in {secs:.0f} s) | '
          f'rocm-smi sclk min/mean/max {min(sclk) if sclk else 0}/{sum(sclk) / max(len(sclk), 1):.0f}/{max(sclk) if sclk else 0}{place}', flush=True)
    if dumped:
        print('    ---- raw tool output in the middle of this run')
        for ln in dumped[0].splitlines():
            print('    ' + ln)
    return ms, mhz


acc = torch.zeros(2, dtype=torch.int64, device=DEV)
print(f'# {torch.cuda.get_device_name(0)}; {secs:.0f} s per configuration; Bc = {Bc}, L = {L}')
print('# idle: ' + json.dumps(smi()))
tests = {}
z = r(M2, 192)
if 'qkv' in kernels:
    W, Wp = r(192, 576) / 14, r(192, 4) / 14
    Cq, bT = torch.empty(M2, 576, device=DEV), torch.empty(Bc, 4, LL, device=DEV)
    bias, csum, W3, bp, csp, Wp3 = r(576), W.sum(0).contiguous(), ops.split_weights(W), r(4), Wp.sum(0).contiguous(), ops.split_weights(Wp)
    tests['gemm_as q|k|v + bias'] = (lambda: ops.gemm_side(ops.gemm(z, W, Cq, bias=bias, ln=(None, csum), B3=W3, exact=2, defer=True, clock_probe=acc),
                                                            ops.gemm(z.view(Bc, LL, 192), Wp, bT.transpose(1, 2), bias=bp, ln=(None, csp), B3=Wp3, exact=2, defer=True)),
                                     2.0 * M2 * 192 * 580)
if 'mlp' in kernels:
    W1, W2 = r(192, 768) / 14, r(768, 192) / 28
    W13, W23p = ops.split_weights(W1), ops.split_weights(ops.permute_k16(W2))
    b1, b2, cs = r(768), r(192), W1.sum(0).contiguous()
    zo = torch.empty(M2, 192, device=DEV)
    tests['gemm3_mlp (transition)'] = (lambda: ops.gemm(z, W1, zo, bias=b1, ln=(None, cs), B3=W13, act=1, resid=z, exact=2, mlp=(W23p, b2), clock_probe=acc),
                                       2.0 * M2 * 192 * 768 * 2)
if 'tri' in kernels:
    x, bT2, mask, o = r(M2, 576), r(Bc, 4, LL), torch.ones(Bc, L, device=DEV), torch.empty(M2, 192, device=DEV)
    tests['tri_attn8'] = (lambda: ops.tri_attn(x, bT2, mask, o, Bc, L, True, bias_is_qk=True, clock_probe=acc), 4.0 * Bc * L * 4 * L * L * 48)
if 'copy' in kernels:
    a_, b_ = r(M2, 768), torch.empty(M2, 768, device=DEV)
    tests['copy 7.6 GB (no clock probe)'] = (lambda: b_.copy_(a_), 0.0)
for name, (fn, fl) in tests.items():
    base = None
    for nhog in (0, 64, 128, 192):
        ms, mhz = run(name, fn, acc, fl, nhog, dump=(nhog in (0, 128)))
        if base is None:
            base = (ms, mhz)
        else:
            print(f'    -> vs full grid: time x {ms / base[0]:.2f} on {(256 - nhog) / 256:.2f} of the CUs = per-CU throughput x {base[0] / ms * 256 / (256 - nhog):.2f}, '
                  f'clock x {mhz / max(base[1], 1):.2f}', flush=True)
