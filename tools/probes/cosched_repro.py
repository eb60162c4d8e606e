#!/usr/bin/env python
"""Minimal repro attempt for the co-scheduling nondeterminism of DESIGN.md section 5.

Each worker process repeats ONE kernel N times on bit-identical inputs (restored from a master copy before every launch) and compares
every output with the output of its own first launch, on the device (torch.equal -> one bool per launch, synchronised in batches).
Kernels: `rigid_update` (no LDS, no MFMA: B*L threads of scalar fp32 math), the dominant split-f16 GEMM (LDS-DMA, raw
s_barrier / s_waitcnt), the triangle attention, and a plain torch elementwise kernel as a control.
Run solo (1 worker) and with 2 / 3 workers started together on the SAME GPU:

    python tools/probes/cosched_repro.py [launches] [workers]

Prints one line per (workers, kernel): launches whose output differed from launch 0, and whether launch 0 itself agreed across workers."""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(n, tag):
    import torch
    from abx_amd import ops
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(1234)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    B, L = 12, 352
    M1 = B * L
    res = {}

    def loop(name, restore, launch, out, reps):
        bad, first = 0, None
        flags = []
        for i in range(reps):
            restore()
            launch()
            o = out()
            if first is None:
                first = o.clone()
                torch.cuda.synchronize()
                h = hashlib.sha1(first.cpu().numpy().tobytes()).hexdigest()[:12]
            else:
                flags.append(torch.equal(o, first) if False else (o == first).all())
            if len(flags) >= 256:
                bad += int((~torch.stack(flags)).sum())
                flags = []
        if flags:
            bad += int((~torch.stack(flags)).sum())
        res[name] = (bad, reps, h)

    # ---- rigid_update
    upd, fixed = 0.1 * rn(M1, 6), (torch.rand(M1, generator=g) > 0.9).int().to(dev)
    q0 = torch.nn.functional.normalize(rn(M1, 4), dim=-1)
    t0 = rn(M1, 3)
    state0 = [q0.clone(), t0.clone(), q0.clone(), t0.clone() / 10, torch.zeros(M1, 9, device=dev), torch.tensor([1., 0, 0, 0], device=dev).repeat(M1, 1)]
    state = [s.clone() for s in state0]

    def restore():
        for s, s0 in zip(state, state0):
            s.copy_(s0)
    loop('rigid_update', restore, lambda: ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0),
         lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1), n)
    # the same kernel with the inputs restored by ELEMENTWISE kernels instead of copy_ (hipMemcpyAsync device-to-device)
    def restore_k():
        for s, s0 in zip(state, state0):
            torch.add(s0, 0.0, out=s)
    if os.environ.get('COSCHED_DIAG'):
        # ---- diagnosis of a differing launch: WHICH elements differ (rows = threads of the kernel: one lane? one
        # workgroup of 128 threads?), were the kernel's read-only inputs and the restored in/out state intact, does an immediate
        # second launch from restored inputs reproduce launch 0
        if os.environ['COSCHED_DIAG'] == 'k':
            restore = restore_k                                    # inputs restored by elementwise kernels (the variant that differs most)
        ro0 = [upd.clone(), fixed.clone(), state0[0].clone(), state0[1].clone()]
        out = lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1)
        restore(); ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0)
        first = out().clone()
        events, done = 0, 0
        while done < n and events < 6:
            outs_, pres_ = [], []
            for _ in range(256):                                   # no host synchronisation inside a batch (that is what lets the processes interleave)
                restore()
                pres_.append([s.clone() for s in state])          # what the kernel is about to read (copies made by other kernels)
                ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0)
                outs_.append(out().clone())
            done += 256
            bad = (~torch.stack([(o == first).all() for o in outs_])).nonzero().flatten().tolist()
            for bi in bad:
                events += 1
                o, pre = outs_[bi], pres_[bi]
                d = (o != first)
                rows = d.any(1).nonzero().flatten()
                pre_ok = [bool((a == b).all()) for a, b in zip(pre, state0)]
                ro_ok = all(bool((a == b).all()) for a, b in zip([upd, fixed, state[0], state[1]], ro0))
                # the same launch again, now from the inputs that launch actually saw
                for s_, p_ in zip(state, pre):
                    s_.copy_(p_)
                ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0)
                again_saw = bool((out() == o).all())
                r0 = int(rows[0])
                cols = d[r0].nonzero().flatten().tolist()
                print(f'DIAG {tag} | launch {done - 256 + bi}: {int(d.sum())} elements in {rows.numel()} rows differ; rows {rows[:12].tolist()} '
                      f'(workgroups of 128 threads: {sorted(set((rows // 128).tolist()))[:8]}); row {r0} columns {cols} got {o[r0, cols].tolist()} '
                      f'first {first[r0, cols].tolist()}; in / out state as restored before the launch (cur_q, cur_t, ...): {pre_ok}; read-only inputs intact: {ro_ok}; '
                      f're-launch from the inputs that launch saw reproduces ITS output: {again_saw}', flush=True)
        print(f'DIAG {tag} | rigid_update: {events} differing launches of {done} (diagnostic loop, 256 launches per synchronisation)', flush=True)
        return
    loop('rigid_update, kernel restore', restore_k, lambda: ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0),
         lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1), n)
    # torch only, the same restore-by-copy_ / launch / concatenate pattern with an elementwise kernel in the middle
    def torch_mid():
        state[2].mul_(1.25); state[3].add_(state[1]); state[4].add_(1.0); state[5].mul_(0.5)
    loop('torch only, copy_ restore', restore, torch_mid, lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1), n)
    loop('torch only, kernel restore', restore_k, torch_mid, lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1), n)
    # ---- trivial kernels from a separately built shared object (tools/probes/probe_axpy.hip), launched through ctypes like libabx_hip
    so = os.path.join(ROOT, 'tools', 'probes', 'bin', 'libprobe_axpy.so')
    if os.path.exists(so):
        import ctypes as C
        lib = C.CDLL(so)
        lib.probe_axpy.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        lib.probe_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        xa, ya0 = rn(M1 * 4), rn(M1 * 4)
        ya = ya0.clone()
        loop('custom axpy (ctypes .so)', lambda: ya.copy_(ya0), lambda: lib.probe_axpy(xa.data_ptr(), ya.data_ptr(), 1.5, M1 * 4, st()), lambda: ya, n)
        xc, yc0 = 0.1 * rn(M1 * 3), torch.nn.functional.normalize(rn(M1, 4), dim=-1).reshape(-1).contiguous()
        yc = yc0.clone()
        loop('custom quaternion chain (ctypes .so)', lambda: yc.copy_(yc0), lambda: lib.probe_chain(xc.data_ptr(), yc.data_ptr(), M1, st()), lambda: yc, n)
    # ---- split-f16 GEMM (N = 768, K = 192), 2 samples of pair rows
    M2 = 2 * L * L
    z, W = rn(M2, 192), rn(192, 768) / 14
    C, bias, csum, W3 = torch.empty(M2, 768, device=dev), rn(768), rn(768), ops.split_weights(W)
    loop('gemm3 768x192', lambda: C.fill_(float('nan')), lambda: ops.gemm(z, W, C, bias=bias, ln=(None, csum), B3=W3, exact=2), lambda: C, max(n // 20, 50))
    # ---- triangle attention
    bT, mask, o = rn(2, 4, L * L), torch.ones(2, L, device=dev), torch.empty(M2, 192, device=dev)
    C.copy_(rn(M2, 768))
    loop('tri_attn4', lambda: o.fill_(float('nan')), lambda: ops.tri_attn(C, bT, mask, o, 2, L, True, bias_is_qk=True), lambda: o, max(n // 50, 20))
    # ---- control: torch elementwise
    x, y = rn(M1, 64), torch.empty(M1, 64, device=dev)
    loop('torch sin*x (control)', lambda: y.fill_(0), lambda: torch.mul(torch.sin(x), x, out=y), lambda: y, n)
    for k, (bad, reps, h) in res.items():
        print(f'RESULT {tag} | {k:24s} | {bad} of {reps - 1} repeats differ from launch 0 | sha1(launch 0) {h}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    for workers in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 2, 3]):
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', str(n), f'{workers} worker(s), #{i}'],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(workers)]
        outs = [p.communicate()[0] for p in procs]
        lines = [ln for o in outs for ln in o.splitlines() if ln.startswith('RESULT')]
        print('\n'.join(ln for o in outs for ln in o.splitlines() if ln.startswith('DIAG')))
        print('\n'.join(sorted(lines, key=lambda s: s.split('|')[1])))
        hashes = {}
        for ln in lines:
            hashes.setdefault(ln.split('|')[1].strip(), set()).add(ln.rsplit(' ', 1)[1])
        print(f'-- {workers} worker(s): launch-0 outputs identical across workers: ' + ', '.join(f'{k}: {len(v) == 1}' for k, v in hashes.items()) +
              f'  ({time.time() - t0:.0f} s)', flush=True)
        for o in outs:
            if 'RESULT' not in o:
                print('worker failed:', o[-800:])
