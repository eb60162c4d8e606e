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
    if os.environ.get('COSCHED_DIAG') in ('a', 'b'):
        # ---- what the recycling of the output buffer has to do with it: the concatenated outputs go into PREALLOCATED buffers, 2 in turn
        # (mode a: the reuse distance of the caching allocator in the plain loop) or 256 in turn (mode b), and every launch is classified
        # on the device, row by row, against f(x) (expected), f(f(x)) (the thread saw the previous launch's output) and x (its stores missing)
        nbuf = 2 if os.environ['COSCHED_DIAG'] == 'a' else 256
        bufs = [torch.empty(M1, 20, device=dev) for _ in range(nbuf)]
        run = lambda: ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0)
        cat = lambda k: torch.cat([state[2], state[3], state[4], state[5]], dim=1, out=bufs[k % nbuf])
        restore_k(); stale = cat(0).clone(); run(); first = cat(0).clone(); run(); twice = cat(0).clone()
        torch.cuda.synchronize()
        tot = torch.zeros(4, dtype=torch.int64)
        done, it = 0, 0
        while done < n:
            stats = []
            for k in range(256):
                restore_k()
                run()
                o = cat(it); it += 1
                d = (o != first).any(1)
                stats.append(torch.stack([d.any().long(), d.sum(), (d & (o == twice).all(1)).sum(), (d & (o == stale).all(1)).sum()]))
            done += 256
            st_ = torch.stack(stats).cpu()
            tot += st_.sum(0)
            for row in st_[st_[:, 0] > 0][:3].tolist():
                if int(tot[0]) <= 12:
                    print(f'DIAG {tag} | a differing launch: {row[1]} rows differ, {row[2]} of them = f(f(x)), {row[3]} = the restored state', flush=True)
        print(f'DIAG {tag} | product rigid_update, kernel restore, outputs into {nbuf} preallocated buffers in turn: {int(tot[0])} differing launches of {done}; '
              f'differing rows {int(tot[1])}: {int(tot[2])} equal f(f(x)), {int(tot[3])} equal the restored state', flush=True)
        return
    if os.environ.get('COSCHED_DIAG') == 'q':
        # ---- the PRODUCT kernel in the pattern that differs (restore by elementwise kernels, launch, concatenate, compare), no extra kernel:
        # the outputs of 256 launches are kept until the batch's one synchronisation; a differing launch is compared, row by row, with
        #   first : f(restored state)            what every launch should give
        #   twice : f(f(restored state))         what a thread gives that still saw the PREVIOUS launch's output instead of the restore
        #   stale : the restored state itself    what is left if the kernel's store for that row never became visible
        out = lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1)
        run = lambda: ops.rigid_update(upd, fixed, state[0], state[1], state[2], state[3], state[4], state[5], M1, 10.0)
        restore_k(); stale = out().clone(); run(); first = out().clone(); run(); twice = out().clone()
        torch.cuda.synchronize()
        NB, events, done = 256, 0, 0
        kinds = {}
        while done < n:
            outs_ = []
            for k in range(NB):
                restore_k()
                run()
                outs_.append(out())
            done += NB
            flags = torch.stack([(o == first).all() for o in outs_])
            for bi in (~flags).nonzero().flatten().tolist():
                events += 1
                o = outs_[bi]
                dmask = (o != first)
                rows = dmask.any(1).nonzero().flatten()
                is_twice = bool((o[rows] == twice[rows]).all())
                is_stale = bool((o[rows] == stale[rows]).all())
                # per differing row: which of the four outputs (cur_q 0-3 | cur_t 4-6 | cur_R 7-15 | delta_q 16-19) differ
                groups = sorted(set(('cur_q' if c < 4 else 'cur_t' if c < 7 else 'cur_R' if c < 16 else 'delta_q') for c in dmask.nonzero()[:, 1].tolist()))
                kind = ('= f(f(x)): the row read the previous launch\'s output' if is_twice else '= the restored state: the row\'s stores are missing' if is_stale else 'neither')
                kinds[kind] = kinds.get(kind, 0) + 1
                if events <= 12:
                    r0 = int(rows[0])
                    print(f'DIAG {tag} | launch {done - NB + bi} (position {bi} of its batch): {rows.numel()} rows differ, first {rows[:8].tolist()} last {int(rows[-1])} '
                          f'(contiguous: {bool(int(rows[-1]) - int(rows[0]) + 1 == rows.numel())}; waves {sorted(set((rows // 64).tolist()))[:6]}; workgroups {sorted(set((rows // 256).tolist()))[:6]}); '
                          f'outputs touched: {groups}; differing rows {kind}; row {r0}: got {o[r0][dmask[r0]].tolist()[:6]} first {first[r0][dmask[r0]].tolist()[:6]} '
                          f'twice {twice[r0][dmask[r0]].tolist()[:6]}', flush=True)
        print(f'DIAG {tag} | product rigid_update, kernel restore: {events} differing launches of {done}; kinds: {kinds}', flush=True)
        return
    if os.environ.get('COSCHED_DIAG') == 'p':
        # ---- the instrumented kernel (tools/probes/probe_rigid.hip: the same rigid_update_row) writes what every thread READ
        # beside its outputs, plus HW_ID / MODE at entry and exit / ticks in the kernel - no extra kernel in the loop (the pattern that
        # differs: restore by elementwise kernels, launch, concatenate, compare; outputs and records of 256 launches are kept until the
        # batch's one synchronisation)
        import ctypes as C
        import numpy as np
        lib = C.CDLL(os.path.join(ROOT, 'tools', 'probes', 'bin', 'libprobe_rigid.so'))
        lib.probe_rigid_update.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        st_ = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        NB = 256
        dbgs = [torch.zeros(M1, 40, dtype=torch.int32, device=dev) for _ in range(NB)]
        def launch(k):
            lib.probe_rigid_update(upd.data_ptr(), fixed.data_ptr(), state[0].data_ptr(), state[1].data_ptr(), state[2].data_ptr(), state[3].data_ptr(),
                                   state[4].data_ptr(), state[5].data_ptr(), M1, 10.0, dbgs[k].data_ptr(), st_())
        out = lambda: torch.cat([state[2], state[3], state[4], state[5]], dim=1)
        expect_in = torch.cat([upd, state0[2], state0[5], state0[3], state0[4], fixed.float().view(-1, 1), state0[0], state0[1]], dim=1)   # dbg[:, :34]
        restore_k(); launch(0)
        first = out().clone()
        torch.cuda.synchronize()
        assert torch.equal(dbgs[0][:, :34].view(torch.float32)[:, :26], expect_in[:, :26]), 'record layout'
        events, done, dts = 0, 0, []
        while done < n and events < 8:
            outs_ = []
            for k in range(NB):
                restore_k()
                launch(k)
                outs_.append(out())
            done += NB
            flags = torch.stack([(o == first).all() for o in outs_])
            bad = (~flags).nonzero().flatten().tolist()
            d37 = torch.stack([d[:, 37] for d in dbgs]).cpu().numpy().astype(np.int64) & 0xffffffff
            dts.append((float(np.median(d37)), float(np.percentile(d37, 99)), float(d37.max())))
            for bi in bad:
                events += 1
                o, dbg = outs_[bi], dbgs[bi]
                dmask = (o != first)
                rows = dmask.any(1).nonzero().flatten()
                rec = dbg[:, :34].view(torch.float32)
                fx = dbg[:, 26].float()
                rec_in = torch.cat([rec[:, :26], fx.view(-1, 1), rec[:, 27:34]], dim=1)
                in_bad = (rec_in.view(torch.int32) != expect_in.view(torch.int32))
                in_bad[:, 26] = dbg[:, 26] != fixed
                in_rows = in_bad.any(1).nonzero().flatten()
                both = set(rows.tolist()) & set(in_rows.tolist())
                r0 = int(rows[0])
                hw = dbg[rows, 34].cpu().numpy() & 0xffffffff
                info = [dict(row=int(r), wave=int(h & 15), simd=int((h >> 4) & 3), cu=int((h >> 8) & 15), sh=int((h >> 12) & 1), se=int((h >> 13) & 7),
                             xcc=int(dbg[r, 38]) & 15, mode0=hex(int(dbg[r, 35]) & 0xffffffff), mode1=hex(int(dbg[r, 36]) & 0xffffffff),
                             ticks=int(dbg[r, 37]) & 0xffffffff, trapsts=hex(int(dbg[r, 39]) & 0xffffffff)) for r, h in list(zip(rows.tolist(), hw))[:6]]
                modes = torch.unique(torch.stack([dbg[:, 35], dbg[:, 36]]))
                print(f'DIAG {tag} | launch {done - NB + bi}: {int(dmask.sum())} output elements in {rows.numel()} rows differ (workgroups '
                      f'{sorted(set((rows // 256).tolist()))[:8]}); rows whose RECORDED INPUTS differ from the restored state: {in_rows.numel()} '
                      f'({len(both)} of them among the differing output rows; input columns {sorted(set(in_bad.nonzero()[:, 1].tolist()))}); '
                      f'row {r0}: out cols {dmask[r0].nonzero().flatten().tolist()} got {o[r0][dmask[r0]].tolist()} first {first[r0][dmask[r0]].tolist()}; '
                      f'recorded inputs of that row {rec_in[r0].tolist() if r0 in both else "= expected"}; MODE values seen in the launch {[hex(int(m) & 0xffffffff) for m in modes.tolist()]}; '
                      f'differing rows ran on {info}', flush=True)
        dts = np.array(dts)
        print(f'DIAG {tag} | probe rigid_update: {events} differing launches of {done}; ticks inside the kernel per thread: median {np.median(dts[:, 0]):.0f}, '
              f'99th percentile {np.median(dts[:, 1]):.0f}, max over all launches {dts[:, 2].max():.0f}', flush=True)
        return
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
