#!/usr/bin/env python
"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) over
the same bench command -> profiles/pmc_traffic.json, read by bench.py for roofline.traffic.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 1 ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json
Corrections (MI355X_MICROARCH.md, HBM section; checked against a known byte count in profiles/r01c_pmc_gemm.txt):
FETCH_SIZE is in KB and counts 128-byte requests as 64 bytes on gfx950 -> bytes = KB * 1024 * 2; WRITE_SIZE bytes = KB * 1024.
The table is stamped (`_meta`) with the library it was collected on (size, sha256 of abx_amd/csrc/libabx_hip.so or $ABX_HIP_LIB) and
whatever `key=value` pairs follow the three paths (bench geometry: samples=100 L=352; git_commit=<hash> from the calling side - the GPU
box has no .git): bench.py reports `traffic_stale` when the running library or the geometry differ."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(Abx\w+\)$', '', name)
    name = re.sub(r'\(.*\)$', '', name)
    return name.strip()


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            key = (f, row['Dispatch_Id'])
            per_dispatch[key] += float(row['Counter_Value'])
            names[key] = short(row['Kernel_Name'])
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    return acc


def source_sha256(root):
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.h, Makefile, include/abx_hip.h; sorted by name): hipcc embeds
    the build directory in the shared object, so the same sources built at another path give another library hash of the same size."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(root, 'abx_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')) + [os.path.join(csrc, 'Makefile'), os.path.join(root, 'include', 'abx_hip.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def library_stamp():
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    path = os.environ.get('ABX_HIP_LIB') or os.path.join(root, 'abx_amd', 'csrc', 'libabx_hip.so')
    if not os.path.exists(path):
        return {}
    return {'lib_bytes': os.path.getsize(path), 'lib_sha256': hashlib.sha256(open(path, 'rb').read()).hexdigest(), 'src_sha256': source_sha256(root)}


def main(fetch_dir, write_dir, out, *meta):
    fe, wr = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f = sum(fe.get(k, [0])) / max(len(fe.get(k, [])), 1) * 1024 * 2
        w = sum(wr.get(k, [0])) / max(len(wr.get(k, [])), 1) * 1024
        res[k] = {'launches': len(fe.get(k, [])), 'hbm_read_bytes_per_launch': f, 'hbm_write_bytes_per_launch': w,
                  'hbm_bytes_per_launch': f + w}
    stamp = library_stamp()
    for kv in meta:
        k, _, v = kv.partition('=')
        stamp[k] = int(v) if v.isdigit() else v
    res['_meta'] = stamp
    json.dump(res, open(out, 'w'), indent=1)
    del res['_meta']
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:5d} read {v['hbm_read_bytes_per_launch'] / 1e6:9.1f} MB write {v['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == '__main__':
    main(*sys.argv[1:])
