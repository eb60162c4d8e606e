"""Registers / scratch / LDS / occupancy per kernel of one HIP source, from the compiler's own remarks.

    python tools/kernel_resources.py abx_amd/csrc/gemm3.hip [filter] [-DNAME ...]

Runs `hipcc -c -Rpass-analysis=kernel-resource-usage` (gfx950, the Makefile's flags) and prints one line per kernel:
VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, waves per SIMD, static LDS.  No GPU needed.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith('-')]
    defs = [a for a in sys.argv[2:] if a.startswith('-')]
    csrc = os.path.join(ROOT, 'abx_amd', 'csrc')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + csrc,
           '-I' + os.path.join(ROOT, 'include'), '-c', src, '-o', '/tmp/_kres.o', '-Rpass-analysis=kernel-resource-usage'] + defs
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode:
        sys.stderr.write(out.stderr)
        sys.exit(out.returncode)
    cur, rows = None, []
    for line in out.stderr.splitlines():
        m = re.search(r'remark:\s+(Function )?Name: (\S+)', line)
        if m:
            name = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = {'name': name.replace('(anonymous namespace)::', '').replace('void ', '')}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?:\s+(\d+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
    for r in rows:
        if flt and not any(f in r['name'] for f in flt):
            continue
        print(f"{r['name'][:72]:72s} {r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('TotalSGPRs', -1):5d} {r.get('VGPRs Spill', -1):6d} "
              f"{r.get('SGPRs Spill', -1):6d} {r.get('ScratchSize', -1):7d} {r.get('Occupancy', -1):4d} {r.get('LDS Size', -1):7d}")


if __name__ == '__main__':
    main()
