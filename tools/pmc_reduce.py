#!/usr/bin/env python
"""Reduce the rocprofv3 --pmc passes of tools/pmc_run.sh to one text summary per kernel (averages per launch):
    python tools/pmc_reduce.py gpurun_out/<prefix> <kernel-name substring> > profiles/r02_pmc_<x>.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main(prefix, sub):
    vals, ndisp, dur = defaultdict(float), defaultdict(set), []
    name = None
    for p in 'abcd':
        for f in glob.glob(os.path.join(prefix + '_' + p, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                if sub not in row['Kernel_Name']:
                    continue
                name = row['Kernel_Name']
                vals[row['Counter_Name']] += float(row['Counter_Value'])
                ndisp[row['Counter_Name']].add((f, row['Dispatch_Id']))
        for f in glob.glob(os.path.join(prefix + '_' + p, '**', '*kernel_trace.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                if sub in row['Kernel_Name']:
                    dur.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    if not vals:
        raise SystemExit('no rows for ' + sub)
    avg = {k: v / max(len(ndisp[k]), 1) for k, v in vals.items()}
    name = re.sub(r'\(anonymous namespace\)::', '', name or sub)
    print(f'# kernel: {name[:160]}')
    print(f'# launches per pass: {len(ndisp[next(iter(ndisp))])}; profiled launch duration (us): mean {sum(dur) / max(len(dur), 1):.1f} min {min(dur):.1f}')
    for k in sorted(avg):
        print(f'{k:34s} {avg[k]:18.1f}')
    g = avg.get
    print('\nderived:')
    if g('SQ_WAVE_CYCLES'):
        w = g('SQ_WAVE_CYCLES')
        print(f"  wave-cycle split: ACTIVE_INST_ANY {100 * g('SQ_ACTIVE_INST_ANY', 0) / w:.1f}%  WAIT_INST_ANY {100 * g('SQ_WAIT_INST_ANY', 0) / w:.1f}%  "
              f"WAIT_ANY {100 * g('SQ_WAIT_ANY', 0) / w:.1f}%   (WAIT_INST_LDS {100 * g('SQ_WAIT_INST_LDS', 0) / w:.1f}%, ACTIVE_INST_VALU {100 * g('SQ_ACTIVE_INST_VALU', 0) / w:.1f}%)")
    if g('GRBM_GUI_ACTIVE') and dur:
        t = sum(dur) / len(dur) * 1e-6
        clk = g('GRBM_GUI_ACTIVE') / 8 / t
        print(f"  GRBM_GUI_ACTIVE / 8 XCDs / duration = {clk / 1e9:.2f} GHz effective clock")
        if g('SQ_VALU_MFMA_BUSY_CYCLES'):
            # counts cycles of busy matrix pipes summed over SIMDs: utilisation = busy / (1024 SIMDs * cycles of the launch)
            cyc = g('GRBM_GUI_ACTIVE') / 8
            print(f"  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x {cyc:.3e} cycles) = {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):.1f}%")
    if g('SQ_LDS_IDX_ACTIVE'):
        print(f"  LDS: BANK_CONFLICT / IDX_ACTIVE = {100 * g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.1f}%")
    if g('FETCH_SIZE') is not None:
        print(f"  HBM read  = FETCH_SIZE(KB) x 1024 x 2 (gfx950 wide-load correction) = {g('FETCH_SIZE') * 2048 / 1e6:.1f} MB per launch")
    if g('WRITE_SIZE') is not None:
        print(f"  HBM write = WRITE_SIZE(KB) x 1024 = {g('WRITE_SIZE') * 1024 / 1e6:.1f} MB per launch")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
