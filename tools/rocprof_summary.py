#!/usr/bin/env python
"""Export the per-kernel statistics of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite database) to CSV.
usage: python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db profiles/x_kernel_stats.csv"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) < 110 else name[:107] + '...'


def main(db, out):
    con = sqlite3.connect(db)
    rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
        for n, c, t, a, p in rows:
            w.writerow([short(n), c, round(t, 1), round(a, 2), round(p, 3)])
    print('wrote', out, len(rows), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
