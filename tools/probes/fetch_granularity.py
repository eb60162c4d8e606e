"""How does FETCH_SIZE (x 2, the gfx950 correction of MI355X_MICROARCH.md) count the access pattern of the triangle attention?  One head's
192-byte slice of every 3 072-byte (q|k|v|gate) row, read by a plain torch copy kernel, against a contiguous read of as many bytes.
Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python tools/probes/fetch_granularity.py
then python tools/probes/fetch_granularity.py reduce <dir>."""
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == 'reduce':
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == 'FETCH_SIZE':
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'][:60], float(r['Counter_Value'])))
    per = {}
    for d, k, v in rows:
        per.setdefault((d, k), 0.0)
        per[(d, k)] += v
    for (d, k), v in sorted(per.items()):
        if v * 2048 > 1e8:
            print(f'dispatch {d:4d}  {k:60s}  FETCH_SIZE x 2 = {v * 2048 / 1e6:9.1f} MB')
    sys.exit(0)

import torch
M = 20 * 352 * 352
x = torch.randn(M, 768, device='cuda')
true_slice = M * 192 / 1e6
print(f'rows {M}; one head slice = {true_slice:.1f} MB of useful bytes, 4 heads x (q, k, v, gate) = {16 * true_slice:.1f} MB')
torch.cuda.synchronize()
outs = []
for part in range(4):                # q, k, v, gate
    for h in range(4):
        outs.append(x[:, part * 192 + h * 48: part * 192 + (h + 1) * 48].contiguous())      # 192-byte slices at a 3 072-byte stride
torch.cuda.synchronize()
flat = torch.randn(M * 48, device='cuda')
c = flat.clone()                     # the same number of bytes, contiguous
torch.cuda.synchronize()
