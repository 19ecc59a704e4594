#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/collect_pmc.sh: pmc_table.py <dir> -> <dir>/pmc.json, <dir>/resources.json
and a text table on stdout. Counters are per launch (mean over the launches of the pass); resources come from the
kernel-trace rows of the same runs (VGPR / accum VGPR / SGPR counts, LDS bytes, workgroup and grid size)."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
clean = lambda s: re.sub(r'\(.*', '', s).replace('rh::', '').replace('void ', '')  # noqa: E731
pmc = collections.defaultdict(dict)
res = {}
for path in sorted(glob.glob(os.path.join(d, 'p*', '*', '*counter_collection.csv'))):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        k = clean(r['Kernel_Name'])
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
        if k not in res:
            res[k] = {f: r.get(f) for f in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Workgroup_Size', 'Grid_Size')}
    for k, cs in acc.items():
        for c, (s, n) in cs.items():
            pmc[k][c] = {'mean': s / n, 'launches': n}
json.dump(pmc, open(os.path.join(d, 'pmc.json'), 'w'), indent=1, sort_keys=True)
json.dump(res, open(os.path.join(d, 'resources.json'), 'w'), indent=1, sort_keys=True)
cols = sorted({c for k in pmc for c in pmc[k]})
for k in sorted(pmc):
    print(k, json.dumps(res.get(k, {})))
    for c in cols:
        if c in pmc[k]:
            print(f"    {c:28s} {pmc[k][c]['mean']:16.1f}  (n={pmc[k][c]['launches']})")
