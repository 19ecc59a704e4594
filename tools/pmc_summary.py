#!/usr/bin/env python3
"""Per-kernel mean of a rocprofv3 --pmc counter_collection.csv (FETCH_SIZE / WRITE_SIZE are reported in KiB)."""
import csv, re, sys, collections, json
out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    cname = None
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('rh::', '').replace('void ', '')
        cname = r['Counter_Name']
        a = acc[name]
        a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (s, n) in acc.items():
        out.setdefault(k, {})[cname] = {"mean_KiB_per_launch": s / n, "launches": n}
print(json.dumps(out, indent=1, sort_keys=True))
