#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into 'kernel calls avg_us min_us max_us pct' lines."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = re.sub(r'\(.*', '', r['Name']).replace('rh::', '').replace('void ', '')
    print(f"{name:28s} calls={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:8.2f}us min={int(r['MinNs'])/1e3:7.2f} max={int(r['MaxNs'])/1e3:7.2f} pct={float(r['Percentage']):5.2f}")
