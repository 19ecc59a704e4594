#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel-trace CSV as a per-queue timeline (diagnostic)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
start = int(sys.argv[2]) if len(sys.argv) > 2 else n // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 70
w = rows[start:start + cnt]
t0 = int(w[0]['Start_Timestamp'])
for r in w:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} dur={(e-s)/1e3:6.1f} q={r.get('Queue_Id')} {r['Kernel_Name'][:44]}")
