#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of a batched run (tools/batch_rate.py): what the scan stream's first kernel of a step waits
for. Per step k: how long after the previous step's last scan kernel (k_dog_mag_b) k_rowscan_b<0> starts, how long before that
the keyline stream's k_join_edges_b of step k - 2 had ended (the buffer-reuse event the scan stream waits on), and where the
keyline stream is at that time. Round 4, 8 lanes: the first kernel starts 21 us (median; p10 7, p90 65) after the stream's
previous kernel, the join it formally waits for ended 68 us earlier - the gap is hand-over packets and the launching thread,
not the dependency.   scan_wait.py <kernel_trace.csv>"""
import csv, sys, re, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
clean = lambda s: re.sub(r'\(.*', '', s).replace('rh::', '').replace('void ', '')
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), clean(r['Kernel_Name'])) for r in rows), key=lambda e: e[0])
rs0 = [e for e in ev if e[2].startswith('k_rowscan_b<0>')]
dog = [e for e in ev if e[2].startswith('k_dog_mag_b')]
join = [e for e in ev if e[2].startswith('k_join_edges_b')]
flag = [e for e in ev if e[2].startswith('k_keyline_flag_b')]
df = [e for e in ev if e[2].startswith('k_df_lists_b')]
n = min(len(rs0), len(dog), len(join)) - 5
out = []
for k in range(300, n):
    start = rs0[k][0]
    prev_dog_end = dog[k - 1][1]
    join2_end = join[k - 2][1]
    out.append(((start - prev_dog_end) / 1e3, (start - join2_end) / 1e3, (flag[k-1][0] - dog[k-1][1]) / 1e3, (join[k-1][1]-flag[k-1][0])/1e3))
a = np.array(out)
print("rowscan<0>(k) start - dog_mag(k-1) end: median %.1f us p10 %.1f p90 %.1f" % (np.median(a[:,0]), np.percentile(a[:,0],10), np.percentile(a[:,0],90)))
print("rowscan<0>(k) start - join(k-2) end:    median %.1f us p10 %.1f p90 %.1f" % (np.median(a[:,1]), np.percentile(a[:,1],10), np.percentile(a[:,1],90)))
print("flag(k-1) start - dog_mag(k-1) end:     median %.1f us" % np.median(a[:,2]))
print("join(k-1) end - flag(k-1) start:        median %.1f us" % np.median(a[:,3]))
