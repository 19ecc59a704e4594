#!/usr/bin/env python3
"""Track-stream timeline from a rocprofv3 kernel_trace.csv: per pair the kernel durations and the gaps between consecutive
kernels (LM -> directedMatch head -> tail -> regularize/EKF -> next LM), medians over the steady state.
  trace_gaps.py <kernel_trace.csv> [skip_first_pairs]"""
import csv
import re
import sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
clean = lambda s: re.sub(r'\(.*', '', s).replace('rh::', '').replace('void ', '')  # noqa: E731
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), clean(r['Kernel_Name'])) for r in rows), key=lambda e: e[0])
trk = [e for e in ev if e[2].startswith(('k_lm_chain', 'k_directed_match', 'k_regularize_ekf', 'k_pair_glue'))]
# split into pairs at every LM kernel
pairs, cur = [], []
for e in trk:
    if e[2].startswith('k_lm_chain') and cur:
        pairs.append(cur)
        cur = []
    cur.append(e)
pairs = pairs[skip:]
if len(pairs) < 10:
    sys.exit("too few pairs in the trace")
names = [e[2] for e in pairs[0]]
durs = {n: [] for n in names}
gaps = {}
period = []
for i in range(len(pairs) - 1):
    p, q = pairs[i], pairs[i + 1]
    if [e[2] for e in p] != names:
        continue
    for e in p:
        durs[e[2]].append((e[1] - e[0]) / 1e3)
    for a, b in zip(p, p[1:] + [q[0]]):
        gaps.setdefault(f"{a[2]} -> {b[2]}", []).append((b[0] - a[1]) / 1e3)
    period.append((q[0][0] - p[0][0]) / 1e3)
print(f"pairs analysed: {len(period)}; pair period (LM start to LM start) median {np.median(period):.2f} us, mean {np.mean(period):.2f}")
for n in names:
    print(f"  {n:28s} median {np.median(durs[n]):7.2f} us  p90 {np.percentile(durs[n], 90):7.2f}")
for g, v in gaps.items():
    print(f"  gap {g:52s} median {np.median(v):6.2f} us  p90 {np.percentile(v, 90):6.2f}")
# what else ran on the device: busy time of the other kernels per pair period
oth = [e for e in ev if e not in trk]
t0, t1 = pairs[0][0][0], pairs[-1][0][0]
busy = sum((min(e[1], t1) - max(e[0], t0)) for e in oth if e[1] > t0 and e[0] < t1) / 1e3
print(f"other streams' kernels: {busy / len(pairs):.1f} us of kernel time per pair period")
