#!/usr/bin/env python3
"""Per-stream timeline of a batched run from a rocprofv3 kernel_trace.csv: how busy each of the batch's three streams is
(kernels grouped by name as the BATCH driver queues them: scan = front end / rowscan / colscan / dog_mag, keyline = flag / emit /
join / df_lists, track = lm_chain / directed_match / tail / regularize), per step (one step = one k_lm_chain*_b launch), medians over the steady state.
A stream that is busy for (nearly) the whole step period is the one that sets it.
  batch_timeline.py <kernel_trace.csv> [skip_first_steps]"""
import csv
import re
import sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 300
clean = lambda s: re.sub(r'\(.*', '', s).replace('rh::', '').replace('void ', '')  # noqa: E731
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), clean(r['Kernel_Name'])) for r in rows), key=lambda e: e[0])


def stream_of(n):
    if n.startswith(('k_rowscan', 'k_colscan', 'k_front_end', 'k_dog_mag')):
        return 'scan'
    if n.startswith(('k_keyline', 'k_join_edges', 'k_df_')):
        return 'keyline'
    if n.startswith(('k_lm_chain', 'k_directed_match', 'k_regularize', 'k_pair_glue')):
        return 'track'
    return 'other'


lm = [e for e in ev if e[2].startswith('k_lm_chain')]
lm = lm[skip:]
if len(lm) < 20:
    sys.exit("too few steps in the trace")
t0, t1 = lm[0][0], lm[-1][0]
steps = len(lm) - 1
print(f"steps analysed: {steps}; step period {(t1 - t0) / 1e3 / steps:.1f} us")
for st in ('scan', 'keyline', 'track', 'other'):
    es = [e for e in ev if stream_of(e[2]) == st and e[1] > t0 and e[0] < t1]
    if not es:
        continue
    busy = sum(min(e[1], t1) - max(e[0], t0) for e in es) / 1e3 / steps
    # idle = time with no kernel of this stream in flight (rocprofv3 durations include the dispatch gap to the predecessor)
    print(f"  {st:8s} {busy:7.1f} us of kernels per step ({100 * busy * steps / ((t1 - t0) / 1e3):5.1f} % of the period), {len(es) / steps:.1f} launches per step")
    by = {}
    for e in es:
        by.setdefault(e[2], []).append((e[1] - e[0]) / 1e3)
    # idle time of the stream in front of each kernel (a kernel that starts when its predecessor ends was queued behind it;
    # one that starts later waited for an event of another stream, or for the host)
    idle = {}
    for a, b2 in zip(es, es[1:]):
        g = (b2[0] - a[1]) / 1e3
        if g > 0.5:
            idle.setdefault(b2[2], []).append(g)
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w = idle.get(n, [])
        print(f"      {n:28s} x{len(v) / steps:4.1f}  median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}"
              f"   | stream idle in front of it: {sum(w) / steps:6.1f} us per step ({len(w) / steps:.2f} waits per step)")
# concurrency: how many of the three streams have a kernel in flight, time-weighted
pts = []
for e in ev:
    if stream_of(e[2]) != 'other' and e[1] > t0 and e[0] < t1:
        pts.append((max(e[0], t0), 1))
        pts.append((min(e[1], t1), -1))
pts.sort()
hist = {}
cur, last = 0, t0
for t, d in pts:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
tot = sum(hist.values())
print("  kernels in flight at once: " + "  ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))
