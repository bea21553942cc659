#!/usr/bin/env python3
"""GPU idle time inside the last `window_ms` of a rocprofv3 kernel trace (csv): union of the kernel intervals, the largest gaps and the
kernels that follow them.  Usage: gpu_idle_gaps.py <kernel_trace.csv> [window_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]) for r in rows)
t1 = max(e[1] for e in ev)
start = t1 - int(win * 1e6)
cur_end, busy, gaps = start, 0, []
for s, e, n in ev:
    if e <= start:
        continue
    s = max(s, start)
    if s > cur_end:
        gaps.append((s - cur_end, n, cur_end - start))
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
tot = cur_end - start
print("window %.1f ms  busy %.1f ms  idle %.1f ms (%.1f %%)" % (tot / 1e6, busy / 1e6, (tot - busy) / 1e6, 100 * (tot - busy) / tot))
gaps.sort(reverse=True)
print("largest gaps: us, following kernel, position in the window (ms)")
for g, n, at in gaps[:14]:
    print("%9.1f  %-44s %8.1f" % (g / 1e3, n, at / 1e6))
c = collections.Counter()
for g, n, at in gaps:
    c[n] += g
print("idle time by following kernel (ms):")
for n, g in c.most_common(8):
    print("%8.2f %s" % (g / 1e6, n))
small = [g for g, n, a in gaps if g < 20e3]
print("gaps < 20 us: %.1f ms in %d gaps" % (sum(small) / 1e6, len(small)))
