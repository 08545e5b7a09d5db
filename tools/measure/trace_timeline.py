#!/usr/bin/env python
"""Timeline of one call from a rocprofv3 --kernel-trace CSV: the trace is cut into calls at every launch of `first_kernel`
(a substring of its name), the last `keep` calls are averaged launch by launch: start offset from the call's first launch,
duration, gap to the launch before.
usage: python tools/measure/trace_timeline.py <dir with *kernel_trace.csv> <first_kernel substring> [keep]"""
import csv
import glob
import sys
from collections import defaultdict

d, first = sys.argv[1], sys.argv[2]
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
calls, cur = [], None
for s, e, n in rows:
    if first in n:
        cur = []
        calls.append(cur)
    if cur is not None:
        cur.append((s, e, n))
if not calls:
    sys.exit(f"no launch of a kernel named *{first}* in {d}")
shape = tuple(n for _, _, n in calls[-1])
same = [c for c in calls if tuple(n for _, _, n in c) == shape][-keep:]
acc = defaultdict(lambda: [0.0, 0.0, 0.0])
for c in same:
    t0, prev_end = c[0][0], c[0][0]
    for i, (s, e, n) in enumerate(c):
        a = acc[i]
        a[0] += (s - t0) / 1e3; a[1] += (e - s) / 1e3; a[2] += (s - prev_end) / 1e3
        prev_end = e
span = sum((c[-1][1] - c[0][0]) / 1e3 for c in same) / len(same)
print(f"{len(same)} calls of {len(shape)} launches averaged; first launch to last end {span:.1f} us")
print(f"{'start':>8} {'dur':>8} {'gap':>7}  kernel")
busy = 0.0
for i, n in enumerate(shape):
    a = [x / len(same) for x in acc[i]]
    busy += a[1]
    print(f"{a[0]:8.1f} {a[1]:8.1f} {a[2]:7.1f}  {n[:90]}")
print(f"kernels busy {busy:.1f} us, gaps {span - busy:.1f} us")
