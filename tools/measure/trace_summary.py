#!/usr/bin/env python
"""Per-kernel mean duration and mean gap to the previous kernel from a rocprofv3 kernel trace csv (last `tail` fraction of the rows)."""
import csv
import sys
import collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
agg = collections.OrderedDict()
prev_end = None
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qoimi::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += ((s - prev_end) / 1e3 if prev_end is not None else 0.0)
    prev_end = e
for k, (n, d, g) in agg.items():
    print(f"{k[:44]:44s} n={n:4d} dur {d / n:8.2f} us  gap-before {g / n:8.2f} us")
