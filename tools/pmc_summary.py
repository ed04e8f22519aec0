#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per kernel from a *_counter_collection.csv: pmc_summary.py FILE [substring]."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
dur = collections.defaultdict(dict)
for r in rows:
    if sub in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:78], r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[r["Kernel_Name"][:78]][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (k, c), v in sorted(agg.items()):
    print(f"{k:78s}  {c:28s} {sum(v) / len(v):16.0f}  x{len(v)}")
for k, d in sorted(dur.items()):
    print(f"{k:78s}  {'duration_ms (profiled pass)':28s} {sum(d.values()) / len(d):16.4f}  x{len(d)}")
