#!/usr/bin/env python
"""Per-kernel averages of every counter in the rocprofv3 --pmc passes under <dir>/pmc*/ (counter_collection.csv)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:90] + " grid=" + r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-44s n=%4d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
