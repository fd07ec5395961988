"""Average SQ counters per launch for kernels matching a substring: python tools/pmc_sq.py <rocprof dir> <substr>."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.0f}  ({n} launches)")
