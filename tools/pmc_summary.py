"""Average rocprofv3 --pmc counter values per kernel:  python tools/pmc_summary.py <rocprof out dir> [kernel substring]
(run the counters in their own pass:  rocprofv3 --pmc A B C --kernel-trace --output-format csv -d DIR -- cmd)"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in sorted(acc.items()):
    if sub not in k:
        continue
    n = max(len(v) for v in cs.values())
    print(f"{k}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        v = v[len(v) // 3:] if len(v) >= 3 else v     # skip the cold launches
        print(f"    {c:28s} {sum(v) / len(v):16.1f}")
