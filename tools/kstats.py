"""python tools/kstats.py <rocprofv3 --stats output dir> [N]: the N heaviest kernels of a `rocprofv3 --kernel-trace --stats` run."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
if not f:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f[0])))[:n]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f} total_ms {float(r['TotalDurationNs']) / 1e6:8.2f}")
