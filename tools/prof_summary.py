"""Summarise a rocprofv3 kernel_stats.csv: per-step ms = total / steps for the top kernels."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))[1:]
tot = sum(float(r[2]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms  = {tot / 1e6 / steps:.2f} ms per step over {steps:.0f} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r[0][:78]:78s} {int(r[1]):6d} calls {float(r[2]) / 1e6 / steps:8.3f} ms/step  avg {float(r[3]) / 1e3:9.1f} us")
