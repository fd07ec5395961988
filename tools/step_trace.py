"""Kernel sequence of ONE training iteration from a rocprofv3 kernel trace:  python tools/step_trace.py <rocprof out dir>
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-paths)
Prints every dispatch between the last two adam_kernel launches: start offset, duration, gap to the previous kernel.
`python tools/step_trace.py DIR render`: the same for one inference FRAME of tools/render_trace.py -- between the last two
deform_infer_kernel launches."""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
MARK = "deform_infer_kernel" if "render" in sys.argv[2:] else "adam_kernel"
adam = [i for i, r in enumerate(rows) if MARK in r[2]]
a, b = adam[-2], adam[-1]
if MARK != "adam_kernel":
    a, b = a - 1, b - 1     # the frame starts WITH its marker kernel
t0 = rows[a][1]
prev = rows[a][1]
busy = 0
small = 0
for s, e, n in rows[a + 1:b + 1]:
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:90]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  {n}")
    busy += e - s
    if "s3g::" not in n:
        small += e - s
    prev = e
print(f"span {(rows[b][1] - t0) / 1e3:.1f} us, kernel-busy {busy / 1e3:.1f} us, non-s3g kernels {small / 1e3:.1f} us, launches {b - a}")
