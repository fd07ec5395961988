"""One training iteration of a rocprofv3 kernel trace with the QUEUE of every dispatch:  python tools/stream_trace.py <rocprof out dir>
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-paths --sustain-steps 0)
The iteration = everything between two consecutive mlp_forward_kernel launches of the HEADLINE loop (the K-th and (K+1)-th from the
start of the trace, counting only intervals that contain a backward; default K = 4: past the warm-up, before the instrumented loop).  For every dispatch: start, duration, queue, and how much of it
ran while a kernel of ANOTHER queue was running (optim.OverlappedStep: Adam of the per-Gaussian groups beside mlp_backward, the weight
gradients beside the HexPlane backward)."""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ALL = "all" in sys.argv[3:]      # every dispatch, the small launches of the main queue included (the zero-edit route's iteration: K past the fused loops)
marks = [i for i, r in enumerate(rows) if "mlp_forward_kernel" in r[2] or "mlp_forward_presplit_kernel" in r[2]]
pairs = [(x, y) for x, y in zip(marks, marks[1:]) if any("mlp_backward" in r[2] for r in rows[x:y])]     # training iterations only
a, b = pairs[K]
step = rows[a:b]
t0 = step[0][0]
queues = sorted({r[3] for r in step})
main = max(queues, key=lambda q: sum(1 for r in step if r[3] == q))
union, cur_s, cur_e = 0, None, None
for s, e, n, q in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
for s, e, n, q in step:
    shared = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, _, q2 in step if q2 != q)
    if e - s < 20_000 and q == main and not shared and not ALL:
        continue        # the small launches of the main queue are in tools/step_trace.py
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  queue {q:>3}{'' if q == main else '*'}  beside another queue {shared / 1e3:8.1f}  {n}")
span = rows[b][0] - t0
prev_end, idle = step[0][1], 0
for s_, e_, _, _ in step[1:]:
    idle += max(0, s_ - prev_end)
    prev_end = max(prev_end, e_)
print(f"idle between kernels {idle / 1e3:.1f} us, launches {len(step)}")
print(f"span {span / 1e3:.1f} us (mlp_forward to mlp_forward), union of kernel time {union / 1e3:.1f} us, sum of kernel time "
      f"{sum(e - s for s, e, _, _ in step) / 1e3:.1f} us, queues {queues} (main {main})")
