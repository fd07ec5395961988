"""How long the GPU idles around the rasterizer forward's one host synchronisation (the D2H read of the instance count that
sizes the binning arena, csrc/raster_forward.hip; the reference has the same sync, rasterizer_impl.cu:281-282).
From a rocprofv3 --kernel-trace CSV: gap between the end of scan_tiles_kernel and the start of the next bin_kernel<true>
(round 4: with the host-asynchronous forward the two are separated by a 32-byte copy and fill_slots_kernel only -- no host wait).

    python tools/sync_gap.py <rocprof output dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps, busy = [], []
for i, (s, e, n) in enumerate(rows):
    if "scan_tiles_kernel" in n:
        for s2, e2, n2 in rows[i + 1:i + 12]:
            if "bin_kernel<true>" in n2:
                gaps.append((s2 - e) / 1e3)
                break
            if "fill_slots_kernel" in n2:   # round 4, asynchronous forward: the slot fill sits between the two (no host wait any more)
                continue
            if not n2.startswith("__amd") and "rocclr" not in n2:   # another kernel ran in between (no instances to bin)
                break
if gaps:
    gaps.sort()
    print(f"forward host sync: {len(gaps)} forwards, GPU idle between scan_tiles and bin<true>: median {gaps[len(gaps) // 2]:.1f} us, "
          f"mean {sum(gaps) / len(gaps):.1f} us, max {gaps[-1]:.1f} us")
# overall: time between consecutive kernels summed, as a fraction of the span
span = rows[-1][1] - rows[0][0]
act = sum(e - s for s, e, _ in rows)
print(f"kernel-active fraction of the traced span: {act / span:.3f} ({len(rows)} kernels)")
