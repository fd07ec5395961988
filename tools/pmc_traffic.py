"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md "HBM") of bench.py into
profiles/rNN_bench_pmc_fetch_write.csv and profiles/kernel_traffic.json (HBM bytes per launch of every s3g kernel).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w <outdir>

gfx950 correction (same guide): FETCH_SIZE counts 64 B per 128-B request -> doubled.  Both counters are in KB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(d, counter):
    """-> {kernel: (training steps used, mean counter value per launch)} over the TRAINING iterations only.
    bench.py also runs inference renders through the same kernels (no activation stash, one image, warm or cold caches).
    Dispatches are walked in order and cut into segments at every adam_kernel (the end of a training step); inside a segment
    the training step is everything from the LAST hexplane_forward launch on.  The first step (cold caches, first sort) is
    dropped when there are more."""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            name = re.sub(r"<.*", "", name)
            rows.append((int(r["Dispatch_Id"]), name, float(r["Counter_Value"])))
    rows.sort()
    steps, seg = [], []
    for _, name, v in rows:
        seg.append((name, v))
        if name == "s3g::adam_kernel":
            last_fwd = max((i for i, (n, _) in enumerate(seg) if n == "s3g::hexplane_forward_kernel"), default=0)
            steps.append(seg[last_fwd:])
            seg = []
    if len(steps) > 1:
        steps = steps[1:]
    acc = defaultdict(list)
    for st in steps:
        for name, v in st:
            if name.startswith("s3g::"):
                acc[name].append(v)
    return {k: (len(steps), sum(v) / len(v)) for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    # optional fourth argument: a pass of the same command with SQ_INSTS_VALU (wave-level VALU instructions per launch): what
    # bench.py prices the blend kernels' VALU roofline with (they are VALU / v_exp-bound, not HBM-bound)
    valu = {k: v[1] for k, v in collect(sys.argv[4], "SQ_INSTS_VALU").items()} if len(sys.argv) > 4 else {}
    os.makedirs(out, exist_ok=True)
    rows, traffic = [], {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1])):
        n = fetch.get(k, write.get(k))[0]
        f, w = fetch.get(k, (0, 0.0))[1], write.get(k, (0, 0.0))[1]
        rows.append((k, n, f, w))
        traffic[k] = int(round((2.0 * f + w) * 1024.0))
    with open(os.path.join(out, "bench_pmc_fetch_write.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg\n")
        for k, n, f, w in rows:
            fh.write(f"{k},{n},{f:.1f},{w:.1f}\n")
    launches = {}   # every hipEvent bracket of bench.py covers ONE launch since round 3 (the nine weight-gradient launches became one)
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 "
                          "--no-cpu-baseline --no-alt-paths (two passes, tools/collect_profiles.sh)",
               "launches_per_bracket": launches,
               "formula": "hbm_bytes = (2 * FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024  (gfx950 FETCH_SIZE correction, "
                          "MI355X_MICROARCH.md 'HBM'); wgrad: average over its template instances",
               "hbm_bytes_per_launch": traffic,
               "valu_wave_instructions_per_launch": {k: int(round(v)) for k, v in valu.items()}},
              open(os.path.join(out, "kernel_traffic.json"), "w"), indent=1)
    for k, n, f, w in rows[:12]:
        print(f"{k:45s} {n:4d} launches  fetch {f / 1024:9.1f} MB  write {w / 1024:9.1f} MB  -> hbm {traffic[k] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
