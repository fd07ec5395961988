"""One table out of the SQ counter passes of a command (rocprofv3 --pmc, 8 SQ slots per pass on gfx950):

    python tools/sq_table.py <pass1 dir> <pass2 dir> [<tcc pass dir>] > profiles/rNN_..._sq_pmc.txt

pass 1: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES
pass 2: SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM
tcc   : TCC_ATOMIC TCC_REQ TCC_HIT TCC_MISS   (optional: L2 requests; TCC_ATOMIC = atomic line operations)
Averages per launch (the first third of a kernel's launches -- cold caches, first sorts -- is skipped when there are >= 3).
Cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) are quad-cycles summed over all waves (MI355X_MICROARCH.md)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("s3g::", "").strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: (sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) if len(v) >= 3 else sum(v) / len(v)) for c, v in cs.items()}
        out[k]["_n"] = max(len(v) for v in cs.values())
    return out


def main():
    p1, p2 = load(sys.argv[1]), load(sys.argv[2])
    p3 = load(sys.argv[3]) if len(sys.argv) > 3 else {}
    print("kernel                                            launches waveMcyc  wait% stall%   act%  valu%   VALU M | vmemRD M vmemWR M    LDS M "
          "bankconf M mfmaBusy M" + (" | atomic M  L2req M  L2hit%" if p3 else ""))
    rows = sorted(p1, key=lambda k: -p1[k].get("SQ_WAVE_CYCLES", 0.0))
    for k in rows:
        a, b, c = p1[k], p2.get(k, {}), p3.get(k, {})
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or "at::" in k or k.startswith("__amd") or "rocblas" in k.lower():
            continue
        pct = lambda x: 100.0 * a.get(x, 0.0) / wc
        line = (f"{k[:48]:49s} {a['_n']:8d} {wc / 1e6:8.1f} {pct('SQ_WAIT_ANY'):6.1f} {pct('SQ_WAIT_INST_ANY'):6.1f} {pct('SQ_ACTIVE_INST_ANY'):6.1f} "
                f"{pct('SQ_ACTIVE_INST_VALU'):6.1f} {a.get('SQ_INSTS_VALU', 0.0) / 1e6:8.1f} | {b.get('SQ_INSTS_VMEM_RD', 0.0) / 1e6:8.2f} "
                f"{b.get('SQ_INSTS_VMEM_WR', 0.0) / 1e6:8.2f} {b.get('SQ_INSTS_LDS', 0.0) / 1e6:8.2f} {b.get('SQ_LDS_BANK_CONFLICT', 0.0) / 1e6:10.2f} "
                f"{b.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1e6:10.1f}")
        if p3:
            req = c.get("TCC_REQ", 0.0)
            line += f" | {c.get('TCC_ATOMIC', 0.0) / 1e6:8.2f} {req / 1e6:8.2f} {100.0 * c.get('TCC_HIT', 0.0) / req if req else 0.0:7.1f}"
        print(line)


if __name__ == "__main__":
    main()
