"""The profile-reduction helpers under tools/ on synthetic rocprofv3 CSVs (the numbers in DESIGN.md / profiles/ pass through
them): dispatch-order attribution of PMC counters to training iterations, per-kernel averages, the one-iteration trace."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def _counter_rows(counter, scale):
    # three "training iterations" (forward ... adam), the first one cold (10x the counter), plus a render-only forward between
    rows, d = [], 0
    for it in range(3):
        for name, v in (("s3g::hexplane_forward_kernel<true>(s3g::HexArgs)", 100), ("void s3g::mlp_wgrad_kernel<64, 64, false, 64>(s3g::WgradArgs)", 7),
                        ("void s3g::mlp_wgrad_kernel<3, 64, false, 64>(s3g::WgradArgs)", 5), ("s3g::adam_kernel(s3g::AdamArgs)", 50)):
            d += 1
            rows.append([d, name, counter, v * scale * (10 if it == 0 else 1)])
        d += 1
        rows.append([d, "s3g::hexplane_forward_kernel<true>(s3g::HexArgs)", counter, 999 * scale])   # a render: not a training step
    return rows


def test_pmc_traffic_attributes_counters_to_training_iterations(tmp_path):
    hdr = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    _write(str(tmp_path / "f" / "x_counter_collection.csv"), hdr, _counter_rows("FETCH_SIZE", 1))
    _write(str(tmp_path / "w" / "x_counter_collection.csv"), hdr, _counter_rows("WRITE_SIZE", 2))
    out = tmp_path / "out"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "f"), str(tmp_path / "w"), str(out)],
                   check=True, capture_output=True)
    db = json.load(open(out / "kernel_traffic.json"))["hbm_bytes_per_launch"]
    # cold first iteration dropped; the render-only forward (999) never counted: each step keeps the LAST forward before adam
    assert db["s3g::adam_kernel"] == (2 * 50 + 2 * 50) * 1024
    assert db["s3g::hexplane_forward_kernel"] == (2 * 100 + 2 * 100) * 1024
    assert db["s3g::mlp_wgrad_kernel"] == (2 * 6 + 2 * 6) * 1024          # average over the template instances


def test_pmc_summary_and_step_trace(tmp_path):
    hdr = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    rows = [[i, "s3g::adam_kernel(s3g::AdamArgs)", "SQ_INSTS_VALU", v] for i, v in enumerate([900, 100, 100, 100, 100, 100])]
    _write(str(tmp_path / "p" / "a_counter_collection.csv"), hdr, rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path / "p"), "adam"],
                       check=True, capture_output=True, text=True)
    assert "SQ_INSTS_VALU" in r.stdout and "100.0" in r.stdout          # the cold third is skipped
    thdr = ["Start_Timestamp", "End_Timestamp", "Kernel_Name"]
    trows = [[0, 1000, "s3g::adam_kernel(s3g::AdamArgs)"], [2000, 5000, "s3g::hexplane_forward_kernel<true>(s3g::HexArgs)"],
             [5000, 6000, "at::native::fill"], [7000, 9000, "s3g::adam_kernel(s3g::AdamArgs)"]]
    _write(str(tmp_path / "t" / "k_kernel_trace.csv"), thdr, trows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_trace.py"), str(tmp_path / "t")],
                       check=True, capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1]
    assert "span 8.0 us" in last and "kernel-busy 6.0 us" in last and "non-s3g kernels 1.0 us" in last and "launches 3" in last


def test_variant_specs_still_apply_to_the_tree():
    """tools/variants/*.py are the A/B specs behind the numbers DESIGN.md quotes (tools/mkvariants.py: exact-once string edits of a
    source file).  A spec whose pattern no longer occurs exactly once has rotted: the experiment could not be repeated."""
    import glob
    import os
    import runpy
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    specs = sorted(glob.glob(os.path.join(root, "tools", "variants", "*.py")))
    assert specs
    for spec in specs:
        for name, (hip, edits) in runpy.run_path(spec)["VARIANTS"].items():
            src = open(os.path.join(root, "s3gaussian_amd", "csrc", hip)).read()
            for old, new in edits:
                assert src.count(old) == 1, (os.path.basename(spec), name, old[:60])
                src = src.replace(old, new)
