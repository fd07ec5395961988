"""bench.py's output contract on a LIVE run ("the contract test tests a file, not the bench"): a small workload through
the real program in a subprocess -- every path (fused / patched / import_swap / zero_diff), the roofline leg with its live hipEvent
times, the CPU baseline -- validated by the same checker the committed line of the round goes through."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_py_prints_one_line_that_follows_the_contract(gpu_device):
    from tests.util import validate_bench_line
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--P", "60000", "--width", "480", "--height", "320", "--frames", "4",
           "--steps", "4", "--warmup", "2"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    validate_bench_line(d, default_workload=False)
    assert d["steps"] == 4 and d["warmup"] == 2 and d["config"]["gaussians"] == 60000 and d["config"]["image"] == [320, 480]
    assert d["roofline"]["traffic"] is None          # PMC traffic exists for the default workload only
