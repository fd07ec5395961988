"""Pretty-print roofline.kernels of a bench.py JSON line (stdin or file)."""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print(f"{d['value']} {d['unit']}  {d['ms_per_step']} ms/step  render {d['config'].get('render_ms_per_frame')} ms/frame")
for k in d["roofline"]["kernels"]:
    print(f"{k['kernel'][:42]:42s} x{k['launches_per_step']:<4} {k['avg_launch_ms']:7.3f} ms  hbm {k['hbm_GBps']:7.1f} GB/s "
          f"({k['hbm_frac']:.3f})  mfma {k.get('mfma_TFLOPs', '-')}  bound {k['bound']} frac {k['frac']}")
