"""K coarse-stage training steps (train.py's first stage: no deformation field, RGB + depth) on the bench's scene: ms per step; meant to run
under `rocprofv3 --kernel-trace --stats` too (tools/kstats.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from s3gaussian_amd.pipeline import training_step  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mult = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
pc, cams, hyper, opt, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev, scale_mult=mult)
targets = {v: bench.make_targets(pc, cams[v], bg, hyper, seed=1000 + v) for v in range(6)}
for i in range(6):
    training_step(pc, cams[i % 6], *targets[i % 6], hyper, opt, bg, stage="coarse", densify_stats=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record()
for i in range(K):
    training_step(pc, cams[i % 6], *targets[i % 6], hyper, opt, bg, stage="coarse", densify_stats=True)
    ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(K))
print(f"coarse stage, scales x {mult}: {sum(t) / K:.3f} ms per step (median {t[K // 2]:.3f}, max {t[-1]:.3f}) = {1000 * K / sum(t):.1f} it/s")
