"""How far ahead of the GPU does the host run?  Enqueue N training steps without synchronising: host ms per step (Python + launch
calls; includes the rasterizer forward's one blocking D2H per step) next to the GPU's ms per step.  A step is host-bound on a box
whose CPU is busy when the two get close (observed once: 103 instead of 121 it/s with identical kernel times)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from s3gaussian_amd.pipeline import training_step  # noqa: E402

dev = torch.device("cuda")
pc, cams, hyper, opt, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev)
gts = bench.make_targets(pc, cams[0], bg, hyper, seed=1)
for i in range(5):
    training_step(pc, cams[i], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    training_step(pc, cams[i % len(cams)], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step")
if "profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        training_step(pc, cams[i % len(cams)], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
