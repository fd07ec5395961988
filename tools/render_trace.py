"""Render-only launch profile at cfg3: one training step (so the sampler has its blocked processing order), then N no_grad renders.
Run under `rocprofv3 --kernel-trace --stats` to get per-kernel averages of an inference frame (tools/README.md)."""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from s3gaussian_amd.pipeline import render, training_step  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
DECOMP = "decomp" in sys.argv[2:]    # evaluation path: return_decomposition=True (gaussian_renderer/__init__.py:168-204)
pc, cams, hyper, opt, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev)
pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
gts = bench.make_targets(pc, cams[0], bg, hyper, seed=1)
training_step(pc, cams[0], *gts, hyper, opt, bg, stage="fine")
with torch.no_grad():
    if DECOMP:   # non-trivial dx so that both classes are populated
        for p in pc._deformation.deformation_net.pos_deform.parameters():
            p.add_(0.05 * torch.randn_like(p))
    for i in range(3):
        render(cams[i], pc, pipe, bg, stage="fine", return_decomposition=DECOMP)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        render(cams[i % len(cams)], pc, pipe, bg, stage="fine", return_decomposition=DECOMP)
    torch.cuda.synchronize()
print(f"render: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/frame over {N} frames")
