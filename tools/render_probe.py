"""Render-only timing (BASELINE's second metric): ms/frame of pipeline.render under no_grad at cfg3, with and without the
feature image."""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from s3gaussian_amd.pipeline import render  # noqa: E402

dev = torch.device("cuda")
pc, cams, hyper, opt, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev)
pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
for feat in (False, True, False, True):
    with torch.no_grad():
        for i in range(3):
            render(cams[i], pc, pipe, bg, stage="fine", render_feat=feat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            render(cams[i % len(cams)], pc, pipe, bg, stage="fine", render_feat=feat)
        torch.cuda.synchronize()
    print(f"render_feat={feat}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/frame")

# evaluation path (gaussian_renderer/__init__.py:168-204): full + dynamic-only + static-only renders
with torch.no_grad():   # non-trivial dx so that both classes are populated
    for p in pc._deformation.deformation_net.pos_deform.parameters():
        p.add_(0.05 * torch.randn_like(p))
for fused in (True, False, True, False):
    pipe_d = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False, fused_decomposition=fused)
    with torch.no_grad():
        for i in range(3):
            render(cams[i], pc, pipe_d, bg, stage="fine", return_decomposition=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            render(cams[i % len(cams)], pc, pipe_d, bg, stage="fine", return_decomposition=True)
        torch.cuda.synchronize()
    print(f"return_decomposition, shared geometry={fused}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/frame")
