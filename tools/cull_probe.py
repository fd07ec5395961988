"""How many (tile, Gaussian) instances of the reference's 3-sigma bounding square can contribute at all?
For a sample of visible Gaussians of the bench scene: exact minimum of the Mahalanobis form over every tile rectangle of
the Gaussian's rect (on a 17 x 17 lattice of the tile's pixel centres, which is what the blend kernels evaluate) against
the alpha >= 1/255 threshold.  Prints the fraction of instances that survive -- the measurement that motivated the exact
tile culling of DESIGN.md section 4.1 (49 % at cfg3).  Runs with the culling switched off to see the reference's lists."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from s3gaussian_amd import _debug, raster_C  # noqa: E402
from s3gaussian_amd.pipeline import render  # noqa: E402
from types import SimpleNamespace  # noqa: E402

dev = torch.device("cuda")
P, W, H = 1_200_000, 1600, 1066
pc, cams, hyper, opt, bg = bench.build_scene(P, W, H, 50, dev)
pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
raster_C.set_exact_cull(False)
with torch.no_grad():
    render(cams[5], pc, pipe, bg, stage="fine")
key, tensors, (R, radii, geom, binning, img) = raster_C._geom_cache
g = _debug.decode_geometry(geom, P)
vis = (radii > 0).nonzero().flatten()
sel = vis[torch.randperm(vis.numel(), device=dev)[:20000]]
m, co, rect = g["means2D"][sel], g["conic_opacity"][sel], g["rect"][sel].int()
a, b, c, o = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
thr = torch.log(255.0 * o).clamp_min(-1e30)           # alpha >= 1/255  <=>  -power <= ln(255 o)
kept = total = 0
lat = torch.arange(16, device=dev, dtype=torch.float32)
for i in range(sel.numel()):
    x0, y0, x1, y1 = [int(v) for v in rect[i]]
    if x1 <= x0 or y1 <= y0:
        continue
    tx = torch.arange(x0, x1, device=dev, dtype=torch.float32)
    ty = torch.arange(y0, y1, device=dev, dtype=torch.float32)
    px = (tx[:, None] * 16 + lat[None, :]).reshape(-1)       # pixel columns of all tiles in the rect
    py = (ty[:, None] * 16 + lat[None, :]).reshape(-1)
    dx, dy = m[i, 0] - px, m[i, 1] - py
    power = -0.5 * (a[i] * dx[None, :] ** 2 + c[i] * dy[:, None] ** 2) - b[i] * dx[None, :] * dy[:, None]   # [rows, cols]
    ok = (power <= 0) & (-power <= thr[i])
    ok = ok.view(y1 - y0, 16, x1 - x0, 16).permute(0, 2, 1, 3).reshape(y1 - y0, x1 - x0, 256).any(-1)
    kept += int(ok.sum())
    total += (x1 - x0) * (y1 - y0)
print(f"sampled {sel.numel()} visible Gaussians: {total} instances in their 3-sigma squares, {kept} can contribute "
      f"({100.0 * kept / total:.1f} %)  -> exact culling cuts R from {R} to ~{int(R * kept / total)}")
