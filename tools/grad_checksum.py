"""Bit-level checksums of every rasterizer gradient (single-image node and the two-image node, cfg3-shaped scene) -- one JSON line, so
that two builds (`S3G_LIB_PATH`, tools/mkvariants.py) can be compared bit for bit across processes:  python tools/grad_checksum.py [P] [scale]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s3gaussian_amd import synth  # noqa: E402
from s3gaussian_amd.rasterizer import _RasterizeGaussiansPair  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0     # larger splats: rects beyond the 24-tile cooperative threshold occur
dev = torch.device("cuda", 0)
sc = synth.street_scene(P=P, n_frames=2)
gs, cam = sc["gaussians"], sc["cameras"][1]
g = torch.Generator().manual_seed(5)
H, W = cam["image_height"], cam["image_width"]
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=sc["bg"].to(dev),
                                   scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev),
                                   sh_degree=0, campos=cam["campos"].to(dev), prefiltered=False, debug=False)
leaf = lambda t: t.to(dev).clone().requires_grad_(True)
bits = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum())
out = {"P": P, "scale": SCALE, "lib": os.environ.get("S3G_LIB_PATH", "tree")}
gc_, gd_, g2_ = (torch.randn(c, H, W, generator=g).to(dev) for c in (3, 1, 3))
for name in ("single", "pair"):
    xyz, sca, rot = leaf(gs["xyz"]), leaf(torch.exp(gs["log_scales"]) * SCALE), leaf(gs["rotations_raw"])
    op, col, col2 = leaf(torch.sigmoid(gs["opacity_logit"])), leaf(torch.rand(P, 3, generator=g)), leaf(torch.rand(P, 3, generator=g))
    m2 = torch.zeros_like(xyz, requires_grad=True)
    if name == "single":
        color, radii, depth = GaussianRasterizer(rs)(means3D=xyz, means2D=m2, opacities=op, colors_precomp=col, scales=sca, rotations=rot)
        ((color * gc_).sum() + (depth * gd_).sum()).backward()
        leaves = dict(xyz=xyz, m2=m2, op=op, col=col, sca=sca, rot=rot)
    else:
        color, radii, depth, color2 = _RasterizeGaussiansPair.apply(xyz, m2, col, col2, op, sca, rot, torch.Tensor([]), rs)
        ((color * gc_).sum() + (depth * gd_).sum() + (color2 * g2_).sum()).backward()
        leaves = dict(xyz=xyz, m2=m2, op=op, col=col, col2=col2, sca=sca, rot=rot)
    out[name] = {k: bits(v.grad) for k, v in leaves.items()}
    out[name]["visible"] = int((radii > 0).sum())
print(json.dumps(out))
