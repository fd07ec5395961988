"""Rasterizer-only probe at BASELINE cfg2/cfg3 size: timings + workload statistics (R, list lengths)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s3gaussian_amd import synth, _debug  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=1_200_000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cam", type=int, default=0)
ap.add_argument("--stats", action="store_true")
ap.add_argument("--scale-mult", type=float, default=1.0, help="every Gaussian's scale multiplied (bench.py --scale-mult: 3.5 = the heavy-raster leg)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = synth.street_scene(P=a.P, n_frames=4)
gs = sc["gaussians"]
cam = sc["cameras"][a.cam]
xyz = gs["xyz"].to(dev).requires_grad_(True)
scales = (torch.exp(gs["log_scales"]) * a.scale_mult).to(dev).requires_grad_(True)
rot = gs["rotations_raw"].to(dev).requires_grad_(True)
op = torch.sigmoid(gs["opacity_logit"]).to(dev).requires_grad_(True)
col = torch.rand(a.P, 3, device=dev).requires_grad_(True)
rs = GaussianRasterizationSettings(image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
                                   tanfovy=cam["tanfovy"], bg=sc["bg"].to(dev), scale_modifier=1.0,
                                   viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev), sh_degree=0,
                                   campos=cam["campos"].to(dev), prefiltered=False, debug=False)
rast = GaussianRasterizer(rs)
H, W = cam["image_height"], cam["image_width"]
gc = torch.randn(3, H, W, device=dev)
gd = torch.randn(1, H, W, device=dev)
means2D = torch.zeros_like(xyz, requires_grad=True)


def step():
    color, radii, depth = rast(means3D=xyz, means2D=means2D, opacities=op, colors_precomp=col, scales=scales, rotations=rot)
    return color, radii, depth


if a.stats:
    from diff_gaussian_rasterization import _C
    e = torch.Tensor([])
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(rs.bg, xyz.detach(), col.detach(), op.detach(), scales.detach(), rot.detach(), 1.0, e,
                                                                        rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, H, W, e, 0, rs.campos, False, False)
    im = _debug.decode_image(img, W, H)
    cnt = (im["ranges"][:, 1] - im["ranges"][:, 0]).float()
    nc = im["n_contrib"].float()
    print(f"P={a.P} visible={(radii > 0).sum().item()} R={R} tiles={cnt.numel()} mean_list={cnt.mean().item():.1f} max_list={cnt.max().item():.0f} "
          f"mean_n_contrib={nc.mean().item():.1f} max_n_contrib={nc.max().item():.0f} mean_final_T={im['final_T'].mean().item():.3f}")
    edges = [0, 1, 256, 1024, 4096, 7424, 16384, 1 << 30]       # the per-tile sort's launches: <= 1024 | <= 4096 | bucket pass <= 7424 | LDS <= 16384 | global
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        m = (cnt >= lo_) & (cnt < hi_)
        print(f"  lists of [{lo_}, {hi_}): {int(m.sum())} tiles, {int(cnt[m].sum())} instances")

for _ in range(3):
    c, r, d = step()
    (c * gc).sum().add((d * gd).sum()).backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(a.iters):
    ev[0].record()
    c, r, d = step()
    ev[1].record()
    (c * gc).sum().add((d * gd).sum()).backward()
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1])
    tb += ev[1].elapsed_time(ev[2])
print(f"forward {tf / a.iters:.3f} ms  backward {tb / a.iters:.3f} ms (incl. torch loss ops)")
