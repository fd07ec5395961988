"""BASELINE full-size checks (cfg3: 1.2 M Gaussians, 1066x1600) through size-independent properties -- the CPU oracle
would need minutes here, so these assert what must hold at any size: tile ranges partition the instance list, every
tile list is sorted by (depth bits, index), the instance->position map is a permutation, blending is linear in the
colours, the backward is linear in the upstream gradient and bit-reproducible, HexPlane gradients are additive over
disjoint point sets."""
import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg3(gpu_device):
    from s3gaussian_amd import synth
    dev = gpu_device
    sc = synth.street_scene(P=1_200_000, seed=0, n_frames=4)
    gs = sc["gaussians"]
    cam = sc["cameras"][4]
    return dict(dev=dev, cam=cam, bg=torch.zeros(3, device=dev), xyz=gs["xyz"].to(dev), scales=torch.exp(gs["log_scales"]).to(dev),
                rot=torch.nn.functional.normalize(gs["rotations_raw"]).to(dev), op=torch.sigmoid(gs["opacity_logit"]).to(dev),
                aabb=sc["aabb"])


def _raw_forward(c, colors):
    from diff_gaussian_rasterization import _C
    cam, dev = c["cam"], c["dev"]
    e = torch.Tensor([])
    return _C.rasterize_gaussians(c["bg"], c["xyz"], colors, c["op"], c["scales"], c["rot"], 1.0, e, cam["viewmatrix"].to(dev),
                                  cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], cam["image_height"],
                                  cam["image_width"], e, 0, cam["campos"].to(dev), False, False)


@pytest.mark.parametrize("cull", [True, False])
def test_binning_and_sort_invariants_at_full_size(cfg3, cull):
    from s3gaussian_amd import _debug
    P = cfg3["xyz"].shape[0]
    H, W = cfg3["cam"]["image_height"], cfg3["cam"]["image_width"]
    from s3gaussian_amd import raster_C
    colors = torch.rand(P, 3, device=cfg3["dev"])
    prev = raster_C.set_exact_cull(cull)
    try:
        R, color, depth, radii, geom, binning, img = _raw_forward(cfg3, colors)
    finally:
        raster_C.set_exact_cull(prev)
    g, im, b = _debug.decode_geometry(geom, P), _debug.decode_image(img, W, H), _debug.decode_binning(binning, R)
    ranges = im["ranges"].long()
    cnt = ranges[:, 1] - ranges[:, 0]
    assert int(cnt.sum()) == R and R > 1_000_000
    assert torch.equal(ranges[1:, 0], ranges[:-1, 1]) and int(ranges[0, 0]) == 0 and int(ranges[-1, 1]) == R
    rect = g["rect"].long()
    touched = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    S = int(touched.sum())                                           # slots: tiles of the 3-sigma bounding squares
    assert (R < 0.7 * S) if cull else (R == S)                       # reference binning: every slot is an instance
    assert torch.equal(touched > 0, radii > 0)
    pl = b["point_list"].long()
    keys = (g["depths"].view(torch.int32).long()[pl] << 32) | pl      # the sort key of every list entry, rebuilt
    tile_of = torch.repeat_interleave(torch.arange(cnt.numel(), device=keys.device), cnt)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool(((keys[1:] > keys[:-1]) | ~same_tile).all())          # (depth bits << 32 | index) strictly ascending inside every tile
    assert torch.equal(torch.sort(b["keys"] & 0xFFFFFFFF).values, torch.sort(pl).values)  # same multiset as was binned
    pos = b["slot_pos"][:S].long()                                   # slot -> list position, -1 for culled tiles
    live = pos[pos >= 0]
    assert live.numel() == R and torch.equal(torch.sort(live).values, torch.arange(R, device=pos.device))
    T = im["final_T"]
    assert float(T.min()) >= 0.0 and float(T.max()) <= 1.0
    assert bool((im["n_contrib"].view(-1).long() <= cnt.max()).all())
    assert torch.isfinite(color).all() and torch.isfinite(depth).all()


def test_blend_is_linear_in_colours_and_backward_is_deterministic(cfg3):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam, dev = cfg3["cam"], cfg3["dev"]
    P = cfg3["xyz"].shape[0]
    rs = GaussianRasterizationSettings(image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
                                       tanfovy=cam["tanfovy"], bg=cfg3["bg"], scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev),
                                       projmatrix=cam["projmatrix"].to(dev), sh_degree=0, campos=cam["campos"].to(dev),
                                       prefiltered=False, debug=False)
    rast = GaussianRasterizer(rs)
    g = torch.Generator().manual_seed(0)
    c1, c2 = torch.rand(P, 3, generator=g).to(dev), torch.rand(P, 3, generator=g).to(dev)

    def run(col, gc, gd):
        leaves = [t.clone().requires_grad_(True) for t in (cfg3["xyz"], col, cfg3["op"], cfg3["scales"], cfg3["rot"])]
        xyz, cc, op, sc, rot = leaves
        m2 = torch.zeros_like(xyz, requires_grad=True)
        color, radii, depth = rast(means3D=xyz, means2D=m2, opacities=op, colors_precomp=cc, scales=sc, rotations=rot)
        ((color * gc).sum() + (depth * gd).sum()).backward()
        return color.detach(), depth.detach(), [t.grad for t in leaves] + [m2.grad]

    H, W = cam["image_height"], cam["image_width"]
    ga, gb = torch.randn(3, H, W, generator=g).to(dev), torch.randn(3, H, W, generator=g).to(dev)
    gd = torch.randn(1, H, W, generator=g).to(dev)
    img1, d1, gr1 = run(c1, ga, gd)
    img2, d2, _ = run(c2, ga, gd)
    img12, d12, _ = run(c1 + c2, ga, gd)
    assert float((img12 - (img1 + img2)).abs().max()) < 1e-4          # bg = 0: blending is linear in the colours
    assert torch.equal(d1, d2) and torch.equal(d1, d12)                # depth does not depend on colour
    _, _, gr1b = run(c1, ga, gd)
    for a, b in zip(gr1, gr1b):
        assert torch.equal(a, b)                                       # no atomics: bit-reproducible gradients
    _, _, gr_b = run(c1, gb, 0 * gd)
    _, _, gr_ab = run(c1, ga + gb, gd)
    for a, b, ab in zip(gr1, gr_b, gr_ab):
        assert rel_l2((a + b).cpu().numpy(), ab.cpu().numpy()) < 1e-4  # backward is linear in the upstream gradient


def test_hexplane_gradients_are_additive_over_points(cfg3):
    from s3gaussian_amd.hexplane import HexPlaneField
    dev = cfg3["dev"]
    torch.manual_seed(0)
    cfg = dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32, resolution=[64, 64, 64, 25])
    field = HexPlaneField(1.6, cfg, [1, 2, 4, 8])
    field.set_aabb(*cfg3["aabb"])
    with torch.no_grad():
        for p in field.grids.parameters():
            p.add_(0.1 * torch.randn_like(p))
    field = field.to(dev)
    xyz = cfg3["xyz"]
    P = xyz.shape[0]
    time = torch.full((P, 1), 0.37, device=dev)
    w = torch.randn(P, 128, device=dev)

    def grads(sel):
        for p in field.parameters():
            p.grad = None
        x = xyz[sel].clone().requires_grad_(True)
        f = field(x, time[sel])
        assert torch.isfinite(f).all()
        (f * w[sel]).sum().backward()
        return [p.grad.clone() for p in field.grids.parameters()], x.grad

    half = torch.arange(P, device=dev) % 2 == 0
    g_all, gx_all = grads(torch.ones(P, dtype=torch.bool, device=dev))
    g_a, gx_a = grads(half)
    g_b, gx_b = grads(~half)
    for ga, gb, gab in zip(g_a, g_b, g_all):
        assert rel_l2((ga + gb).cpu().numpy(), gab.cpu().numpy()) < 2e-5
    assert torch.allclose(gx_all[half], gx_a, rtol=1e-5, atol=1e-7) and torch.allclose(gx_all[~half], gx_b, rtol=1e-5, atol=1e-7)


def test_training_converges_on_a_small_scene(gpu_device):
    """End-to-end sanity of the whole iteration (sampler -> MLP -> glue -> two-image raster -> fused losses -> backward ->
    Adam): fitting a perturbed copy of a small street scene back to the renders of the original must raise the PSNR of
    every training view and lower the loss.  (PSNR against the reference's own training cannot be measured here -- no
    CUDA -- so this only guards the optimisation loop as a whole; kernel parity is covered by the other tests.)"""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, psnr, render, training_step
    dev = gpu_device
    scn = synth.street_scene(P=20_000, seed=4, width=192, height=128, n_frames=2)
    hyper, opt = default_hyper(), default_opt()
    torch.manual_seed(0)
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    pc.training_setup(opt)
    bg = scn["bg"].to(dev)
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in scn["cameras"][:3]]
    with torch.no_grad():
        targets = []
        for cam in cams:
            pkg = render(cam, pc, pipe, bg, stage="fine", render_feat=True)
            targets.append((pkg["render"].clamp(0, 1).clone(), pkg["depth"].clone(), pkg["feat"].clone()))
        g = torch.Generator().manual_seed(1)
        pc._features_dc.add_(0.3 * torch.randn(pc._features_dc.shape, generator=g).to(dev))     # wrong colours
        pc._opacity.add_(0.5 * torch.randn(pc._opacity.shape, generator=g).to(dev))             # wrong opacities
        before = [psnr(render(c, pc, pipe, bg, stage="fine")["render"].clamp(0, 1), t[0]).mean().item()
                  for c, t in zip(cams, targets)]
    losses = []
    for it in range(90):
        v = it % len(cams)
        loss, _ = training_step(pc, cams[v], *targets[v], hyper, opt, bg, stage="fine")
        losses.append(loss.item())
    with torch.no_grad():
        after = [psnr(render(c, pc, pipe, bg, stage="fine")["render"].clamp(0, 1), t[0]).mean().item()
                 for c, t in zip(cams, targets)]
    assert all(np.isfinite(losses))
    assert np.mean(losses[-9:]) < 0.7 * np.mean(losses[:9]), (losses[:9], losses[-9:])
    assert all(a > b + 1.0 for a, b in zip(after, before)), (before, after)
