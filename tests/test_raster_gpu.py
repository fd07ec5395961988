"""Parity of the HIP rasterizer (through the C ABI / drop-in Python surface) against the CPU oracle.

Bars:
  * integer / index work (radii, tile rects, num_rendered, per-tile sorted lists): BIT-EXACT;
  * fp32 render / depth: max abs error <= 1e-4 (colour in [0,1]); depth relative <= 1e-4 -- except pixels where a
    borderline skip test (alpha ~ 1/255, T ~ 1e-4) flips because the GPU's exp2/rcp differ from libm by an ulp:
    at most 0.05% of pixels may exceed the tolerance and n_contrib may differ there;
  * gradients: relative L2 error per tensor <= 1e-4 versus the fp32 oracle (which accumulates in double).
"""
import math

import numpy as np
import pytest
import torch

from tests.util import cam_kwargs, oracle_forward, rel_l2, settings_from, tiny_scene, to_np

pytestmark = pytest.mark.gpu

COLOR_TOL = 1e-4
GRAD_TOL = 1e-4
OUTLIER_FRAC = 5e-4


@pytest.fixture(scope="module")
def oracle():
    from oracle.oracle import RasterOracle
    return RasterOracle(np.float32)


def run_gpu(s, dev, mode="precomp", sh_degree=3, debug=False, grads=None, cov3D=None, seed_grad=5):
    from diff_gaussian_rasterization import GaussianRasterizer
    rs = settings_from(s, dev, sh_degree=sh_degree if mode == "sh" else 0, debug=debug)
    rast = GaussianRasterizer(raster_settings=rs)
    t = lambda x: x.to(dev).clone().requires_grad_(True)
    means3D, op = t(s["means3D"]), t(s["opacities"])
    means2D = torch.zeros_like(means3D, requires_grad=True)
    kw = {}
    leaves = dict(means3D=means3D, means2D=means2D, opacities=op)
    if cov3D is None:
        leaves["scales"], leaves["rotations"] = t(s["scales"]), t(s["rotations"])
        kw.update(scales=leaves["scales"], rotations=leaves["rotations"])
    else:
        leaves["cov3D"] = t(cov3D)
        kw.update(cov3D_precomp=leaves["cov3D"])
    if mode == "precomp":
        leaves["colors"] = t(s["colors_precomp"])
        kw.update(colors_precomp=leaves["colors"])
    else:
        leaves["shs"] = t(s["shs"])
        kw.update(shs=leaves["shs"])
    color, radii, depth = rast(means3D=means3D, means2D=means2D, opacities=op, **kw)
    out = dict(color=color, radii=radii, depth=depth, leaves=leaves)
    if grads is not None:
        gc, gd = grads
        (color * gc.to(dev)).sum().add((depth * gd.to(dev)).sum()).backward()
    return out


def check_forward(gpu, ref, H, W):
    np.testing.assert_array_equal(gpu["radii"].cpu().numpy(), ref["radii"])
    c, d = gpu["color"].detach().cpu().numpy(), gpu["depth"].detach().cpu().numpy()
    bad = (np.abs(c - ref["color"]).max(0) > COLOR_TOL) | (np.abs(d - ref["depth"])[0] > 1e-4 * (1 + np.abs(ref["depth"][0])))
    assert bad.mean() <= OUTLIER_FRAC, f"{bad.sum()} of {bad.size} pixels outside tolerance"
    good = ~bad
    assert np.abs(c - ref["color"])[:, good].max() <= COLOR_TOL
    return bad


def grad_pair(H, W, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g)


def backprop_masked(gpu, gc, gd, bad, dev):
    """Back-propagates the upstream gradient with the flipped / out-of-tolerance pixels `bad` ([H,W] bool) zeroed, and
    returns the same masked pair for the oracle: both sides then differentiate the same set of blended pairs, so the
    gradient comparison never depends on the seed producing no borderline flip."""
    keep = torch.from_numpy(~bad).to(torch.float32)
    gcm, gdm = gc * keep, gd * keep
    (gpu["color"] * gcm.to(dev)).sum().add((gpu["depth"] * gdm.to(dev)).sum()).backward()
    return gcm.numpy(), gdm.numpy()


@pytest.mark.parametrize("mode", ["precomp", "sh"])
@pytest.mark.parametrize("shape", [(48, 40), (64, 64), (33, 17)])
def test_forward_backward_vs_oracle(gpu_device, oracle, mode, shape):
    W, H = shape
    s = tiny_scene(P=400, W=W, H=H, seed=1)
    gc, gd = grad_pair(H, W)
    gpu = run_gpu(s, gpu_device, mode=mode)
    ref = oracle_forward(oracle, s, mode=mode)
    bad = check_forward(gpu, ref, H, W)
    rb = oracle.backward(ref, *backprop_masked(gpu, gc, gd, bad, gpu_device))
    L = gpu["leaves"]
    assert rel_l2(L["means3D"].grad.cpu().numpy(), rb["dL_dmeans3D"]) <= GRAD_TOL
    assert rel_l2(L["means2D"].grad.cpu().numpy(), rb["dL_dmeans2D"]) <= GRAD_TOL
    assert rel_l2(L["opacities"].grad.cpu().numpy(), rb["dL_dopacity"]) <= GRAD_TOL
    assert rel_l2(L["scales"].grad.cpu().numpy(), rb["dL_dscales"]) <= GRAD_TOL
    assert rel_l2(L["rotations"].grad.cpu().numpy(), rb["dL_drotations"]) <= GRAD_TOL
    if mode == "precomp":
        assert rel_l2(L["colors"].grad.cpu().numpy(), rb["dL_dcolors"]) <= GRAD_TOL
    else:
        assert rel_l2(L["shs"].grad.cpu().numpy(), rb["dL_dsh"]) <= GRAD_TOL


def test_internal_state_bit_exact(gpu_device, oracle, reference_binning):
    """Geometry state, instance count and per-tile sorted lists are integer/index work: exact equality."""
    from s3gaussian_amd import _debug
    from diff_gaussian_rasterization import _C
    W, H = 96, 80
    s = tiny_scene(P=2000, W=W, H=H, seed=3, scale=0.06)
    dev = gpu_device
    cam = s["cam"]
    e = torch.Tensor([])
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        s["bg"].to(dev), s["means3D"].to(dev), s["colors_precomp"].to(dev), s["opacities"].to(dev), s["scales"].to(dev),
        s["rotations"].to(dev), 1.0, e, cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"],
        H, W, e, 0, cam["campos"].to(dev), False, False)
    ref = oracle_forward(oracle, s)
    st = ref["state"]
    assert R == ref["num_rendered"]
    P = 2000
    g = _debug.decode_geometry(geom, P)
    vis = ref["radii"] > 0
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    # contraction is off in the per-Gaussian kernels: fp32 results are bit-identical to the oracle's
    np.testing.assert_array_equal(g["means2D"].cpu().numpy()[vis], st["means2D"][vis])
    np.testing.assert_array_equal(g["depths"].cpu().numpy()[vis], st["depths"][vis])
    np.testing.assert_array_equal(g["conic_opacity"].cpu().numpy()[vis], st["conic_opacity"][vis])
    np.testing.assert_array_equal(g["cov3D"].cpu().numpy()[vis], st["cov3D"][vis])
    rect = g["rect"].cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])), st["tiles_touched"])
    im = _debug.decode_image(img, W, H)
    ranges = im["ranges"].cpu().numpy()
    ref_ranges = st["ranges"].astype(np.int64)
    np.testing.assert_array_equal(ranges[:, 1] - ranges[:, 0], ref_ranges[:, 1] - ref_ranges[:, 0])
    nonempty = (ref_ranges[:, 1] - ref_ranges[:, 0]) > 0
    np.testing.assert_array_equal(ranges[nonempty], ref_ranges[nonempty])
    b = _debug.decode_binning(binning, R)
    np.testing.assert_array_equal(b["point_list"].cpu().numpy().astype(np.uint32), st["point_list"])


def test_depth_ties_resolve_by_index(gpu_device, oracle, reference_binning):
    """Equal depths: order inside a tile must be ascending Gaussian index (stable radix order of the reference)."""
    from s3gaussian_amd import _debug
    from diff_gaussian_rasterization import _C
    W, H = 32, 32
    s = tiny_scene(P=500, W=W, H=H, seed=7, scale=0.2)
    s["means3D"][:, 2] = 4.0  # all at the same view depth
    dev = gpu_device
    cam = s["cam"]
    e = torch.Tensor([])
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        s["bg"].to(dev), s["means3D"].to(dev), s["colors_precomp"].to(dev), s["opacities"].to(dev), s["scales"].to(dev),
        s["rotations"].to(dev), 1.0, e, cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"],
        H, W, e, 0, cam["campos"].to(dev), False, False)
    ref = oracle_forward(oracle, s)
    assert R == ref["num_rendered"]
    pl = _debug.decode_binning(binning, R)["point_list"].cpu().numpy().astype(np.uint32)
    np.testing.assert_array_equal(pl, ref["state"]["point_list"])


def test_long_tile_lists_use_every_sort_path(gpu_device, oracle, reference_binning):
    """> 4096 and > 16384 instances in one tile: second LDS launch and the in-global-memory network."""
    from s3gaussian_amd import _debug
    from diff_gaussian_rasterization import _C
    W, H = 32, 16
    P = 18000
    s = tiny_scene(P=P, W=W, H=H, seed=11, scale=0.02, spread=0.25)
    s["opacities"] = s["opacities"] * 0.05  # keep transmittance alive so many Gaussians contribute
    dev = gpu_device
    cam = s["cam"]
    e = torch.Tensor([])
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        s["bg"].to(dev), s["means3D"].to(dev), s["colors_precomp"].to(dev), s["opacities"].to(dev), s["scales"].to(dev),
        s["rotations"].to(dev), 1.0, e, cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"],
        H, W, e, 0, cam["campos"].to(dev), False, False)
    ref = oracle_forward(oracle, s)
    counts = ref["state"]["ranges"][:, 1].astype(np.int64) - ref["state"]["ranges"][:, 0]
    assert counts.max() > 16384 or counts.max() > 4096, counts.max()
    assert R == ref["num_rendered"]
    pl = _debug.decode_binning(binning, R)["point_list"].cpu().numpy().astype(np.uint32)
    np.testing.assert_array_equal(pl, ref["state"]["point_list"])
    bad = (np.abs(color.cpu().numpy() - ref["color"]).max(0) > COLOR_TOL)
    assert bad.mean() <= 0.01


def test_cov3d_precomp_path(gpu_device, oracle):
    W, H = 48, 48
    s = tiny_scene(P=300, W=W, H=H, seed=2)
    f0 = oracle_forward(oracle, s)
    cov3D = torch.from_numpy(f0["state"]["cov3D"].copy())
    gc, gd = grad_pair(H, W)
    gpu = run_gpu(s, gpu_device, cov3D=cov3D)
    ref = oracle_forward(oracle, s, scales=None, rotations=None, cov3D_precomp=cov3D.numpy())
    bad = check_forward(gpu, ref, H, W)
    rb = oracle.backward(ref, *backprop_masked(gpu, gc, gd, bad, gpu_device))
    assert rel_l2(gpu["leaves"]["cov3D"].grad.cpu().numpy(), rb["dL_dcov3D"]) <= GRAD_TOL
    assert rel_l2(gpu["leaves"]["means3D"].grad.cpu().numpy(), rb["dL_dmeans3D"]) <= GRAD_TOL


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees(gpu_device, oracle, deg):
    W, H = 32, 32
    s = tiny_scene(P=200, W=W, H=H, seed=4)
    gc, gd = grad_pair(H, W)
    gpu = run_gpu(s, gpu_device, mode="sh", sh_degree=deg)
    ref = oracle_forward(oracle, s, mode="sh", sh_degree=deg)
    bad = check_forward(gpu, ref, H, W)
    rb = oracle.backward(ref, *backprop_masked(gpu, gc, gd, bad, gpu_device))
    assert rel_l2(gpu["leaves"]["shs"].grad.cpu().numpy(), rb["dL_dsh"]) <= GRAD_TOL
    assert rel_l2(gpu["leaves"]["means3D"].grad.cpu().numpy(), rb["dL_dmeans3D"]) <= GRAD_TOL


def test_empty_scene_returns_zero_image(gpu_device):
    """P == 0: all-zero colour/depth, background NOT applied (rasterize_points.cu:81-116)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=4, W=32, H=32)
    rs = settings_from(s, gpu_device)
    rast = GaussianRasterizer(raster_settings=rs)
    z = torch.zeros(0, 3, device=gpu_device)
    color, radii, depth = rast(means3D=z, means2D=z.clone(), opacities=torch.zeros(0, 1, device=gpu_device),
                               colors_precomp=z.clone(), scales=z.clone(), rotations=torch.zeros(0, 4, device=gpu_device))
    assert color.shape == (3, 32, 32) and depth.shape == (1, 32, 32) and radii.shape == (0,)
    assert float(color.abs().max()) == 0.0 and float(depth.abs().max()) == 0.0


def test_all_culled_gives_background(gpu_device, oracle):
    s = tiny_scene(P=50, W=32, H=32)
    s["means3D"][:, 2] = -5.0  # behind the camera
    gpu = run_gpu(s, gpu_device)
    ref = oracle_forward(oracle, s)
    assert ref["num_rendered"] == 0
    assert int(gpu["radii"].abs().sum()) == 0
    np.testing.assert_allclose(gpu["color"].detach().cpu().numpy(), ref["color"], atol=0)
    assert float(gpu["depth"].detach().abs().max()) == 0.0


def test_single_huge_gaussian(gpu_device, oracle):
    s = tiny_scene(P=1, W=80, H=48)
    s["means3D"][:] = torch.tensor([0.0, 0.0, 2.0])
    s["scales"][:] = 5.0
    s["opacities"][:] = 0.9
    gc, gd = grad_pair(48, 80)
    gpu = run_gpu(s, gpu_device)
    ref = oracle_forward(oracle, s)
    bad = check_forward(gpu, ref, 48, 80)
    rb = oracle.backward(ref, *backprop_masked(gpu, gc, gd, bad, gpu_device))
    assert rel_l2(gpu["leaves"]["opacities"].grad.cpu().numpy(), rb["dL_dopacity"]) <= GRAD_TOL


def test_argument_validation_matches_reference(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=8, W=16, H=16)
    rast = GaussianRasterizer(raster_settings=settings_from(s, gpu_device))
    d = lambda k: s[k].to(gpu_device)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=d("means3D"), means2D=d("means3D"), opacities=d("opacities"), scales=d("scales"), rotations=d("rotations"))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=d("means3D"), means2D=d("means3D"), opacities=d("opacities"), colors_precomp=d("colors_precomp"))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        rast(means3D=d("means3D")[:, :2], means2D=d("means3D"), opacities=d("opacities"), colors_precomp=d("colors_precomp"),
             scales=d("scales"), rotations=d("rotations"))


def test_non_contiguous_and_masked_inputs(gpu_device, oracle):
    """Callers slice inputs with boolean masks (gaussian_renderer/__init__.py:176-195)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=300, W=48, H=48, seed=9)
    mask = torch.arange(300) % 3 != 0
    rast = GaussianRasterizer(raster_settings=settings_from(s, gpu_device))
    d = lambda k: s[k].to(gpu_device)
    wide = torch.zeros(300, 6, device=gpu_device)
    wide[:, ::2] = d("means3D")
    color, radii, depth = rast(means3D=wide[:, ::2][mask], means2D=torch.zeros(int(mask.sum()), 3, device=gpu_device),
                               opacities=d("opacities")[mask], colors_precomp=d("colors_precomp")[mask],
                               scales=d("scales")[mask], rotations=d("rotations")[mask])
    sub = dict(s)
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations", "shs"):
        sub[k] = s[k][mask].contiguous()
    ref = oracle_forward(oracle, sub)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    bad = np.abs(color.cpu().numpy() - ref["color"]).max(0) > COLOR_TOL
    assert bad.mean() <= OUTLIER_FRAC


def test_mark_visible(gpu_device, oracle):
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=500, W=32, H=32, zmin=-1.0, zmax=1.0)
    rast = GaussianRasterizer(raster_settings=settings_from(s, gpu_device))
    vis = rast.markVisible(s["means3D"].to(gpu_device))
    assert vis.dtype == torch.bool
    ref = oracle.mark_visible(s["means3D"].numpy(), s["cam"]["viewmatrix"].numpy(), s["cam"]["projmatrix"].numpy())
    np.testing.assert_array_equal(vis.cpu().numpy(), ref)


def test_debug_mode_runs(gpu_device, oracle):
    s = tiny_scene(P=100, W=32, H=32)
    gc, gd = grad_pair(32, 32)
    gpu = run_gpu(s, gpu_device, debug=True, grads=(gc, gd))
    ref = oracle_forward(oracle, s)
    check_forward(gpu, ref, 32, 32)


def test_cfg1_full_size_vs_oracle(gpu_device, oracle):
    """BASELINE config #1 at full size: 10k Gaussians, 400x400, SH degree 3 evaluated in-kernel."""
    from s3gaussian_amd import synth
    sc = synth.cfg1_scene()
    gs = sc["gaussians"]
    s = dict(means3D=gs["xyz"], scales=torch.exp(gs["log_scales"]), rotations=gs["rotations_raw"],
             opacities=torch.sigmoid(gs["opacity_logit"]), shs=gs["shs"], colors_precomp=torch.rand(10_000, 3),
             cam=sc["cameras"][0], bg=sc["bg"])
    gc, gd = grad_pair(400, 400)
    gpu = run_gpu(s, gpu_device, mode="sh")
    ref = oracle_forward(oracle, s, mode="sh")
    bad = check_forward(gpu, ref, 400, 400)
    rb = oracle.backward(ref, *backprop_masked(gpu, gc, gd, bad, gpu_device))
    tol = GRAD_TOL
    assert rel_l2(gpu["leaves"]["means3D"].grad.cpu().numpy(), rb["dL_dmeans3D"]) <= tol
    assert rel_l2(gpu["leaves"]["shs"].grad.cpu().numpy(), rb["dL_dsh"]) <= tol
    assert rel_l2(gpu["leaves"]["scales"].grad.cpu().numpy(), rb["dL_dscales"]) <= tol


def test_geometry_cache_second_render_identical(gpu_device):
    """The feature render of an iteration reuses the RGB render's geometry (preprocess/binning/sort skipped): results and
    gradients must equal those of an un-cached call bit for bit."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import raster_C
    s = tiny_scene(P=600, W=80, H=64, seed=5)
    dev = gpu_device
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    g = torch.Generator().manual_seed(0)
    feat_cols = torch.rand(600, 3, generator=g)
    gc, gd = grad_pair(64, 80)

    def run(cache_on):
        raster_C._GEOM_CACHE_ON = cache_on
        raster_C._geom_cache = None
        t = lambda x: x.to(dev).clone().requires_grad_(True)
        m3, op, sc, rot = t(s["means3D"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])
        c1, c2 = t(s["colors_precomp"]), t(feat_cols)
        m2 = torch.zeros_like(m3, requires_grad=True)
        img1, r1, d1 = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=c1, scales=sc, rotations=rot)
        img2, r2, d2 = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=c2, scales=sc, rotations=rot)
        ((img1 * gc.to(dev)).sum() + (d1 * gd.to(dev)).sum() + (img2 * gc.to(dev).flip(0)).sum()).backward()
        return [img1, img2, d1, d2, r1, r2], [x.grad for x in (m3, op, sc, rot, c1, c2, m2)]

    try:
        hits0 = raster_C._geom_cache_hits
        o_on, g_on = run(True)
        hit = raster_C._geom_cache_hits == hits0 + 1   # the second render of the pair, through the autograd path
        o_off, g_off = run(False)
    finally:
        raster_C._GEOM_CACHE_ON = True
        raster_C._geom_cache = None
    assert hit
    for a, b in zip(o_on + g_on, o_off + g_off):
        assert torch.equal(a, b)


@pytest.mark.parametrize("P,W,H", [(600, 80, 64), (3000, 131, 77), (1, 32, 32)])
def test_forward_pair_matches_two_separate_renders(gpu_device, P, W, H):
    """GaussianRasterizer.forward_pair (one node, fused two-image backward) vs the reference's way -- two rasterizer calls
    whose gradients autograd adds.  Images are bit-identical (same forward kernels); gradients agree to fp32 round-off of
    the re-associated sums (rel-L2 1e-5; the oracle-level tolerance of the single-image path is 1e-4)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=P, W=W, H=H, seed=11)
    dev = gpu_device
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    g = torch.Generator().manual_seed(1)
    feat_cols = torch.rand(P, 3, generator=g)
    gc, gd = grad_pair(H, W)
    gc2 = torch.randn(3, H, W, generator=g)

    def run(pair):
        t = lambda x: x.to(dev).clone().requires_grad_(True)
        m3, op, sc, rot = t(s["means3D"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])
        c1, c2 = t(s["colors_precomp"]), t(feat_cols)
        m2 = torch.zeros_like(m3, requires_grad=True)
        if pair:
            img1, r1, d1, img2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=c1, colors_b=c2, scales=sc,
                                                   rotations=rot)
        else:
            img1, r1, d1 = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=c1, scales=sc, rotations=rot)
            img2, _, _ = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=c2, scales=sc, rotations=rot)
        ((img1 * gc.to(dev)).sum() + (d1 * gd.to(dev)).sum() + (img2 * gc2.to(dev)).sum()).backward()
        return [img1, img2, d1, r1], [x.grad for x in (m3, op, sc, rot, c1, c2, m2)]

    o_pair, g_pair = run(True)
    o_sep, g_sep = run(False)
    for a, b in zip(o_pair, o_sep):
        assert torch.equal(a, b)
    names = ["means3D", "opacity", "scales", "rotations", "colors_a", "colors_b", "means2D"]
    for n, a, b in zip(names, g_pair, g_sep):
        assert a.shape == b.shape, n
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, n


def test_forward_pair_only_second_image_has_gradient(gpu_device):
    """Upstream gradient on the second image alone (first image and depth unused -> None grads from autograd)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    s = tiny_scene(P=400, W=64, H=48, seed=3)
    dev = gpu_device
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    t = lambda x: x.to(dev).clone().requires_grad_(True)
    cols_b = torch.rand(400, 3, generator=torch.Generator().manual_seed(5))
    res = []
    for pair in (True, False):
        m3, op, sc, rot, c1, c2 = (t(s["means3D"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"]),
                                   t(s["colors_precomp"]), t(cols_b))
        m2 = torch.zeros_like(m3, requires_grad=True)
        if pair:
            _, _, _, img2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=c1, colors_b=c2, scales=sc, rotations=rot)
        else:
            img2, _, _ = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=c2, scales=sc, rotations=rot)
        img2.square().sum().backward()
        res.append([m3.grad, op.grad, sc.grad, rot.grad, c2.grad, m2.grad, c1.grad])
    for a, b in zip(res[0][:6], res[1][:6]):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    assert res[0][6] is None or float(res[0][6].abs().max()) == 0.0


@pytest.mark.parametrize("scene", ["elongated", "round", "faint"])
def test_exact_tile_culling_changes_nothing_but_the_lists(gpu_device, scene):
    """Exact (tile, Gaussian) culling at binning time (include/s3g_raster.h): images, depth, radii and every gradient are
    bit-identical to the reference's bounding-square binning; the instance list is a subset of it."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import _debug, raster_C
    W, H, P = 176, 120, 3000
    s = tiny_scene(P=P, W=W, H=H, seed=21, scale=0.08)
    g = torch.Generator().manual_seed(4)
    if scene == "elongated":      # needles: most tiles of the bounding square are never touched
        s["scales"] = s["scales"] * torch.tensor([6.0, 0.15, 0.15])
    elif scene == "faint":        # opacities around and below 1/255: some Gaussians can never pass the alpha test
        s["opacities"] = 0.012 * torch.rand(P, 1, generator=g)
    dev = gpu_device
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    gc, gd = grad_pair(H, W)
    res = {}
    for cull in (True, False):
        prev = raster_C.set_exact_cull(cull)
        prev_async = raster_C.set_async(False)   # the test reads the instance COUNT next to the private lists: synchronous forward
        try:
            t = lambda k: s[k].to(dev).clone().requires_grad_(True)
            m3, op, sc, rot, col = t("means3D"), t("opacities"), t("scales"), t("rotations"), t("colors_precomp")
            m2 = torch.zeros_like(m3, requires_grad=True)
            color, radii, depth = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)
            R, geom, binning, img = raster_C._geom_cache[2][0], *raster_C._geom_cache[2][2:]
            ((color * gc.to(dev)).sum() + (depth * gd.to(dev)).sum()).backward()
            im = _debug.decode_image(img, W, H)
            b = _debug.decode_binning(binning, R)
            lists = set((int(tile), int(gid)) for tile, (lo, hi) in enumerate(im["ranges"].cpu().tolist())
                        for gid in b["point_list"][lo:hi].cpu().tolist())
            res[cull] = (R, [color, depth, radii], [x.grad for x in (m3, m2, op, sc, rot, col)], lists)
        finally:
            raster_C.set_exact_cull(prev)
            raster_C.set_async(prev_async)
    (R1, o1, g1, l1), (R0, o0, g0, l0) = res[True], res[False]
    for a, b in zip(o1 + g1, o0 + g0):
        assert torch.equal(a, b)
    assert l1 <= l0 and R1 == len(l1) and R0 == len(l0)
    if scene != "round":
        assert R1 < 0.8 * R0, (R1, R0)     # the cull actually removes instances


@pytest.mark.parametrize("band", [1, 7, 29])
def test_banded_binning_is_identical_to_the_single_pass(gpu_device, band):
    """Tile grids beyond the LDS histogram are binned in bands of consecutive tiles (include/s3g_raster.h); forcing a small
    band on a small image must not change one bit of the lists, images or gradients."""
    import ctypes
    from diff_gaussian_rasterization import _C
    from s3gaussian_amd import _debug, _lib
    L = _lib.lib()
    L.s3g_raster_set_bin_band.restype = ctypes.c_int
    W, H, P = 96, 80, 2500
    s = tiny_scene(P=P, W=W, H=H, seed=31, scale=0.1)
    s["scales"][:40] *= 12.0                      # some rects larger than 32 tiles: the wave-cooperative walk
    dev = gpu_device
    cam = s["cam"]
    e = torch.Tensor([])
    gc, gd = grad_pair(H, W)

    def run():
        d = lambda k: s[k].to(dev)
        args = (s["bg"].to(dev), d("means3D"), d("colors_precomp"), d("opacities"), d("scales"), d("rotations"), 1.0, e,
                cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], H, W, e, 0, cam["campos"].to(dev),
                False, False)
        R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(*args)
        grads = _C.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], 1.0, e, args[8], args[9],
                                                cam["tanfovx"], cam["tanfovy"], gc.to(dev), gd.to(dev), e, 0, args[16], geom, R, binning,
                                                img, False)
        im, b = _debug.decode_image(img, W, H), _debug.decode_binning(binning, R)
        return [torch.tensor(R), color, depth, radii, im["ranges"].clone(), b["point_list"].clone(), im["n_contrib"].clone()] + list(grads)

    prev = L.s3g_raster_set_bin_band(band)
    try:
        banded = run()
    finally:
        L.s3g_raster_set_bin_band(prev)
    single = run()
    for a, b in zip(banded, single):
        assert torch.equal(a, b)


def test_image_beyond_the_lds_histogram_matches_the_oracle(gpu_device, oracle, reference_binning):
    """4096 x 2400 = 38 400 tiles > the 38 000 the binning histogram holds in LDS: two bands."""
    from diff_gaussian_rasterization import _C
    from s3gaussian_amd import _debug
    W, H, P = 4096, 2400, 3000
    s = tiny_scene(P=P, W=W, H=H, seed=41, scale=0.05)
    dev = gpu_device
    cam = s["cam"]
    e = torch.Tensor([])
    R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
        s["bg"].to(dev), s["means3D"].to(dev), s["colors_precomp"].to(dev), s["opacities"].to(dev), s["scales"].to(dev),
        s["rotations"].to(dev), 1.0, e, cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"],
        H, W, e, 0, cam["campos"].to(dev), False, False)
    ref = oracle_forward(oracle, s)
    assert R == ref["num_rendered"]
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["radii"])
    pl = _debug.decode_binning(binning, R)["point_list"].cpu().numpy().astype(np.uint32)
    np.testing.assert_array_equal(pl, ref["state"]["point_list"])
    bad = np.abs(color.cpu().numpy() - ref["color"]).max(0) > COLOR_TOL
    assert bad.mean() <= OUTLIER_FRAC
