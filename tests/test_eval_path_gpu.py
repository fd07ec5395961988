"""SURVEY 8f rows 2 and 4 on the GPU: the shared-geometry decomposition renders (gaussian_renderer/__init__.py:168-204), the
densification bookkeeping fused into the rasterizer backward (train.py:489-493, scene/gaussian_model.py:693-695), and
the geometry cache staying coherent across optimizer steps."""
import numpy as np
import pytest
import torch

from tests.util import settings_from, tiny_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["precomp", "sh"])
@pytest.mark.parametrize("P,W,H,frac", [(3000, 131, 77, 0.5), (800, 64, 48, 0.1), (50, 32, 32, 0.0), (50, 32, 32, 1.0)])
def test_decomposition_renders_equal_three_separate_rasterizations(gpu_device, mode, P, W, H, frac):
    """One preprocess/binning/sort + one extra blend pass == the reference's way (the rasterizer called on the
    boolean-masked inputs), bit for bit: a subset's tile lists are the full lists minus the other class, same order."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = gpu_device
    s = tiny_scene(P=P, W=W, H=H, seed=17, scale=0.12)
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev, sh_degree=3 if mode == "sh" else 0))
    d = lambda k: s[k].to(dev)
    g = torch.Generator().manual_seed(2)
    mask = (torch.rand(P, generator=g) < frac).to(dev)
    col = dict(shs=d("shs")) if mode == "sh" else dict(colors_precomp=d("colors_precomp"))
    with torch.no_grad():
        out = rast.forward_decomposed(means3D=d("means3D"), opacities=d("opacities"), dynamic_mask=mask, scales=d("scales"),
                                      rotations=d("rotations"), **col)
        full = rast(means3D=d("means3D"), means2D=torch.zeros(P, 3, device=dev), opacities=d("opacities"),
                    scales=d("scales"), rotations=d("rotations"), **col)
        assert torch.equal(out["render"], full[0]) and torch.equal(out["radii"], full[1]) and torch.equal(out["depth"], full[2])
        for tag, m in (("d", mask), ("s", ~mask)):
            sub = {k: v[m] for k, v in col.items()}
            img, rad, dep = rast(means3D=d("means3D")[m], means2D=torch.zeros(int(m.sum()), 3, device=dev),
                                 opacities=d("opacities")[m], scales=d("scales")[m], rotations=d("rotations")[m], **sub)
            assert torch.equal(out[f"render_{tag}"], img), tag
            assert torch.equal(out[f"depth_{tag}"], dep), tag
            assert torch.equal(out["radii"][m], rad), tag


@pytest.mark.parametrize("scene,frac", [("tiny", 0.5), ("tiny", 0.0), ("tiny", 1.0), ("street", 0.3)])
def test_decomposition_renders_match_the_reference_build(gpu_device, scene, frac):
    """render_d / depth_d / render_s / depth_s of ONE forward_decomposed call vs the REFERENCE rasterizer itself
    (oracle/_ref = its forward.cu / rasterizer_impl.cu built for gfx950) run on the boolean-masked subsets, which is literally
    what gaussian_renderer/__init__.py:168-204 does.  An empty class is the reference's P == 0 early-out: an all-zero image
    WITHOUT background (rasterize_points.cu:81-116).  Bars as in test_raster_ref_gpu.py: colour <= 1e-4 abs, depth <= 1e-4 rel
    outside at most 0.05 % of pixels (borderline alpha / transmittance skip tests between exp implementations)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle import ref_raster
    from tests.util import cam_kwargs, to_np
    if not ref_raster.available():
        pytest.fail("oracle/_ref/libref_raster.so missing: run oracle/build_ref.sh (needs /root/reference) before gpurun")
    ref = ref_raster.RefRaster()
    dev = gpu_device
    if scene == "tiny":
        s = tiny_scene(P=4000, W=131, H=77, seed=29, scale=0.12)
    else:
        from s3gaussian_amd import synth
        sc = synth.street_scene(P=200_000, seed=2, width=960, height=640, n_frames=4)
        gs = sc["gaussians"]
        s = dict(means3D=gs["xyz"], scales=torch.exp(gs["log_scales"]), rotations=torch.nn.functional.normalize(gs["rotations_raw"]),
                 opacities=torch.sigmoid(gs["opacity_logit"]), colors_precomp=torch.rand(200_000, 3, generator=torch.Generator().manual_seed(1)),
                 cam=sc["cameras"][4], bg=torch.tensor([0.1, 0.3, 0.2]))
    P = s["means3D"].shape[0]
    H, W = s["cam"]["image_height"], s["cam"]["image_width"]
    mask = torch.rand(P, generator=torch.Generator().manual_seed(5)) < frac
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    d = lambda k: s[k].to(dev)
    out = rast.forward_decomposed(means3D=d("means3D"), opacities=d("opacities"), dynamic_mask=mask.to(dev), scales=d("scales"),
                                  rotations=d("rotations"), colors_precomp=d("colors_precomp"))
    kw = to_np(cam_kwargs(s))
    for tag, m in (("d", mask), ("s", ~mask)):
        img, dep = out[f"render_{tag}"].cpu().numpy(), out[f"depth_{tag}"].cpu().numpy()
        if int(m.sum()) == 0:
            assert not img.any() and not dep.any(), tag          # zeros, no background (reference P == 0 early-out)
            continue
        r = ref.forward(means3D=s["means3D"][m].numpy(), opacities=s["opacities"][m].numpy(), scales=s["scales"][m].numpy(),
                        rotations=s["rotations"][m].numpy(), colors_precomp=s["colors_precomp"][m].numpy(), sh_degree=0, **kw)
        assert (out["radii"][m.to(dev)].cpu().numpy() == r["radii"]).all(), tag
        bad_c = np.abs(img - r["color"]).max(0) > 1e-4
        bad_d = np.abs(dep - r["depth"])[0] > 1e-4 * np.maximum(np.abs(r["depth"][0]), 1.0)
        assert (bad_c | bad_d).mean() <= 5e-4, (tag, bad_c.mean(), bad_d.mean())
        assert np.abs(img - r["color"]).max(0)[~bad_c].max() <= 1e-4


def test_render_with_decomposition_matches_the_unfused_path(gpu_device):
    """pipeline.render(return_decomposition=True) under no_grad: fused decomposition vs pipe.fused_decomposition=False."""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, render
    dev = gpu_device
    scn = synth.street_scene(P=30_000, seed=1, width=320, height=208, n_frames=2)
    hyper = default_hyper()
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    with torch.no_grad():   # make dx non-trivial so both classes are populated
        for p in pc._deformation.deformation_net.pos_deform.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][1].items()}
    bg = scn["bg"].to(dev)
    res = {}
    with torch.no_grad():
        for fused in (True, False):
            pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False, fused_decomposition=fused)
            res[fused] = render(cam, pc, pipe, bg, stage="fine", return_decomposition=True)
    for k in ("render", "depth", "radii", "render_d", "depth_d", "render_s", "depth_s", "visibility_filter_d", "visibility_filter_s"):
        assert torch.equal(res[True][k], res[False][k]), k
    assert 0 < int(res[True]["visibility_filter_d"].numel()) < 30_000


def test_densify_stats_kernel_and_fused_backward_match_the_reference_formulas(gpu_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd.optim import densify_stats
    dev = gpu_device
    P, W, H = 2500, 96, 64
    s = tiny_scene(P=P, W=W, H=H, seed=23, scale=0.1, spread=3.5, zmin=-1.0)   # some Gaussians culled
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    t = lambda k: s[k].to(dev).clone().requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    colb = torch.rand(P, 3, generator=g).to(dev)
    gc, gd, gc2 = (torch.randn(c, H, W, generator=g).to(dev) for c in (3, 1, 3))
    prev = [torch.rand(P, 1, generator=g).to(dev), torch.randint(0, 4, (P, 1), generator=g).float().to(dev),
            (torch.rand(P, generator=g) * 6).to(dev)]

    def run(acc):
        m3, op, sc, rot, col = t("means3D"), t("opacities"), t("scales"), t("rotations"), t("colors_precomp")
        m2 = torch.zeros_like(m3, requires_grad=True)
        a, radii, depth, b = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=colb, scales=sc,
                                               rotations=rot, densify_accum=acc)
        ((a * gc).sum() + (depth * gd).sum() + (b * gc2).sum()).backward()
        return m2.grad, radii

    vg, radii = run(None)
    vis = radii > 0
    assert 0 < int(vis.sum()) < P
    # the reference's three updates (train.py:491, gaussian_model.py:693-695)
    want = [x.clone() for x in prev]
    want[2][vis] = torch.max(want[2][vis], radii[vis].float())
    want[0][vis] += torch.norm(vg[vis, :2], dim=-1, keepdim=True)
    want[1][vis] += 1
    # (a) stand-alone pass over the viewspace gradient
    got = [x.clone() for x in prev]
    densify_stats(got[0], got[1], got[2], vg, radii)
    # (b) fused into the two-image backward
    fused = [x.clone() for x in prev]
    vg2, _ = run(tuple(fused))
    assert torch.equal(vg, vg2)
    for w, a, b in zip(want, got, fused):
        assert torch.equal(a, b)                              # same arithmetic in both kernels
        np.testing.assert_allclose(a.cpu().numpy(), w.cpu().numpy(), rtol=2e-7, atol=0)
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    # explicit visibility mask (data-parallel: visible in ANY rank's view)
    got2 = [x.clone() for x in prev]
    anyvis = vis | (torch.arange(P, device=dev) % 7 == 0)
    densify_stats(got2[0], got2[1], got2[2], vg, radii, anyvis)
    assert torch.equal(got2[1], prev[1] + anyvis[:, None].float())


def test_training_step_densify_stats_fused_equals_separate_pass(gpu_device):
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, training_step
    dev = gpu_device
    scn = synth.street_scene(P=20_000, seed=0, width=320, height=208, n_frames=2)
    hyper, opt = default_hyper(), default_opt()
    H, W = 208, 320
    g = torch.Generator().manual_seed(0)
    gt = [torch.rand(3, H, W, generator=g).to(dev), (torch.rand(1, H, W, generator=g) * 50).to(dev), torch.rand(3, H, W, generator=g).to(dev)]
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][0].items()}
    res = {}
    for fused_pair in (True, False):
        torch.manual_seed(0)
        pc = GaussianParams(3, hyper)
        gs = scn["gaussians"]
        pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
        pc._deformation.deformation_net.set_aabb(*scn["aabb"])
        pc.training_setup(opt)
        pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False, fused_pair=fused_pair)
        for _ in range(2):
            loss, pkg = training_step(pc, cam, *gt, hyper, opt, scn["bg"].to(dev), pipe=pipe, densify_stats=True)
        assert pkg["densify_stats_fused"] == fused_pair
        res[fused_pair] = (pc.xyz_gradient_accum.clone(), pc.denom.clone(), pc.max_radii2D.clone())
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    assert float(res[True][1].max()) == 2.0
    np.testing.assert_allclose(res[True][0].cpu().numpy(), res[False][0].cpu().numpy(), rtol=1e-4, atol=1e-9)


def test_adam_step_between_two_renders_of_the_same_tensors_misses_the_geometry_cache(gpu_device):
    """optim.Adam writes parameters through raw pointers; it must bump their version counters (and drop the cached arenas)
    so that a render of the SAME tensors afterwards is not served the previous step's preprocess / binning."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.optim import Adam
    dev = gpu_device
    P = 600
    s = tiny_scene(P=P, W=80, H=64, seed=5)
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    prm = {k: torch.nn.Parameter(s[k].to(dev).clone()) for k in ("means3D", "opacities", "scales", "rotations")}
    col = s["colors_precomp"].to(dev)
    opt = Adam([{"params": list(prm.values()), "lr": 0.05}], lr=0.0, eps=1e-15)

    def draw():
        return rast(means3D=prm["means3D"], means2D=torch.zeros(P, 3, device=dev), opacities=prm["opacities"], colors_precomp=col,
                    scales=prm["scales"], rotations=prm["rotations"])

    with torch.no_grad():
        a = draw()[0].clone()
        h0 = raster_C._geom_cache_hits
        b = draw()[0].clone()
        assert raster_C._geom_cache_hits == h0 + 1 and torch.equal(a, b)       # unchanged tensors: served from the cache
    v0 = prm["means3D"]._version
    for p in prm.values():
        p.grad = torch.ones_like(p)
    opt.step()
    assert prm["means3D"]._version > v0
    with torch.no_grad():
        h1 = raster_C._geom_cache_hits
        c = draw()[0].clone()
        assert raster_C._geom_cache_hits == h1                                  # parameters moved: no hit
        raster_C._GEOM_CACHE_ON = False
        try:
            d = draw()[0].clone()
        finally:
            raster_C._GEOM_CACHE_ON = True
    assert torch.equal(c, d) and not torch.equal(a, c)
