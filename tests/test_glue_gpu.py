"""Fused render glue + full pipeline.render() vs the plain-PyTorch restatement of the reference's glue (pinned to the
reference by tests/golden/glue.npz): outputs rtol 1e-5, gradients rel-L2 1e-5."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_glue_matches_reference_golden(gpu_device):
    from s3gaussian_amd.glue import activations_and_colors
    z = np.load(os.path.join(GOLD, "hexplane_deform.npz"))
    gl = np.load(os.path.join(GOLD, "glue.npz"))
    dev = gpu_device
    shs = torch.from_numpy(z["shs"]).to(dev)
    P = shs.shape[0]
    cols, *_ = activations_and_colors(3, shs[:, :1].contiguous(), shs[:, 1:].contiguous(), None, torch.from_numpy(z["xyz"]).to(dev),
                                      torch.from_numpy(gl["campos"]).to(dev), torch.zeros(P, 3, device=dev),
                                      torch.ones(P, 4, device=dev), torch.zeros(P, 1, device=dev))
    np.testing.assert_allclose(cols.cpu().numpy(), gl["colors"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("live", [1.0, 0.3, 0.15])
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("with_dshs", [True, False])
def test_glue_forward_backward_vs_restatement(gpu_device, deg, with_dshs, live):
    """live: fraction of the Gaussians whose colour receives a gradient (a view sees ~18 % of them).  The backward stages the
    coefficient rows of up to 64 live Gaussians per workgroup through LDS (csrc/glue.hip, round 6): 0.15 = every live row staged,
    0.3 = staged rows and per-lane rows in one workgroup, 1.0 = mostly per-lane rows."""
    from oracle import hexplane_ref as hr
    from s3gaussian_amd.glue import activations_and_colors
    g = torch.Generator().manual_seed(10 * deg + int(with_dshs))
    P = 1000
    mk = lambda *s: torch.randn(*s, generator=g)
    f_dc, f_rest, dshs, xyz = mk(P, 1, 3), 0.3 * mk(P, 15, 3), 0.1 * mk(P, 16, 3), 3 * mk(P, 3)
    ls, rr, ol, campos = 0.5 * mk(P, 3), mk(P, 4), mk(P, 1), torch.tensor([0.3, -0.2, 1.1])
    ws = [mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 1)]
    if live < 1.0:
        ws[0] = ws[0] * (torch.rand(P, 1, generator=g) < live).float()
    leaf = lambda t, dev: t.clone().to(dev).requires_grad_(True)

    def run(dev, fused):
        L = [leaf(t, dev) for t in (f_dc, f_rest, dshs, xyz, ls, rr, ol)]
        a, b, d, x, s_, r_, o_ = L
        # with dshs present the lambda_dshs * mean|dshs| regulariser (train.py:407-410) rides along in the fused pass
        if fused:
            outs = activations_and_colors(deg, a, b, d if with_dshs else None, x, campos.to(dev), s_, r_, o_,
                                          with_dshs_l1=with_dshs)
            l1 = outs[4] if with_dshs else 0.0
            outs = outs[:4]
        else:
            shs = torch.cat((a, b), dim=1) + (d if with_dshs else 0)
            outs = (hr.shs_to_colors(deg, shs, x, campos), torch.exp(s_), torch.nn.functional.normalize(r_), torch.sigmoid(o_))
            l1 = torch.mean(torch.abs(d)) if with_dshs else 0.0
        (sum((o * w.to(dev)).sum() for o, w in zip(outs, ws)) + 700.0 * l1).backward()
        return outs + ((l1,) if with_dshs else ()), L

    outs_r, Lr = run("cpu", False)
    outs_g, Lg = run(gpu_device, True)
    for a, b in zip(outs_g, outs_r):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=2e-6)
    for i, (a, b) in enumerate(zip(Lg, Lr)):
        if i == 2 and not with_dshs:
            assert a.grad is None
            continue
        if b.grad is None:  # e.g. degree 0 does not depend on the view direction
            assert float(a.grad.abs().max()) == 0.0
            continue
        assert rel_l2(a.grad.cpu().numpy(), b.grad.numpy()) < 1e-5, i


def test_render_fused_and_unfused_paths_agree(gpu_device):
    """pipeline.render(): fused glue + deform_heads path vs the step-by-step path (convert_SHs_python glue in torch)."""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, render
    dev = gpu_device
    scn = synth.street_scene(P=5000, seed=1, width=160, height=112, n_frames=2)
    torch.manual_seed(0)
    pc = GaussianParams(3, default_hyper())
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][1].items()}
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    bg = scn["bg"].to(dev)

    def run(force_unfused):
        for p in pc.parameters():
            p.grad = None
        if force_unfused:
            real = pc._deformation.deformation_net._fused_ok
            pc._deformation.deformation_net._fused_ok = lambda: False
        try:
            pkg = render(cam, pc, pipe, bg, stage="fine", return_dx=True, render_feat=True)
        finally:
            if force_unfused:
                pc._deformation.deformation_net._fused_ok = real
        (pkg["render"].sum() + 0.1 * pkg["depth"].sum() + pkg["feat"].sum() + pkg["dx"].abs().sum() + pkg["dshs"].abs().sum()).backward()
        return pkg, {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None}

    pk1, g1 = run(False)
    pk2, g2 = run(True)
    assert torch.equal(pk1["radii"], pk2["radii"])
    for k in ("render", "depth", "feat", "dx", "dshs"):
        np.testing.assert_allclose(pk1[k].detach().cpu().numpy(), pk2[k].detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert set(g1) == set(g2)
    for k in g1:
        assert rel_l2(g1[k].cpu().numpy(), g2[k].cpu().numpy()) < 2e-4, k


def test_coarse_stage_render_and_step(gpu_device):
    """Coarse stage (train.py's first phase: deformation network bypassed, gaussian_renderer/__init__.py:82-84): the fused
    glue path equals the step-by-step path (activations + eval_sh evaluated by torch), the deformation parameters get no
    gradient, and a full training_step runs (single-image raster path: there is no feature render in this stage)."""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, render, training_step
    dev = gpu_device
    scn = synth.street_scene(P=4000, seed=7, width=144, height=96, n_frames=2)
    torch.manual_seed(0)
    hyper, opt = default_hyper(), default_opt()
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][0].items()}
    bg = scn["bg"].to(dev)
    res = []
    for fused in (True, False):
        for p in pc.parameters():
            p.grad = None
        pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False, fused_glue=fused)
        pkg = render(cam, pc, pipe, bg, stage="coarse", return_dx=True, render_feat=True)
        (pkg["render"].sum() + 0.1 * pkg["depth"].sum()).backward()
        res.append((pkg, {n: p.grad.clone() for n, p in pc.named_parameters() if p.grad is not None}))
    (p1, g1), (p2, g2) = res
    assert torch.equal(p1["radii"], p2["radii"]) and "feat" not in p1 and "dx" not in p1
    for k in ("render", "depth"):
        np.testing.assert_allclose(p1[k].detach().cpu().numpy(), p2[k].detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert set(g1) == set(g2) and g1 and not any(k.startswith("_deformation") for k in g1)
    for k in g1:
        assert rel_l2(g1[k].cpu().numpy(), g2[k].cpu().numpy()) < 2e-4, k
    pc.training_setup(opt)
    before = pc._features_dc.detach().clone()
    loss, pkg = training_step(pc, cam, torch.rand(3, 96, 144, device=dev), torch.rand(1, 96, 144, device=dev) * 50, None,
                              hyper, opt, bg, stage="coarse")
    assert torch.isfinite(loss) and pkg["render"].shape == (3, 96, 144)
    assert not torch.equal(before, pc._features_dc.detach())          # the optimizer stepped


def test_render_decomposition_outputs(gpu_device):
    """return_decomposition=True (gaussian_renderer/__init__.py:168-205): dynamic / static subsets rendered separately."""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, render
    dev = gpu_device
    scn = synth.street_scene(P=3000, seed=9, width=128, height=96, n_frames=2)
    torch.manual_seed(0)
    pc = GaussianParams(3, default_hyper())
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][1].items()}
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        pkg = render(cam, pc, pipe, scn["bg"].to(dev), stage="fine", return_decomposition=True, return_dx=True)
    for tag in ("d", "s"):
        assert pkg[f"render_{tag}"].shape == (3, 96, 128) and pkg[f"depth_{tag}"].shape == (1, 96, 128)
        assert torch.isfinite(pkg[f"render_{tag}"]).all()
    n_d, n_s = pkg["visibility_filter_d"].numel(), pkg["visibility_filter_s"].numel()
    assert n_d + n_s == 3000 and 0 < n_d < 3000
    assert int(pkg["visibility_filter_d"].sum()) + int(pkg["visibility_filter_s"].sum()) == int(pkg["visibility_filter"].sum())
