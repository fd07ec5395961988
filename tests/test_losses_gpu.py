"""Fused SSIM (HIP) vs the reference's utils/loss_utils.py::ssim (golden) and vs the plain-PyTorch restatement with
autograd.  Tolerance: |value| 2e-6 (separable window vs the reference's materialised 11x11 product), grad rel-L2 1e-4."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ssim_matches_reference_golden(gpu_device):
    from s3gaussian_amd.losses import ssim
    z = np.load(os.path.join(GOLD, "losses.npz"))
    a, b = torch.from_numpy(z["a"]).to(gpu_device), torch.from_numpy(z["b"]).to(gpu_device)
    assert abs(ssim(a, b).item() - float(z["ssim"])) < 2e-6


@pytest.mark.parametrize("shape", [(3, 40, 56), (3, 17, 33), (1, 16, 16), (3, 1066, 1600)])
def test_ssim_value_and_gradient_vs_restatement(gpu_device, shape):
    from oracle import hexplane_ref as hr
    from s3gaussian_amd.losses import ssim
    g = torch.Generator().manual_seed(shape[1])
    a = torch.rand(shape, generator=g)
    b = (a + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    ar = a.clone().requires_grad_(True)
    vr = hr.ssim(ar[None], b[None])
    (3.0 * vr).backward()
    ag = a.to(gpu_device).requires_grad_(True)
    vg = ssim(ag[None], b.to(gpu_device)[None])
    (3.0 * vg).backward()
    assert abs(vg.item() - vr.item()) < 2e-6
    assert rel_l2(ag.grad.cpu().numpy(), ar.grad.numpy()) < 1e-4


def test_plane_regulation_value_and_gradient(gpu_device):
    """Fused regulariser vs the reference formulas (golden value from scene/regulation.py + gaussian_model.py:710-749
    run in-container, and autograd of the restatement for the gradient)."""
    from oracle import hexplane_ref as hr
    from s3gaussian_amd.hexplane import HexPlaneField
    from s3gaussian_amd.losses import plane_regulation
    z = np.load(os.path.join(GOLD, "hexplane_deform.npz"))
    zl = np.load(os.path.join(GOLD, "losses.npz"))
    cfg = dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32, resolution=[8, 8, 8, 5])
    ref = hr.HexPlaneField(1.6, cfg, [1, 2])
    sd = {k[len("sd::deformation_net.grid."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::deformation_net.grid.")}
    ref.load_state_dict(sd)
    mine = HexPlaneField(1.6, cfg, [1, 2])
    mine.load_state_dict(sd)
    mine = mine.to(gpu_device)
    v = plane_regulation(mine.grids, 0.01, 0.0001, 0.0001)
    assert abs(v.item() - float(zl["regulation"])) < 1e-6 * max(1.0, abs(float(zl["regulation"])))
    (2.5 * v).backward()
    vr = hr.plane_regulation(ref.grids, 0.01, 0.0001, 0.0001)
    (2.5 * vr).backward()
    for (k, pr), (_, pg) in zip(ref.named_parameters(), mine.named_parameters()):
        if pr.grad is not None:
            assert rel_l2(pg.grad.cpu().numpy(), pr.grad.numpy()) < 1e-5, k


def test_plane_regulation_default_resolution(gpu_device):
    from oracle import hexplane_ref as hr
    from s3gaussian_amd.hexplane import HexPlaneField
    from s3gaussian_amd.losses import plane_regulation
    torch.manual_seed(0)
    cfg = dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32, resolution=[64, 64, 64, 25])
    ref = hr.HexPlaneField(1.6, cfg, [1, 2])
    with torch.no_grad():
        for p in ref.grids.parameters():
            p.add_(0.1 * torch.randn_like(p))
    mine = HexPlaneField(1.6, cfg, [1, 2])
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(gpu_device)
    v = plane_regulation(mine.grids, 0.01, 0.0001, 0.0001)
    vr = hr.plane_regulation(ref.grids, 0.01, 0.0001, 0.0001)
    assert abs(v.item() - vr.item()) < 2e-6 * max(1.0, abs(vr.item()))
    v.backward(); vr.backward()
    for (k, pr), (_, pg) in zip(ref.named_parameters(), mine.named_parameters()):
        if pr.grad is not None:
            assert rel_l2(pg.grad.cpu().numpy(), pr.grad.numpy()) < 1e-5, k


def _photometric_ref(hr, img, gt, dep, gdep, ft, gft, w_ssim, w_depth, w_feat):
    """train.py:395-425 per-pixel terms with the restated loss_utils functions (CPU, autograd)."""
    loss = hr.l1_loss(img[None], gt[None])
    if dep is not None and w_depth != 0:
        loss = loss + w_depth * hr.depth_l2(dep[None], gdep[None])
    if w_ssim != 0:
        loss = loss + w_ssim * (1.0 - hr.ssim(img[None], gt[None]))
    if ft is not None and w_feat != 0:
        loss = loss + w_feat * hr.l2_loss(ft, gft)
    return loss


@pytest.mark.parametrize("case", ["all", "no_depth", "no_feat", "l1_only", "full_size"])
def test_photometric_loss_value_and_gradients(gpu_device, case):
    """Fused L1 + depth-L2 + DSSIM + feature-L2 vs the step-by-step restatement.  Value 5e-6 abs (double accumulation on
    the GPU vs fp32 means on the CPU), gradients rel-L2 1e-4 (SSIM part) with exact agreement of the mask logic."""
    from oracle import hexplane_ref as hr
    from s3gaussian_amd.losses import photometric_loss
    H, W = (1066, 1600) if case == "full_size" else (37, 53)
    g = torch.Generator().manual_seed(len(case))
    img = torch.rand(3, H, W, generator=g)
    gt = (img + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    # depths straddle every branch: below 0.01, inside, beyond max_depth (mask), predictions below 0 and above max (clamp)
    gdep = 100.0 * torch.rand(1, H, W, generator=g)
    gdep[0, ::7, ::5] = 0.0
    dep = gdep + 30.0 * torch.randn(1, H, W, generator=g)
    ft, gft = torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g)
    w_ssim, w_depth, w_feat = 0.2, 0.5, 0.001
    if case == "no_depth":
        w_depth = 0.0
    if case == "no_feat":
        ft = gft = None
    if case == "l1_only":
        w_ssim = w_depth = w_feat = 0.0
    leaf = lambda t, dev: None if t is None else t.clone().to(dev).requires_grad_(True)
    ri, rd, rf = leaf(img, "cpu"), leaf(dep, "cpu"), leaf(ft, "cpu")
    vr = _photometric_ref(hr, ri, gt, rd, gdep, rf, gft, w_ssim, w_depth, w_feat)
    (2.5 * vr).backward()
    dev = gpu_device
    gi, gd, gf = leaf(img, dev), leaf(dep, dev), leaf(ft, dev)
    to = lambda t: None if t is None else t.to(dev)
    vg = photometric_loss(gi, to(gt), gd, to(gdep), gf, to(gft), lambda_dssim=w_ssim, lambda_depth=w_depth, lambda_feat=w_feat)
    (2.5 * vg).backward()
    assert abs(vg.item() - vr.item()) < 5e-6 * max(1.0, abs(vr.item()))
    assert rel_l2(gi.grad.cpu().numpy(), ri.grad.numpy()) < 1e-4
    if w_depth != 0:
        assert rel_l2(gd.grad.cpu().numpy(), rd.grad.numpy()) < 1e-5
        assert torch.equal(gd.grad.cpu() == 0, rd.grad == 0)          # mask and clamp pass/stop exactly the same pixels
    else:
        assert gd.grad is None
    if ft is not None and w_feat != 0:
        assert rel_l2(gf.grad.cpu().numpy(), rf.grad.numpy()) < 1e-5
    elif gf is not None:
        assert gf.grad is None


def test_photometric_loss_empty_depth_mask_is_nan_like_reference(gpu_device):
    from s3gaussian_amd.losses import photometric_loss
    dev = gpu_device
    img, gt = torch.rand(3, 8, 8, device=dev), torch.rand(3, 8, 8, device=dev)
    dep, gdep = torch.rand(1, 8, 8, device=dev), torch.zeros(1, 8, 8, device=dev)   # no gt depth in (0.01, 80)
    v = photometric_loss(img, gt, dep, gdep, lambda_depth=0.5)
    assert torch.isnan(v).item()      # F.mse_loss over an empty selection is NaN (utils/loss_utils.py:32-45)


def test_training_loss_fused_and_stepwise_agree(gpu_device):
    """pipeline.training_loss with the fused per-pixel pass vs its step-by-step branch on a small scene."""
    from types import SimpleNamespace
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, render, training_loss
    dev = gpu_device
    scn = synth.street_scene(P=4000, seed=2, width=128, height=96, n_frames=2)
    torch.manual_seed(0)
    hyper, opt = default_hyper(), default_opt()
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][1].items()}
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    bg = scn["bg"].to(dev)
    g = torch.Generator().manual_seed(3)
    gt_img = torch.rand(3, 96, 128, generator=g).to(dev)
    gt_dep = (60.0 * torch.rand(1, 96, 128, generator=g)).to(dev)
    gt_ft = torch.rand(3, 96, 128, generator=g).to(dev)
    res = []
    for fused in (True, False):
        for p in pc.parameters():
            p.grad = None
        pkg = render(cam, pc, pipe, bg, stage="fine", return_dx=True, render_feat=True)
        loss = training_loss(pc, pkg, gt_img, gt_dep, gt_ft, hyper, opt, stage="fine", fused_pixel_terms=fused)
        loss.backward()
        res.append((loss.item(), [p.grad.clone() for p in pc.parameters() if p.grad is not None]))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * max(1.0, abs(res[1][0]))
    assert len(res[0][1]) == len(res[1][1])
    for a, b in zip(res[0][1], res[1][1]):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
