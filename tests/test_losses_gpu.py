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
