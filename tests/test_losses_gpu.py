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
