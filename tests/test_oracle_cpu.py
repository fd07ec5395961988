"""CPU-only tests: pin the oracle (it is the checker for every GPU parity test).

  * fp64 oracle forward == float64 autograd restatement; oracle hand-derived backward == autograd (all 8 grads)
  * fp32 oracle ~ fp64 oracle
  * in-kernel SH path == the reference's utils/sh_utils.py::eval_sh (golden, generated in-container)
  * hexplane/deformation/loss restatements == the reference's own modules (golden, generated in-container)
  * simple-knn restatement == brute force
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref
from oracle.oracle import RasterOracle, knn_mean_dist2
from tests.util import cam_kwargs, oracle_forward, rel_l2, tiny_scene, to_np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("mode", ["precomp", "sh"])
@pytest.mark.parametrize("seed,scale", [(0, 0.15), (1, 0.3)])
def test_fp64_oracle_matches_autograd(mode, seed, scale):
    s = tiny_scene(P=250, W=48, H=40, seed=seed, scale=scale)
    o64 = RasterOracle(np.float64)
    T = lambda t: t.double().clone().requires_grad_(True)
    m3, sc, q, op = T(s["means3D"]), T(s["scales"]), T(s["rotations"]), T(s["opacities"])
    m2 = torch.zeros(250, 3, dtype=torch.float64, requires_grad=True)
    kw = {k: (v.double() if torch.is_tensor(v) else v) for k, v in cam_kwargs(s).items()}
    if mode == "precomp":
        col = T(s["colors_precomp"]); extra = dict(colors_precomp=col, sh_degree=0)
    else:
        sh = T(s["shs"]); extra = dict(shs=sh, sh_degree=3)
    c, r, d = torch_ref.rasterize(means3D=m3, opacities=op, scales=sc, rotations=q, means2D=m2, **kw, **extra)
    f = o64.forward(means3D=m3.detach().numpy(), opacities=op.detach().numpy(), scales=sc.detach().numpy(),
                    rotations=q.detach().numpy(), **to_np(kw), **to_np(extra))
    np.testing.assert_array_equal(f["radii"], r.numpy())
    assert np.abs(f["color"] - c.detach().numpy()).max() < 1e-12
    assert np.abs(f["depth"] - d.detach().numpy()).max() < 1e-11
    g = torch.Generator().manual_seed(5)
    gc = torch.randn(c.shape, generator=g, dtype=torch.float64)
    gd = torch.randn(d.shape, generator=g, dtype=torch.float64)
    ((c * gc).sum() + (d * gd).sum()).backward()
    b = o64.backward(f, gc.numpy(), gd.numpy())
    assert rel_l2(b["dL_dmeans2D"][:, :2], m2.grad.numpy()[:, :2]) < 1e-12
    assert rel_l2(b["dL_dopacity"], op.grad.numpy()) < 1e-12
    # the covariance path differs from autograd only by the reference's 1/(det^2+1e-7) (backward.cu:203)
    assert rel_l2(b["dL_dmeans3D"], m3.grad.numpy()) < 1e-7
    assert rel_l2(b["dL_dscales"], sc.grad.numpy()) < 1e-6
    assert rel_l2(b["dL_drotations"], q.grad.numpy()) < 1e-6
    if mode == "precomp":
        assert rel_l2(b["dL_dcolors"], col.grad.numpy()) < 1e-12
    else:
        assert rel_l2(b["dL_dsh"], sh.grad.numpy()) < 1e-12


def test_fp32_oracle_close_to_fp64():
    s = tiny_scene(P=400, W=64, H=48, seed=3)
    f32 = oracle_forward(RasterOracle(np.float32), s)
    f64 = oracle_forward(RasterOracle(np.float64), s)
    same = f32["radii"] == f64["radii"]
    assert same.mean() > 0.99
    err = np.abs(f32["color"] - f64["color"]).max(0)
    assert (err > 1e-4).mean() < 2e-3  # borderline skip flips only
    assert np.median(err) < 1e-6


def test_oracle_sh_path_matches_reference_eval_sh():
    """Golden from the reference's utils/sh_utils.py::eval_sh (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLD, "sh_eval.npz"))
    shs, dirs = z["shs"], z["dirs"]
    N = shs.shape[0]
    from s3gaussian_amd import synth
    W = H = 64
    fov = 2 * math.atan(2.5)
    cam = synth.make_camera(np.eye(3), np.zeros(3), fov, fov, W, H)
    means = (4.0 * dirs).astype(np.float32)
    o = RasterOracle(np.float32)
    for deg in range(4):
        f = o.forward(bg=np.zeros(3), means3D=means, opacities=np.full((N, 1), 0.5), scales=np.full((N, 3), 0.3),
                      rotations=np.tile([1.0, 0, 0, 0], (N, 1)), shs=shs, sh_degree=deg, viewmatrix=cam["viewmatrix"].numpy(),
                      projmatrix=cam["projmatrix"].numpy(), campos=np.zeros(3), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                      image_height=H, image_width=W)
        vis = f["radii"] > 0
        assert vis.sum() > N // 2
        want = np.maximum(z[f"rgb_deg{deg}"] + 0.5, 0.0)
        np.testing.assert_allclose(f["state"]["rgb"][vis], want[vis], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(f["state"]["clamped"].reshape(N, 3)[vis], (z[f"rgb_deg{deg}"] + 0.5 < 0)[vis])


def _load_golden_net():
    from oracle import hexplane_ref as hr
    z = np.load(os.path.join(GOLD, "hexplane_deform.npz"))
    hyper = hr.default_hyper(kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32,
                                                 resolution=[8, 8, 8, 5]), multires=[1, 2])
    net = hr.deform_network(hyper)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    net.deformation_net.grid.set_aabb(z["aabb"][0].tolist(), z["aabb"][1].tolist())
    missing, unexpected = net.load_state_dict(sd, strict=True)
    return net, z


def test_hexplane_deformation_restatement_matches_reference_modules():
    net, z = _load_golden_net()
    t = lambda k: torch.from_numpy(z[k])
    xyz = t("xyz").clone().requires_grad_(True)
    shs = t("shs").clone().requires_grad_(True)
    feat = net.deformation_net.grid(xyz.detach(), t("time"))
    np.testing.assert_allclose(feat.detach().numpy(), z["hexplane_features"], rtol=1e-5, atol=1e-6)
    outs = net(xyz, t("scales"), t("rotations"), t("opacity"), shs, t("time"))
    names = ["means3D", "scales", "rotations", "opacity", "shs", "dx", "feat", "dshs"]
    loss = 0
    for n, o in zip(names, outs):
        np.testing.assert_allclose(o.detach().numpy(), z["out_" + n], rtol=1e-4, atol=1e-5)
        loss = loss + (o * t("w_" + n)).sum()
    loss.backward()
    assert rel_l2(xyz.grad.numpy(), z["grad_xyz"]) < 1e-4
    assert rel_l2(shs.grad.numpy(), z["grad_shs"]) < 1e-6
    for k, p in net.named_parameters():
        if "grad::" + k in z.files:
            assert rel_l2(p.grad.numpy(), z["grad::" + k]) < 1e-4, k
        else:
            assert p.grad is None, k


def test_loss_restatements_match_reference():
    from oracle import hexplane_ref as hr
    z = np.load(os.path.join(GOLD, "losses.npz"))
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    assert abs(hr.l1_loss(a, b).item() - z["l1"]) < 1e-7
    assert abs(hr.l2_loss(a, b).item() - z["l2"]) < 1e-7
    assert abs(hr.ssim(a, b).item() - z["ssim"]) < 1e-6
    assert abs(hr.depth_l2(torch.from_numpy(z["dp"]), torch.from_numpy(z["dg"])).item() - z["depth_l2"]) < 1e-7
    net, _ = _load_golden_net()
    reg = hr.plane_regulation(net.deformation_net.grid.grids, 0.01, 0.0001, 0.0001)
    assert abs(reg.item() - z["regulation"]) < 1e-7 * max(1.0, abs(float(z["regulation"])))


def test_glue_sh_to_colors_matches_reference():
    from oracle import hexplane_ref as hr
    z = np.load(os.path.join(GOLD, "hexplane_deform.npz"))
    gl = np.load(os.path.join(GOLD, "glue.npz"))
    cols = hr.shs_to_colors(3, torch.from_numpy(z["shs"]), torch.from_numpy(z["xyz"]), torch.from_numpy(gl["campos"]))
    np.testing.assert_allclose(cols.numpy(), gl["colors"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("P", [1, 2, 5, 1500, 5000])
def test_knn_oracle_equals_brute_force(P):
    g = np.random.default_rng(P)
    pts = g.normal(size=(P, 3)).astype(np.float32) * np.array([3, 1, 0.5], np.float32)
    got = knn_mean_dist2(pts)
    d = ((pts[:, None, :] - pts[None, :, :]) ** 2)
    d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
    np.fill_diagonal(d2, np.inf)
    if P >= 4:
        best = np.sort(d2, axis=1)[:, :3].astype(np.float32)
        want = (best[:, 0] + best[:, 1] + best[:, 2]) / np.float32(3.0)
        np.testing.assert_allclose(got, want, rtol=1e-6)
    else:
        assert np.all(np.isinf(got) | (got > 1e30))  # fewer than 3 neighbours: FLT_MAX sums overflow like the reference
