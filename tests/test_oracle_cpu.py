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


def _tile_can_contribute_np(mx, my, a, b, c, o, tx, ty, W, H):
    """float32 numpy mirror of s3gaussian_amd/csrc/geom_math.hpp::tile_can_contribute (exact tile culling)."""
    f = np.float32
    det = f(a * c - b * b)
    if not (det > 0 and a > 0 and c > 0 and o == o):
        return True
    L = f(np.log(f(255.0) * o))
    if L < 0:
        return False
    budget = f(2.0) * f(L * f(1.001) + f(1.0e-5))
    x0, x1 = f(tx * 16) - f(0.01), f(min(tx * 16 + 15, W - 1)) + f(0.01)
    y0, y1 = f(ty * 16) - f(0.01), f(min(ty * 16 + 15, H - 1)) + f(0.01)
    dx_lo, dx_hi, dy_lo, dy_hi = f(mx - x1), f(mx - x0), f(my - y1), f(my - y0)
    in_x, in_y = dx_lo <= 0 <= dx_hi, dy_lo <= 0 <= dy_hi
    if in_x and in_y:
        return True

    def edge(X, A, B, C, lo, hi):
        y = f(min(hi, max(lo, f(-(B * X) * f(1.0 / C)))))
        return f(A * X * X + f(2.0) * B * X * y + C * y * y)

    q = f(3.0e38)
    if not in_x:
        q = edge(dx_lo if dx_lo > 0 else dx_hi, a, b, c, dy_lo, dy_hi)
    if not in_y:
        q = min(q, edge(dy_lo if dy_lo > 0 else dy_hi, c, b, a, dx_lo, dx_hi))
    return bool(q <= budget)


def test_exact_tile_cull_never_drops_a_contributing_pixel():
    """The binning-time cull (include/s3g_raster.h: s3g_raster_set_exact_cull) must be conservative: whenever it drops a
    (tile, Gaussian) pair, NO pixel centre of that tile passes the blend loops' test alpha = min(0.99, o * exp(power)) >=
    1/255 with power <= 0 (forward.cu:330-341), evaluated here in float32 like the kernels.  Brute force over random
    conics (round, elongated, rotated), opacities around the 1/255 threshold and all tiles of a 5 x 5 neighbourhood; also
    checks that the cull is not vacuous."""
    rng = np.random.default_rng(0)
    f = np.float32
    W, H = 200, 150                      # right / bottom tiles are partial (200 = 12.5 tiles, 150 = 9.4 tiles)
    lat = np.arange(16, dtype=np.float32)
    dropped = kept = 0
    for _ in range(1500):
        sx, sy = np.exp(rng.uniform(np.log(0.6), np.log(40.0), 2))
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([sx * sx, sy * sy]) @ R.T + 0.3 * np.eye(2)
        inv = np.linalg.inv(cov)
        a, b, c = f(inv[0, 0]), f(inv[0, 1]), f(inv[1, 1])
        o = f(rng.choice([rng.uniform(0.003, 0.006), rng.uniform(0.006, 0.05), rng.uniform(0.05, 1.0)]))
        mx, my = f(rng.uniform(-20, W + 20)), f(rng.uniform(-20, H + 20))
        ctx, cty = int(np.clip(mx // 16, 0, 12)), int(np.clip(my // 16, 0, 9))
        for ty in range(max(0, cty - 2), min(10, cty + 3)):
            for tx in range(max(0, ctx - 2), min(13, ctx + 3)):
                px = (f(tx * 16) + lat)[(tx * 16 + lat) < W]
                py = (f(ty * 16) + lat)[(ty * 16 + lat) < H]
                dx, dy = (mx - px)[None, :], (my - py)[:, None]
                power = f(-0.5) * (a * dx * dx + c * dy * dy) - b * dx * dy
                alpha = np.minimum(f(0.99), o * np.exp(power, dtype=np.float32))
                contributes = bool(((power <= 0) & (alpha >= f(1.0 / 255.0))).any())
                keep = _tile_can_contribute_np(mx, my, a, b, c, o, tx, ty, W, H)
                assert keep or not contributes, (mx, my, a, b, c, o, tx, ty)
                dropped += not keep
                kept += keep
    assert dropped > 0.3 * (dropped + kept)      # the test above is not vacuous: a third of the pairs are culled
