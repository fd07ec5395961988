"""Fused inference kernel HexPlane sampler (+) MLP heads (include/s3g_mlp.h::s3g_deform_infer) vs the two separate kernels it
replaces under torch.no_grad(): same arithmetic in the same order, so dx / dshs must be BIT-IDENTICAL; and through
pipeline.render() the image must not change.  (The two-kernel path itself is oracle-checked in test_hexplane_gpu.py,
test_mlp_gpu.py and at BASELINE size in test_parity_fullsize_gpu.py; scene/deformation.py:108-166.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(dev, aabb, seed=0):
    from s3gaussian_amd.deformation import deform_network
    from s3gaussian_amd.pipeline import default_hyper
    torch.manual_seed(seed)
    net = deform_network(default_hyper())
    net.deformation_net.set_aabb(*aabb)
    d = net.deformation_net
    with torch.no_grad():
        for p in d.grid.grids.parameters():
            p.add_(0.2 * torch.randn_like(p))
        for m in d.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.bias, std=0.1)
    return net.to(dev).deformation_net


@pytest.mark.parametrize("tmode", ["uniform", "per_point"])
@pytest.mark.parametrize("P", [1, 7, 31, 32, 33, 255, 257, 5000, 70_001, 1_200_000])
def test_fused_inference_is_bit_identical_to_the_two_kernels(gpu_device, P, tmode):
    from s3gaussian_amd import synth
    from s3gaussian_amd.mlp import deform_infer, deform_mlp
    dev = gpu_device
    sc = synth.street_scene(P=max(P, 64), seed=1, n_frames=2)
    d = _net(dev, sc["aabb"], seed=P % 7)
    g = torch.Generator().manual_seed(P)
    xyz = sc["gaussians"]["xyz"][:P].to(dev).contiguous()
    xyz[::5] += (torch.rand(xyz[::5].shape, generator=g).to(dev) - 0.5) * torch.tensor([200.0, 80.0, 30.0], device=dev)   # some outside the aabb
    time = (torch.rand(P, 1, generator=g).to(dev) * 1.2 - 0.1) if tmode == "per_point" else torch.full((P, 1), 0.41, device=dev)
    ut = tmode == "uniform"
    with torch.no_grad():
        feats = d.grid(xyz, time, uniform_time=ut)
        dx0, dshs0, none = deform_mlp(feats, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, need_feat=False)
        assert none is None
        dx1, dshs1 = deform_infer(d.grid, xyz, time, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=ut)
        assert torch.equal(dx0, dx1) and torch.equal(dshs0, dshs1)
        if P >= 5000:   # and again in the blocked processing order a backward leaves behind
            with torch.enable_grad():
                x = xyz.clone().requires_grad_(True)
                d.grid(x, time, uniform_time=ut).sum().backward()
            assert d.grid._order_cache.get("order") is not None
            dx2, dshs2 = deform_infer(d.grid, xyz, time, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=ut)
            assert torch.equal(dx0, dx2) and torch.equal(dshs0, dshs2)


def test_render_uses_the_fused_inference_path_and_is_unchanged(gpu_device, monkeypatch):
    from types import SimpleNamespace
    import s3gaussian_amd.deformation as dm
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, render
    dev = gpu_device
    scn = synth.street_scene(P=40_000, seed=2, width=320, height=208, n_frames=2)
    torch.manual_seed(0)
    pc = GaussianParams(3, default_hyper())
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    with torch.no_grad():
        for p in pc._deformation.deformation_net.pos_deform.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cam = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scn["cameras"][1].items()}
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    calls = []
    real = dm.deform_infer
    monkeypatch.setattr(dm, "deform_infer", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        a = render(cam, pc, pipe, scn["bg"].to(dev), stage="fine", return_dx=True)
        assert len(calls) == 1
        monkeypatch.setattr(dm, "FUSED_INFERENCE", False)
        b = render(cam, pc, pipe, scn["bg"].to(dev), stage="fine", return_dx=True)
        assert len(calls) == 1
    for k in ("render", "depth", "radii", "dx", "dshs"):
        assert torch.equal(a[k], b[k]), k
    # with autograd on (training) or the feature image requested the fused path is not taken
    render(cam, pc, pipe, scn["bg"].to(dev), stage="fine", render_feat=True)
    with torch.no_grad():
        render(cam, pc, pipe, scn["bg"].to(dev), stage="fine", render_feat=True)
    assert len(calls) == 1
