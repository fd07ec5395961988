"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON MODULES (CPU) in the build
container.  /root/reference is read-only and is only ever imported here (never copied, never read by tests,
smoke() or bench.py at run time -- it does not exist on the GPU box).

    python tests/golden/make_golden.py        # rewrites *.npz next to this file

Fixtures (small, committed):
  sh_eval.npz          utils/sh_utils.py::eval_sh for degrees 0..3 (pins the in-kernel SH path of the oracle)
  hexplane_deform.npz  scene/deformation.py::deform_network (+ scene/hexplane.py) on a reduced-resolution config:
                       full state_dict, inputs, every output and the gradients of a fixed scalar loss
  losses.npz           utils/loss_utils.py l1/l2/ssim/compute_depth, scene/regulation.py plane smoothness and
                       scene/gaussian_model.py:710-749 regulation restated over the same grids
  glue.npz             gaussian_renderer/__init__.py:99-115 activations + SH->RGB glue (restated call sequence on
                       the reference's eval_sh)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


SWITCH_VARIANTS = {
    # no_grid=True cannot be pinned: the reference itself raises UnboundLocalError there (scene/deformation.py:79-91 assigns
    # `h` and then reads `hidden`)
    "grid_pe": dict(grid_pe=2),
    "static_mlp": dict(static_mlp=True, no_ds=False, no_do=False),
    "empty_voxel": dict(empty_voxel=True, no_ds=False),
    "apply_rotation": dict(no_dr=False, apply_rotation=True),
    "all_heads": dict(no_ds=False, no_dr=False, no_do=False),
    "no_dx_no_dshs": dict(no_dx=True, no_dshs=True, feat_head=False),
}


def empty_voxel_pattern():
    """Deterministic non-trivial content for DenseGrid.grid [1,1,64,64,64] (regenerated identically by the test)."""
    a = torch.linspace(0, 3.0, 64)
    return (1.0 + 0.3 * torch.sin(a[:, None, None] * 1.3 + a[None, :, None] * 0.7 - a[None, None, :]))[None, None]


def import_reference():
    # stub modules the reference imports for side effects only (SURVEY.md 7 "hard parts" vii)
    tk = types.ModuleType("tkinter"); tk.W = "w"; sys.modules["tkinter"] = tk
    for name in ("plyfile", "open3d", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    # a synthetic `scene` package object: sub-modules import from /root/reference/scene without executing
    # scene/__init__.py (which pulls the dataset readers)
    scene = types.ModuleType("scene"); scene.__path__ = [os.path.join(REF, "scene")]; sys.modules["scene"] = scene
    utils = types.ModuleType("utils"); utils.__path__ = [os.path.join(REF, "utils")]; sys.modules["utils"] = utils
    sys.path.insert(0, REF)
    mods = {n: importlib.import_module(n) for n in ("utils.sh_utils", "scene.hexplane", "scene.deformation",
                                                     "utils.loss_utils", "scene.regulation")}
    return mods


def main():
    m = import_reference()
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(7)

    # ---- eval_sh -------------------------------------------------------------------------------------
    N = 64
    shs = torch.randn(N, 16, 3, generator=g)
    dirs = torch.randn(N, 3, generator=g); dirs[:, 2] = dirs[:, 2].abs() + 0.5  # in front of a +z camera
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"shs": shs.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = m["utils.sh_utils"].eval_sh(deg, shs.transpose(1, 2), dirs).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), **out)

    # ---- deform_network on a reduced config ---------------------------------------------------------------
    from types import SimpleNamespace
    hyper = SimpleNamespace(net_width=64, timebase_pe=4, defor_depth=1, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2,
                            timenet_width=64, timenet_output=32, bounds=1.6, plane_tv_weight=0.0001,
                            time_smoothness_weight=0.01, l1_time_planes=0.0001,
                            kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                            "resolution": [8, 8, 8, 5]},
                            multires=[1, 2], no_dx=False, no_grid=False, no_ds=True, no_dr=True, no_do=True, no_dshs=False,
                            feat_head=True, empty_voxel=False, grid_pe=0, static_mlp=False, apply_rotation=False)
    net = m["scene.deformation"].deform_network(hyper)
    # NOTE feature_out expects grid feat_dim = 32 * len(multires) = 64 here
    with torch.no_grad():  # make planes non-trivial (time planes are all-ones at init)
        for p in net.deformation_net.grid.grids.parameters():
            p.add_(0.2 * torch.randn(p.shape, generator=g))
    net.deformation_net.set_aabb([2.0, 1.5, 1.0], [-1.0, -1.5, -0.5])
    P = 96
    xyz = (torch.rand(P, 3, generator=g) * torch.tensor([3.4, 3.4, 1.9]) + torch.tensor([-1.2, -1.7, -0.7])).requires_grad_(True)
    scales = torch.randn(P, 3, generator=g); rot = torch.randn(P, 4, generator=g); op = torch.randn(P, 1, generator=g)
    shs_in = torch.randn(P, 16, 3, generator=g).requires_grad_(True)
    time = torch.full((P, 1), 0.37)
    outs = net(xyz, scales, rot, op, shs_in, time)
    names = ["means3D", "scales", "rotations", "opacity", "shs", "dx", "feat", "dshs"]
    w = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * wi).sum() for o, wi in zip(outs, w))
    loss.backward()
    fx = {"xyz": xyz.detach().numpy(), "scales": scales.numpy(), "rotations": rot.numpy(), "opacity": op.numpy(),
          "shs": shs_in.detach().numpy(), "time": time.numpy(), "aabb": net.deformation_net.grid.aabb.detach().numpy(),
          "grad_xyz": xyz.grad.numpy(), "grad_shs": shs_in.grad.numpy(), "loss": np.float64(loss.item())}
    for n, o, wi in zip(names, outs, w):
        fx["out_" + n] = o.detach().numpy(); fx["w_" + n] = wi.numpy()
    for k, v in net.state_dict().items():
        fx["sd::" + k] = v.numpy()
    for k, p in net.named_parameters():
        if p.grad is not None:
            fx["grad::" + k] = p.grad.numpy()
    # hexplane features alone
    feat = net.deformation_net.grid(xyz.detach(), time)
    fx["hexplane_features"] = feat.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "hexplane_deform.npz"), **fx)

    # ---- losses ------------------------------------------------------------------------------------------
    L = m["utils.loss_utils"]; Rg = m["scene.regulation"]
    a = torch.rand(1, 3, 40, 56, generator=g); b = (a + 0.1 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    dp = torch.rand(1, 1, 40, 56, generator=g) * 100; dg = torch.rand(1, 40, 56, generator=g) * 100
    dg[0, :5] = 0.0
    grids = net.deformation_net.grid.grids
    sp = sum(Rg.compute_plane_smoothness(gr[i]) for gr in grids for i in (0, 1, 3))
    tm = sum(Rg.compute_plane_smoothness(gr[i]) for gr in grids for i in (2, 4, 5))
    l1t = sum(torch.abs(1 - gr[i]).mean() for gr in grids for i in (2, 4, 5))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), a=a.numpy(), b=b.numpy(), dp=dp.numpy(), dg=dg.numpy(),
                        l1=L.l1_loss(a, b).item(), l2=L.l2_loss(a, b).item(), ssim=L.ssim(a, b).item(),
                        depth_l2=L.compute_depth("l2", dp, dg).item(), plane_smooth_spatial=sp.item(),
                        plane_smooth_time=tm.item(), l1_time=l1t.item(),
                        regulation=(0.0001 * sp + 0.01 * tm + 0.0001 * l1t).item())

    # ---- render glue (gaussian_renderer/__init__.py:99-115) -----------------------------------------------
    campos = torch.tensor([0.3, -0.2, 1.1])
    shs_view = shs_in.detach().transpose(1, 2).view(-1, 3, 16)
    dir_pp = xyz.detach() - campos.repeat(P, 1)
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    cols = torch.clamp_min(m["utils.sh_utils"].eval_sh(3, shs_view, dir_pp) + 0.5, 0.0)
    np.savez_compressed(os.path.join(HERE, "glue.npz"), campos=campos.numpy(), colors=cols.numpy())

    # ---- non-default switches of scene/deformation.py (grid_pe, no_grid, static_mlp, empty_voxel, apply_rotation, all heads)
    sw = {}
    for tag, over in SWITCH_VARIANTS.items():
        torch.manual_seed(99)
        hy = SimpleNamespace(**{**vars(hyper), "kplanes_config": {"grid_dimensions": 2, "input_coordinate_dim": 4,
                                                                  "output_coordinate_dim": 32, "resolution": [4, 4, 4, 3]}, **over})
        netv = m["scene.deformation"].deform_network(hy)
        gv = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for pp in netv.deformation_net.grid.grids.parameters():
                pp.add_(0.2 * torch.randn(pp.shape, generator=gv))
            if over.get("empty_voxel"):
                netv.deformation_net.empty_voxel.grid.copy_(empty_voxel_pattern())
        netv.deformation_net.set_aabb([2.0, 1.5, 1.0], [-1.0, -1.5, -0.5])
        Pv = 64
        xv = (torch.rand(Pv, 3, generator=gv) * torch.tensor([2.6, 2.6, 1.2]) + torch.tensor([-0.8, -1.3, -0.4])).requires_grad_(True)
        sv, rv, ov = torch.randn(Pv, 3, generator=gv), torch.randn(Pv, 4, generator=gv), torch.randn(Pv, 1, generator=gv)
        shv = torch.randn(Pv, 16, 3, generator=gv)
        tv = torch.full((Pv, 1), 0.61)
        outs = netv(xv, sv, rv, ov, shv, tv)
        wv = [None if o is None else torch.randn(o.shape, generator=gv) for o in outs]
        sum((o * wi).sum() for o, wi in zip(outs, wv) if o is not None).backward()
        sw[tag + "::xyz"], sw[tag + "::scales"], sw[tag + "::rotations"] = xv.detach().numpy(), sv.numpy(), rv.numpy()
        sw[tag + "::opacity"], sw[tag + "::shs"], sw[tag + "::time"] = ov.numpy(), shv.numpy(), tv.numpy()
        sw[tag + "::grad_xyz"] = xv.grad.numpy()
        for n, o, wi in zip(names, outs, wv):
            if o is not None:
                sw[f"{tag}::out_{n}"] = o.detach().numpy(); sw[f"{tag}::w_{n}"] = wi.numpy()
        for k, v in netv.state_dict().items():
            if "empty_voxel" not in k:       # regenerated from empty_voxel_pattern() by the test
                sw[f"{tag}::sd::{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "deform_switches.npz"), **sw)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
