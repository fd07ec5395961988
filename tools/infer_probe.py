"""Fused inference kernel (HexPlane (+) MLP heads) alone at 1.2 M points, blocked processing order: in-library hipEvent time."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s3gaussian_amd import _lib, synth  # noqa: E402
from s3gaussian_amd.deformation import deform_network  # noqa: E402
from s3gaussian_amd.mlp import deform_infer, deform_mlp  # noqa: E402
from s3gaussian_amd.pipeline import default_hyper  # noqa: E402

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
sc = synth.street_scene(P=P, n_frames=2)
net = deform_network(default_hyper())
net.deformation_net.set_aabb(*sc["aabb"])
d = net.to(dev).deformation_net
xyz = sc["gaussians"]["xyz"].to(dev)
t = torch.full((P, 1), 0.37, device=dev)
x = xyz.clone().requires_grad_(True)
d.grid(x, t, uniform_time=True).sum().backward()     # leaves the blocked processing order in the field's cache
L = _lib.lib()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
with torch.no_grad():
    for _ in range(3):
        deform_infer(d.grid, xyz, t, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=True)
    torch.cuda.synchronize()
    for i in (2, 5, 9):
        L.s3g_profile_read(i, None, None, None)
    for arith in ("bf16x3",):     # the GEMM layers on the bf16 matrix pipe (s3g_deform_infer_split), bracket 9 as well
        for _ in range(3):
            deform_infer(d.grid, xyz, t, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=True, arithmetic=arith)
        torch.cuda.synchronize()
        L.s3g_profile_read(9, None, None, None)
        L.s3g_profile_enable(1)
        for _ in range(10):
            deform_infer(d.grid, xyz, t, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=True, arithmetic=arith)
        torch.cuda.synchronize()
        L.s3g_profile_enable(0)
        ms = C.c_double()
        n = L.s3g_profile_read(9, C.byref(ms), None, None)
        print(f"deform_infer (fused, {arith}): {ms.value / max(n, 1):.4f} ms avg over {n}")
    L.s3g_profile_enable(1)
    for _ in range(10):
        deform_infer(d.grid, xyz, t, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, uniform_time=True)
    for _ in range(10):
        f = d.grid(xyz, t, uniform_time=True)
        deform_mlp(f, d.feature_out, d.pos_deform, d.shs_deform, d.dino_head, need_feat=False)
    torch.cuda.synchronize()
    L.s3g_profile_enable(0)
for i, name in ((9, "deform_infer (fused)"), (2, "hexplane_forward"), (5, "mlp_forward (no feature head, no stash)")):
    ms = C.c_double()
    n = L.s3g_profile_read(i, C.byref(ms), None, None)
    print(f"{name}: {ms.value / max(n, 1):.4f} ms avg over {n}")
