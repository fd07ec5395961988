"""Fused MLP-only probe at 1.2 M points: forward + backward timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s3gaussian_amd.pipeline import default_hyper  # noqa: E402
from s3gaussian_amd.deformation import deform_network  # noqa: E402
from s3gaussian_amd.mlp import deform_mlp  # noqa: E402

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
net = deform_network(default_hyper()).to(dev).deformation_net
x = torch.randn(P, 128, device=dev, requires_grad=True)
ws = [torch.randn(P, n, device=dev) for n in (3, 48, 3)]
for it in range(4):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    outs = deform_mlp(x, net.feature_out, net.pos_deform, net.shs_deform, net.dino_head)
    e[1].record()
    torch.autograd.backward(outs, ws)
    e[2].record()
    torch.cuda.synchronize()
    print(f"iter {it}: forward {e[0].elapsed_time(e[1]):.3f} ms  backward {e[1].elapsed_time(e[2]):.3f} ms")

# in-library hipEvent timings of the three MLP kernels (ids in include/s3g_raster.h)
import ctypes as C  # noqa: E402
from s3gaussian_amd import _lib  # noqa: E402
L = _lib.lib()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.s3g_profile_enable(1)
for it in range(5):
    outs = deform_mlp(x, net.feature_out, net.pos_deform, net.shs_deform, net.dino_head)
    torch.autograd.backward(outs, ws)
torch.cuda.synchronize()
for i, name in ((5, "mlp_forward"), (6, "mlp_backward"), (7, "mlp_wgrad")):
    ms = C.c_double()
    n = L.s3g_profile_read(i, C.byref(ms), None, None)
    print(f"{name}: {ms.value / max(n, 1):.4f} ms avg over {n}")
L.s3g_profile_enable(0)
