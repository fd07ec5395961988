"""Why is hexplane_scatter_kernel slower inside a training step (0.85-1.07 ms) than in tools/hex_probe.py (0.71-0.74 ms)?

One process, the bench's own scene: (A) K training steps with the in-library brackets on; (B) the field alone (forward, a weighted
sum, backward) on the SAME field object, the same points and the same cached walk orders; (C) the same with the plane regulariser on
the sampler's node (plane-gradient buffers pre-filled during the forward, as in a training step); (D) the same after the points were
put back to where the walk orders were sorted (fresh orders, no movement).  Prints the bracket averages of the three HexPlane kernels.
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from s3gaussian_amd import _lib, raster_C  # noqa: E402
from s3gaussian_amd.pipeline import training_step  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
NV = int(sys.argv[3]) if len(sys.argv) > 3 else 6      # distinct views the training steps cycle through (the bench line: steps + warm-up)
pc, cams, hyper, opt, bg = bench.build_scene(P, 1600, 1066, 50, dev)
L = _lib.lib()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.s3g_profile_enable(1)
for i in range(10):
    L.s3g_profile_read(i, None, None, None)
targets = {v: bench.make_targets(pc, cams[v], bg, hyper, seed=1000 + v) for v in range(6)}
torch.cuda.synchronize()
_ms = C.c_double()
_n = L.s3g_profile_read(2, C.byref(_ms), None, None)
print(f"0. the bench's make_targets renders (no backward has run yet: no processing order cached)   hexplane_forward {_ms.value / max(_n, 1):.4f} ({_n})", flush=True)
L.s3g_profile_enable(0)
NAMES = ((2, "hexplane_forward"), (3, "hexplane_backward_point"), (4, "hexplane_scatter"))


def clear():
    for i in range(10):
        L.s3g_profile_read(i, None, None, None)


def report(tag):
    torch.cuda.synchronize()
    parts = []
    for i, name in NAMES:
        ms = C.c_double()
        n = L.s3g_profile_read(i, C.byref(ms), None, None)
        parts.append(f"{name} {ms.value / max(n, 1):.4f} ({n})")
    print(f"{tag:58s} " + "  ".join(parts), flush=True)


raster_C.set_async(True) if hasattr(raster_C, "set_async") else None
VIEWS = [(7 * i) % len(cams) for i in range(NV)] if NV > 6 else list(range(6))     # NV > 6: spread over cameras AND timestamps


def steps(n):
    for i in range(n):
        training_step(pc, cams[VIEWS[i % len(VIEWS)]], *targets[i % 6], hyper, opt, bg, stage="fine", densify_stats=True)


steps(max(6, len(VIEWS)))       # warm-up (capacity history, walk orders)
torch.cuda.synchronize()
L.s3g_profile_enable(1)
clear()
steps(K)
report(f"A. {K} training steps")

grid = pc._deformation.deformation_net.grid
xyz = pc._xyz.detach().clone().requires_grad_(True)
t = torch.full((1,), 0.37, device=dev)
w = torch.randn(P, 128, device=dev)


def field_only(n, reg=False, x=xyz):
    for _ in range(n):
        for p in grid.parameters():
            p.grad = None
        if reg:
            out, r = grid(x, t, uniform_time=True, reg_weights=(hyper.time_smoothness_weight, hyper.l1_time_planes, hyper.plane_tv_weight))
            ((out * w).sum() + r).backward()
        else:
            out = grid(x, t, uniform_time=True)
            (out * w).sum().backward()


field_only(2)
clear()
field_only(10)
report("B. field alone, the step's points + cached orders")
try:
    field_only(2, reg=True)
    clear()
    field_only(10, reg=True)
    report("C. field alone + plane regulariser on the node")
except Exception as e:  # noqa: BLE001
    print("C. skipped:", repr(e)[:200])
grid._order_cache.clear()
field_only(2)
clear()
field_only(10)
report("D. field alone, walk orders re-sorted on these points")
gw = torch.randn(P, 128, device=dev) * 1e-9
w = gw
field_only(2)
clear()
field_only(10)
report("E. field alone, upstream gradient scaled by 1e-9")
# F. the training steps again (is A reproducible after B-E?)
clear()
steps(K)
report(f"F. {K} training steps again")
os.environ["S3G_HEX_SORT_REFRESH"] = "1"
import s3gaussian_amd.hexplane as hx  # noqa: E402
hx.SORT_REFRESH = 1
clear()
steps(K)
report(f"G. {K} training steps, walk orders re-sorted on every backward")
L.s3g_profile_enable(0)

# H. how far do the points move in one training step, and what does a displacement of that size cost the walk?
hx.SORT_REFRESH = 16
x0 = pc._xyz.detach().clone()
steps(1)
dx1 = (pc._xyz.detach() - x0).abs()
steps(7)
dx8 = (pc._xyz.detach() - x0).abs()
print("H. |xyz movement| per coordinate: 1 step mean %.3e max %.3e;  8 steps mean %.3e max %.3e   (finest cells: %.3e %.3e %.3e)" % (
    dx1.mean().item(), dx1.max().item(), dx8.mean().item(), dx8.max().item(), 100 / 511, 40 / 511, 10 / 511), flush=True)
L.s3g_profile_enable(1)
w = torch.randn(P, 128, device=dev)
base = pc._xyz.detach().clone()
for eps in (0.0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2):
    grid._order_cache.clear()
    xs = base.clone().requires_grad_(True)
    field_only(1, x=xs)                      # sorts the orders on `base`
    xm = (base + eps * (2 * torch.rand_like(base) - 1)).requires_grad_(True)
    field_only(1, x=xm)
    clear()
    field_only(6, x=xm)                      # ages 2 .. 7: no re-sort
    report(f"H. orders sorted on x, walked on x + U(-{eps:g}, {eps:g})")
L.s3g_profile_enable(0)
