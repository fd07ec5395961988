"""HexPlane sampler-only probe at 1.2 M points (cfg3 aabb): forward + backward timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s3gaussian_amd import synth  # noqa: E402
from s3gaussian_amd.hexplane import HexPlaneField  # noqa: E402

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
sc = synth.street_scene(P=P, n_frames=2)
cfg = dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32, resolution=[64, 64, 64, 25])
f = HexPlaneField(1.6, cfg, [1, 2, 4, 8])
f.set_aabb(*sc["aabb"])
f = f.to(dev)
xyz = sc["gaussians"]["xyz"].to(dev)
if "morton" in sys.argv[2:]:   # the Gaussians THEMSELVES in Morton order (pipeline.GaussianParams.reorder_spatially's key)
    lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
    q = ((xyz - lo) / (hi - lo) * 1023).long().clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    xyz = xyz[torch.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), stable=True)].contiguous()
    print("points in Morton order")
xyz.requires_grad_(True)
t = torch.full((P, 1), 0.37, device=dev)
w = torch.randn(P, 128, device=dev)
for it in range(4):
    for p in f.parameters():
        p.grad = None
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    out = f(xyz, t)
    e[1].record()
    (out * w).sum().backward()
    e[2].record()
    torch.cuda.synchronize()
    print(f"iter {it}: forward {e[0].elapsed_time(e[1]):.3f} ms  backward(+loss) {e[1].elapsed_time(e[2]):.3f} ms")

import ctypes as C  # noqa: E402
from s3gaussian_amd import _lib  # noqa: E402
L = _lib.lib()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
L.s3g_profile_enable(1)
tot = []
for it in range(10):      # SORT_REFRESH = 8: one of these re-sorts the walk orders, the others reuse them
    for p in f.parameters():
        p.grad = None
    out = f(xyz, t, uniform_time=True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    (out * w).sum().backward()
    e[1].record()
    torch.cuda.synchronize()
    tot.append(e[0].elapsed_time(e[1]))
print("backward(+loss) per iteration, ms: " + " ".join(f"{x:.3f}" for x in tot) + f"   (max - median = the re-sort: {max(tot) - sorted(tot)[len(tot) // 2]:.3f} ms)")
for i, name in ((2, "hexplane_forward"), (3, "hexplane_backward_point"), (4, "hexplane_scatter")):
    ms = C.c_double()
    n = L.s3g_profile_read(i, C.byref(ms), None, None)
    print(f"{name}: {ms.value / max(n, 1):.4f} ms avg over {n}")
L.s3g_profile_enable(0)
print("S3G_HEX_BACKWARD =", os.environ.get("S3G_HEX_BACKWARD", "slab"))

if "walks" in sys.argv[2:]:
    # the twelve (orientation, level) scatter walks timed ONE BY ONE (include/s3g_hexplane.h::
    # s3g_hexplane_debug_walk_mask; the gradients of such a run are incomplete -- timing only).  Level 0 has ~290 points per cell,
    # level 3 ~4.6: is one serial-walker design right for both?
    L.s3g_hexplane_debug_walk_mask.argtypes = [C.c_uint32]
    L.s3g_hexplane_debug_walk_mask.restype = None
    SC = 4   # S3G_PROFILE_HEXPLANE_SCATTER
    L.s3g_profile_enable(1)
    rows = []
    for mask_name, mask in [("all", 0xffffffff)] + [(f"o{o} l{l}", 1 << (o * 4 + l)) for o in range(3) for l in range(4)] + \
            [(f"level {l} (3 orientations)", sum(1 << (o * 4 + l) for o in range(3))) for l in range(4)]:
        L.s3g_hexplane_debug_walk_mask(mask)
        for i in range(10):
            L.s3g_profile_read(i, None, None, None)
        for it in range(6):
            for p in f.parameters():
                p.grad = None
            out = f(xyz, t, uniform_time=True)
            (out * w).sum().backward()
        torch.cuda.synchronize()
        ms = C.c_double()
        n = L.s3g_profile_read(SC, C.byref(ms), None, None)
        rows.append((mask_name, ms.value / max(n, 1)))
    L.s3g_hexplane_debug_walk_mask(0xffffffff)
    L.s3g_profile_enable(0)
    print("scatter walks one by one (ms per launch; the launch always has the full grid, masked walks exit at once):")
    for name, ms in rows:
        print(f"  {name:28s} {ms:.4f}")
