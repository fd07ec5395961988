"""Why does bench.py's bf16x3 render loop read 4-5 ms/frame in round 4 when round 3 measured 1.45?  Exact / split render loops
alternated, wall time and the inference kernel's own bracket, with the synchronous and the asynchronous rasterizer forward."""
import ctypes as C
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import s3gaussian_amd.deformation as deformation  # noqa: E402
from s3gaussian_amd import _lib, raster_C  # noqa: E402
from s3gaussian_amd.pipeline import render, training_step  # noqa: E402

dev = torch.device("cuda")
L = _lib.lib()
L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
pc, cams, hyper, opt, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev)
pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
gts = bench.make_targets(pc, cams[0], bg, hyper, seed=1)
for i in range(3):
    training_step(pc, cams[i], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
torch.cuda.synchronize()


def loop(arith, asyn, n=20, bracket=False):
    deformation.INFER_ARITHMETIC = arith
    raster_C.set_async(asyn)
    with torch.no_grad():
        for i in range(3):
            render(cams[i], pc, pipe, bg, stage="fine")
        torch.cuda.synchronize()
        L.s3g_profile_read(9, None, None, None)
        L.s3g_profile_enable(1 if bracket else 0)
        t0 = time.perf_counter()
        for i in range(n):
            render(cams[i % len(cams)], pc, pipe, bg, stage="fine")
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        L.s3g_profile_enable(0)
    ms = C.c_double()
    k = L.s3g_profile_read(9, C.byref(ms), None, None)
    print(f"{arith:7s} raster {'async' if asyn else 'sync '} brackets {'on ' if bracket else 'off'}: wall {1e3 * (t2 - t0) / n:6.3f} ms/frame, "
          f"host enqueue {1e3 * (t1 - t0) / n:6.3f}, infer kernel {ms.value / k if k else float('nan'):6.3f} ms", flush=True)


for rnd in range(2):
    for asyn in (True, False):
        for arith in ("f32", "bf16x3"):
            for br in (False, True):
                loop(arith, asyn, bracket=br)
deformation.INFER_ARITHMETIC = "f32"
