"""Why did the driver's round-3 headline loop run 2.1 ms/step slower than every later loop of the same process?  (VERDICT r3,
Weak #1.)  One process, BASELINE cfg3, the SAME training step timed in loops that differ in exactly one thing each:

    raster   sync  = the reference's one host wait per rasterizer forward (round 3's only mode)
             async = s3g_raster_forward_async (round 4 default): no host wait inside an iteration
    brackets on    = the nine in-library hipEvent brackets inside the timed region (round 3's headline loop), off = none
    views    cold  = the first loop after the scene is built, 5 warm-ups, 20 steps over 25 DISTINCT views (what the driver's
                     `bench.py --steps 20 --warmup 5` times);  replay = the same 20 views again

Per loop: wall ms/step, the host's enqueue ms/step (time until the last step is queued), stream ms/step from one event pair per
step (median / max), and the number of generation-2 garbage collections of the interpreter that ran inside the loop.  The order is
cold first (it can only be first), then every combination twice, second time reversed, then six 60-step loops.

    python tools/headline_variance.py > gpurun_out/r04_headline_variance.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from s3gaussian_amd import _lib, raster_C  # noqa: E402
from s3gaussian_amd.pipeline import training_step  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
L = _lib.lib()
pc, cams, hyper, opt, bg = bench.build_scene(P, 1600, 1066, 50, dev)
from s3gaussian_amd import dp  # noqa: E402
views = dp.shard_views(len(cams), 0, 1, seed=0)[:25]
targets = {v: bench.make_targets(pc, cams[v], bg, hyper, seed=1000 + v) for v in views[:12]}
tk = list(targets)


def step(i):
    v = views[i % len(views)]
    gts = targets[v] if v in targets else targets[tk[i % len(tk)]]
    return training_step(pc, cams[v], *gts, hyper, opt, bg, stage="fine", densify_stats=True)


def clear():
    for i in range(10):
        L.s3g_profile_read(i, None, None, None)


rows = []


def loop(label, asyn, brackets, idx, warm, collect=False):
    raster_C.set_async(asyn)
    L.s3g_profile_enable(0)
    for i in warm:
        step(i)
    clear()
    L.s3g_profile_enable(1 if brackets else 0)
    dt, t_enq, per = bench.timed_loop(step, idx, 1, dev, collect=collect)
    L.s3g_profile_enable(0)
    clear()
    per = sorted(per)
    rows.append((label, "async" if asyn else "sync", "on" if brackets else "off", 1e3 * dt / len(idx), 1e3 * t_enq / len(idx),
                 per[len(per) // 2], per[-1], bench.GC_PASSES[-1]))
    print("%-36s raster %-5s brackets %-3s  wall %7.3f  host-enqueue %7.3f  stream median %7.3f  max %7.3f  ms/step  gen-2 GC passes inside: %d"
          % rows[-1], flush=True)


print(f"# BASELINE cfg3-shaped scene, P = {P}, 1066x1600; 20 timed steps per loop; torch {torch.__version__}; "
      f"host cores {os.cpu_count()}", flush=True)
cold = list(range(5, 25))
loop("cold (first loop, r3 headline shape)", False, True, cold, range(5))
combos = [(a_, b_) for a_ in (False, True) for b_ in (True, False)]
for rnd, order in enumerate((combos, combos[::-1])):
    for asyn, br in order:
        loop(f"replay #{rnd + 1}", asyn, br, cold, range(3))
# the interpreter's generation-2 collections: a pass over this process's heap is a host stall of tens of milliseconds; with the
# synchronous forward the GPU idles for all of it, with the asynchronous one only for what exceeds the host's lead
import gc  # noqa: E402
import time  # noqa: E402
t0 = time.perf_counter()
n = gc.collect()
print(f"# one full gc.collect() of this process: {1e3 * (time.perf_counter() - t0):.1f} ms ({n} objects freed, {len(gc.get_objects())} tracked)")
for rnd in range(3):
    for asyn in (False, True):
        loop(f"long loop #{rnd + 1} (60 steps, no pre-collect)", asyn, False, [5 + (i % 20) for i in range(60)], range(2), collect=False)
st = raster_C.async_status(dev, block=True)
print(f"# asynchronous forwards: {st['calls']} calls, overflows {st['overflows']}, capacity {st['capacity']}")
