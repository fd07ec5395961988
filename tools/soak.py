"""Sustained throughput of the fused training step at BASELINE cfg3: N steps over all 150 views (every view has its own instance
count, so the arena capacities and the allocator see the whole range), the interpreter's collector running normally (no freeze,
no pre-collect), re-sorts of the HexPlane walk orders every 16 steps included.  One JSON line.

    python tools/soak.py [steps=600] [sync] [P=2500000] [scale=3.5]"""
import gc
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from s3gaussian_amd import raster_C  # noqa: E402
from s3gaussian_amd.pipeline import training_step  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
if "sync" in sys.argv[2:]:
    raster_C.set_async(False)
P = next((int(a[2:]) for a in sys.argv[2:] if a.startswith("P=")), 1_200_000)            # P=2500000: BASELINE cfg5 size
SCALE = next((float(a[6:]) for a in sys.argv[2:] if a.startswith("scale=")), 1.0)         # scale=3.5: ~10 M instances per view
dev = torch.device("cuda", 0)
pc, cams, hyper, opt, bg = bench.build_scene(P, 1600, 1066, 50, dev, scale_mult=SCALE)
targets = {v: bench.make_targets(pc, cams[v], bg, hyper, seed=1000 + v) for v in range(0, len(cams), 13)}
tk = list(targets)
g = torch.Generator().manual_seed(0)
order = torch.randperm(len(cams), generator=g).tolist()


def step(i):
    v = order[i % len(order)]
    return training_step(pc, cams[v], *targets[tk[i % len(tk)]], hyper, opt, bg, stage="fine", densify_stats=True)


for i in range(10):
    step(i)
torch.cuda.synchronize()
raster_C.async_reset_statistics(dev)
gen2 = gc.get_stats()[2]["collections"]
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.perf_counter()
evs[0].record()
losses = []
for i in range(N):
    loss, _ = step(10 + i)
    evs[i + 1].record()
    if i % 50 == 0:
        losses.append(loss)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
per = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(N))
st = raster_C.async_status(dev, block=True)
print(json.dumps({"what": f"sustained fused training step, {P} Gaussians (scale x{SCALE}), 1066x1600, all 150 views in random order", "steps": N,
                  "rasterizer_forward": "asynchronous" if st["enabled"] else "synchronous",
                  "iters_per_s": round(N / dt, 2), "ms_per_step": round(1e3 * dt / N, 3), "host_enqueue_ms_per_step": round(1e3 * t_enq / N, 3),
                  "step_ms_min": round(per[0], 3), "step_ms_median": round(per[N // 2], 3), "step_ms_p99": round(per[int(N * 0.99)], 3),
                  "step_ms_max": round(per[-1], 3), "steps_over_2x_median": sum(x > 2 * per[N // 2] for x in per),
                  "gc_gen2_passes": gc.get_stats()[2]["collections"] - gen2, "arena_overflows": len(st["overflows"]),
                  "mean_instances_per_view": round(st["mean_instances"] or 0), "capacity": {str(k): v for k, v in st["capacity"].items()},
                  "losses_finite": bool(all(torch.isfinite(x) for x in losses)), "loss_first_last": [float(losses[0]), float(losses[-1])]}))
