"""Where does the HOST spend an iteration of the reference's own train.py::scene_reconstruction under patch_reference()?  The reference
loop waits for the device once per iteration (`loss.item()`), so between that wait and the last launch of the next iteration every host
microsecond is GPU idle time (profiles/r05_patched_iteration_trace_after.txt: 1.6 ms idle of a 9.6-ms traced iteration).  cProfile of 200
iterations at cfg3 size; prints the functions by cumulative and by own time."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref_py  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pc, cams, hyper0, opt0, bg = bench.build_scene(1_200_000, 1600, 1066, 50, dev)
targets = {v: bench.make_targets(pc, cams[v], bg, hyper0, seed=1000 + v) for v in range(6)}
aabb = tuple(pc._deformation.deformation_net.grid.aabb.detach().cpu().tolist())
state = {k: v.detach().clone() for k, v in pc._deformation.state_dict().items()}
gs = dict(xyz=pc._xyz.detach(), log_scales=pc._scaling.detach(), rotations_raw=pc._rotation.detach(),
          opacity_logit=pc._opacity.detach(), shs=torch.cat([pc._features_dc.detach(), pc._features_rest.detach()], dim=1))
ref = ref_py.load(patch=True)
args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
dataset.render_process = False
gm = ref_py.make_gaussians(ref, gs, aabb, hyper)
gm._deformation.load_state_dict(state)
cam_objs = [ref_py.make_camera(ref, cams[v], targets[v], uid=v) for v in range(6)]
timer = ref_py.RecordingTimer(record_locals=False)
ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cam_objs), dataset, hyper, opt, pipe, 10, "fine", timer)     # warm-up
torch.cuda.synchronize()
timer = ref_py.RecordingTimer(record_locals=False)
prof = cProfile.Profile()
prof.enable()
ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cam_objs), dataset, hyper, opt, pipe, N, "fine", timer)
torch.cuda.synchronize()
prof.disable()
st = timer.stamps
print(f"{N} iterations: {1000.0 * (st[-1] - st[0]) / (len(st) - 1):.3f} ms per iteration (under cProfile)")
for key in ("cumulative", "tottime"):
    print(f"---- by {key} (ms per iteration)")
    ps = pstats.Stats(prof)
    rows = sorted(ps.stats.items(), key=lambda kv: -(kv[1][3] if key == "cumulative" else kv[1][2]))[:45]
    for (fn, line, name), (cc, nc, tt, ct, callers) in rows:
        short = fn.replace(ROOT + "/", "").split("site-packages/")[-1]
        print(f"{1000.0 * (ct if key == 'cumulative' else tt) / N:8.3f}  calls/it {nc / N:7.1f}  {short}:{line} {name}")
