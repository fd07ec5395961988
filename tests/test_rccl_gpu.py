"""RCCL executed on the leased MI355X.  One GPU per lease rules out N > 1, and RCCL refuses two ranks on one
device, so the `nccl` backend is exercised as a process group of ONE rank (`S3G_FORCE_DIST=1`, s3gaussian_amd/dp.py::force_dist):
library load, communicator init, in-place all-reduces on row-major and channels_last gradient views, the post-accumulate hooks
firing during backward, the two-phase optimizer step, the 4-byte skip-flag reduce and the densification-statistics reduces, all
ordered against the product's kernels on the stream.  A sum over one rank is the identity: every all-reduce is checked to leave real
gradients (row-major and channels_last) bit-identical, and four training steps through the reducers are checked against four plain
steps.  That second comparison cannot be bitwise -- two PLAIN runs of the product already differ after the first optimizer step (the
HexPlane scatter and the weight-gradient flush sum with float atomics) -- so it is held to the run-to-run noise floor, which the
test measures itself.  (Still no scaling curve: that needs more than one GPU.)

Each test runs in a fresh interpreter: the environment variable and the process group must not leak into the rest of the suite."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.update(S3G_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    return env


SCRIPT = r'''
import json, sys, torch
import torch.distributed as dist
from s3gaussian_amd import dp, synth, raster_C
from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, training_step

rank, world, local = dp.init_from_env()
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl" and dp.active()
dev = torch.device("cuda:0")
scn = synth.street_scene(P=30_000, seed=0, width=320, height=208, n_frames=2)
hyper, opt = default_hyper(), default_opt()
H, W = 208, 320
g = torch.Generator().manual_seed(0)
gts = (torch.rand(3, H, W, generator=g).to(dev), (torch.rand(1, H, W, generator=g) * 60).to(dev), torch.rand(3, H, W, generator=g).to(dev))
cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in scn["cameras"][:4]]

def model():
    torch.manual_seed(0)
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    pc.training_setup(opt)
    return pc

out = {}
for mode in ("plain", "plain_again", "rccl_two_phase", "rccl_single_phase"):
    pc = model()
    calls = {"n": 0, "elems": 0, "flag": 0}
    reducer = dp.OverlappedGradAllReducer(pc.optimizer, average=False) if mode.startswith("rccl") else None
    if reducer is not None:
        assert len(reducer._hooks) >= 6          # the big per-Gaussian arrays and planes are hooked
        pc.optimizer.grad_scale = 1.0 / world

    def hook(pc_, pkg):
        g_xy, any_vis, rmax = dp.reduce_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])
        dp.add_densification_stats(pc_.xyz_gradient_accum, pc_.denom, pc_.max_radii2D, g_xy, any_vis, rmax)
        calls["n"] += 1
        calls["flag"] += int(raster_C.async_skip_flag(dev) is not None)

    def opt_step():
        if mode == "rccl_two_phase":
            calls["elems"] += reducer.finish_and_step(pc.optimizer)
        else:
            calls["elems"] += reducer.finish()
            pc.optimizer.step()

    losses = []
    for it in range(4):
        loss, pkg = training_step(pc, cams[it % 4], *gts, hyper, opt, scn["bg"].to(dev), stage="fine",
                                  grad_hook=hook if reducer is not None else None, densify_stats=(reducer is None),
                                  optimizer_step=opt_step if reducer is not None else None)
        losses.append(loss)
    torch.cuda.synchronize()
    out[mode] = dict(losses=[float(x) for x in losses], params={n: p.detach().clone() for n, p in pc.named_parameters()},
                     accum=pc.xyz_gradient_accum.clone(), denom=pc.denom.clone(), radii=pc.max_radii2D.clone(), calls=dict(calls))
    if reducer is not None:
        reducer.remove_hooks()

res = {"rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()), "backend": dist.get_backend()}
n_params = sum(p.numel() for p in out["plain"]["params"].values())

def distance(a, b):
    """Two runs of the product are NOT bit-identical beyond the first step even without any collective: the HexPlane scatter and the
    weight-gradient flush sum with float atomics (order varies run to run), and Adam's first steps turn a last-bit difference of
    a near-zero gradient into +-lr.  So: first loss exact, later losses relative, parameters by the fraction of elements that are
    not close (the measure tests/test_patch_gpu.py uses), and the same distance between two PLAIN runs printed beside it."""
    frac = max(float((~torch.isclose(a["params"][n], b["params"][n], rtol=1e-4, atol=1e-6)).float().mean()) for n in a["params"])
    return dict(first_loss_equal=a["losses"][0] == b["losses"][0],
                loss_max_rel=max(abs(x - y) / abs(x) for x, y in zip(a["losses"], b["losses"])), params_not_close_frac=frac,
                denom_equal=bool(torch.equal(a["denom"], b["denom"])), radii_equal=bool(torch.equal(a["radii"], b["radii"])),
                accum_max_rel=float(((a["accum"] - b["accum"]).abs() / a["accum"].abs().clamp_min(1e-9)).max()))

res["noise_floor_plain_vs_plain"] = distance(out["plain"], out["plain_again"])
for mode in ("rccl_two_phase", "rccl_single_phase"):
    b = out[mode]
    res[mode] = dict(distance(out["plain"], b), hook_calls=b["calls"]["n"], elems_reduced_per_step=b["calls"]["elems"] / 4,
                     flag_seen=b["calls"]["flag"], n_param_elems=n_params)
# the collectives themselves ARE the identity, bit for bit, on real gradients in both memory layouts the reducers touch
pc = model()
ident = True
from s3gaussian_amd.pipeline import render, training_loss
from types import SimpleNamespace
pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
pkg = render(cams[1], pc, pipe, scn["bg"].to(dev), stage="fine", return_dx=True, render_feat=True)
training_loss(pc, pkg, *gts, hyper, opt, "fine").backward()
layouts = set()
for n, p in pc.named_parameters():
    if p.grad is None:
        continue
    g0 = p.grad.clone(memory_format=torch.preserve_format)
    v = dp._flat_view(p.grad)
    assert v is not None, n
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    ident = ident and bool(torch.equal(p.grad, g0))
    layouts.add("channels_last" if (p.dim() == 4 and not p.grad.is_contiguous()) else "row_major")
res["allreduce_identity"] = ident
res["layouts"] = sorted(layouts)
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''


def test_one_rank_rccl_group_runs_the_data_parallel_step(gpu_device):
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["backend"] == "nccl" and res["rccl_version"].count(".") >= 1
    assert res["allreduce_identity"] and res["layouts"] == ["channels_last", "row_major"]      # in-place RCCL all-reduce == identity, bitwise
    floor = res["noise_floor_plain_vs_plain"]
    print("noise floor, two plain runs:", floor)
    for mode in ("rccl_two_phase", "rccl_single_phase"):
        m = res[mode]
        print(mode, m)
        assert m["first_loss_equal"] and m["loss_max_rel"] <= max(1e-4, 3 * floor["loss_max_rel"]), (mode, m, floor)
        assert m["params_not_close_frac"] <= max(1e-3, 3 * floor["params_not_close_frac"]), (mode, m, floor)
        assert m["denom_equal"] and m["radii_equal"] and m["accum_max_rel"] <= max(1e-4, 3 * floor["accum_max_rel"]), (mode, m, floor)
        assert m["hook_calls"] == 4 and m["flag_seen"] == 4                 # the skip word existed and was all-reduced every step
        # every gradient of the step went through a collective (the unused heads have none): >= 95 % of the parameter elements
        assert m["elems_reduced_per_step"] >= 0.95 * m["n_param_elems"] - 20_000, (mode, m)


def test_bench_under_the_forced_group_prints_a_comm_block_from_executed_collectives(gpu_device):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--P", "60000", "--width", "480",
           "--height", "320", "--frames", "4", "--sustain-steps", "0"]
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cm = d["comm"]
    assert d["n_gpus"] == 1 and cm["backend"] == "nccl" and cm["world_size"] == 1 and cm["forced_single_rank_group"] is True
    assert cm["rccl_version"] and cm["bytes_reduced_per_step_per_rank"] > 4 * 59 * 60000
    assert cm["optimizer_step"].startswith("two phases") and cm["scaling_curve_measured_by_the_builder"] is False
    assert "cpu_baseline" not in d and "paths" not in d["config"]
