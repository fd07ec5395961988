#!/usr/bin/env python
"""bench.py -- headline benchmark of the S3Gaussian hot path on MI355X.

One "step" = one training iteration of the reference (train.py:291-522, fine stage, batch of ONE view per rank):
HexPlane sample -> deformation MLP -> activations + SH->RGB -> rasterize RGB+depth -> rasterize feature image ->
L1 + DSSIM + depth-L2 + feat-L2 + dx/dshs + plane regularisers -> backward -> (RCCL grad all-reduce) -> Adam step.

Workload at N=1 = BASELINE.json configs[2] (the configuration the metric is quoted on): 1.2 M Gaussians, 1066x1600,
3 cameras x 50 frames, depth + RGB, hexplane + deformation ON; synthetic street scene, random-init
network of the reference architecture.  N>1: every rank holds a replica and takes a different view per step (weak
scaling, `value` = views processed by all ranks per second).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32), the dtype the MLP computes in


def build_scene(P, width, height, n_frames, device, seed=0):
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt
    sc = synth.street_scene(P=P, seed=seed, width=width, height=height, n_frames=n_frames)
    hyper, opt = default_hyper(), default_opt()
    torch.manual_seed(seed)
    pc = GaussianParams(sc["sh_degree"], hyper)
    gs = sc["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], device)
    pc._deformation.deformation_net.set_aabb(*sc["aabb"])
    pc.training_setup(opt)
    cams = []
    for c in sc["cameras"]:
        c = dict(c)
        for k in ("viewmatrix", "projmatrix", "campos"):
            c[k] = c[k].to(device)
        cams.append(c)
    return pc, cams, hyper, opt, sc["bg"].to(device)


@torch.no_grad()
def make_targets(pc, cam, bg, hyper, seed):
    """GT image / lidar-like depth / feature map = render of a perturbed copy of the scene (non-trivial losses)."""
    from types import SimpleNamespace
    from s3gaussian_amd.pipeline import render
    g = torch.Generator(device="cpu").manual_seed(seed)
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    xyz0 = pc._xyz.data.clone()
    pc._xyz.data.add_(0.01 * torch.randn(xyz0.shape, generator=g).to(xyz0.device))
    pkg = render(cam, pc, pipe, bg, stage="fine", render_feat=True)
    pc._xyz.data.copy_(xyz0)
    return pkg["render"].clamp(0, 1).clone(), pkg["depth"].clone(), pkg["feat"].clone()


def cpu_baseline(P_full, width, height, sample_div=120, seed=0):
    """The same iteration on the host cores through the ORACLE (oracle/: plain-PyTorch restatement of the reference's
    hexplane+MLP+glue+losses, C restatement of the tile rasterizer, OpenMP), on a bounded sample: P_full/sample_div
    Gaussians at the full image size, time scaled linearly in P."""
    import numpy as np
    from oracle import hexplane_ref as hr
    from oracle.oracle import RasterOracle
    from s3gaussian_amd import synth
    P = max(1000, P_full // sample_div)
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 64))
    sc = synth.street_scene(P=P, seed=seed, width=width, height=height, n_frames=2)
    gs, cam = sc["gaussians"], sc["cameras"][0]
    hyper = hr.default_hyper()
    torch.manual_seed(seed)
    net = hr.deform_network(hyper)
    net.deformation_net.grid.set_aabb(*sc["aabb"])
    orc = RasterOracle(np.float32)
    leaves = {k: v.clone().requires_grad_(True) for k, v in
              dict(xyz=gs["xyz"], sc=gs["log_scales"], rot=gs["rotations_raw"], op=gs["opacity_logit"], shs=gs["shs"]).items()}
    H, W = cam["image_height"], cam["image_width"]
    gt = torch.rand(3, H, W)
    gtd = torch.rand(1, H, W) * 60
    gtf = torch.rand(3, H, W)
    kw = dict(bg=np.zeros(3, np.float32), viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
              campos=cam["campos"].numpy(), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W)

    def one_iter():
        for v in leaves.values():
            v.grad = None
        net.zero_grad(set_to_none=True)
        time_t = torch.full((P, 1), cam["time"])
        m3, s, r, o, shs, dx, feat, dshs = net(leaves["xyz"], leaves["sc"], leaves["rot"], leaves["op"], leaves["shs"], time_t)
        scales, rots, opac = torch.exp(s), torch.nn.functional.normalize(r), torch.sigmoid(o)
        cols = hr.shs_to_colors(3, shs, leaves["xyz"], cam["campos"])
        outs, fwd = [], []
        for c in (cols, feat):
            f = orc.forward(means3D=m3.detach().numpy(), opacities=opac.detach().numpy(), scales=scales.detach().numpy(),
                            rotations=rots.detach().numpy(), colors_precomp=c.detach().numpy(), sh_degree=0, **kw)
            fwd.append(f)
            outs.append((torch.from_numpy(f["color"]).requires_grad_(True), torch.from_numpy(f["depth"]).requires_grad_(True)))
        (img, dep), (fimg, _) = outs
        loss = (hr.l1_loss(img[None], gt[None]) + 0.2 * (1 - hr.ssim(img[None], gt[None])) + 0.5 * hr.depth_l2(dep, gtd)
                + 0.001 * hr.l2_loss(fimg, gtf))
        loss.backward()
        g1 = orc.backward(fwd[0], img.grad.numpy(), dep.grad.numpy())
        g2 = orc.backward(fwd[1], fimg.grad.numpy(), np.zeros((1, H, W), np.float32))
        t = torch.from_numpy
        surrogate = ((m3 * t(g1["dL_dmeans3D"] + g2["dL_dmeans3D"])).sum() + (scales * t(g1["dL_dscales"] + g2["dL_dscales"])).sum()
                     + (rots * t(g1["dL_drotations"] + g2["dL_drotations"])).sum() + (opac * t(g1["dL_dopacity"] + g2["dL_dopacity"])).sum()
                     + (cols * t(g1["dL_dcolors"])).sum() + (feat * t(g2["dL_dcolors"])).sum()
                     + 0.001 * dx.abs().mean() + 0.001 * dshs.abs().mean()
                     + hr.plane_regulation(net.deformation_net.grid.grids, 0.01, 0.0001, 0.0001))
        surrogate.backward()

    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < 8.0 and n < 10):
        one_iter()
        n += 1
    per_iter_sample = (time.perf_counter() - t0) / n
    est_full = per_iter_sample * (P_full / P)
    return {"value": 1.0 / est_full, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"{n} iterations of the oracle path (plain-PyTorch hexplane+MLP+losses, C/OpenMP tile rasterizer fwd+bwd x2) "
                      f"on {P} of {P_full} Gaussians at {width}x{height}: {per_iter_sample:.2f} s/iter, scaled linearly in P"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--P", type=int, default=1_200_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1066)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    from s3gaussian_amd import _lib, dp
    from s3gaussian_amd.pipeline import training_step
    rank, world, local = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU; the modulo only matters for functional tests that oversubscribe one GPU (S3G_DIST_BACKEND=gloo)
    device = torch.device("cuda", local % torch.cuda.device_count() if world > 1 else 0)
    torch.cuda.set_device(device)
    import ctypes as C
    L = _lib.lib()
    L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]

    pc, cams, hyper, opt, bg = build_scene(a.P, a.width, a.height, a.frames, device)
    my_views = dp.shard_views(len(cams), rank, world, seed=0)
    n_needed = a.steps + a.warmup
    views = [my_views[i % len(my_views)] for i in range(n_needed)]
    uniq = sorted(set(views))
    targets = {}
    for v in uniq[:min(len(uniq), 12)]:          # bounded target cache (each is 4 images of 1066x1600)
        targets[v] = make_targets(pc, cams[v], bg, hyper, seed=1000 + v)
    tkeys = list(targets)
    # large gradients are all-reduced as soon as backward produces them (overlaps the rest of the backward pass)
    reducer = dp.OverlappedGradAllReducer(pc.parameters(), average=False) if world > 1 else None
    if world > 1:
        pc.optimizer.grad_scale = 1.0 / world   # the SUM all-reduce is averaged inside the Adam kernel

    def hook(pc_, pkg):
        if reducer is not None:
            reducer()
            dp.reduce_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])

    visible, instances = [], []

    def step(i):
        v = views[i]
        gt_img, gt_depth, gt_feat = targets[v] if v in targets else targets[tkeys[i % len(tkeys)]]
        loss, pkg = training_step(pc, cams[v], gt_img, gt_depth, gt_feat, hyper, opt, bg, stage="fine", grad_hook=hook)
        return loss, pkg

    for i in range(a.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    for i in range(9):
        L.s3g_profile_read(i, None, None, None)
    L.s3g_profile_enable(1)
    vis_acc = torch.zeros((), device=device, dtype=torch.float64)
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        loss, pkg = step(i)
        vis_acc += pkg["visibility_filter"].sum()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    L.s3g_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # render ms/frame (the second half of BASELINE's metric): one render(stage="fine") under no_grad, SURVEY.md 3.5
    from types import SimpleNamespace
    from s3gaussian_amd.pipeline import render as render_fn
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        for i in range(3):
            render_fn(cams[views[i % len(views)]], pc, pipe, bg, stage="fine")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_frames = 20
        for i in range(n_frames):
            render_fn(cams[views[i % len(views)]], pc, pipe, bg, stage="fine")
        torch.cuda.synchronize()
        render_ms = 1000.0 * (time.perf_counter() - t1) / n_frames

    if rank == 0:
        # ---- roofline leg: every hot kernel timed in-library with hipEvents on the launch stream (include/s3g_raster.h),
        # priced against its ALGORITHMIC bytes / flops (DESIGN.md section 7 states each model) -----------------------------
        V = float(vis_acc.item()) / max(a.steps, 1)    # mean visible Gaussians per step (both raster calls share them)
        P = float(a.P)
        net = pc._deformation.deformation_net
        planes = [p for p in net.grid.grids.parameters()] if hasattr(net.grid, "grids") else []
        plane_bytes = float(sum(p.numel() for p in planes) * 4)
        levels = float(len(getattr(net.grid, "grids", [])) or 4)
        FEAT = 32.0 * levels
        MLP_FLOP = 2.0 * (128 * 64 + 4 * 64 * 64 + 2 * 64 * 3 + 64 * 48)   # per point and direction: 56064
        pair = bool(hyper.feat_head)   # pipeline.render blends both images of an iteration in one pass each way

        def read(i):
            ms, x, y = C.c_double(), C.c_double(), C.c_double()
            n = L.s3g_profile_read(i, C.byref(ms), C.byref(x), C.byref(y))
            return (n, ms.value / n, x.value / n, y.value / n) if n else (0, 0.0, 0.0, 0.0)

        # id -> (kernel, bytes(R or P, pixels), flops)
        models = {
            # two-image pass (RGB+depth and feature image from one geometry): + colors2 per instance, + one image per pixel
            0: ("s3g::blend_forward_kernel", lambda R, N: (56.0 if pair else 44.0) * R + (36.0 if pair else 24.0) * N, None),
            1: ("s3g::blend_backward_kernel", lambda R, N: (56.0 if pair else 44.0) * R + (36.0 if pair else 24.0) * N + 40.0 * V, None),
            2: ("s3g::hexplane_forward_kernel", lambda n, l: n * (16.0 + 4.0 * FEAT) + plane_bytes, None),
            3: ("s3g::hexplane_backward_point_kernel", lambda n, l: n * (28.0 + 4.0 * FEAT + 6.0 * l * 128.0) + plane_bytes, None),
            4: ("s3g::hexplane_scatter_kernel", lambda n, l: n * (60.0 + 6.0 * l * 128.0) + plane_bytes, None),
            5: ("s3g::mlp_forward_kernel", lambda n, _: n * (512.0 + 5 * 256.0 + 216.0), lambda n: n * MLP_FLOP),
            6: ("s3g::mlp_backward_kernel", lambda n, _: n * (5 * 256.0 + 216.0 + 5 * 256.0 + 512.0), lambda n: n * MLP_FLOP),
            7: ("s3g::mlp_wgrad_kernel (9 launches)", lambda n, _: n * 4.0 * (2 * 67 + 6 * 128 + 112), lambda n: n * MLP_FLOP),
            8: ("s3g::adam_kernel", lambda n, _: n * 28.0, None),   # p, g, m, v read; p, m, v written
        }
        traffic_db = {}
        default_workload = (a.P, a.width, a.height, a.frames) == (1_200_000, 1600, 1066, 50)
        pmc = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        if os.path.exists(pmc) and default_workload:   # the PMC passes were collected on the default workload only
            try:
                traffic_db = json.load(open(pmc)).get("hbm_bytes_per_launch", {})
            except Exception:
                traffic_db = {}
        kernels = []
        for i, (name, fbytes, fflops) in models.items():
            n, avg_ms, x, y = read(i)
            if not n:
                continue
            nbytes = fbytes(x, y)
            t = avg_ms * 1e-3
            gbs = nbytes / t / 1e9
            ent = {"kernel": name, "launches_per_step": round(n / a.steps, 2), "avg_launch_ms": round(avg_ms, 4),
                   "algorithmic_bytes_per_launch": round(nbytes), "hbm_GBps": round(gbs, 1),
                   "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)}
            bound, frac = "hbm", gbs / PEAK_HBM_GBS
            if fflops is not None:
                tf = fflops(x) / t / 1e12
                ent.update({"flops_per_launch": round(fflops(x)), "mfma_TFLOPs": round(tf, 2),
                            "mfma_frac": round(tf / PEAK_MFMA_F32_TFLOPS, 4)})
                if tf / PEAK_MFMA_F32_TFLOPS > frac:   # the roof this kernel sits closer to
                    bound, frac = "mfma", tf / PEAK_MFMA_F32_TFLOPS
            ent["bound"] = bound
            ent["frac"] = round(frac, 4)
            ent["ms_per_step"] = round(avg_ms * n / a.steps, 4)
            base = name.split(" ")[0]
            if base in traffic_db:   # PMC bytes per launch; the wgrad entry brackets nine launches
                ent["traffic"] = traffic_db[base] * (9 if i == 7 else 1)
            kernels.append(ent)
        roof = None
        if kernels:
            dom = max(kernels, key=lambda e: e["ms_per_step"])   # the kernel the step spends most time in
            if dom["bound"] == "mfma":
                roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["mfma_TFLOPs"], "peak": PEAK_MFMA_F32_TFLOPS,
                        "unit": "TFLOP/s", "frac": dom["mfma_frac"], "traffic": dom.get("traffic")}
            else:
                roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["hbm_GBps"], "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": dom["hbm_frac"], "traffic": dom.get("traffic")}
            roof.update({"avg_launch_ms": dom["avg_launch_ms"], "launches_per_step": dom["launches_per_step"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "kernels": kernels})
        fwd = next((k for k in kernels if k["kernel"] == "s3g::blend_forward_kernel"), None)
        out = {
            "metric": "train_iters_per_sec", "value": round(world * a.steps / dt, 3), "unit": "iters/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE cfg3: {a.P} Gaussians, {a.height}x{a.width}, 3 cams x {a.frames} frames, fine stage "
                                   "(hexplane+deformation ON), RGB+depth render + feature render, L1+DSSIM+depthL2+featL2+regs, Adam",
                       "gaussians": a.P, "image": [a.height, a.width], "views_per_step_per_rank": 1,
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU",
                       "blend_forward_avg_ms": fwd["avg_launch_ms"] if fwd else None,
                       "render_ms_per_frame": round(render_ms, 3)},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.P, a.width, a.height)
            except Exception as ex:  # the baseline must never take the headline number down with it
                out["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(ex).__name__}: {ex}"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
