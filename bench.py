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

Prints ONE JSON line (rank 0).  `value` is the FUSED path (every edit of INTEGRATION.md section 4/4b applied:
`config.path = "fused"`); at N=1 two more paths are timed for the record (`config.paths`): "zero_diff" = the reference's
own files on the drop-in packages with no edit (our rasterizer called twice + geometry cache, everything else plain
PyTorch: 24 grid_samples, nn.Linear stack, torch glue, conv2d SSIM, torch.optim.Adam) and "import_swap" = INTEGRATION
section 4's one-line import swap only (fused HexPlane + MLP, the rest as zero_diff).

`roofline`: hipEvent times are measured live in this run; `algorithmic_bytes_per_launch` is the STRICT model of SURVEY 8(d)
(inputs and outputs the math needs -- scratch this implementation chose to write, e.g. the HexPlane backward's G slab or the
MLP's activation stash, is `implementation_bytes_per_launch`, never algorithmic); `frac` is priced on the strict figure.
`traffic` / the VALU instruction counts: collected BY THIS RUN when `rocprofv3` is on the box (round 6) -- three extra passes of a
3-step child (`--pmc-child`) under `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU --kernel-trace` after the timed loops --
else read from profiles/kernel_traffic.json (an earlier collection of the same passes); `traffic_source` says which.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32): the exact chain and the weight-gradient GEMMs
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA; the bf16x3 chains issue SIX bf16 piece products per fp32 product
VALU_CLOCK_GHZ = 2.4  # MI355X_MICROARCH.md: peak engine clock; 256 CUs x 4 SIMDs, a wave64 VALU instruction occupies a SIMD for 4 cycles


def build_scene(P, width, height, n_frames, device, seed=0, scale_mult=1.0):
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt
    sc = synth.street_scene(P=P, seed=seed, width=width, height=height, n_frames=n_frames)
    hyper, opt = default_hyper(), default_opt()
    torch.manual_seed(seed)
    pc = GaussianParams(sc["sh_degree"], hyper)
    gs = sc["gaussians"]
    import math
    pc.init_from_tensors(gs["xyz"], gs["log_scales"] + math.log(scale_mult), gs["rotations_raw"], gs["opacity_logit"], gs["shs"],
                         device)
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
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.pipeline import render
    prev_async = raster_C.set_async(False)   # targets must be right whatever the capacity policy has seen so far: synchronous forward
    g = torch.Generator(device="cpu").manual_seed(seed)
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    xyz0 = pc._xyz.data.clone()
    pc._xyz.data.add_(0.01 * torch.randn(xyz0.shape, generator=g).to(xyz0.device))
    pkg = render(cam, pc, pipe, bg, stage="fine", render_feat=True)
    pc._xyz.data.copy_(xyz0)
    raster_C.invalidate_geometry_cache()     # writes through .data bump no version counter (geometry + inference-deformation caches)
    raster_C.set_async(prev_async)
    return pkg["render"].clamp(0, 1).clone(), pkg["depth"].clone(), pkg["feat"].clone()


def _oracle_iteration_factory(P, width, height, seed, point_splat, ref=None):
    """One training iteration on the host cores at P Gaussians and the full image.  point_splat=False: the oracle path
    (plain-PyTorch restatement of the reference's hexplane+MLP+glue+losses, C/OpenMP restatement of the tile rasterizer,
    fwd+bwd x2).  point_splat=True: the north_star's baseline -- the same PyTorch hexplane+MLP+glue+losses with the
    rasterizer stubbed to a point splat (each Gaussian -> its nearest pixel, depth-sorted alpha = opacity compositing)."""
    import numpy as np
    from types import SimpleNamespace
    from oracle import hexplane_ref as hr
    from oracle.oracle import RasterOracle
    from s3gaussian_amd import synth
    sc = synth.street_scene(P=P, seed=seed, width=width, height=height, n_frames=2)
    gs, cam = sc["gaussians"], sc["cameras"][0]
    torch.manual_seed(seed)
    if ref is not None:
        # the reference's OWN modules on the host cores (oracle/_ref/reference_py.tar.gz: scene/deformation.py + scene/hexplane.py,
        # utils/sh_utils.py::eval_sh, utils/loss_utils.py, GaussianModel.compute_regulation) -- `kind: "reference"`
        from oracle import ref_py
        _, _, ref_hyper, _, _ = ref_py.default_arguments(ref)
        gm = ref.gaussian_model.GaussianModel(3, ref_hyper)          # constructor only: nothing is moved to a GPU
        net = gm._deformation
        net.deformation_net.set_aabb(*sc["aabb"])
        LU = ref.loss_utils

        def _colors(deg, shs, xyz, campos):                           # gaussian_renderer/__init__.py:104-110 on the reference's eval_sh
            d = xyz - campos.repeat(xyz.shape[0], 1)
            return torch.clamp_min(ref.sh_utils.eval_sh(deg, shs.transpose(1, 2).view(-1, 3, 16), d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)

        hr = SimpleNamespace(shs_to_colors=_colors, l1_loss=LU.l1_loss, ssim=LU.ssim, l2_loss=LU.l2_loss,
                             depth_l2=lambda a, b: LU.compute_depth("l2", a, b),
                             plane_regulation=lambda grids, tw, l1w, pw: gm.compute_regulation(tw, l1w, pw))
    else:
        net = hr.deform_network(hr.default_hyper())
        net.deformation_net.grid.set_aabb(*sc["aabb"])
    orc = None if point_splat else RasterOracle(np.float32)
    leaves = {k: v.clone().requires_grad_(True) for k, v in
              dict(xyz=gs["xyz"], sc=gs["log_scales"], rot=gs["rotations_raw"], op=gs["opacity_logit"], shs=gs["shs"]).items()}
    H, W = cam["image_height"], cam["image_width"]
    gt, gtd, gtf = torch.rand(3, H, W), torch.rand(1, H, W) * 60, torch.rand(3, H, W)
    kw = dict(bg=np.zeros(3, np.float32), viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
              campos=cam["campos"].numpy(), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W)
    view, proj = cam["viewmatrix"], cam["projmatrix"]

    def splat(m3, opac, cols, feat):
        """Point-splat stub: nearest pixel, front-to-back alpha compositing per pixel with alpha = opacity (segmented
        exclusive cumprod of (1 - alpha) over the depth-sorted Gaussians of each pixel).  Differentiable in colour,
        feature and opacity; positions only pick the pixel."""
        with torch.no_grad():
            hom = torch.cat([m3, torch.ones(P, 1)], 1)
            pv, pp = hom @ view, hom @ proj
            z = pv[:, 2]
            ndc = pp[:, :2] / (pp[:, 3:4] + 1e-7)
            px = (((ndc[:, 0] + 1) * W - 1) * 0.5).round().long()
            py = (((ndc[:, 1] + 1) * H - 1) * 0.5).round().long()
            ok = (z > 0.2) & (px >= 0) & (px < W) & (py >= 0) & (py < H)
            idx = ok.nonzero(as_tuple=True)[0]
            pix = py[idx] * W + px[idx]
            order = torch.argsort(pix.double() * 4096.0 + z[idx].double().clamp(0, 4000.0))   # by pixel, then by depth
            idx, pix = idx[order], pix[order]
            first = torch.ones_like(pix, dtype=torch.bool)
            first[1:] = pix[1:] != pix[:-1]
            seg = torch.cumsum(first.long(), 0) - 1
        a = opac[idx, 0].clamp(max=0.99)
        logt = torch.log1p(-a)
        cs = torch.cumsum(logt, 0)
        seg_start = (cs - logt)[first]                       # exclusive cumsum at each segment head
        T = torch.exp(cs - logt - seg_start[seg])            # transmittance in front of each Gaussian
        wgt = (a * T)[:, None]
        img = torch.zeros(H * W, 3).index_add(0, pix, wgt * cols[idx]).t().reshape(3, H, W)
        fimg = torch.zeros(H * W, 3).index_add(0, pix, wgt * feat[idx]).t().reshape(3, H, W)
        dep = torch.zeros(H * W).index_add(0, pix, wgt[:, 0] * z[idx]).reshape(1, H, W)
        return img, dep, fimg

    def one_iter():
        for v in leaves.values():
            v.grad = None
        net.zero_grad(set_to_none=True)
        time_t = torch.full((P, 1), cam["time"])
        m3, s, r, o, shs, dx, feat, dshs = net(leaves["xyz"], leaves["sc"], leaves["rot"], leaves["op"], leaves["shs"], time_t)
        scales, rots, opac = torch.exp(s), torch.nn.functional.normalize(r), torch.sigmoid(o)
        cols = hr.shs_to_colors(3, shs, leaves["xyz"], cam["campos"])
        regs = (0.001 * dx.abs().mean() + 0.001 * dshs.abs().mean()
                + hr.plane_regulation(net.deformation_net.grid.grids, 0.01, 0.0001, 0.0001))
        if point_splat:
            img, dep, fimg = splat(m3, opac, cols, feat)
            loss = (hr.l1_loss(img[None], gt[None]) + 0.2 * (1 - hr.ssim(img[None], gt[None])) + 0.5 * hr.depth_l2(dep, gtd)
                    + 0.001 * hr.l2_loss(fimg, gtf) + regs)
            loss.backward()
            return
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
                     + (cols * t(g1["dL_dcolors"])).sum() + (feat * t(g2["dL_dcolors"])).sum() + regs)
        surrogate.backward()

    return one_iter


def _time_iteration(P, width, height, seed, point_splat, repeats=1, warm=True, ref=None):
    it = _oracle_iteration_factory(P, width, height, seed, point_splat, ref=ref)
    if warm:
        it()                                   # warm-up (allocator, OpenMP team, lazy inits)
    t0 = time.perf_counter()
    for _ in range(repeats):
        it()
    return (time.perf_counter() - t0) / repeats


def _fit(P_full, width, height, seed, point_splat, sizes):
    """Least-squares t(P) = a + b * P over `sizes` samples at the FULL image: `a` carries everything that scales with the pixels
    (tile walk / splat image, SSIM's five convolutions, pixel losses, the 143 MB of plane regularisers), `b` everything that scales
    with the Gaussians (HexPlane gathers, MLP, glue, per-Gaussian raster work, blending work per instance).  The residuals of the
    fit at the sampled sizes are reported next to the estimate for P_full."""
    import numpy as np
    times = [_time_iteration(P, width, height, seed, point_splat, repeats=2 if P <= 40_000 else 1) for P in sizes]
    A = np.stack([np.ones(len(sizes)), np.asarray(sizes, np.float64)], 1)
    (a_, b), *_ = np.linalg.lstsq(A, np.asarray(times, np.float64), rcond=None)
    a_, b = max(float(a_), 0.0), max(float(b), 0.0)
    est = a_ + b * P_full
    return est, {"sample_P": list(sizes), "sample_s_per_iter": [round(x, 3) for x in times], "pixel_term_s": round(a_, 3),
                 "per_gaussian_term_us": round(b * 1e6, 3), "estimate_s_per_iter": round(est, 2),
                 "fit_residual_rel": [round((a_ + b * P - t_) / t_, 4) for P, t_ in zip(sizes, times)]}


def cpu_baseline(P_full, width, height, seed=0):
    """Reported baseline, not the target: the same fine-stage iteration on the host cores of this box.
      top level ("port", what = "point_splat"): BASELINE.json north_star's baseline -- the reference-architecture PyTorch
          hexplane + MLP + glue + losses (oracle/hexplane_ref.py) with the rasterizer stubbed to a nearest-pixel point splat.
          ONE real iteration at the FULL workload (P_full Gaussians, full image), measured, nothing extrapolated
          (a warm-up iteration at P/40 first: allocator, thread pools);
      "tile_rasterizer_port": the oracle path proper (same PyTorch front end, C/OpenMP restatement of the tile rasterizer
          fwd+bwd x2) sampled at THREE sizes of P up to P_full/4 at the full image, least-squares t = a + b*P with the residuals."""
    cores = os.cpu_count() or 1
    ref = None
    try:
        from oracle import ref_py
        if ref_py.available():
            ref = ref_py.load(patch=False)
    except Exception:
        ref = None
    # BASELINE.md section 4 plans "all host cores"; whether torch's intra-op pools pay beyond ~64 threads for these element-wise /
    # gather ops is a property of the box, so it is MEASURED here: the same P/40 iteration at every candidate
    # thread count, the full-size iteration at the fastest; all timings go into the line
    # -- ASCENDING, and stopping at the first count that is slower than its predecessor: on the 256-core host of this pool the same
    # iteration took 0.76 s with 32 threads, 1.41 s with 64, 3.9 s with 128 and 157 s (!) with all 256 (profiles/r06_bench_line_thread_probe.json:
    # oversubscribed intra-op pools on tensors this small), so probing downwards from "all cores" costs minutes.
    probe_P, probe = max(2000, P_full // 40), {}
    try:
        torch.set_num_threads(min(cores, 16))
        _time_iteration(probe_P, width, height, seed, True, repeats=1, warm=False, ref=ref)       # allocator, lazy inits
        for n in sorted({min(cores, 16), min(cores, 32), min(cores, 64), min(cores, 128), cores}):
            torch.set_num_threads(n)
            probe[n] = _time_iteration(probe_P, width, height, seed, True, repeats=1, warm=True, ref=ref)
            if len(probe) > 1 and probe[n] > sorted(probe.items())[-2][1]:
                break
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        t_full = _time_iteration(P_full, width, height, seed, True, repeats=1, warm=False, ref=ref)
    finally:
        if ref is not None:
            ref_py.unload()
    front = ("the reference's OWN scene/deformation.py + scene/hexplane.py + utils/sh_utils.py + utils/loss_utils.py + "
             "GaussianModel.compute_regulation (oracle/_ref/reference_py.tar.gz)" if ref is not None else
             "reference-architecture PyTorch hexplane+MLP+glue+losses (oracle/hexplane_ref.py)")
    out = {"value": round(1.0 / t_full, 5), "unit": "iters/s", "cores": threads, "host_cores": cores,
           "kind": "reference" if ref is not None else "port", "what": "point_splat",
           "sample": f"measured, not extrapolated: ONE full iteration of the workload itself ({P_full} Gaussians, {width}x{height}) in "
                     f"{t_full:.2f} s -- {front}, rasterizer stubbed to a nearest-pixel "
                     f"point splat (torch threads {threads} of {cores} cores: the fastest of {sorted(probe)} on a {probe_P}-Gaussian "
                     f"iteration, s/iter {[round(probe[k], 2) for k in sorted(probe)]}; the probe goes up in thread count and stops at the "
                     f"first count that is slower than the one before it)",
           "thread_probe_s_per_iter": {str(k): round(v, 3) for k, v in sorted(probe.items())}}
    try:
        sizes = (max(2000, P_full // 120), max(6000, P_full // 40), max(20000, P_full // 4))
        est, model = _fit(P_full, width, height, seed, False, sizes)
        out["tile_rasterizer_port"] = {
            "value": round(1.0 / est, 5), "unit": "iters/s", "cores": cores, "kind": "port", "model": model,
            "sample": f"oracle path (PyTorch front end + C/OpenMP tile rasterizer fwd+bwd x2) at P = {model['sample_P']} of {P_full} "
                      f"Gaussians, full image: {model['sample_s_per_iter']} s/iter; least squares t(P) = {model['pixel_term_s']} s + "
                      f"{model['per_gaussian_term_us']} us * P -> {model['estimate_s_per_iter']} s/iter at full size, residuals "
                      f"{model['fit_residual_rel']}"}
    except Exception as ex:
        out["tile_rasterizer_port"] = {"value": None, "kind": "port", "sample": f"failed: {type(ex).__name__}: {ex}"}
    return out


# ---- the two slower call paths, timed for the record (N=1 only) -------------------------------------------------------
def _torch_hexplane(grid, xyz, time_col):
    """scene/hexplane.py:73-106,151-175 as the reference runs it: 24 F.grid_sample launches + products + concat."""
    import itertools
    import torch.nn.functional as F
    aabb = grid.aabb
    pts = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    pts = torch.cat([pts, time_col], dim=-1)
    combs = list(itertools.combinations(range(4), 2))
    feats = []
    for planes in grid.grids:
        prod = 1.0
        for ci, comb in enumerate(combs):
            coords = pts[:, list(comb)].view(1, -1, 1, 2)
            s = F.grid_sample(planes[ci], coords, align_corners=True, mode="bilinear", padding_mode="border")
            prod = prod * s.view(s.shape[1], -1).t()
        feats.append(prod)
    return torch.cat(feats, dim=-1)


def _alt_step(path, pc, cam, gts, hyper, opt, bg, torch_adam):
    """One fine-stage iteration the way the UNMODIFIED reference files run it on the drop-in packages.
    path "zero_diff":   reference scene/deformation.py + scene/hexplane.py in plain PyTorch (stand-in below: the same
                        grid_sample / nn.Linear ops on this model's parameters), torch glue + eval_sh, TWO rasterizer calls
                        (ours; the second one is served by the geometry cache), torch losses (conv2d SSIM), torch plane
                        regularisers, torch.optim.Adam.
    path "import_swap": INTEGRATION.md section 4 only -- the deformation network is this package's (fused HexPlane sampler
                        + fused MFMA MLP); everything else as zero_diff."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from s3gaussian_amd import pipeline as pl
    gt_image, gt_depth, gt_feat = gts
    P = pc._xyz.shape[0]
    dev = pc._xyz.device
    net = pc._deformation.deformation_net
    time_col = torch.full((P, 1), float(cam["time"]), device=dev)
    screenspace = torch.zeros_like(pc._xyz, requires_grad=True) + 0
    screenspace.retain_grad()
    if path == "zero_diff":
        hidden = net.feature_out(_torch_hexplane(net.grid, pc._xyz, time_col))
        dx = net.pos_deform(hidden)
        dshs = net.shs_deform(hidden).reshape(P, 16, 3)
        feat = net.dino_head(hidden)
        means3D, shs = pc._xyz + dx, pc.get_features + dshs
    else:
        means3D, _, _, _, shs, dx, feat, dshs = pc._deformation(pc._xyz, pc._scaling, pc._rotation, pc._opacity,
                                                                 pc.get_features, time_col)
    scales, rots, opac = torch.exp(pc._scaling), torch.nn.functional.normalize(pc._rotation), torch.sigmoid(pc._opacity)
    shs_view = shs.transpose(1, 2).view(-1, 3, 16)
    d = pc._xyz - cam["campos"].repeat(P, 1)
    colors = torch.clamp_min(pl.eval_sh(3, shs_view, d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    rs = GaussianRasterizationSettings(image_height=int(cam["image_height"]), image_width=int(cam["image_width"]),
                                       tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
                                       viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], sh_degree=3, campos=cam["campos"],
                                       prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    img, radii, depth = rast(means3D=means3D, means2D=screenspace, shs=None, colors_precomp=colors, opacities=opac, scales=scales,
                             rotations=rots, cov3D_precomp=None)
    fimg, _, _ = rast(means3D=means3D, means2D=screenspace, shs=None, colors_precomp=feat, opacities=opac, scales=scales,
                      rotations=rots, cov3D_precomp=None)
    grids = net.grid.grids
    loss = (pl.l1_loss(img[None], gt_image[None, :3]) + opt.lambda_depth * pl.compute_depth_l2(depth[None], gt_depth[None])
            + opt.lambda_dssim * (1.0 - pl.ssim(img[None], gt_image[None])) + opt.lambda_feat * pl.l2_loss(fimg, gt_feat)
            + opt.lambda_dx * dx.abs().mean() + opt.lambda_dshs * dshs.abs().mean()
            + hyper.plane_tv_weight * sum(pl._plane_smoothness(g[i]) for g in grids for i in (0, 1, 3))
            + hyper.time_smoothness_weight * sum(pl._plane_smoothness(g[i]) for g in grids for i in (2, 4, 5))
            + hyper.l1_time_planes * sum(torch.abs(1 - g[i]).mean() for g in grids for i in (2, 4, 5)))
    loss.backward()
    torch_adam.step()
    torch_adam.zero_grad(set_to_none=True)
    return loss.detach()


def camera_object(cam, gts):
    """A stand-in for the reference's Camera (scene/cameras.py:20-70) carrying what train.py / render() read from it."""
    import math
    from types import SimpleNamespace
    gt_image, gt_depth, gt_feat = gts
    return SimpleNamespace(image_height=cam["image_height"], image_width=cam["image_width"], FoVx=2.0 * math.atan(cam["tanfovx"]),
                           FoVy=2.0 * math.atan(cam["tanfovy"]), world_view_transform=cam["viewmatrix"],
                           full_proj_transform=cam["projmatrix"], camera_center=cam["campos"], time=cam["time"],
                           original_image=gt_image, depth_map=gt_depth, feat_map=gt_feat.permute(1, 2, 0))


def patched_reference_step(gaussians, viewpoint_cam, hyper, opt, background, pipe=None, stage="fine"):
    """One iteration of the UNMODIFIED train.py (:372-437, :489-522; batch_size 1, below densify_until_iter) on the names
    `s3gaussian_amd.patch.patch_reference()` rebinds -- what `python -m s3gaussian_amd.patch train.py ...` runs per iteration.
    /root/reference does not exist on the GPU box, so the iteration body is restated here statement by statement: the calls are the
    patch's replacements, everything train.py does inline (loss assembly one `loss +=` at a time, psnr, the NaN check and
    `loss.item()` host syncs, the boolean-mask max_radii2D update) is kept as it is there."""
    from types import SimpleNamespace
    from s3gaussian_amd import patch
    from s3gaussian_amd.pipeline import psnr
    pipe = pipe or SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    args = hyper
    render_pkg = patch.render(viewpoint_cam, gaussians, pipe, background, stage=stage, return_dx=True,
                              render_feat=True if ('fine' in stage and args.feat_head) else False)
    image, viewspace_point_tensor, visibility_filter, radii = (render_pkg["render"], render_pkg["viewspace_points"],
                                                               render_pkg["visibility_filter"], render_pkg["radii"])
    depth_pred_tensor = render_pkg["depth"].unsqueeze(0)
    image_tensor = image.unsqueeze(0)
    gt_image_tensor = viewpoint_cam.original_image.cuda().unsqueeze(0)
    gt_depth_tensor = viewpoint_cam.depth_map.cuda().unsqueeze(0).float()
    radii = radii.unsqueeze(0).max(dim=0).values
    visibility_filter = visibility_filter.unsqueeze(0).any(dim=0)
    Ll1 = patch.l1_loss(image_tensor, gt_image_tensor[:, :3, :, :])
    psnr_ = psnr(image_tensor, gt_image_tensor).mean().double()
    loss = Ll1
    if 'fine' in stage and not args.no_dx and opt.lambda_dx != 0:
        loss += torch.mean(torch.abs(render_pkg['dx'])) * opt.lambda_dx
    if 'fine' in stage and not args.no_dshs and opt.lambda_dshs != 0:
        loss += torch.mean(torch.abs(render_pkg['dshs'])) * opt.lambda_dshs
    if opt.lambda_depth != 0:
        loss += patch.compute_depth("l2", depth_pred_tensor, gt_depth_tensor) * opt.lambda_depth
    if stage == "fine" and hyper.time_smoothness_weight != 0:
        loss += patch.compute_regulation(gaussians, hyper.time_smoothness_weight, hyper.l1_time_planes, hyper.plane_tv_weight)
    if opt.lambda_dssim != 0:
        loss += opt.lambda_dssim * (1.0 - patch.ssim(image_tensor, gt_image_tensor))
    if stage == 'fine' and args.feat_head:
        feat = render_pkg['feat'].to('cuda')
        gt_feat = viewpoint_cam.feat_map.permute(2, 0, 1).to('cuda')
        loss += patch.l2_loss(feat, gt_feat) * opt.lambda_feat
    loss.backward()
    if torch.isnan(loss).any():
        raise RuntimeError("loss is nan")
    viewspace_point_tensor_grad = torch.zeros_like(viewspace_point_tensor) + viewspace_point_tensor.grad
    with torch.no_grad():
        _ = 0.4 * loss.item()                              # train.py:442 (progress bar EMA): a host sync per iteration
        _ = 0.4 * psnr_
        gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter].float())
        patch.add_densification_stats(gaussians, viewspace_point_tensor_grad, visibility_filter)
        gaussians.optimizer.step()
        gaussians.optimizer.zero_grad(set_to_none=True)
    return loss.detach()


def time_alt_paths(pc, cams, views, targets, tkeys, hyper, opt, bg, steps=4, warmup=1):
    """FALLBACK when oracle/_ref/reference_py.tar.gz is absent (see time_reference_paths): restatements of train.py's iteration body.
    -> {"patched": {...}, "import_swap": {...}, "zero_diff": {...}}: ms/step and it/s of the slower call paths on the same scene."""
    groups = [{"params": g["params"], "lr": g["lr"], "name": g.get("name", "")} for g in pc.optimizer.param_groups]
    torch_adam = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    out = {}
    for path in ("patched", "import_swap", "zero_diff"):
        try:
            def one(i):
                v = views[i % len(views)]
                gts = targets[v] if v in targets else targets[tkeys[i % len(tkeys)]]
                if path == "patched":
                    return patched_reference_step(pc, camera_object(cams[v], gts), hyper, opt, bg)
                return _alt_step(path, pc, cams[v], gts, hyper, opt, bg, torch_adam)
            for i in range(warmup):
                one(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(warmup, warmup + steps):
                one(i)
            torch.cuda.synchronize()
            ms = 1000.0 * (time.perf_counter() - t0) / steps
            n = steps * (4 if path == "patched" else 1)
            if path == "patched":     # the fast one: a longer sample
                t0 = time.perf_counter()
                for i in range(warmup, warmup + n):
                    one(i)
                torch.cuda.synchronize()
                ms = 1000.0 * (time.perf_counter() - t0) / n
            out[path] = {"ms_per_step": round(ms, 2), "iters_per_s": round(1000.0 / ms, 2), "steps": n, "executed": "restatement in bench.py"}
        except Exception as ex:   # never take the headline down
            out[path] = {"ms_per_step": None, "error": f"{type(ex).__name__}: {ex}"}
    return out


def time_reference_paths(pc, cams, targets, bg, aabb):
    """The reference's OWN `train.py::scene_reconstruction` (train.py:216-560: update_learning_rate, the view stack, render(), the loss
    assembly one `loss +=` at a time, psnr, the NaN check, `loss.item()`, max_radii2D / add_densification_stats, the optimizer step),
    UNCHANGED, on real `scene.cameras.Camera` objects and a real `scene.gaussian_model.GaussianModel` holding this bench's scene --
    executed from oracle/_ref/reference_py.tar.gz (the reference's files packed by oracle/ref_py.py; /root/reference itself does not
    exist on the GPU box).  The reference's Python is the CALLER here; what is timed underneath is the product (libs3g.so).
      "zero_diff"    nothing but the two drop-in packages: the reference's own plain-PyTorch HexPlane / MLP / glue / SSIM / Adam
      "import_swap"  + `scene.gaussian_model.deform_network` bound to s3gaussian_amd.deformation.deform_network
      "patched"      `s3gaussian_amd.patch.patch_reference()` before train.py is imported, zero file edits
    ms/step = host time between the iteration body's own `timer.pause()` calls (train.py:469) after a warm-up; the body waits for the
    device once per iteration (`loss.item()`), so host time is step time.  No densify / prune event falls into these short runs
    (densify_from_iter = 500); tests/test_reference_py_gpu.py runs the body through such events."""
    from oracle import ref_py
    out = {}
    tk = list(targets)[:6]
    state = {k: v.detach().clone() for k, v in pc._deformation.state_dict().items()}
    gs = dict(xyz=pc._xyz.detach(), log_scales=pc._scaling.detach(), rotations_raw=pc._rotation.detach(),
              opacity_logit=pc._opacity.detach(), shs=torch.cat([pc._features_dc.detach(), pc._features_rest.detach()], dim=1))
    for route, (warm, n) in (("patched", (5, 60)), ("import_swap", (2, 8)), ("zero_diff", (1, 3))):
        gm = None
        try:
            ref = ref_py.load(patch=(route == "patched"))
            if route == "import_swap":
                from s3gaussian_amd import deformation as _deformation
                ref.gaussian_model.deform_network = _deformation.deform_network
            args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
            dataset.render_process = False
            gm = ref_py.make_gaussians(ref, gs, aabb, hyper)
            gm._deformation.load_state_dict(state)
            cam_objs = [ref_py.make_camera(ref, cams[v], targets[v], uid=v) for v in tk]
            timer = ref_py.RecordingTimer(record_locals=False)
            ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cam_objs), dataset, hyper, opt, pipe, warm + n, "fine", timer)
            torch.cuda.synchronize()
            st = timer.stamps
            ms = 1000.0 * (st[-1] - st[warm - 1]) / n
            out[route] = {"ms_per_step": round(ms, 2), "iters_per_s": round(1000.0 / ms, 2), "steps": n,
                          "executed": "the reference's train.py::scene_reconstruction, unchanged (oracle/_ref/reference_py.tar.gz)",
                          "optimizer": type(gm.optimizer).__module__ + "." + type(gm.optimizer).__name__,
                          "deformation": type(gm._deformation).__module__}
        except Exception as ex:   # never take the headline down
            out[route] = {"ms_per_step": None, "error": f"{type(ex).__name__}: {ex}"}
        finally:
            del gm
            ref_py.unload()
            gc.collect()
            torch.cuda.empty_cache()
    return out


def collect_counters_in_run(a, per_pass_timeout_s=170):
    """HBM-side traffic (FETCH_SIZE, WRITE_SIZE: they cannot share a pass) and wave-level VALU instruction counts (SQ_INSTS_VALU) of
    every kernel of a training step, collected BY THIS RUN: three passes of `rocprofv3 --pmc <counter> --kernel-trace` (counters in
    their own runs with the kernel trace only, MI355X_MICROARCH.md / the pool's rule) over a child of this script that builds the same
    scene and issues 2 + 3 training steps (`--pmc-child`); attributed to the training iterations by dispatch order
    (tools/pmc_traffic.py::collect, the code that wrote profiles/kernel_traffic.json in earlier rounds).
    -> (dict like profiles/kernel_traffic.json | None, note)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on this box"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic
    except Exception as ex:
        return None, f"tools/pmc_traffic.py not importable: {ex}"
    base = tempfile.mkdtemp(prefix="s3g_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "2", "--P", str(a.P), "--width", str(a.width),
             "--height", str(a.height), "--frames", str(a.frames), "--scale-mult", str(a.scale_mult)]
    env = dict(os.environ, TMPDIR="/tmp")
    got, t0 = {}, time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(base, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=per_pass_timeout_s)
            if r.returncode != 0:
                return None, f"{counter} pass: rocprofv3 exited with {r.returncode}: {r.stderr.decode('utf-8', 'replace')[-300:]}"
            got[counter] = pmc_traffic.collect(d, counter)
            if not got[counter]:
                return None, f"{counter} pass produced no counter rows"
    except subprocess.TimeoutExpired:
        return None, f"a counter pass exceeded {per_pass_timeout_s} s"
    except Exception as ex:
        return None, f"{type(ex).__name__}: {ex}"
    finally:
        shutil.rmtree(base, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"], got["WRITE_SIZE"]
    traffic = {k: int(round((2.0 * fetch.get(k, (0, 0.0))[1] + write.get(k, (0, 0.0))[1]) * 1024.0)) for k in set(fetch) | set(write)}
    return ({"hbm_bytes_per_launch": traffic, "launches_per_bracket": {},
             "fetch_bytes_per_launch_uncorrected": {k: int(round(v[1] * 1024.0)) for k, v in fetch.items()},
             "write_bytes_per_launch": {k: int(round(v[1] * 1024.0)) for k, v in write.items()},
             "valu_wave_instructions_per_launch": {k: int(round(v[1])) for k, v in got["SQ_INSTS_VALU"].items()},
             "seconds": round(time.perf_counter() - t0, 1)},
            "collected IN THIS RUN: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU --kernel-trace (three passes) -- python bench.py "
            "--pmc-child --steps 3 --warmup 2; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (the guide's gfx950 correction for 16 B/lane "
            "streaming reads; profiles/r06_fetch_calib.txt holds the factor measured for the 4 B/lane and 8 x 16 B row gathers of the HexPlane kernels)")


def workload_label(a, world=None):
    """Name of the workload, derived from the arguments: a BASELINE.json config name only when the run IS that config."""
    world = a.gpus if world is None else world
    std = (a.width, a.height, a.frames, a.scale_mult) == (1600, 1066, 50, 1.0)
    if std and a.P == 1_200_000:
        return "BASELINE cfg3 (configs[2])" if world == 1 else f"BASELINE cfg4 (configs[3]: cfg3 view-parallel over {world} ranks)"
    if std and a.P == 600_000:
        return "BASELINE cfg2 (configs[1])"
    if std and a.P == 2_500_000:
        return "BASELINE cfg5 size (configs[4], densified count)"
    return f"synthetic street scene, NOT a BASELINE config (scale_mult {a.scale_mult})"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--P", type=int, default=1_200_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1066)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-paths", action="store_true", help="skip timing the zero_diff / import_swap call paths")
    ap.add_argument("--scale-mult", type=float, default=1.0,
                    help="multiply every Gaussian's scale: larger splats -> more (tile, Gaussian) instances R per view (R-sweep)")
    ap.add_argument("--sparse-rows", action="store_true",
                    help="N > 1: exchange the SH / opacity / scale / rotation gradients as compact rows of the union of the ranks' "
                         "visible sets (dp.SparseRowExchange) instead of dense all-reduces")
    ap.add_argument("--single-phase-step", action="store_true",
                    help="N > 1: finish every collective, then one optimizer step (default: dp.finish_and_step, two phases)")
    ap.add_argument("--reorder", action="store_true",
                    help="keep the Gaussians themselves in Morton order (GaussianParams.reorder_spatially() once after the scene is "
                         "built; a real run repeats it after every densification)")
    ap.add_argument("--sustain-steps", type=int, default=300,
                    help="after the K timed steps: this many more in one untended loop -> sustained_iters_per_s, step_ms_p99 (0: skip)")
    ap.add_argument("--sync-raster", action="store_true",
                    help="rasterizer forward with the reference's one host wait per call (default: host-asynchronous, raster_C.ASYNC)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect HBM traffic / VALU counters with rocprofv3 after the timed loops")
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) the 3-step run the counter passes profile: scene, warm-up, K steps, nothing else, no JSON line")
    ap.add_argument("--no-heavy-raster", action="store_true", help="skip the rasterizer-heavy leg (config.paths.heavy_raster)")
    ap.add_argument("--heavy-scale-mult", type=float, default=3.5)
    ap.add_argument("--heavy-steps", type=int, default=60)
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per GPU,
    `torch.distributed.run --standalone`-style rendezvous on 127.0.0.1 and a FREE port) and return the launcher's exit code.
    Refuses when the node has fewer than N GPUs, unless S3G_DIST_BACKEND=gloo asks for the functional oversubscribed run
    (RCCL rejects two ranks on one device)."""
    import subprocess
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < a.gpus and os.environ.get("S3G_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {a.gpus} but this node has {n_dev} GPU(s); refusing to print a mislabelled line "
              "(S3G_DIST_BACKEND=gloo runs the ranks on the GPUs there are, functional only)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this image's driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


GC_PASSES = []     # generation-2 collections that ran inside each timed_loop call, in call order
GC1_PASSES = []    # generation-1 collections, same order (diagnostics: one run in 25 of round 5 had a 42-ms step in its 20-step headline with no
                   # generation-2 pass inside -- 109 instead of 137 it/s, profiles/r05_bench_line_host_stall.json; the container may not raise its
                   # scheduling priority (os.setpriority is refused), so a host thread that loses its core in the first ~7 steps, while the host
                   # is less than 35 ms ahead of the device, shows up 1 : 1; `sustained` is the number that does not depend on it)
HOST_STALLS = []   # per timed_loop call: the longest host-side gap between two consecutive step() calls' returns, in ms


def timed_loop(step, indices, world, device, after_step=None, collect=True):
    """Enqueue step(i) for i in indices.  -> (wall seconds from the barrier before to the barrier after -- MAX over ranks is taken by
    the caller --, host seconds spent enqueuing, per-step stream milliseconds from one hipEvent pair per step).  Nothing in the
    loop waits for the device (the rasterizer forward is host-asynchronous by default); the events are read afterwards."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(indices) + 1)]
    # pay the interpreter's collection debt BEFORE the clock starts: a generation-2 pass over this process's heap takes 50-90 ms
    # (measured: one render frame of 89 ms in a loop of 1.5 ms frames, profiles/r04_headline_variance.txt), i.e. a quarter of a
    # 20-step timed region if it happens to fall inside it.  The collector stays ENABLED during the loop; GC_PASSES counts what ran.
    if collect:
        gc.collect()
    gen2_before = gc.get_stats()[2]["collections"]
    gen1_before = gc.get_stats()[1]["collections"]
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    t_prev, longest = t0, 0.0
    for k, i in enumerate(indices):
        out = step(i)
        evs[k + 1].record()
        if after_step is not None:
            after_step(out)
        t_now = time.perf_counter()
        longest, t_prev = max(longest, t_now - t_prev), t_now
    t_enq = time.perf_counter() - t0
    HOST_STALLS.append(round(1000.0 * longest, 3))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    per_step = [evs[k].elapsed_time(evs[k + 1]) for k in range(len(indices))]
    GC_PASSES.append(gc.get_stats()[2]["collections"] - gen2_before)
    GC1_PASSES.append(gc.get_stats()[1]["collections"] - gen1_before)
    return dt, t_enq, per_step


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's N > 1 invocation goes through torch.distributed.run and sets WORLD_SIZE; a plain `bench.py --gpus N`
        # must not print a 1-GPU line under an N-GPU label: launch the ranks here
        raise SystemExit(launch_ranks(a, argv))

    from s3gaussian_amd import _lib, dp, raster_C
    from s3gaussian_amd.pipeline import training_step
    rank, world, local = dp.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the label would not match the run")
    dist_on = dp.active()      # world > 1, or a process group of ONE rank under S3G_FORCE_DIST=1 (executes the RCCL path on a one-GPU lease)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU; the modulo only matters for functional tests that oversubscribe one GPU (S3G_DIST_BACKEND=gloo)
    device = torch.device("cuda", local % torch.cuda.device_count() if world > 1 else 0)
    torch.cuda.set_device(device)
    if a.sync_raster:
        raster_C.set_async(False)
    # the loops of this script check for arena overflows AFTER each timed region and repeat it if there was one (and `sustained`
    # reports its count): they run the speculative form of the asynchronous forward (nothing waits for a verdict).  The drop-in
    # boundary's default is "verified" (raster_C.POLICY); the reference's own train.py legs below run under that default.
    raster_C.set_async_policy("speculative")
    import ctypes as C
    L = _lib.lib()
    L.s3g_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def clear_profile_slots():
        for i in range(10):
            L.s3g_profile_read(i, None, None, None)

    pc, cams, hyper, opt, bg = build_scene(a.P, a.width, a.height, a.frames, device, scale_mult=a.scale_mult)
    if a.reorder:
        pc.reorder_spatially()
    scene_aabb = tuple(pc._deformation.deformation_net.grid.aabb.detach().cpu().tolist())     # (xyz_max, xyz_min) as set_aabb takes them
    my_views = dp.shard_views(len(cams), rank, world, seed=0)
    n_needed = a.steps + a.warmup
    views = [my_views[i % len(my_views)] for i in range(n_needed)]
    uniq = sorted(set(views))
    targets = {}
    for v in uniq[:min(len(uniq), 12)]:          # bounded target cache (each is 4 images of 1066x1600)
        targets[v] = make_targets(pc, cams[v], bg, hyper, seed=1000 + v)
    tkeys = list(targets)
    # large gradients are all-reduced as soon as backward produces them (overlaps the rest of the backward pass)
    SPARSE = ("f_dc", "f_rest", "opacity", "scaling", "rotation")   # gradients that are exactly zero for Gaussians no rank sees
    reducer = sparse = None
    comm = {"elems": 0, "events": [], "sparse_rows": 0}
    if dist_on:
        if a.sparse_rows:
            dense = lambda: [p for g in pc.optimizer.param_groups if g.get("name") not in SPARSE for p in g["params"]]
            reducer = dp.OverlappedGradAllReducer(dense, average=False)
            sparse = dp.SparseRowExchange(average=False)
        else:
            reducer = dp.OverlappedGradAllReducer(pc.optimizer, average=False)   # follows densify/prune
        pc.optimizer.grad_scale = 1.0 / world   # the SUM all-reduce is averaged inside the Adam kernel

    def hook(pc_, pkg):
        if reducer is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()            # backward's last kernel is queued: from here on the stream waits for collectives + steps
            comm["events"].append([ev])
            # (a rank whose asynchronous forward overflowed makes EVERY replica drop this step: the reducer's finish() /
            #  finish_and_step() all-reduce the flag themselves since round 5)
            g_xy, any_vis, rmax = dp.reduce_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])
            dp.add_densification_stats(pc_.xyz_gradient_accum, pc_.denom, pc_.max_radii2D, g_xy, any_vis, rmax)
            if sparse is not None:
                rows = [p for g in pc_.optimizer.param_groups if g.get("name") in SPARSE for p in g["params"]]
                comm["elems"] += sparse(rows, pkg["visibility_filter"])
                comm["sparse_rows"] += sparse.last_rows

    def optimizer_step():
        if a.single_phase_step or sparse is not None:
            comm["elems"] += reducer.finish()
            pc.optimizer.step()
        else:
            comm["elems"] += reducer.finish_and_step(pc.optimizer)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        comm["events"][-1].append(ev)

    def step(i):
        v = views[i]
        gt_img, gt_depth, gt_feat = targets[v] if v in targets else targets[tkeys[i % len(tkeys)]]
        # densification bookkeeping (train.py:489-493) is part of every iteration below densify_until_iter: single GPU ->
        # inside the rasterizer's per-Gaussian backward; data parallel -> after the all-reduce of the statistics (hook)
        loss, pkg = training_step(pc, cams[v], gt_img, gt_depth, gt_feat, hyper, opt, bg, stage="fine", grad_hook=hook,
                                  densify_stats=(not dist_on), optimizer_step=optimizer_step if dist_on else None)
        return loss, pkg

    if a.pmc_child:
        # the run the counter passes profile (collect_counters_in_run): warm-up + K training steps of the workload, nothing else
        L.s3g_profile_enable(0)
        for i in range(a.warmup + a.steps):
            step(i)
        torch.cuda.synchronize()
        return

    # everything built so far (scene, targets, modules, the torch / ctypes machinery) is long-lived: take it out of the collector's
    # working set, so that a generation-2 pass during the loops below walks this step's garbage, not the whole heap
    gc.collect()
    gc.freeze()

    # ---- 1. the headline: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, with the
    #         in-library kernel brackets OFF (they are hipEventCreate + hipEventRecord pairs inside the timed region otherwise) ----
    L.s3g_profile_enable(0)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    raster_C.async_reset_statistics(device)
    comm.update(elems=0, events=[], sparse_rows=0)
    vis_masks, losses = [], []     # read after the timed region (workload statistics are not part of the step)

    def keep(out):
        losses.append(out[0])
        vis_masks.append(out[1]["visibility_filter"])

    headline_idx = list(range(a.warmup, a.warmup + a.steps))
    dt, t_enq, per_step = timed_loop(step, headline_idx, world, device, after_step=keep)
    astat = raster_C.async_status(device, block=True)
    n_over = len(astat["overflows"])
    if world > 1:     # every rank must take the same branch below (the loops contain barriers and collectives)
        t = torch.tensor([n_over], device=device, dtype=torch.int32)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        n_over = int(t.item())
    if n_over:
        # a step whose forward overflowed its speculative arena did no work: the capacity has grown by now, time the loop again
        vis_masks.clear()
        losses.clear()
        GC_PASSES.clear()
        GC1_PASSES.clear()
        HOST_STALLS.clear()
        comm.update(elems=0, events=[], sparse_rows=0)
        raster_C.async_reset_statistics(device)
        dt, t_enq, per_step = timed_loop(step, headline_idx, world, device, after_step=keep)
        astat2 = raster_C.async_status(device, block=True)
        astat2["overflows_in_discarded_first_attempt"] = n_over
        astat = astat2
        n_over = len(astat["overflows"])
        if world > 1:
            t = torch.tensor([n_over], device=device, dtype=torch.int32)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            n_over = int(t.item())
        if n_over:
            raise SystemExit("bench.py: the asynchronous rasterizer overflowed its arena twice in the timed region; run with --sync-raster")
    if not all(bool(torch.isfinite(x)) for x in losses):
        raise SystemExit("bench.py: a timed step produced a non-finite loss -- the number would not be a training throughput")
    if world > 1:
        t = torch.tensor([dt, t_enq], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, t_enq = float(t[0].item()), float(t[1].item())
    comm_timed = {"elems": comm["elems"], "events": list(comm["events"]), "sparse_rows": comm["sparse_rows"]}

    # ---- 1b. sustained throughput: a.sustain_steps more steps (default 300, ~2.3 s) over the same views, the interpreter's collector
    #          running as it pleases (no collect before, nothing frozen anew), walk-order re-sorts included -- the 20-step headline is a
    #          0.15-s burst; this is the number a long training run sees (tools/soak.py is the 600 / 3000-step form) ----------------
    sustained = None
    if a.sustain_steps > 0:
        s_idx = [(a.warmup + k) % n_needed for k in range(a.sustain_steps)]
        raster_C.async_reset_statistics(device)
        dts, t_enq_s, per_s = timed_loop(step, s_idx, world, device, collect=False)
        s_over = len(raster_C.async_status(device, block=True)["overflows"])
        if world > 1:
            t = torch.tensor([dts, float(s_over)], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dts, s_over = float(t[0].item()), int(t[1].item())
        ps = sorted(per_s)
        sustained = {"steps": a.sustain_steps, "iters_per_s": round(world * a.sustain_steps / dts, 3),
                     "ms_per_step": round(1000.0 * dts / a.sustain_steps, 3), "step_ms_median": round(ps[len(ps) // 2], 3),
                     "step_ms_p99": round(ps[min(len(ps) - 1, int(0.99 * len(ps)))], 3), "step_ms_max": round(ps[-1], 3),
                     "host_enqueue_ms_per_step": round(1000.0 * t_enq_s / a.sustain_steps, 3),
                     "gc_gen2_passes": GC_PASSES[-1], "arena_overflows": s_over}
        comm.update(elems=0, events=[], sparse_rows=0)

    # ---- 1c. the OTHER arithmetic of the MLP kernels, A/B at the SAME point of the process: the default step again, then the other one,
    #          same views.  Since round 6 the default is "bf16x3" (the bf16 matrix pipe on operands split exactly into three bf16 pieces,
    #          weight fragments split once per call, fp32 accumulation: fp32 results) and the leg times the EXACT fp32 fma chains --
    #          rounds 3-5 had it the other way round.  (Every loop after the first ~2 s of load runs ~0.2 ms/step slower than the opening
    #          20-step burst on these boxes: read the pair against each other, not against the headline.)
    mlp_ab = None
    from s3gaussian_amd import mlp as _mlp
    mlp_default = _mlp.get_mlp_arithmetic()
    mlp_other = "f32" if mlp_default != "f32" else "bf16x3"
    if world == 1 and not dist_on and not a.no_alt_paths:
        try:
            ab_idx = [i % n_needed for i in range(a.steps)]
            dt_a, _, ps_a = timed_loop(step, ab_idx, 1, device)
            _mlp.set_mlp_arithmetic(mlp_other)
            for i in range(3):
                step(i % n_needed)
            dt_b, _, ps_b = timed_loop(step, ab_idx, 1, device)
            ms_b = 1000.0 * dt_b / a.steps
            mlp_ab = {"arithmetic": mlp_other, "ms_per_step": round(ms_b, 3), "iters_per_s": round(1000.0 / ms_b, 2), "steps": a.steps,
                      "default_arithmetic": mlp_default, "default_back_to_back_ms_per_step": round(1000.0 * dt_a / a.steps, 3),
                      "gpu_ms_per_step": round(sum(ps_b) / len(ps_b), 3), "default_back_to_back_gpu_ms_per_step": round(sum(ps_a) / len(ps_a), 3)}
        except Exception as ex:   # never take the headline down
            mlp_ab = {"arithmetic": mlp_other, "ms_per_step": None, "error": f"{type(ex).__name__}: {ex}"}
        finally:
            _mlp.set_mlp_arithmetic(mlp_default)
        comm.update(elems=0, events=[], sparse_rows=0)

    # ---- 1d. a rasterizer-HEAVY workload in the driver-visible line (VERDICT r5 missing #4): the same 1.2 M Gaussians with every scale
    #          multiplied (default 3.5: R ~ 11 M instances per view, SURVEY 8a-a11's estimate for a trained cfg3 scene is 5-12 M, the
    #          headline scene has 1.46 M), own model, own targets, 60 timed steps --------------------------------------------------
    heavy = None
    if world == 1 and not dist_on and not a.no_alt_paths and not a.no_heavy_raster:
        pc_h = None
        try:
            pc_h, cams_h, hyper_h, opt_h, bg_h = build_scene(a.P, a.width, a.height, a.frames, device, scale_mult=a.scale_mult * a.heavy_scale_mult)
            hv = [my_views[i % len(my_views)] for i in range(6)]
            tg_h = {v: make_targets(pc_h, cams_h[v], bg_h, hyper_h, seed=2000 + v) for v in sorted(set(hv))}
            hvis = []

            def step_h(i):
                v = hv[i % len(hv)]
                return training_step(pc_h, cams_h[v], *tg_h[v], hyper_h, opt_h, bg_h, stage="fine", densify_stats=True)

            for i in range(6):
                step_h(i)
            torch.cuda.synchronize()
            for attempt in range(2):          # an overflowed step did no work: the capacity has grown by now, time again
                hvis.clear()
                raster_C.async_reset_statistics(device)
                dt_h, enq_h, ps_h = timed_loop(step_h, list(range(a.heavy_steps)), 1, device, after_step=lambda o: hvis.append(o[1]["visibility_filter"]))
                st_h = raster_C.async_status(device, block=True)
                if not st_h["overflows"]:
                    break
            ms_h = 1000.0 * dt_h / a.heavy_steps
            srt_h = sorted(ps_h)
            R_h = st_h["mean_instances"] or 0.0
            heavy = {"ms_per_step": round(ms_h, 3), "iters_per_s": round(1000.0 / ms_h, 2), "steps": a.heavy_steps,
                     "scale_mult": a.scale_mult * a.heavy_scale_mult, "gaussians": a.P, "instances_R_per_view": round(R_h),
                     "visible_V_per_view": round(float(sum(int(m.sum()) for m in hvis)) / max(len(hvis), 1)),
                     "mean_tile_list_length": round(R_h / (((a.width + 15) // 16) * ((a.height + 15) // 16)), 1),
                     "gpu_ms_per_step": round(sum(ps_h) / len(ps_h), 3), "step_ms_median": round(srt_h[len(srt_h) // 2], 3),
                     "step_ms_max": round(srt_h[-1], 3), "host_enqueue_ms_per_step": round(1000.0 * enq_h / a.heavy_steps, 3),
                     "arena_overflows": len(st_h["overflows"]),
                     "what": "the fused step on the same scene with every Gaussian's scale multiplied: the blend / sort / binning kernels "
                             "carry ~8 x the instances of the headline scene"}
        except Exception as ex:   # never take the headline down
            heavy = {"ms_per_step": None, "error": f"{type(ex).__name__}: {ex}"}
        finally:
            del pc_h
            gc.collect()
            torch.cuda.empty_cache()
            raster_C.invalidate_geometry_cache()
        comm.update(elems=0, events=[], sparse_rows=0)

    # ---- 2. roofline leg: the SAME steps again with the nine hot kernels bracketed by hipEvent pairs inside libs3g.so -------
    clear_profile_slots()
    L.s3g_profile_enable(1)
    _, _, per_step_instrumented = timed_loop(step, headline_idx, world, device)
    L.s3g_profile_enable(0)
    prof = {}
    for i in range(9):
        ms, x, y = C.c_double(), C.c_double(), C.c_double()
        n = L.s3g_profile_read(i, C.byref(ms), C.byref(x), C.byref(y))
        prof[i] = (n, ms.value / n, x.value / n, y.value / n) if n else (0, 0.0, 0.0, 0.0)

    # ---- 2b. the scatter walk on FRESH walk orders: a few more instrumented steps with the orders re-sorted on every backward.  The
    #          walk orders are refreshed every 16th backward only (a re-sort costs ~0.9 ms) and the points move under Adam in between: the
    #          difference to the bracket above is what the age of the orders costs (profiles/r06_scatter_context*.txt) ----
    scatter_fresh_ms = None
    if world == 1 and not dist_on:
        import s3gaussian_amd.hexplane as _hx
        keep_refresh = _hx.SORT_REFRESH
        try:
            _hx.SORT_REFRESH = 1
            step(headline_idx[0])
            clear_profile_slots()
            L.s3g_profile_enable(1)
            for i in headline_idx[:8]:
                step(i)
            torch.cuda.synchronize()
            L.s3g_profile_enable(0)
            ms_ = C.c_double()
            n_ = L.s3g_profile_read(4, C.byref(ms_), None, None)
            scatter_fresh_ms = round(ms_.value / n_, 4) if n_ else None
        finally:
            _hx.SORT_REFRESH = keep_refresh
            L.s3g_profile_enable(0)
            clear_profile_slots()

    # ---- 3. render ms/frame (the second half of BASELINE's metric): one render(stage="fine") under no_grad, SURVEY.md 3.5 ----
    from types import SimpleNamespace
    from s3gaussian_amd.pipeline import render as render_fn
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    n_frames = 20

    render_outliers = []

    def render_loop():
        """-> (wall ms per frame over n_frames frames, median stream ms per frame from one event pair per frame)."""
        for i in range(5):
            render_fn(cams[views[i % len(views)]], pc, pipe, bg, stage="fine")
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_frames + 1)]
        gc.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        evs[0].record()
        for i in range(n_frames):
            render_fn(cams[views[i % len(views)]], pc, pipe, bg, stage="fine")
            evs[i + 1].record()
        torch.cuda.synchronize()
        wall = 1000.0 * (time.perf_counter() - t1) / n_frames
        raw = [evs[k].elapsed_time(evs[k + 1]) for k in range(n_frames)]
        per = sorted(raw)
        if per[-1] > 3.0 * per[n_frames // 2]:     # a one-off stall inside the loop: say where, so that it can be explained
            render_outliers.append({"frame": raw.index(per[-1]), "ms": round(per[-1], 3), "median_ms": round(per[n_frames // 2], 3)})
        return wall, per[n_frames // 2]

    import s3gaussian_amd.deformation as _deformation

    def render_loop_by_timestamp():
        """The evaluation loops of the reference visit the cameras in dataset order -- the 3 Waymo cameras of one frame back to back
        (utils/video_utils.py:116-349) -- and the deformation depends on (xyz, t) only: under no_grad its outputs are reused for the 2nd
        and 3rd camera of a timestamp (deformation.INFER_CACHE).  -> median stream ms of a frame that OPENS a timestamp (evaluates the
        field), of a frame that shares its predecessor's timestamp, and the mean over the video order (3 cameras per timestamp)."""
        order = [3 * k + c for k in range(min(a.frames, 8)) for c in range(3) if 3 * k + c < len(cams)]
        if len(order) < 6:
            return None
        for v in order[:6]:
            render_fn(cams[v], pc, pipe, bg, stage="fine")
        hits0 = _deformation.infer_cache_hits
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        evs[0].record()
        for k, v in enumerate(order):
            render_fn(cams[v], pc, pipe, bg, stage="fine")
            evs[k + 1].record()
        torch.cuda.synchronize()
        wall = 1000.0 * (time.perf_counter() - t1) / len(order)
        raw = [evs[k].elapsed_time(evs[k + 1]) for k in range(len(order))]
        fresh = sorted(raw[k] for k in range(len(order)) if k % 3 == 0)
        shared = sorted(raw[k] for k in range(len(order)) if k % 3 != 0)
        return {"fresh_timestamp_ms": round(fresh[len(fresh) // 2], 3), "same_timestamp_ms": round(shared[len(shared) // 2], 3),
                "video_order_ms_per_frame": round(wall, 3), "frames": len(order), "deformation_reused_for": _deformation.infer_cache_hits - hits0}

    with torch.no_grad():
        render_ms, render_median_ms = render_loop()
        try:
            render_by_timestamp = render_loop_by_timestamp()
        except Exception as ex:
            render_by_timestamp = {"error": f"{type(ex).__name__}: {ex}"}
        clear_profile_slots()
        L.s3g_profile_enable(1)          # a separate, instrumented pass for the inference kernel's own time
        for i in range(8):
            render_fn(cams[views[i % len(views)]], pc, pipe, bg, stage="fine")
        torch.cuda.synchronize()
        L.s3g_profile_enable(0)
        _ms = C.c_double()
        _n = L.s3g_profile_read(9, C.byref(_ms), None, None)
        infer_kernel_ms = (_ms.value / _n) if _n else None   # s3g::deform_infer_kernel (HexPlane (+) MLP heads), per frame
        clear_profile_slots()
        # the same frames with the inference kernel in the OTHER arithmetic (default since round 6: the GEMM layers on the bf16 matrix
        # pipe, exactly split operands, include/s3g_mlp.h::s3g_deform_infer_split; other: the exact fp32 chain) -- reported beside it
        _arith = _deformation.INFER_ARITHMETIC
        infer_other = "f32" if _arith != "f32" else "bf16x3"
        _deformation.INFER_ARITHMETIC = infer_other
        try:
            render_other_ms, render_other_median_ms = render_loop()
        finally:
            _deformation.INFER_ARITHMETIC = _arith

    # ---- 4. counters of this very run (rank 0 of a one-GPU run; the timed loops above are over) --------------------------------
    pmc_in_run, pmc_in_run_note = None, "not attempted (--no-pmc, N > 1 or --no-alt-paths)"
    if world == 1 and not dist_on and not a.no_pmc and not a.no_alt_paths:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        pmc_in_run, pmc_in_run_note = collect_counters_in_run(a)

    dev_ids = None
    if dist_on:     # collectives are called by EVERY rank, never inside the rank-0 block below
        dev_ids = [None] * world
        torch.distributed.all_gather_object(dev_ids, f"{os.uname().nodename}:cuda:{device.index}")
    if rank == 0:
        # ---- roofline: every hot kernel timed in-library with hipEvents on the launch stream (include/s3g_raster.h),
        # priced against its ALGORITHMIC bytes / flops (DESIGN.md section 7 states each model) -----------------------------
        V = float(sum(int(m.sum()) for m in vis_masks)) / max(a.steps, 1)    # mean visible Gaussians per step (both raster calls share them)
        P = float(a.P)
        net = pc._deformation.deformation_net
        planes = [p for p in net.grid.grids.parameters()] if hasattr(net.grid, "grids") else []
        plane_bytes = float(sum(p.numel() for p in planes) * 4)
        levels = float(len(getattr(net.grid, "grids", [])) or 4)
        FEAT = 32.0 * levels
        MLP_FLOP = 2.0 * (128 * 64 + 4 * 64 * 64 + 2 * 64 * 3 + 64 * 48)   # per point and direction: 56064
        pair = bool(hyper.feat_head)   # pipeline.render blends both images of an iteration in one pass each way
        N_pix = float(a.width * a.height)
        # instances per view: the asynchronous forward does not know R when it records its bracket (it reports -1); the true
        # counts arrive through the status ring (raster_C.async_status) -- synchronous forward: the bracket carries them
        R_mean = astat["mean_instances"] if (astat.get("drained") and astat.get("mean_instances")) else max(prof[0][2], 0.0)

        # id -> (kernel, STRICT algorithmic bytes, implementation bytes incl. scratch, flops) as functions of the bracket's (x, y).
        # Strict = SURVEY 8(d): what the math must read and write (inputs, outputs, parameters once); scratch that exists only
        # because of how this implementation is split into kernels (G slab, activation stash, gradient signals) is counted
        # under implementation bytes and never enters `frac`.
        G_ROWS = float(L.s3g_hexplane_backward_scratch_rows(int(levels)))   # 128-byte rows of scratch per point (r2: 24, r3: 4)
        from s3gaussian_amd import hexplane as _hx
        POINT_KERNEL = ("s3g::hexplane_backward_pointdiv_kernel" if _hx.BACKWARD_MODE == "slab" else "s3g::hexplane_backward_point_kernel")
        RB = lambda R_, N_: (56.0 if pair else 44.0) * R_mean + (36.0 if pair else 24.0) * N_
        split_chains = mlp_default == "bf16x3"     # the two chain kernels run on the bf16 pipe (6 piece products per product)
        models = {
            # two-image pass (RGB+depth and feature image from one geometry): + colors2 per instance, + one image per pixel
            0: ("s3g::blend_forward_kernel", RB, None, None),
            1: ("s3g::blend_backward_kernel", lambda R_, N_: RB(R_, N_) + 40.0 * V,
                lambda R_, N_: RB(R_, N_) + (56.0 if pair else 40.0) * R_mean, None),
            2: ("s3g::hexplane_forward_kernel", lambda n, l: n * (16.0 + 4.0 * FEAT) + plane_bytes, None, None),
            # xyz,t 16 B + dL/dfeatures 4F B read, dL/dxyz 12 B written, planes read once; the scratch rows are implementation
            # (round 4: the division-form pass also reads the forward's feature rows, 4F B per point: implementation)
            3: (POINT_KERNEL, lambda n, l: n * (28.0 + 4.0 * FEAT) + plane_bytes,
                lambda n, l: n * (28.0 + 4.0 * FEAT + (4.0 * FEAT if POINT_KERNEL.endswith("pointdiv_kernel") else 0.0) + G_ROWS * 128.0) + plane_bytes, None),
            # plane gradients written once; reading the scratch back is implementation: every (orientation, level) walk reads
            # index + position (8 B), the coordinates (12 B) and the level's T row (128 B) of each point
            4: ("s3g::hexplane_scatter_kernel", lambda n, l: plane_bytes,
                lambda n, l: n * (3.0 * l * 20.0 + (3.0 if G_ROWS < 24 else 1.0) * G_ROWS * 128.0) + plane_bytes, None),
            # features in, three heads out; the 5 stashed activations are implementation
            # implementation: + 5 stashed activation planes (for the weight gradients) + 5 ReLU mask words per lane (40 B/point)
            5: ("s3g::mlp_forward_presplit_kernel" if split_chains else "s3g::mlp_forward_kernel",
                lambda n, _: n * (512.0 + 216.0), lambda n, _: n * (512.0 + 5 * 256.0 + 40.0 + 216.0), lambda n: n * MLP_FLOP),
            # implementation: mask words in, 5 gradient-signal planes out (read back by the weight-gradient launches)
            6: ("s3g::mlp_backward_presplit_kernel" if split_chains else "s3g::mlp_backward_kernel",
                lambda n, _: n * (216.0 + 512.0), lambda n, _: n * (40.0 + 216.0 + 5 * 256.0 + 512.0), lambda n: n * MLP_FLOP),
            # ONE launch for all nine GEMMs (mlp_wgrad_all_kernel): every stash / signal plane and the feature rows read once:
            # 5 x 256 (stash) + 5 x 256 (signals) + 512 (features) + 12 + 12 + 192 (head gradients) = 3288 B per point
            7: ("s3g::mlp_wgrad_all_kernel", lambda n, _: n * 512.0, lambda n, _: n * 3288.0, lambda n: n * MLP_FLOP),
            8: ("s3g::adam_kernel", lambda n, _: n * 28.0, None, None),   # p, g, m, v read; p, m, v written
        }
        traffic_db, traffic_launches, traffic_source, valu_db = {}, {}, None, {}
        default_workload = (a.P, a.width, a.height, a.frames, a.scale_mult) == (1_200_000, 1600, 1066, 50, 1.0)
        pmc_note = None
        if pmc_in_run is not None:     # counters of THIS run on THIS box (collect_counters_in_run)
            traffic_db = pmc_in_run.get("hbm_bytes_per_launch", {})
            valu_db = pmc_in_run.get("valu_wave_instructions_per_launch", {})
            traffic_source = pmc_in_run_note
        else:
            pmc_note = pmc_in_run_note
        pmc = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        if not traffic_db and os.path.exists(pmc) and default_workload:   # earlier collection of the same passes, default workload only
            try:
                db = json.load(open(pmc))
                traffic_db = db.get("hbm_bytes_per_launch", {})
                traffic_launches = db.get("launches_per_bracket", {})
                valu_db = db.get("valu_wave_instructions_per_launch", {})   # SQ_INSTS_VALU pass of the same command
                traffic_source = ("NOT collected in this run (the hipEvent times are): profiles/kernel_traffic.json: " +
                                  db.get("command", "rocprofv3 --pmc passes"))
            except Exception:
                traffic_db = {}
        kernels = []
        for i, (name, fbytes, fimpl, fflops) in models.items():
            n, avg_ms, x, y = prof[i]
            if not n:
                continue
            if i in (0, 1):
                y = N_pix
            nbytes = fbytes(x, y)
            t = avg_ms * 1e-3
            gbs = nbytes / t / 1e9
            ent = {"kernel": name, "launches_per_step": round(n / a.steps, 2), "avg_launch_ms": round(avg_ms, 4),
                   "algorithmic_bytes_per_launch": round(nbytes),
                   "implementation_bytes_per_launch": round((fimpl or fbytes)(x, y)), "hbm_GBps": round(gbs, 1),
                   "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)}
            bound, frac = "hbm", gbs / PEAK_HBM_GBS
            if fflops is not None:
                tf = fflops(x) / t / 1e12
                if split_chains and i in (5, 6):
                    # algorithmic fp32 FLOPs stay the unit of `mfma_TFLOPs`; the pipe executes six bf16 piece products for each
                    mf = 6.0 * tf / PEAK_MFMA_BF16_TFLOPS
                    ent.update({"mfma_pipe": "bf16 (6 piece products per fp32 product)", "mfma_pipe_TFLOPs": round(6.0 * tf, 1),
                                "mfma_peak_TFLOPs": PEAK_MFMA_BF16_TFLOPS})
                else:
                    mf = tf / PEAK_MFMA_F32_TFLOPS
                ent.update({"flops_per_launch": round(fflops(x)), "mfma_TFLOPs": round(tf, 2), "mfma_frac": round(mf, 4)})
                # the chain kernels also stream their stash / signal planes: implementation bytes, priced for information
                ent["implementation_hbm_frac"] = round((fimpl or fbytes)(x, y) / t / 1e9 / PEAK_HBM_GBS, 4)
                if mf > frac:   # the roof this kernel sits closer to
                    bound, frac = "mfma", mf
            base = name.split(" ")[0]
            # The third roof: VALU issue.  A wave64 VALU instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs at
            # VALU_CLOCK_GHZ = 614.4 G wave-instructions/s.  Priced for EVERY kernel that has a SQ_INSTS_VALU count (VERDICT r5 weak #2:
            # rounds 4-5 applied it to the two blend kernels only and labelled the HexPlane backward kernels "hbm 0.016" when the same
            # counters put them at 0.83 / 0.73 of the VALU issue roof); `bound` = the roof the kernel sits closest to.  SURVEY 8(d)
            # predicted it for the blend passes (~80 FLOP/B >> the 25 FLOP/B machine balance).  fp32 MFMA shares the vector datapath
            # (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD = the FP32 vector rate), so for the MLP kernels the two figures are read together.
            if base in valu_db and valu_db[base] > 0:
                busy_ms = valu_db[base] * 4.0 / (1024.0 * VALU_CLOCK_GHZ * 1e9) * 1e3
                ent.update({"valu_wave_instructions_per_launch": valu_db[base], "valu_issue_ms": round(busy_ms, 4),
                            "valu_frac": round(busy_ms / avg_ms, 4)})
                if busy_ms / avg_ms > frac:
                    bound, frac = "valu", busy_ms / avg_ms
            elif i in (0, 1):
                bound = "valu"               # no SQ_INSTS_VALU count for this workload: still not an HBM kernel
                ent["valu_frac"] = None      # ... and the HBM figure is NOT its roof
            ent["bound"] = bound
            ent["frac"] = round(frac, 4)
            ent["ms_per_step"] = round(avg_ms * n / a.steps, 4)
            if base in traffic_db:   # PMC bytes per launch (the wgrad bracket may cover several launches)
                ent["traffic"] = traffic_db[base] * (traffic_launches.get(base, 1) if i == 7 else 1)
            kernels.append(ent)
        roof = None
        if kernels:
            dom = max(kernels, key=lambda e: e["ms_per_step"])   # the kernel the step spends most time in
            if dom["bound"] == "mfma":
                roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom.get("mfma_pipe_TFLOPs", dom["mfma_TFLOPs"]),
                        "peak": dom.get("mfma_peak_TFLOPs", PEAK_MFMA_F32_TFLOPS), "unit": "TFLOP/s", "frac": dom["mfma_frac"],
                        "traffic": dom.get("traffic")}
            elif dom["bound"] == "valu" and dom.get("valu_frac"):
                peak_gi = 1024.0 * VALU_CLOCK_GHZ / 4.0        # G wave-instructions/s the chip can issue
                ach = dom["valu_wave_instructions_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9
                roof = {"kernel": dom["kernel"], "bound": "valu", "achieved": round(ach, 1), "peak": round(peak_gi, 1),
                        "unit": "G wave-instructions/s", "frac": round(ach / peak_gi, 4), "traffic": dom.get("traffic"),
                        # the contract's own roofs for the same kernel, for a reader who prices everything against HBM
                        "hbm": {"achieved": dom["hbm_GBps"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dom["hbm_frac"]}}
            else:
                roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["hbm_GBps"], "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": dom["hbm_frac"], "traffic": dom.get("traffic")}
            if pmc_note:
                roof["counters_in_this_run"] = pmc_note
            elif pmc_in_run is not None:
                roof["counters_in_this_run"] = {"seconds": pmc_in_run.get("seconds"), "kernels": len(traffic_db)}
            roof.update({"avg_launch_ms": dom["avg_launch_ms"], "launches_per_step": dom["launches_per_step"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                         "implementation_bytes_per_launch": dom["implementation_bytes_per_launch"],
                         "traffic_source": traffic_source, "kernels": kernels,
                         "bracketed_kernels_ms_per_step": round(sum(k["ms_per_step"] for k in kernels), 4),
                         "timed_in": "a separate instrumented loop over the same steps (the headline loop runs with the brackets off)"})
        # the HexPlane backward is ONE operation split into two kernels by this implementation (per-point pass + scatter walks):
        # their combined figures, for information next to the per-kernel entries (the top-level fields stay per kernel)
        hb = [k for k in kernels if k["kernel"] in (POINT_KERNEL, "s3g::hexplane_scatter_kernel")]
        if roof is not None and len(hb) == 2:
            ms = sum(k["avg_launch_ms"] for k in hb)
            nb = sum(k["algorithmic_bytes_per_launch"] for k in hb)
            tr = [k.get("traffic") for k in hb]
            roof["hexplane_backward_pair"] = {"ms": round(ms, 4), "algorithmic_bytes": nb, "hbm_GBps": round(nb / (ms * 1e-3) / 1e9, 1),
                                              "frac": round(nb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                              "traffic": sum(tr) if all(t is not None for t in tr) else None,
                                              "scatter_ms_on_fresh_walk_orders": scatter_fresh_ms,
                                              "scatter_note": "avg_launch_ms of the scatter kernel is taken with walk orders up to 16 iterations old "
                                                              "(re-sorted every 16th backward: a re-sort costs ~0.9 ms); scatter_ms_on_fresh_walk_orders "
                                                              "is the same kernel over 8 more steps with the orders re-sorted on every backward"}
        fwd = next((k for k in kernels if k["kernel"] == "s3g::blend_forward_kernel"), None)
        srt = sorted(per_step)
        out = {
            "metric": "train_iters_per_sec", "value": round(world * a.steps / dt, 3), "unit": "iters/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # one hipEvent pair per step on the launch stream (rank 0): stream time between step boundaries; next to the host's
            # own time to ENQUEUE a step -- the step is GPU-bound while the second stays below the first
            "step_ms_min": round(srt[0], 3), "step_ms_median": round(srt[len(srt) // 2], 3), "step_ms_max": round(srt[-1], 3),
            "gpu_ms_per_step": round(sum(per_step) / len(per_step), 3),
            "host_enqueue_ms_per_step": round(1000.0 * t_enq / a.steps, 3),
            "instrumented_loop_ms_per_step": round(sum(per_step_instrumented) / len(per_step_instrumented), 3),
            "gc_gen2_passes_in_timed_loop": GC_PASSES[0] if GC_PASSES else None,
            "gc_gen1_passes_in_timed_loop": GC1_PASSES[0] if GC1_PASSES else None,
            "longest_host_enqueue_of_one_step_ms": HOST_STALLS[0] if HOST_STALLS else None,
            "sustained_iters_per_s": sustained["iters_per_s"] if sustained else None,
            "step_ms_p99": sustained["step_ms_p99"] if sustained else None,
            "sustained": sustained,
            "gc_policy": "gc.collect() + gc.freeze() after the warm-up and gc.collect() before each timed loop; the collector stays enabled "
                         "inside the loops (tools/soak.py times 600 steps with no such care: profiles/r04_soak.jsonl)",
            "config": {"workload": workload_label(a, world) + f": {a.P} Gaussians, {a.height}x{a.width}, 3 cams x {a.frames} frames, fine stage "
                                   "(hexplane+deformation ON), RGB+depth render + feature render, L1+DSSIM+depthL2+featL2+regs, Adam",
                       "path": "fused", "gaussians": a.P, "image": [a.height, a.width], "views_per_step_per_rank": 1,
                       "scale_mult": a.scale_mult, "gaussians_in_morton_order": bool(a.reorder), "instances_R_per_view": round(R_mean), "visible_V_per_view": round(V),
                       "mean_tile_list_length": round(R_mean / (((a.width + 15) // 16) * ((a.height + 15) // 16)), 1),
                       "densify_bookkeeping_in_step": True,
                       "rasterizer_forward": ("host-asynchronous, policy 'speculative' (arena sized for a speculative capacity, overflows checked after "
                                              "each timed loop: s3g_raster_forward_async; the drop-in default 'verified' reads every forward's verdict)" if astat.get("enabled")
                                              else "synchronous (one host wait per forward, like the reference)"),
                       "raster_async": {k: astat.get(k) for k in ("enabled", "policy", "calls", "drained", "overflows", "overflows_in_discarded_first_attempt")
                                        if k in astat},
                       "parallelism": f"view-parallel dp{world}" if world > 1 else "single GPU",
                       "blend_forward_avg_ms": fwd["avg_launch_ms"] if fwd else None,
                       "render_ms_per_frame": round(render_ms, 3), "render_ms_per_frame_median": round(render_median_ms, 3),
                       "render_deform_infer_kernel_ms": round(infer_kernel_ms, 4) if infer_kernel_ms else None,
                       "render_by_timestamp": render_by_timestamp,
                       "render_arithmetic": _arith,
                       f"render_ms_per_frame_{infer_other}": round(render_other_ms, 3),
                       f"render_ms_per_frame_{infer_other}_median": round(render_other_median_ms, 3),
                       "mlp_arithmetic": (mlp_default + (": per-point GEMM chains on the bf16 matrix pipe, every fp32 operand split exactly into three "
                                          "bf16 pieces, fp32 accumulation -- fp32 results (DESIGN 4.3); weight-gradient GEMMs exact fp32 MFMA"
                                          if mlp_default == "bf16x3" else ": exact fp32 fma chains (v_mfma_f32_32x32x2_f32)")),
                       "render_loop_outliers": render_outliers},
            "roofline": roof,
        }
        if dist_on:
            # what the all-reduce cost this rank: stream time from "backward queued" to "optimizer step queued" minus the Adam
            # kernel itself = waiting for collectives + the small statistics reduces.  NO multi-GPU run has been measured in the
            # build sandbox (one GPU per lease); these fields exist so that the first SCALE run explains itself.
            tail = [e[0].elapsed_time(e[1]) for e in comm_timed["events"] if len(e) == 2]
            adam = next((k["ms_per_step"] for k in kernels if k["kernel"] == "s3g::adam_kernel"), 0.0)
            backend = torch.distributed.get_backend()
            try:
                lib_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:
                lib_version = None
            out["comm"] = {"backend": backend, "world_size": torch.distributed.get_world_size(), "rccl_version": lib_version,
                           "devices": dev_ids, "distinct_devices": len(set(dev_ids)),
                           "bytes_reduced_per_step_per_rank": round(4.0 * comm_timed["elems"] / max(a.steps, 1)),
                           "tail_ms_backward_end_to_step_end": round(sum(tail) / max(len(tail), 1), 3), "adam_ms_per_step": adam,
                           "comm_ms_exposed": round(max(sum(tail) / max(len(tail), 1) - adam, 0.0), 3),
                           "optimizer_step": "single phase" if (a.single_phase_step or a.sparse_rows) else "two phases (dp.finish_and_step)",
                           "sparse_row_exchange": bool(a.sparse_rows),
                           "sparse_rows_per_step": round(comm_timed["sparse_rows"] / max(a.steps, 1)) if a.sparse_rows else None,
                           "forced_single_rank_group": bool(dp.force_dist() and world == 1),
                           "collectives_executed_in_this_run": True,
                           "scaling_curve_measured_by_the_builder": False}
        if world == 1 and not dist_on and not a.no_alt_paths:
            out["config"]["paths"] = {"fused": {"ms_per_step": out["ms_per_step"], "iters_per_s": out["value"]}}
            from oracle import ref_py as _ref_py     # the reference's own files as the CALLER of the product (never the other way round)
            if _ref_py.available():
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):      # train.py and the reference's modules print; stdout carries ONE JSON line
                    out["config"]["paths"].update(time_reference_paths(pc, cams, targets, bg, scene_aabb))
                out["config"]["paths_note"] = ("`patched` / `import_swap` / `zero_diff` EXECUTE the reference's own train.py::scene_reconstruction "
                                               "(train.py:216-560), unchanged, from oracle/_ref/reference_py.tar.gz on real Camera / GaussianModel "
                                               "objects; see each entry's `executed`")
            else:
                out["config"]["paths"].update(time_alt_paths(pc, cams, views, targets, tkeys, hyper, opt, bg))
                out["config"]["paths_note"] = ("oracle/_ref/reference_py.tar.gz absent: `patched` / `import_swap` / `zero_diff` are RESTATEMENTS of "
                                               "train.py's iteration body inside bench.py")
            if mlp_ab is not None:
                out["config"]["paths"]["fused_mlp_" + mlp_other] = mlp_ab
            if heavy is not None:
                out["config"]["paths"]["heavy_raster"] = heavy
        psnr_file = os.path.join(ROOT, "profiles", "psnr_parity.json")
        if os.path.exists(psnr_file):   # the third part of BASELINE's metric: written by tools/psnr_parity.py on the GPU box
            try:
                pj = json.load(open(psnr_file))
                md = pj.get("mean_psnr_delta_db")     # per split (train / held-out views): what an evaluation reports
                # rounds 1-2 reported the worst single view under "psnr_delta_vs_oracle_db"; since round 3 the 0.1 dB bar is on the
                # split MEANS (two fp32 trainings diverge chaotically per view): both are printed under names that say which
                out["config"]["psnr_mean_delta_vs_oracle_db"] = max(abs(v) for v in md.values()) if md else None
                out["config"]["psnr_worst_single_view_delta_db"] = pj.get("max_abs_delta_db")
                out["config"]["psnr_parity_source"] = "profiles/psnr_parity.json: " + pj.get("what", "")
            except Exception:
                pass
        cfg2_file = os.path.join(ROOT, "profiles", "psnr_parity_cfg2.json")
        if os.path.exists(cfg2_file):   # round 5: against the WHOLE reference on this GPU at BASELINE cfg2 size (tools/psnr_parity_cfg2.py)
            try:
                pj = json.load(open(cfg2_file))
                out["config"]["psnr_cfg2_mean_delta_vs_reference_stack_db"] = pj["summary"]["delta_of_means_db"]
                out["config"]["psnr_cfg2_runs"] = {"reference": pj["summary"]["train"]["reference"]["n"], "product": pj["summary"]["train"]["product"]["n"]}
                out["config"]["psnr_cfg2_single_run_std_db"] = {"reference": round(pj["summary"]["train"]["reference"]["std"], 4),
                                                                "product": round(pj["summary"]["train"]["product"]["std"], 4)}
                out["config"]["psnr_cfg2_source"] = "profiles/psnr_parity_cfg2.json (NOT collected in this run)"
                out["config"]["reference_stack_on_this_gpu_ms_per_iteration_cfg2"] = pj["ms_per_iteration"]["reference_stack_on_mi355x"]
            except Exception:
                pass
        cfg3_file = os.path.join(ROOT, "profiles", "psnr_parity_cfg3.json")
        if os.path.exists(cfg3_file):   # round 6: the same experiment at the HEADLINE configuration (1.2 M Gaussians), across an opacity reset
            try:
                pj = json.load(open(cfg3_file))
                out["config"]["psnr_cfg3_mean_delta_vs_reference_stack_db"] = pj["mean_psnr_delta_db"]
                out["config"]["psnr_cfg3_reference_vs_itself_db"] = pj.get("reference_vs_reference_again", {}).get("mean_psnr_delta_db")
                out["config"]["psnr_cfg3_source"] = "profiles/psnr_parity_cfg3.json (NOT collected in this run): " + pj.get("what", "")[:200]
                out["config"]["reference_stack_on_this_gpu_ms_per_iteration_cfg3"] = pj["ms_per_iteration"]["reference_stack_on_mi355x"]
            except Exception:
                pass
        if world == 1 and not dist_on and not a.no_cpu_baseline:
            try:
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):      # the reference's modules print; stdout carries ONE JSON line
                    out["cpu_baseline"] = cpu_baseline(a.P, a.width, a.height)
            except Exception as ex:  # the baseline must never take the headline number down with it
                out["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(ex).__name__}: {ex}"}
        print(json.dumps(out))
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
