"""PSNR parity (third part of BASELINE.json's metric, north_star: "PSNR within 0.1 dB of the reference").

Two trainings from the SAME initial state, the same view order and the same targets:

  GPU   : s3gaussian_amd.pipeline.training_step -- the fused product path (HIP HexPlane + MFMA MLP + glue + two-image
          rasterizer + fused losses + single-launch Adam), exactly what bench.py times;
  oracle: the reference's algorithm on the CPU -- oracle/hexplane_ref.py (restatement of scene/hexplane.py +
          scene/deformation.py + glue + losses, pinned by goldens generated from the reference's own modules) + the
          reference-pinned C rasterizer oracle fwd/bwd called twice (RGB+depth, feature image) + torch.optim.Adam with the
          reference's groups and eps (scene/gaussian_model.py:177-189), train.py:395-437,521-522 loss assembly.

After K iterations both models are rendered on every training view AND on held-out views.  Bars: the MEAN PSNR over the training
views and over the held-out views (what an evaluation reports) within 0.1 dB of the oracle path's; every single view within 0.3 dB;
the two loss trajectories within 1e-3 relative over the first 20 iterations.  Two fp32 trainings that sum in different orders are
a chaotic pair: Adam's first steps move every parameter by lr * sign(g), so round-off in a gradient that is itself a cancellation
decides a direction, and the trajectories separate gradually (recorded: `rel_loss_gap_at`) -- the reference's own backward is not
even run-to-run reproducible (float atomics, backward.cu:550-587).  At 12 k Gaussians / 160 iterations (round 2) the worst view
differed by 0.006 dB; at 100 k / 300 iterations single views drift to 0.2 dB while the means stay within 0.03 dB.
Results go to gpurun_out/psnr_parity.json (committed copy: profiles/psnr_parity.json, which bench.py reports as
config.psnr_delta_vs_oracle_db)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# > 65 536 points: every MLP wave loops over more than one tile; V < P.  The oracle side costs ~2.2 s per iteration on the GPU box's
# host cores: the routine suite runs 40 iterations (~1.5 min; round 6: the whole suite has to stay inside the driver's window, and the
# PSNR evidence proper now comes from the WHOLE reference on the GPU at cfg2 / cfg3 size -- test_psnr_parity_cfg2_gpu.py,
# profiles/psnr_parity_cfg3*.json; 80, 120 and 200 were run and recorded in earlier rounds: same bars, deltas of 0.004-0.006 dB);
# S3G_PSNR_ITERS=200 reproduces the committed profiles/psnr_parity.json
P, W, H, K = 100_000, 480, 320, int(os.environ.get("S3G_PSNR_ITERS", "40"))


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * np.log10(1.0 / max(mse, 1e-12))


def test_training_on_the_gpu_path_reaches_the_psnr_of_the_oracle_path(gpu_device):
    from types import SimpleNamespace
    from oracle import hexplane_ref as hr
    from oracle.oracle import RasterOracle
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, render, training_step
    dev = gpu_device
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    small = dict(kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32, resolution=[16, 16, 16, 8]))
    hyper, opt = default_hyper(**small), default_opt()
    scn = synth.street_scene(P=P, seed=3, width=W, height=H, n_frames=3)
    gs = scn["gaussians"]
    cams_cpu = scn["cameras"]                      # 9 views (3 frames x 3 cameras): 6 train, 3 held out
    train_ids, test_ids = [0, 1, 2, 6, 7, 8], [3, 4, 5]
    bg = scn["bg"]

    # ---- GPU model + targets (render of a perturbed copy, like bench.py) ----
    torch.manual_seed(0)
    pc = GaussianParams(3, hyper)
    pc.init_from_tensors(gs["xyz"], gs["log_scales"] + 0.9, gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    init_state = {k: v.detach().cpu().clone() for k, v in pc._deformation.state_dict().items()}
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in cams_cpu]
    targets = {}
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        xyz0, sh0 = pc._xyz.data.clone(), pc._features_dc.data.clone()
        pc._xyz.data.add_(0.05 * torch.randn(xyz0.shape, generator=g).to(dev))
        pc._features_dc.data.add_(0.5 * torch.randn(sh0.shape, generator=g).to(dev))
        for v in train_ids + test_ids:
            pk = render(cams[v], pc, pipe, bg.to(dev), stage="fine", render_feat=True)
            targets[v] = (pk["render"].clamp(0, 1).clone(), pk["depth"].clone(), pk["feat"].clone())
        pc._xyz.data.copy_(xyz0)
        pc._features_dc.data.copy_(sh0)
    pc.training_setup(opt)
    order = [train_ids[i] for i in torch.randint(0, len(train_ids), (K,), generator=torch.Generator().manual_seed(1)).tolist()]

    # ---- oracle model: identical initial values, the reference's optimizer ----
    net = hr.deform_network(hr.default_hyper(**small))
    net.load_state_dict(init_state)
    leaves = dict(xyz=gs["xyz"].clone(), f_dc=gs["shs"][:, :1].clone(), f_rest=gs["shs"][:, 1:].clone(),
                  sc=(gs["log_scales"] + 0.9).clone(), rot=gs["rotations_raw"].clone(), op=gs["opacity_logit"].clone())
    leaves = {k: v.float().contiguous().requires_grad_(True) for k, v in leaves.items()}
    mlp_params = [p for n, p in net.named_parameters() if "grid" not in n]
    grid_params = [p for n, p in net.named_parameters() if "grid" in n]
    adam = torch.optim.Adam([
        {"params": [leaves["xyz"]], "lr": opt.position_lr_init}, {"params": mlp_params, "lr": opt.deformation_lr_init},
        {"params": grid_params, "lr": opt.grid_lr_init}, {"params": [leaves["f_dc"]], "lr": opt.feature_lr},
        {"params": [leaves["f_rest"]], "lr": opt.feature_lr / 20.0}, {"params": [leaves["op"]], "lr": opt.opacity_lr},
        {"params": [leaves["sc"]], "lr": opt.scaling_lr}, {"params": [leaves["rot"]], "lr": opt.rotation_lr}], lr=0.0, eps=1e-15)
    orc = RasterOracle(np.float32)
    tcpu = {v: tuple(t.cpu() for t in targets[v]) for v in targets}

    def oracle_render(cam, want_grad):
        time_t = torch.full((P, 1), cam["time"])
        shs0 = torch.cat([leaves["f_dc"], leaves["f_rest"]], 1)
        m3, s, r, o, shs, dx, feat, dshs = net(leaves["xyz"], leaves["sc"], leaves["rot"], leaves["op"], shs0, time_t)
        scales, rots, opac = torch.exp(s), torch.nn.functional.normalize(r), torch.sigmoid(o)
        cols = hr.shs_to_colors(3, shs, leaves["xyz"], cam["campos"])
        kw = dict(bg=bg.numpy(), viewmatrix=cam["viewmatrix"].numpy(), projmatrix=cam["projmatrix"].numpy(),
                  campos=cam["campos"].numpy(), tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], image_height=H, image_width=W)
        fwd = [orc.forward(means3D=m3.detach().numpy(), opacities=opac.detach().numpy(), scales=scales.detach().numpy(),
                           rotations=rots.detach().numpy(), colors_precomp=c.detach().numpy(), sh_degree=0, **kw) for c in (cols, feat)]
        return fwd, (m3, scales, rots, opac, cols, feat, dx, dshs)

    def oracle_step(v):
        cam = cams_cpu[v]
        gt, gtd, gtf = tcpu[v]
        adam.zero_grad(set_to_none=True)
        fwd, (m3, scales, rots, opac, cols, feat, dx, dshs) = oracle_render(cam, True)
        img = torch.from_numpy(fwd[0]["color"]).requires_grad_(True)
        dep = torch.from_numpy(fwd[0]["depth"]).requires_grad_(True)
        fimg = torch.from_numpy(fwd[1]["color"]).requires_grad_(True)
        loss = (hr.l1_loss(img[None], gt[None]) + opt.lambda_dssim * (1 - hr.ssim(img[None], gt[None]))
                + opt.lambda_depth * hr.depth_l2(dep, gtd) + opt.lambda_feat * hr.l2_loss(fimg, gtf))
        loss.backward()
        g1 = orc.backward(fwd[0], img.grad.numpy(), dep.grad.numpy())
        g2 = orc.backward(fwd[1], fimg.grad.numpy(), np.zeros((1, H, W), np.float32))
        t = torch.from_numpy
        regs = (opt.lambda_dx * dx.abs().mean() + opt.lambda_dshs * dshs.abs().mean()
                + hr.plane_regulation(net.deformation_net.grid.grids, hyper.time_smoothness_weight, hyper.l1_time_planes, hyper.plane_tv_weight))
        surrogate = ((m3 * t(g1["dL_dmeans3D"] + g2["dL_dmeans3D"])).sum() + (scales * t(g1["dL_dscales"] + g2["dL_dscales"])).sum()
                     + (rots * t(g1["dL_drotations"] + g2["dL_drotations"])).sum() + (opac * t(g1["dL_dopacity"] + g2["dL_dopacity"])).sum()
                     + (cols * t(g1["dL_dcolors"])).sum() + (feat * t(g2["dL_dcolors"])).sum() + regs)
        surrogate.backward()
        adam.step()
        # the SAME loss expression as pipeline.training_loss reports (train.py:395-425): pixel terms + dx / dshs / plane regularisers
        return float(loss.detach() + regs.detach())

    losses_gpu, losses_orc = [], []
    for v in order:
        lg, _ = training_step(pc, cams[v], *targets[v], hyper, opt, bg.to(dev), stage="fine", pipe=pipe)
        losses_gpu.append(float(lg))
        losses_orc.append(oracle_step(v))

    # ---- PSNR of both trained models on every view ----
    rows = []
    with torch.no_grad():
        for v in train_ids + test_ids:
            gt = targets[v][0]
            mine = render(cams[v], pc, pipe, bg.to(dev), stage="fine")["render"].clamp(0, 1)
            fwd, _ = oracle_render(cams_cpu[v], False)
            theirs = torch.from_numpy(fwd[0]["color"]).clamp(0, 1)
            rows.append(dict(view=v, split="train" if v in train_ids else "test", psnr_gpu=_psnr(mine.cpu(), gt.cpu()),
                             psnr_oracle=_psnr(theirs, gt.cpu())))
    for r in rows:
        r["delta_db"] = r["psnr_gpu"] - r["psnr_oracle"]
    worst = max(abs(r["delta_db"]) for r in rows)
    mean_delta = {sp: float(np.mean([r["psnr_gpu"] for r in rows if r["split"] == sp]) - np.mean([r["psnr_oracle"] for r in rows if r["split"] == sp]))
                  for sp in ("train", "test")}
    gaps = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses_gpu, losses_orc)]
    first = float(np.mean(losses_gpu[:10]))
    rec = dict(what=f"{K} fine-stage iterations, {P} Gaussians, {H}x{W}, {len(train_ids)} train + {len(test_ids)} held-out views, "
                    "same init / view order / targets: fused GPU path vs oracle path (CPU, reference algorithm, torch Adam)",
               max_abs_delta_db=worst, mean_psnr_delta_db=mean_delta, views=rows,
               rel_loss_gap_at={str(i): float(gaps[i]) for i in sorted({0, 1, 2, 5, 10, 20, 50, 100, 150, K - 1}) if i < K},
               max_rel_loss_gap_first20=float(max(gaps[:20])), loss_first10_mean=first, loss_last10_mean_gpu=float(np.mean(losses_gpu[-10:])),
               loss_last10_mean_oracle=float(np.mean(losses_orc[-10:])),
               max_rel_loss_gap=float(max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses_gpu, losses_orc))))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        name = "psnr_parity.json" if K >= 200 else f"psnr_parity_{K}it.json"
        json.dump(rec, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
    except OSError:
        pass
    assert rec["loss_last10_mean_gpu"] < 0.9 * first, rec          # it actually trained
    assert max(abs(v) for v in mean_delta.values()) <= 0.1, rec    # north_star: PSNR within 0.1 dB (mean over a split's views)
    assert worst <= 0.3, rec                                       # no single view drifts further than this
    assert rec["max_rel_loss_gap_first20"] <= 1e-3, rec            # same loss expression, same trajectory until chaos separates them
