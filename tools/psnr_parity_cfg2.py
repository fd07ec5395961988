#!/usr/bin/env python
"""PSNR parity at a BASELINE configuration (north_star: "PSNR within 0.1 dB of the reference").

Two trainings of the SAME scene by the reference's OWN `train.py::scene_reconstruction` (train.py:216-560, unchanged, from
oracle/_ref/reference_py.tar.gz), same initial state, same view order (shared `random` seed), same targets, same densify / prune
schedule (the reference's own rules, scene/gaussian_model.py:661-678) and the same torch RNG seed for densify_and_split:

  reference : the WHOLE reference on the MI355X -- its Python, its `diff_gaussian_rasterization` wrapper, its own kernels
              (forward.cu / backward.cu / rasterizer_impl.cu compiled for gfx950 by oracle/build_ref.sh, bound by
              oracle/ref_diff_raster_C.py), its plain-PyTorch HexPlane + MLP, torch SSIM, torch.optim.Adam.  No product code.
  product   : the same train.py on this repo's drop-in packages under `s3gaussian_amd.patch.patch_reference()` (fused sampler + MFMA
              MLP, two-image rasterizer passes, fused losses, one-launch Adam) -- zero file edits.

Afterwards every training view and every held-out view is rendered by each side's own `render()` and compared with the target.
Reported: split-mean PSNR difference (bar 0.1 dB), per-view rows, loss curves, point counts across the densify / prune events, and
the wall time per iteration of both stacks (the first number in this repository for the reference running end to end on this GPU).

    python tools/psnr_parity_cfg2.py                      # cfg2: 600 k Gaussians, 1066x1600, 1000 iterations  -> profiles/psnr_parity_cfg2.json
    python tools/psnr_parity_cfg2.py --P 60000 --width 480 --height 320 --iters 150      # what the -m gpu suite runs
    python tools/psnr_parity_cfg2.py --P 1200000 --opacity-reset-at 600 --reference-twice --product-runs 3 --out gpurun_out/psnr_parity_cfg3.json
                                                          # round 6: the headline configuration (cfg3), across an opacity reset
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _psnr(a, b):
    return float(10.0 * torch.log10(1.0 / ((a - b) ** 2).mean().clamp_min(1e-12)))


def run(P=600_000, W=1600, H=1066, iters=1000, n_frames=6, seed=0, densify_at=None, prune_at=None, grad_threshold=0.0002,
        opacity_threshold=0.005, verbose=True, reference_twice=False, eval_at=(), product_runs=1, reference=True, opacity_reset_at=None):
    """reference_twice: train the reference stack a second time from the same state and seeds -- its backward sums with float atomics
    (backward.cu:550-587), so two runs of the REFERENCE ITSELF are a chaotic pair too; their distance is the yardstick the
    product-vs-reference distance has to be read against.  eval_at: iterations at which every view is also rendered and scored
    inside the run (from the loop's own `timer.pause()` call, under its no_grad block; rendering draws no random numbers).
    product_runs > 1: the product is trained that many times from the same state and seeds (its HexPlane scatter and weight-gradient
    flush sum with float atomics, so it is not bit-reproducible either): the spread of ITS split-mean PSNR.  reference=False: skip
    the reference stack (200 ms per iteration) and return only the product runs (-> {"product_runs": [...]})."""
    from types import SimpleNamespace
    from oracle import ref_py
    from s3gaussian_amd import raster_C, synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, render as product_render
    dev = torch.device("cuda:0")
    densify_at = densify_at if densify_at is not None else iters // 2 + 1     # (twice that is past the last iteration: ONE event)
    prune_at = prune_at if prune_at is not None else (3 * iters) // 4
    scn = synth.street_scene(P=P, seed=seed, width=W, height=H, n_frames=n_frames)
    cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in scn["cameras"]]
    ids = list(range(len(cams)))
    test_ids = ids[1::3]                      # one of the three cameras of every frame is held out
    train_ids = [i for i in ids if i not in test_ids]
    bg = scn["bg"].to(dev)

    # ---- targets: the scene with perturbed positions / colours, rendered once (synchronous product forward; both sides share them) ----
    hyper0 = default_hyper()
    torch.manual_seed(seed)
    pc = GaussianParams(3, hyper0)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    init_state = {k: v.detach().clone() for k, v in pc._deformation.state_dict().items()}
    pipe0 = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    targets = {}
    prev = raster_C.set_async(False)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        pc._xyz.data.add_(0.05 * torch.randn(pc._xyz.shape, generator=g).to(dev))
        pc._features_dc.data.add_(0.5 * torch.randn(pc._features_dc.shape, generator=g).to(dev))
        for v in ids:
            pk = product_render(cams[v], pc, pipe0, bg, stage="fine", render_feat=True)
            targets[v] = (pk["render"].clamp(0, 1).clone(), pk["depth"].clone(), pk["feat"].clone())
    raster_C.set_async(prev)
    del pc
    torch.cuda.empty_cache()

    out = {}
    sides = (("reference", "reference_again") if reference_twice else ("reference",)) if reference else ()
    sides = sides + ("product",) + tuple(f"product_{k}" for k in range(2, product_runs + 1))
    for side in sides:
        is_product = side.startswith("product")
        ref = ref_py.load(patch=is_product, rasterizer="dropin" if is_product else "reference")
        try:
            args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
            dataset.render_process = False
            # one densify and one prune event inside the run, by the reference's own schedule arithmetic (train.py:500-509)
            opt.densify_from_iter, opt.densification_interval = densify_at - 1, densify_at
            opt.pruning_from_iter, opt.pruning_interval = prune_at - 1, prune_at
            # opacity_reset_at = N: reset_opacity at iteration N (train.py:514-516; only multiples of N below the last iteration), and
            # every densify / prune after N runs with size_threshold = 20 (train.py:502-508) -- VERDICT r5: "never across an opacity reset"
            opt.opacity_reset_interval = opacity_reset_at if opacity_reset_at else 10 ** 9
            opt.densify_grad_threshold_fine_init = opt.densify_grad_threshold_after = grad_threshold
            opt.opacity_threshold_fine_init = opt.opacity_threshold_fine_after = opacity_threshold
            torch.manual_seed(seed)
            gm = ref_py.make_gaussians(ref, gs, scn["aabb"], hyper)
            gm._deformation.load_state_dict(init_state)
            cam_objs = {v: ref_py.make_camera(ref, cams[v], targets[v], uid=v) for v in ids}
            scene = ref_py.SceneStub([cam_objs[v] for v in train_ids], cameras_extent=50.0)
            random.seed(seed + 1)
            torch.manual_seed(seed + 1)
            timer = ref_py.RecordingTimer(record_locals=True)
            checkpoints = {}

            def score(gm=gm, ref=ref, cam_objs=cam_objs, pipe=pipe):
                with torch.no_grad():
                    return {v: _psnr(ref.gaussian_renderer.render(cam_objs[v], gm, pipe, bg, stage="fine")["render"].clamp(0, 1),
                                     targets[v][0]) for v in ids}

            if eval_at:
                def at_iteration(it):
                    if it in eval_at:
                        checkpoints[it] = score()

                timer.after_pause = at_iteration
            t0 = time.perf_counter()
            ref_py.run_scene_reconstruction(ref, gm, scene, dataset, hyper, opt, pipe, iterations=iters, stage="fine", timer=timer)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            rows = score()
            st = timer.stamps
            steady = 1000.0 * (st[-1] - st[len(st) // 10]) / max(len(st) - 1 - len(st) // 10, 1)
            out[side] = dict(psnr=rows, checkpoints=checkpoints, losses=timer.losses, points=timer.points, wall_s=wall, ms_per_iteration=steady,
                             optimizer=type(gm.optimizer).__module__ + "." + type(gm.optimizer).__name__,
                             deformation=type(gm._deformation).__module__,
                             rasterizer=sys.modules["diff_gaussian_rasterization"].__file__.replace(ref.root, "<archive>").replace(ROOT, "<repo>"))
            if verbose:
                print(f"[{side}] {iters} iterations in {wall:.1f} s ({steady:.2f} ms/iteration), points {timer.points[0]} -> {timer.points[-1]}, "
                      f"loss {np.mean(timer.losses[:10]):.4f} -> {np.mean(timer.losses[-10:]):.4f}", file=sys.stderr)
            del gm, scene, cam_objs
        finally:
            ref_py.unload()
            torch.cuda.empty_cache()

    def split_means(rows):
        return {sp: float(np.mean([rows[v] for v in ids if (v in test_ids) == (sp == "test")])) for sp in ("train", "test")}

    runs = [dict(run=sd, mean_psnr_db=split_means(out[sd]["psnr"]), points_final=out[sd]["points"][-1],
                 loss_last10_mean=float(np.mean(out[sd]["losses"][-10:])), ms_per_iteration=round(out[sd]["ms_per_iteration"], 3))
            for sd in out if sd.startswith("product")]
    if not reference:
        return dict(what=f"the product trained {product_runs} times from the same state and seeds: {iters} iterations, {P} Gaussians, {H}x{W}",
                    product_runs=runs)
    a, b = out["product"], out["reference"]
    views = [dict(view=v, split="test" if v in test_ids else "train", psnr_product=a["psnr"][v], psnr_reference=b["psnr"][v],
                  delta_db=a["psnr"][v] - b["psnr"][v]) for v in ids]
    mean_delta = {sp: float(np.mean([r["psnr_product"] for r in views if r["split"] == sp]) -
                            np.mean([r["psnr_reference"] for r in views if r["split"] == sp])) for sp in ("train", "test")}
    la, lb = np.array(a["losses"]), np.array(b["losses"])
    gaps = np.abs(la - lb) / np.maximum(np.abs(lb), 1e-12)
    k = max(iters // 50, 1)
    rec = dict(
        what=f"{iters} fine-stage iterations of the reference's own train.py::scene_reconstruction, {P} Gaussians, {H}x{W}, "
             f"{len(train_ids)} train + {len(test_ids)} held-out views, one densify event (iteration {densify_at}) and one prune event "
             f"(iteration {prune_at})" + (f", reset_opacity at every multiple of {opacity_reset_at} (screen-size pruning on from there)" if opacity_reset_at else "") +
             f" by the reference's own rules, shared seeds: product (drop-in packages + patch_reference) vs the "
             "whole reference on this GPU (its Python + its own kernels built for gfx950 + torch.optim.Adam)",
        mean_psnr_delta_db=mean_delta, max_abs_delta_db=float(max(abs(r["delta_db"]) for r in views)), views=views,
        mean_psnr_db={sp: {"product": float(np.mean([r["psnr_product"] for r in views if r["split"] == sp])),
                           "reference": float(np.mean([r["psnr_reference"] for r in views if r["split"] == sp]))} for sp in ("train", "test")},
        points={"product": [a["points"][0], a["points"][min(densify_at, iters - 1)], a["points"][-1]],
                "reference": [b["points"][0], b["points"][min(densify_at, iters - 1)], b["points"][-1]],
                "at_iterations": [1, densify_at + 1, iters]},
        max_rel_loss_gap_first20=float(gaps[:20].max()), rel_loss_gap_at={str(i): float(gaps[i]) for i in (0, 1, 5, 20, iters // 4, iters // 2, iters - 1)},
        loss_first10_mean=float(lb[:10].mean()), loss_last10_mean={"product": float(la[-10:].mean()), "reference": float(lb[-10:].mean())},
        loss_curve_every=k, loss_curve={"product": [round(float(x), 5) for x in la[::k]], "reference": [round(float(x), 5) for x in lb[::k]]},
        ms_per_iteration={"product": round(a["ms_per_iteration"], 3), "reference_stack_on_mi355x": round(b["ms_per_iteration"], 3)},
        stacks={sd: {kk: out[sd][kk] for kk in ("optimizer", "deformation", "rasterizer")} for sd in out})

    if product_runs > 1:
        rec["product_runs"] = runs
    if eval_at:     # where along the run do the two trainings separate?
        rec["mean_psnr_delta_db_at_iteration"] = {
            str(it): {sp: split_means(a["checkpoints"][it])[sp] - split_means(b["checkpoints"][it])[sp] for sp in ("train", "test")}
            for it in sorted(a["checkpoints"]) if it in b["checkpoints"]}
    if reference_twice:
        c = out["reference_again"]
        lc = np.array(c["losses"])
        rec["reference_vs_reference_again"] = dict(
            what="the reference stack trained twice from the same state and seeds (its float-atomic backward is not reproducible): the "
                 "distance between two runs of the REFERENCE ITSELF, same measures as above",
            mean_psnr_delta_db={sp: split_means(c["psnr"])[sp] - split_means(b["psnr"])[sp] for sp in ("train", "test")},
            max_abs_delta_db=float(max(abs(c["psnr"][v] - b["psnr"][v]) for v in ids)),
            per_view_delta_db=[round(c["psnr"][v] - b["psnr"][v], 4) for v in ids],
            points=[c["points"][0], c["points"][min(densify_at, iters - 1)], c["points"][-1]],
            rel_loss_gap_at={str(i): float(abs(lc[i] - lb[i]) / max(abs(lb[i]), 1e-12)) for i in (0, 1, 5, 20, iters // 4, iters // 2, iters - 1)},
            mean_psnr_delta_db_at_iteration={
                str(it): {sp: split_means(c["checkpoints"][it])[sp] - split_means(b["checkpoints"][it])[sp] for sp in ("train", "test")}
                for it in sorted(c["checkpoints"]) if it in b["checkpoints"]})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=600_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1066)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_parity_cfg2.json"))
    ap.add_argument("--reference-twice", action="store_true", help="also train the reference stack a second time: its own run-to-run spread")
    ap.add_argument("--eval-at", type=int, nargs="*", default=[], help="iterations at which every view is scored inside the run")
    ap.add_argument("--product-runs", type=int, default=1, help="train the product this many times (its own run-to-run spread)")
    ap.add_argument("--no-reference", action="store_true", help="product runs only (the reference stack costs 200 ms per iteration)")
    ap.add_argument("--opacity-reset-at", type=int, default=None, help="opt.opacity_reset_interval (default: never inside the run)")
    a = ap.parse_args()
    rec = run(a.P, a.width, a.height, a.iters, a.frames, reference_twice=a.reference_twice, eval_at=tuple(a.eval_at),
              product_runs=a.product_runs, reference=not a.no_reference, opacity_reset_at=a.opacity_reset_at)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("mean_psnr_delta_db", "max_abs_delta_db", "mean_psnr_db", "points", "ms_per_iteration",
                                           "max_rel_loss_gap_first20", "loss_last10_mean", "mean_psnr_delta_db_at_iteration",
                                           "reference_vs_reference_again", "product_runs") if k in rec}))


if __name__ == "__main__":
    main()
