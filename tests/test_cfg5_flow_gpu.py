"""BASELINE cfg5 flow on the GPU (arguments/stage2_nvs.py:1-11, scene/gaussian_model.py:397-494): training steps -> a densify +
prune cycle that edits the optimizer the way the reference does (P grows, then shrinks; nn.Parameter objects replaced, Adam
moments carried for survivors) -> more training steps; and capture() -> torch.save -> restore() -> identical next step (the
stage-2 warm start from `prior_checkpoint`).

The densify / prune DECISIONS are the reference's control plane and stay with it; the helpers below only replay the optimizer
surgery those decisions perform (`cat_tensors_to_optimizer` / `_prune_optimizer` semantics: single-parameter groups edited,
`exp_avg` / `exp_avg_sq` extended with zeros or boolean-masked, multi-parameter groups -- the deformation network -- skipped),
because every per-point buffer of the accelerated path has to survive it: the HexPlane `sort_state` (25 words per point), the MLP
stash and mask words, the rasterizer's geometry cache and arenas, the fused Adam's tensor list, the densification accumulators.
Also: training_step(densify_stats=True) under pipe.debug=True and with P == 0 after a prune."""
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

PER_GAUSSIAN = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                "rotation": "_rotation"}


def _rebind(pc, tensors):
    for name, attr in PER_GAUSSIAN.items():
        setattr(pc, attr, tensors[name])


def _append_points(pc, new):
    """scene/gaussian_model.py:446-492: extend every single-parameter group and its moments (zeros for the newcomers)."""
    out = {}
    for group in pc.optimizer.param_groups:
        if len(group["params"]) > 1:
            continue
        old = group["params"][0]
        ext = new[group["name"]]
        st = pc.optimizer.state.get(old, None)
        fresh = nn.Parameter(torch.cat((old, ext), dim=0).requires_grad_(True))
        if st is not None:
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del pc.optimizer.state[old]
            pc.optimizer.state[fresh] = st
        group["params"][0] = fresh
        out[group["name"]] = fresh
    _rebind(pc, out)
    n = pc._xyz.shape[0]
    dev = pc._xyz.device
    pc._deformation_table = torch.cat([pc._deformation_table, torch.ones(n - pc._deformation_table.shape[0], dtype=torch.bool, device=dev)])
    pc.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    pc.denom = torch.zeros((n, 1), device=dev)
    pc.max_radii2D = torch.zeros(n, device=dev)


def _remove_points(pc, drop):
    """scene/gaussian_model.py:413-444: boolean-mask every single-parameter group, its moments and the accumulators."""
    keep = ~drop
    out = {}
    for group in pc.optimizer.param_groups:
        if len(group["params"]) > 1:
            continue
        old = group["params"][0]
        st = pc.optimizer.state.get(old, None)
        fresh = nn.Parameter(old[keep].requires_grad_(True))
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
            del pc.optimizer.state[old]
            pc.optimizer.state[fresh] = st
        group["params"][0] = fresh
        out[group["name"]] = fresh
    _rebind(pc, out)
    pc.xyz_gradient_accum, pc.denom = pc.xyz_gradient_accum[keep], pc.denom[keep]
    pc._deformation_table, pc.max_radii2D = pc._deformation_table[keep], pc.max_radii2D[keep]


def _setup(dev, P=60_000, W=480, H=320, seed=3):
    from s3gaussian_amd import synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, render
    scn = synth.street_scene(P=P, seed=seed, width=W, height=H, n_frames=3)
    hyper, opt = default_hyper(), default_opt()
    torch.manual_seed(0)
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    pc.training_setup(opt)
    bg = scn["bg"].to(dev)
    cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in scn["cameras"][:6]]
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        targets = []
        for cam in cams:
            pkg = render(cam, pc, pipe, bg, stage="fine", render_feat=True)
            targets.append((pkg["render"].clamp(0, 1).clone(), pkg["depth"].clone(), pkg["feat"].clone()))
        g = torch.Generator().manual_seed(1)
        pc._features_dc.add_(0.3 * torch.randn(pc._features_dc.shape, generator=g).to(dev))
        pc._opacity.add_(0.5 * torch.randn(pc._opacity.shape, generator=g).to(dev))
    return pc, cams, targets, hyper, opt, bg


def test_densify_prune_cycle_between_training_steps(gpu_device):
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.pipeline import training_step
    dev = gpu_device
    pc, cams, targets, hyper, opt, bg = _setup(dev)
    grid = pc._deformation.deformation_net.grid
    losses = []

    def steps(n, it0):
        for it in range(it0, it0 + n):
            v = it % len(cams)
            loss, _ = training_step(pc, cams[v], *targets[v], hyper, opt, bg, stage="fine", densify_stats=True)
            losses.append(loss.item())

    steps(30, 0)
    P0 = pc._xyz.shape[0]
    from s3gaussian_amd.hexplane import sort_state_words
    W_ = sort_state_words(4)          # 6 * levels + 1 words per point since round 4 (one walk order per orientation and level)
    assert W_ == 25 and grid._order_cache["sort_state"].numel() == W_ * P0
    assert float(pc.denom.max()) > 0 and float(pc.xyz_gradient_accum.max()) > 0          # bookkeeping ran inside the backward
    # --- densify (clone the 20 % with the largest mean viewspace gradient), then prune (drop the 10 % least opaque) ------------
    score = (pc.xyz_gradient_accum / pc.denom.clamp_min(1)).squeeze(1)
    sel = score >= torch.quantile(score, 0.8)
    n_new = int(sel.sum())
    m_before = {n: pc.optimizer.state[getattr(pc, a)]["exp_avg"].clone() for n, a in PER_GAUSSIAN.items()}
    step_before = float(pc.optimizer.state[pc._xyz]["step"])
    _append_points(pc, {n: getattr(pc, a).detach()[sel].clone() for n, a in PER_GAUSSIAN.items()})
    assert pc._xyz.shape[0] == P0 + n_new
    drop = pc._opacity.detach().squeeze(1) < torch.quantile(pc._opacity.detach().squeeze(1), 0.1)
    survivors_of_old = ~drop[:P0]
    _remove_points(pc, drop)
    P1 = pc._xyz.shape[0]
    assert P1 != P0 and P1 == P0 + n_new - int(drop.sum())
    for n, a in PER_GAUSSIAN.items():      # Adam moments carried for the survivors, zeros for the clones
        st = pc.optimizer.state[getattr(pc, a)]
        k = int(survivors_of_old.sum())
        assert torch.equal(st["exp_avg"][:k], m_before[n][survivors_of_old]), n
        assert float(st["step"]) == step_before
    hits = raster_C._geom_cache_hits
    steps(1, 30)
    assert raster_C._geom_cache_hits == hits                                              # new tensors: geometry cache missed
    assert grid._order_cache["sort_state"].numel() == W_ * P1 and grid._order_cache["sort_age"] == 0   # re-sorted at the new size
    assert pc.xyz_gradient_accum.shape == (P1, 1) and pc.max_radii2D.shape == (P1,)
    assert float(pc.optimizer.state[pc._xyz]["step"]) == step_before + 1
    steps(29, 31)
    assert all(np.isfinite(losses))
    assert np.mean(losses[-6:]) < np.mean(losses[:6]), (losses[:6], losses[-6:])
    # --- the same surgery down to nothing: P == 0 must render zeros and step without touching the per-Gaussian groups ----------
    _remove_points(pc, torch.ones(P1, dtype=torch.bool, device=dev))
    loss, pkg = training_step(pc, cams[0], *targets[0], hyper, opt, bg, stage="fine", densify_stats=True)
    assert torch.isfinite(loss) and pkg["radii"].numel() == 0 and not pkg["densify_stats_fused"]
    assert float(pkg["render"].abs().max()) == 0.0                                        # rasterize_points.cu:81-116


def test_training_step_with_debug_snapshots_on(gpu_device, tmp_path, monkeypatch):
    """pipe.debug=True: the two-image node falls back to two ordinary nodes, so the bookkeeping must take the separate pass
    instead of raising (pipeline.py:372); results equal the fused run."""
    from s3gaussian_amd.pipeline import training_step
    dev = gpu_device
    monkeypatch.chdir(tmp_path)          # a failing debug forward would write snapshot_fw.dump into the cwd
    res = {}
    for debug in (False, True):
        pc, cams, targets, hyper, opt, bg = _setup(dev, P=8000, W=160, H=112, seed=5)
        pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=debug)
        loss, pkg = training_step(pc, cams[0], *targets[0], hyper, opt, bg, stage="fine", pipe=pipe, densify_stats=True)
        assert pkg["densify_stats_fused"] == (not debug)
        res[debug] = (loss.item(), pc.xyz_gradient_accum.clone(), pc.denom.clone(), pc.max_radii2D.clone())
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0])
    assert torch.equal(res[True][2], res[False][2]) and torch.equal(res[True][3], res[False][3])
    np.testing.assert_allclose(res[True][1].cpu().numpy(), res[False][1].cpu().numpy(), rtol=1e-4, atol=1e-9)


def test_capture_save_restore_then_identical_next_step(gpu_device):
    """stage-2 warm start (arguments/stage2_nvs.py: `prior_checkpoint`): the 14-tuple of capture() through torch.save /
    torch.load into a FRESH GaussianParams; the next training step of the restored model is bit-identical to the original's."""
    from s3gaussian_amd.pipeline import GaussianParams, training_step
    dev = gpu_device
    pc, cams, targets, hyper, opt, bg = _setup(dev, P=20_000, W=240, H=160, seed=9)
    for it in range(6):
        training_step(pc, cams[it % 3], *targets[it % 3], hyper, opt, bg, stage="fine", densify_stats=True)
    buf = io.BytesIO()
    torch.save((pc.capture(), 6), buf)
    buf.seek(0)
    model_args, first_iter = torch.load(buf, map_location=dev, weights_only=False)
    assert first_iter == 6
    pc2 = GaussianParams(3, hyper)
    pc2._deformation = pc2._deformation.to(dev)
    pc2.restore(model_args, opt)
    assert pc2._xyz.is_cuda and torch.equal(pc2._xyz, pc._xyz) and torch.equal(pc2.denom, pc.denom)
    for p in pc2._deformation.deformation_net.grid.grids.parameters():
        assert p.is_contiguous(memory_format=torch.channels_last)
    assert float(pc2.optimizer.state[pc2._xyz]["step"]) == 6.0
    l1, k1 = training_step(pc, cams[0], *targets[0], hyper, opt, bg, stage="fine", densify_stats=True)
    l2, k2 = training_step(pc2, cams[0], *targets[0], hyper, opt, bg, stage="fine", densify_stats=True)
    # rasterizer, MLP chain, glue and losses are bit-reproducible; plane / weight gradients are flushed with float atomics,
    # so the step itself is equal to summation-order round-off
    assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
    assert torch.equal(k1["radii"], k2["radii"]) and torch.equal(k1["render"], k2["render"])
    for (n, a), (_, b) in zip(pc.named_parameters(), pc2.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), n


def test_an_overflow_after_a_densify_event_is_replayed_not_dropped(gpu_device, monkeypatch):
    """VERDICT r4 item 8.  Densification is exactly when the instance count jumps.  Segment 1: 6 iterations; "densify": every
    Gaussian cloned with the reference's optimizer surgery (P doubles) while the capacity policy is made to believe that small
    views are all it has seen; segment 2: 6 more iterations through pipeline.run_training_steps.  The first forward after the event
    overflows its arena, the device freezes the model (sticky word), the host notices a few iterations later, rewinds and re-issues:
    the model ends where a run with the SYNCHRONOUS forward (the reference's one wait per call) ends, every parameter's Adam step
    count is 12 and the statistics count 12 views -- nothing dropped, nothing applied twice."""
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.pipeline import run_training_steps, training_step
    dev = gpu_device
    res = {}
    for mode in ("sync", "sync_again", "replay"):
        prev_async = raster_C.set_async(mode == "replay")
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()
        try:
            pc, cams, targets, hyper, opt, bg = _setup(dev, P=30_000, W=320, H=208, seed=5)
            losses = {}

            def issue(i):
                v = i % len(cams)
                loss, _ = training_step(pc, cams[v], *targets[v], hyper, opt, bg, stage="fine", densify_stats=True)
                losses[i] = loss            # a re-issued iteration overwrites the (meaningless) loss of its dropped first attempt

            log = []
            out1 = run_training_steps(issue, 1, 6, optimizer=pc.optimizer, device=dev, log=log)
            assert out1["rewinds"] == [] and log == [1, 2, 3, 4, 5, 6]
            torch.cuda.synchronize()
            with torch.no_grad():       # "densify_and_clone" of every Gaussian (scene/gaussian_model.py:522-560 + :446-492)
                _append_points(pc, {n: getattr(pc, a).detach().clone() for n, a in PER_GAUSSIAN.items()})
            if mode == "replay":
                st = raster_C._async_state(dev)
                st.drain(block=True)
                key = (320, 208)
                true_R = st.hist[key][0]
                monkeypatch.setattr(raster_C, "_ASYNC_MIN_INSTANCES", 1)
                st.hist[key] = [true_R // 8, true_R // 8, st.hist[key][2]]      # 4 x headroom < the doubled scene's count
                assert st.caps(key)[0] < 2 * true_R
            log = []
            out2 = run_training_steps(issue, 7, 12, optimizer=pc.optimizer, device=dev, log=log)
            torch.cuda.synchronize()
            if mode == "replay":
                assert len(out2["rewinds"]) >= 1 and out2["rewinds"][0][0] == 7, out2      # iteration 7 overflowed and was re-issued
                assert log[0] == 7 and log.count(7) >= 2 and log[-1] == 12 and out2["issued"] == len(log) > 6
                stt = raster_C.async_status(dev, block=True)
                assert len(stt["overflows"]) >= 1 and stt["replay"] is False      # (the loop restored the previous mode)
                assert int(raster_C._async_state(dev).sticky_dev.item()) == 0      # thawed
                print("replay:", out2, "issue order:", log, "frozen forwards:", len(stt["frozen"]))
            else:
                assert out2["rewinds"] == [] and log == [7, 8, 9, 10, 11, 12]
            assert all(float(s["step"]) == 12.0 for s in pc.optimizer.state.values() if "step" in s)
            res[mode] = dict(losses=[float(losses[i]) for i in range(1, 13)], params={n: p.detach().clone() for n, p in pc.named_parameters()},
                             denom=pc.denom.clone(), accum=pc.xyz_gradient_accum.clone())
        finally:
            raster_C.set_async(prev_async)
            raster_C._async_states.pop(dev.index or 0, None)
            raster_C.invalidate_geometry_cache()

    def distance(a, b):
        frac = max(float((~torch.isclose(a["params"][n], b["params"][n], rtol=1e-4, atol=1e-6)).float().mean()) for n in a["params"])
        lrel = max(abs(x - y) / abs(x) for x, y in zip(a["losses"], b["losses"]))
        return frac, lrel

    floor = distance(res["sync"], res["sync_again"])        # two synchronous runs differ too (float atomics in the HexPlane scatter)
    got = distance(res["sync"], res["replay"])
    print("not-close fraction / max relative loss difference: sync vs sync", floor, " sync vs replay", got)
    assert got[0] <= max(2e-3, 3 * floor[0]) and got[1] <= max(1e-4, 3 * floor[1]), (got, floor)
    assert torch.equal(res["sync"]["denom"], res["replay"]["denom"])      # views 7..12 counted once each for the 60 000 Gaussians
    assert float(res["replay"]["denom"].sum()) > 0
