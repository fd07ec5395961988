"""Configurations of the reference's training loop that tests/test_reference_py_gpu.py does not reach (VERDICT r5 "missing" #1), run
through the reference's OWN `train.py` (oracle/_ref/reference_py.tar.gz, unchanged) on both routes -- the two drop-in packages alone
("zero_diff") and under `s3gaussian_amd.patch.patch_reference()` ("patched"):

  (a) `opt.batch_size = 2`  (train.py:331-392: images concatenated, radii max, visibility any; :435-437 viewspace gradients summed)
      -- and the densification statistics such an iteration leaves behind equal what TWO view-parallel ranks leave behind through
      `dp.reduce_densification_stats` + `dp.add_densification_stats` for the same two views (two real processes over gloo on the one
      GPU of the box; RCCL refuses two ranks on one device);
  (b) `reset_opacity` -> `replace_tensor_to_optimizer` (scene/gaussian_model.py:350-353,397-410; train.py:514-516) and the
      `size_threshold = 20` branch of densify / prune once `iteration > opt.opacity_reset_interval` (train.py:502-508);
  (c) `training()` itself (train.py:553-641): coarse stage -> fine stage on the same model, optimizer rebuilt in between.

Tolerances: between the routes the first losses agree to 2e-3 (plain-PyTorch deformation field vs the fused kernels), every later one
to 10 %, point counts to 1 %; statistics: `denom` exact, `xyz_gradient_accum` rel-L2 1e-3 (zero_diff) / 2e-5 (patched: same kernels).
"""
import os
import random
import socket

import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu

P_SCENE, W_IMG, H_IMG = 20_000, 320, 208


@pytest.fixture(scope="module")
def ref_py():
    from oracle import ref_py as rp
    if not rp.available():
        pytest.skip("oracle/_ref/reference_py.tar.gz not built (needs /root/reference once: python oracle/ref_py.py)")
    yield rp
    rp.unload()


def _scene(seed, frames=3):
    from s3gaussian_amd import synth
    return synth.street_scene(P=P_SCENE, seed=seed, width=W_IMG, height=H_IMG, n_frames=frames)


def _dev_cam(c, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}


def _targets(dev, seed):
    """image, depth (every pixel inside the loss's 0.01 < gt < 80 mask: the masked mean over a batch is then the mean of the views'
    means, which is what makes the batch loss comparable with the view-parallel one), feature map."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(3, H_IMG, W_IMG, generator=g).to(dev), (torch.rand(1, H_IMG, W_IMG, generator=g) * 50 + 1).to(dev),
            torch.rand(3, H_IMG, W_IMG, generator=g).to(dev))


def _route(ref_py, route, scn, dev, state, views, target_seed=0, rendered_targets=False):
    """-> (ref, args, dataset, hyper, opt, pipe, gm, cams): the reference's modules loaded for `route`, its GaussianModel holding the
    scene with the deformation weights `state` (None: returned for the next route to share), real Camera objects for `views`."""
    ref = ref_py.load(patch=(route == "patched"))
    args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
    dataset.render_process = False
    torch.manual_seed(4)
    gm = ref_py.make_gaussians(ref, scn["gaussians"], scn["aabb"], hyper)
    if state:
        gm._deformation.load_state_dict(state)
    bg = scn["bg"].to(dev)
    cams = []
    if rendered_targets:
        with torch.no_grad():        # targets = the scene itself with perturbed positions, through whatever render() the route binds
            x0 = gm._xyz.data.clone()
            gm._xyz.data.add_(0.2 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(9)).to(dev))
            blank = _targets(dev, 0)
            for v in views:
                c = _dev_cam(scn["cameras"][v], dev)
                pkg = ref.gaussian_renderer.render(ref_py.make_camera(ref, c, blank, uid=v), gm, pipe, bg, stage="fine", render_feat=True)
                cams.append(ref_py.make_camera(ref, c, (pkg["render"].clamp(0, 1).clone(), pkg["depth"].clone(), pkg["feat"].clone()), uid=v))
            gm._xyz.data.copy_(x0)
    else:
        cams = [ref_py.make_camera(ref, _dev_cam(scn["cameras"][v], dev), _targets(dev, target_seed + v), uid=v) for v in views]
    return ref, args, dataset, hyper, opt, pipe, gm, cams


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_rank(rank, world, port, seed, views, state_file, out_file):
    """One view-parallel rank (its own process, gloo, the box's one GPU): the product's fused iteration for ITS view up to backward,
    then the data-parallel densification bookkeeping of bench.py's hook (dp.reduce_densification_stats -> dp.add_densification_stats)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    from s3gaussian_amd import dp
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, render, training_loss
    dp.init_from_env(backend="gloo")
    dev = torch.device("cuda", torch.cuda.current_device())
    scn = _scene(seed)
    hyper, opt = default_hyper(), default_opt(lambda_feat=0.0)
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    pc._deformation.load_state_dict(torch.load(state_file, map_location=dev))
    pc.training_setup(opt)
    v = views[rank]
    gt = _targets(dev, 100 + v)
    pipe = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False)
    pkg = render(_dev_cam(scn["cameras"][v], dev), pc, pipe, scn["bg"].to(dev), stage="fine", return_dx=True, render_feat=True)
    loss = training_loss(pc, pkg, gt[0], gt[1], gt[2], hyper, opt, "fine")      # this rank's OWN, un-averaged loss
    loss.backward()
    g_xy, any_vis, rmax = dp.reduce_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])
    dp.add_densification_stats(pc.xyz_gradient_accum, pc.denom, pc.max_radii2D, g_xy, any_vis, rmax)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"accum": pc.xyz_gradient_accum.cpu(), "denom": pc.denom.cpu(), "max_radii2D": pc.max_radii2D.cpu(),
                    "loss": float(loss)}, out_file)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_batch_size_two_on_both_routes_and_its_statistics_equal_two_view_parallel_ranks(gpu_device, ref_py, tmp_path):
    """(a).  First half: ONE iteration of scene_reconstruction at batch_size 2 over a two-camera scene (both views are popped, in
    either order) with lambda_feat = 0 -- train.py:419-422 applies the feature loss to the LAST view of a batch only, at full weight,
    which no batch-mean semantics reproduces --, against two dp ranks.  Second half: 24 iterations at batch_size 2 with every default
    loss term through a densify and a prune event, route against route."""
    import torch.multiprocessing as mp
    dev = gpu_device
    seed, views = 6, (0, 4)
    scn = _scene(seed)
    stats, state = {}, None
    for route in ("zero_diff", "patched"):
        ref, args, dataset, hyper, opt, pipe, gm, cams = _route(ref_py, route, scn, dev, state, views, target_seed=100)
        if state is None:
            state = {k: v.detach().clone() for k, v in gm._deformation.state_dict().items()}
        opt.batch_size, opt.lambda_feat = 2, 0.0
        random.seed(3)
        torch.manual_seed(3)
        timer = ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cams), dataset, hyper, opt, pipe, iterations=1, stage="fine")
        assert len(timer.losses) == 1 and np.isfinite(timer.losses[0])
        stats[route] = {"accum": gm.xyz_gradient_accum.detach().cpu(), "denom": gm.denom.detach().cpu(),
                        "max_radii2D": gm.max_radii2D.detach().cpu().float(), "loss": timer.losses[0]}
    ref_py.unload()
    state_file, out_file = str(tmp_path / "deform_state.pt"), str(tmp_path / "dp_stats.pt")
    torch.save({k: v.cpu() for k, v in state.items()}, state_file)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, seed, views, state_file, out_file)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"view-parallel rank exited with {p.exitcode}"
    dpst = torch.load(out_file)
    seen_twice = int((dpst["denom"] > 0).sum())
    assert seen_twice > 2000 and float(dpst["denom"].max()) == 1.0        # ANY over the batch: a Gaussian counts once per iteration
    for route, tol in (("zero_diff", 1e-3), ("patched", 2e-5)):
        st = stats[route]
        assert float((st["denom"] != dpst["denom"]).float().mean()) <= (1e-3 if route == "zero_diff" else 0.0), route
        assert rel_l2(st["accum"].numpy(), dpst["accum"].numpy()) <= tol, (route, rel_l2(st["accum"].numpy(), dpst["accum"].numpy()))
        assert float((st["max_radii2D"] != dpst["max_radii2D"]).float().mean()) <= (1e-3 if route == "zero_diff" else 0.0), route
    # the two routes against each other, every loss term on, through densify (iterations 10, 20) and prune (15) events
    runs = {}
    scn2 = _scene(seed + 1)
    state = None
    for route in ("zero_diff", "patched"):
        ref, args, dataset, hyper, opt, pipe, gm, cams = _route(ref_py, route, scn2, dev, state, range(6), rendered_targets=True)
        if state is None:
            state = {k: v.detach().clone() for k, v in gm._deformation.state_dict().items()}
        opt.batch_size = 2
        opt.densify_from_iter, opt.densification_interval = 5, 10
        opt.pruning_from_iter, opt.pruning_interval = 5, 15
        opt.densify_grad_threshold_fine_init = opt.densify_grad_threshold_after = 2e-5
        opt.opacity_threshold_fine_init = opt.opacity_threshold_fine_after = 0.05
        random.seed(12)
        torch.manual_seed(12)
        timer = ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cams), dataset, hyper, opt, pipe, iterations=24, stage="fine")
        assert len(timer.losses) == 24 and all(np.isfinite(timer.losses))
        runs[route] = (np.array(timer.losses), timer.points)
    a, b = runs["zero_diff"][0], runs["patched"][0]
    print("batch_size 2 losses, zero_diff:", np.round(a, 5).tolist())
    print("batch_size 2 losses, patched:  ", np.round(b, 5).tolist())
    print("points:", runs["zero_diff"][1][::4], runs["patched"][1][::4])
    assert np.all(np.abs(a[:8] - b[:8]) <= 2e-3 * np.abs(a[:8])), (a[:8], b[:8])
    assert np.all(np.abs(a - b) <= 0.1 * np.abs(a)), np.abs(a - b) / np.abs(a)
    pa, pb = runs["zero_diff"][1], runs["patched"][1]
    assert pa[0] == pb[0] == P_SCENE and pa[-1] != pa[0]
    assert abs(pa[-1] - pb[-1]) <= 0.01 * pa[-1], (pa[-1], pb[-1])


def test_opacity_reset_and_the_screen_size_prune_run_on_both_routes(gpu_device, ref_py):
    """(b) opacity_reset_interval = 12: iteration 12 calls reset_opacity (a NEW opacity Parameter with zeroed moments put into the
    optimizer by replace_tensor_to_optimizer -- on the patched route that optimizer is s3gaussian_amd.optim.Adam); from iteration 13 on
    densify (iterations 15, 20, 25, 30; the one at 10 still runs without) and prune (15, 30) run with size_threshold = 20, i.e. max_radii2D (which the rasterizer's radii feed) and
    the world-space extent test decide what is pruned."""
    from s3gaussian_amd import optim
    dev = gpu_device
    scn = _scene(8)
    runs, state = {}, None
    for route in ("zero_diff", "patched"):
        ref, args, dataset, hyper, opt, pipe, gm, cams = _route(ref_py, route, scn, dev, state, range(6), rendered_targets=True)
        if state is None:
            state = {k: v.detach().clone() for k, v in gm._deformation.state_dict().items()}
        opt.opacity_reset_interval = 12
        opt.densify_from_iter, opt.densification_interval = 5, 5
        opt.pruning_from_iter, opt.pruning_interval = 5, 15
        opt.densify_grad_threshold_fine_init = opt.densify_grad_threshold_after = 2e-5
        calls = {"reset": [], "prune_size": [], "densify_size": [], "pruned": []}
        real_reset, real_prune, real_densify = gm.reset_opacity, gm.prune, gm.densify

        def reset_opacity(_gm=gm, _real=real_reset, _calls=calls):
            before = _gm._opacity
            _real()
            st = _gm.optimizer.state[_gm._opacity]
            _calls["reset"].append((_gm._opacity is not before, float(st["exp_avg"].abs().max()), float(st["exp_avg_sq"].abs().max()),
                                    float(torch.sigmoid(_gm._opacity).max())))

        def prune(max_grad, min_opacity, extent, max_screen_size, _gm=gm, _real=real_prune, _calls=calls):
            n0 = _gm.get_xyz.shape[0]
            big = 0
            if max_screen_size:       # the two tests of scene/gaussian_model.py:664-669 that only run with a size threshold
                big = int(((_gm.max_radii2D > max_screen_size) | (_gm.get_scaling.max(dim=1).values > 0.1 * extent)).sum())
            _real(max_grad, min_opacity, extent, max_screen_size)
            _calls["prune_size"].append(max_screen_size)
            _calls["pruned"].append((n0 - _gm.get_xyz.shape[0], big))

        def densify(max_grad, min_opacity, extent, max_screen_size, *a, _real=real_densify, _calls=calls, **k):
            _calls["densify_size"].append(max_screen_size)
            _real(max_grad, min_opacity, extent, max_screen_size, *a, **k)

        gm.reset_opacity, gm.prune, gm.densify = reset_opacity, prune, densify      # instance attributes: the class is untouched
        random.seed(21)
        torch.manual_seed(21)
        # (cameras_extent 1.5: the world-space half of the size test -- scaling > 0.1 * extent -- then has something to prune in a scene
        #  whose splats are ~0.05 m; at 320 x 208 no splat reaches a 20-pixel radius)
        timer = ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cams, cameras_extent=1.5), dataset, hyper, opt, pipe,
                                                iterations=32, stage="fine")
        assert isinstance(gm.optimizer, optim.Adam) == (route == "patched")
        assert len(timer.losses) == 32 and all(np.isfinite(timer.losses))
        # reset_opacity ran at iterations 12 and 24: new Parameter, zeroed moments, every opacity <= 0.01 right after
        assert len(calls["reset"]) == 2
        for replaced, m, v, omax in calls["reset"]:
            assert replaced and m == 0.0 and v == 0.0 and omax <= 0.0100001
        assert calls["densify_size"] == [None, 20, 20, 20, 20] and calls["prune_size"] == [20, 20]      # densify at 10 | 15, 20, 25, 30
        assert calls["pruned"][0][1] > 0 and calls["pruned"][0][0] >= calls["pruned"][0][1]     # screen-size pruning had something to prune
        for grp in gm.optimizer.param_groups:
            if grp["name"] == "opacity":
                st = gm.optimizer.state[grp["params"][0]]
                # 32 iterations minus the 7 in which the opacity Parameter was replaced (densify 10 15 20 25 30, reset 12 24) before
                # that iteration's optimizer.step() could see a gradient on it: the reference's own order of operations
                assert grp["params"][0] is gm._opacity and st["exp_avg"].shape == gm._opacity.shape and float(st["step"]) == 25.0
        runs[route] = (np.array(timer.losses), timer.points, calls)
    a, b = runs["zero_diff"][0], runs["patched"][0]
    print("opacity-reset run losses, zero_diff:", np.round(a, 5).tolist())
    print("opacity-reset run losses, patched:  ", np.round(b, 5).tolist())
    print("points:", runs["zero_diff"][1][::4], runs["patched"][1][::4], "pruned:", runs["zero_diff"][2]["pruned"], runs["patched"][2]["pruned"])
    assert np.all(np.abs(a[:8] - b[:8]) <= 2e-3 * np.abs(a[:8])), (a[:8], b[:8])
    assert np.all(np.abs(a - b) <= 0.1 * np.abs(a)), np.abs(a - b) / np.abs(a)
    pa, pb = runs["zero_diff"][1], runs["patched"][1]
    assert pa[0] == pb[0] == P_SCENE
    assert abs(pa[-1] - pb[-1]) <= 0.01 * pa[-1], (pa[-1], pb[-1])
    for x, y in zip(runs["zero_diff"][2]["pruned"], runs["patched"][2]["pruned"]):
        assert abs(x[0] - y[0]) <= max(0.02 * x[0], 20) and abs(x[1] - y[1]) <= max(0.02 * x[1], 20), (x, y)


def test_training_of_train_py_hands_over_from_the_coarse_to_the_fine_stage(gpu_device, ref_py):
    """(c) train.py:553-641 `training()`: GaussianModel constructed by the reference, 8 coarse iterations (no deformation, no feature
    render, coarse thresholds), then 12 fine iterations on the same model with the optimizer rebuilt by the second training_setup, then
    its evaluation call.  See oracle/ref_py.py::run_training for the three names rebound around it."""
    from s3gaussian_amd import optim
    dev = gpu_device
    scn = _scene(10)
    runs, state = {}, {}
    for route in ("zero_diff", "patched"):
        ref = ref_py.load(patch=(route == "patched"))
        args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
        dataset.render_process = False
        opt.coarse_iterations, opt.iterations = 8, 12
        cams = [ref_py.make_camera(ref, _dev_cam(scn["cameras"][v], dev), _targets(dev, 200 + v), uid=v) for v in range(6)]

        def fill(gaussians, _ref=ref, _hyper=hyper):
            torch.manual_seed(10)
            ref_py.make_gaussians(_ref, scn["gaussians"], scn["aabb"], _hyper, model=gaussians)
            if state:
                gaussians._deformation.load_state_dict(state)
            else:
                state.update({k: v.detach().clone() for k, v in gaussians._deformation.state_dict().items()})

        random.seed(31)
        torch.manual_seed(31)
        res = ref_py.run_training(ref, fill, cams, dataset, hyper, opt, pipe)
        gm = res.gaussians
        assert type(gm).__module__ == "scene.gaussian_model" and isinstance(gm.optimizer, optim.Adam) == (route == "patched")
        assert res.evaluations == [12]                                   # train.py:630-641: evaluation at step = opt.iterations
        assert len(res.timer.losses) == 20 and all(np.isfinite(res.timer.losses))
        # the fine stage's optimizer is a fresh one (training_setup ran again): 12 steps on the Gaussians; the deformation field's
        # parameters only ever received gradients in the fine stage
        for grp in gm.optimizer.param_groups:
            for p in grp["params"]:
                st = gm.optimizer.state.get(p)
                if st:
                    assert float(st["step"]) == 12.0, (grp["name"], float(st["step"]))
        runs[route] = np.array(res.timer.losses)
    a, b = runs["zero_diff"], runs["patched"]
    print("training() losses (8 coarse + 12 fine), zero_diff:", np.round(a, 5).tolist())
    print("training() losses (8 coarse + 12 fine), patched:  ", np.round(b, 5).tolist())
    assert np.all(np.abs(a[:8] - b[:8]) <= 1e-4 * np.abs(a[:8])), (a[:8], b[:8])          # coarse: no deformation field on either route
    assert np.all(np.abs(a[8:14] - b[8:14]) <= 2e-3 * np.abs(a[8:14])), (a[8:14], b[8:14])
    assert np.all(np.abs(a - b) <= 0.1 * np.abs(a)), np.abs(a - b) / np.abs(a)
    assert a[8] != a[7]                                                  # the hand-over happened: the fine stage's loss has more terms


def test_the_reference_loop_never_loses_an_iteration_to_a_capacity_overflow(gpu_device, ref_py, monkeypatch):
    """VERDICT r5 next #6.  train.py::scene_reconstruction on the patched route, 24 iterations through a densify event; right after
    iteration 10 the rasterizer's capacity history is doctored so that the NEXT forward cannot hold its instances (what a densify
    event or a new camera can do to a speculative capacity).  Default policy ("verified"): the forward is issued again with room
    before anyone has seen its outputs -- same losses and point counts as a run with the SYNCHRONOUS forward (S3G_RASTER_ASYNC=0, the
    reference's own one wait per call), no iteration rendered background, none skipped its optimizer step."""
    from s3gaussian_amd import raster_C
    dev = gpu_device
    scn = _scene(14)
    runs, state = {}, None
    for mode in ("sync", "verified"):
        prev = raster_C.set_async(mode != "sync")
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()
        try:
            ref, args, dataset, hyper, opt, pipe, gm, cams = _route(ref_py, "patched", scn, dev, state, range(6), rendered_targets=True)
            if state is None:
                state = {k: v.detach().clone() for k, v in gm._deformation.state_dict().items()}
            opt.densify_from_iter, opt.densification_interval = 5, 10
            opt.densify_grad_threshold_fine_init = opt.densify_grad_threshold_after = 2e-5
            doctored = []

            def after_pause(n, _mode=mode):
                if _mode == "verified" and n in (10, 17):
                    st = raster_C._async_state(dev)
                    st.drain(block=True)
                    key = (W_IMG, H_IMG)
                    monkeypatch.setattr(raster_C, "_ASYNC_MIN_INSTANCES", 1)
                    true_R = st.hist[key][0]
                    st.hist[key] = [true_R // 16, true_R // 16, 4]
                    doctored.append((n, true_R, st.caps(key)[0]))

            random.seed(41)
            torch.manual_seed(41)
            timer = ref_py.RecordingTimer(after_pause=after_pause)
            ref_py.run_scene_reconstruction(ref, gm, ref_py.SceneStub(cams), dataset, hyper, opt, pipe, iterations=24, stage="fine", timer=timer)
            # every deformation parameter took 24 steps; the per-Gaussian parameters 22: densify (iterations 10, 20) puts fresh
            # nn.Parameters into the groups BEFORE that iteration's optimizer.step(), which skips them (no .grad yet) -- the reference's
            # own order of operations (train.py:489-522), the same on both runs
            steps = {float(st_["step"]) for st_ in gm.optimizer.state.values() if "step" in st_}
            assert steps == {22.0, 24.0}, steps
            if mode == "verified":
                stt = raster_C.async_status(dev, block=True)
                assert len(doctored) == 2 and all(c < r for _, r, c in doctored)
                assert stt["reissued"] >= 2 and stt["overflows"] == [] and stt["policy"] == "verified"
                print("verified run:", {k: stt[k] for k in ("calls", "reissued", "overflows")}, "doctored at", doctored)
            runs[mode] = (np.array(timer.losses), timer.points, np.array(timer.psnrs))
        finally:
            raster_C.set_async(prev)
            raster_C._async_states.pop(dev.index or 0, None)
            raster_C.invalidate_geometry_cache()
    a, b = runs["sync"][0], runs["verified"][0]
    print("sync     losses:", np.round(a, 5).tolist())
    print("verified losses:", np.round(b, 5).tolist())
    # the two runs execute the same kernels on the same inputs (an overflowed first attempt leaves nothing behind); what separates them
    # is the summation order of the HexPlane / weight-gradient flushes, if anything
    assert np.all(np.abs(a - b) <= 2e-3 * np.abs(a)), np.abs(a - b) / np.abs(a)
    assert runs["sync"][1][0] == runs["verified"][1][0] == P_SCENE
    assert abs(runs["sync"][1][-1] - runs["verified"][1][-1]) <= 0.002 * runs["sync"][1][-1], (runs["sync"][1][-1], runs["verified"][1][-1])
    assert runs["verified"][2].min() > 0.5 * runs["sync"][2].min()        # no background-only frame slipped into the loop
