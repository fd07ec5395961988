"""Host-asynchronous rasterizer forward (include/s3g_raster.h::s3g_raster_forward_async, raster_C.ASYNC).

The reference's forward waits for the device once per call (rasterizer_impl.cu:281-282: the instance count sizes the binning
arena); the asynchronous variant sizes the arenas for a speculative capacity and never waits.  Checked here:
  * bit-identical images / radii / gradients to the synchronous path (same kernels, same lists);
  * an overflow of the capacity is a well-defined no-op on the device -- background-only image, zero gradients, no densification
    bookkeeping, Adam step dropped -- that the host learns about later, and the capacity then grows;
  * the host really runs ahead of the device: several status rows are outstanding while training steps are being enqueued.
"""
import warnings
from types import SimpleNamespace

import pytest
import torch

from tests.util import settings_from, tiny_scene

pytestmark = pytest.mark.gpu


def _render_and_grads(s, dev, pair=False):
    from diff_gaussian_rasterization import GaussianRasterizer
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    t = lambda k: s[k].to(dev).clone().requires_grad_(True)
    m3, op, sc, rot, col = t("means3D"), t("opacities"), t("scales"), t("rotations"), t("colors_precomp")
    col2 = (s["colors_precomp"].flip(0).to(dev).clone()).requires_grad_(True)
    m2 = torch.zeros_like(m3, requires_grad=True)
    H, W = s["cam"]["image_height"], s["cam"]["image_width"]
    g = torch.Generator().manual_seed(5)
    gc, gd, gc2 = (torch.randn(c, H, W, generator=g).to(dev) for c in (3, 1, 3))
    if pair:
        color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=col2, scales=sc,
                                                        rotations=rot)
        ((color * gc).sum() + (depth * gd).sum() + (color2 * gc2).sum()).backward()
        outs = [color, depth, color2, radii]
    else:
        color, radii, depth = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)
        ((color * gc).sum() + (depth * gd).sum()).backward()
        outs = [color, depth, radii]
    grads = [x.grad for x in (m3, m2, op, sc, rot, col)] + ([col2.grad] if pair else [])
    return [o.detach().clone() for o in outs], [x.clone() for x in grads]


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("P,W,H", [(700, 96, 64), (20_000, 320, 208)])
def test_async_forward_is_bit_identical_to_the_synchronous_one(gpu_device, P, W, H, pair):
    from s3gaussian_amd import raster_C
    s = tiny_scene(P=P, W=W, H=H, seed=3)
    prev = raster_C.set_async(False)
    try:
        raster_C.invalidate_geometry_cache()
        o_sync, g_sync = _render_and_grads(s, gpu_device, pair)
        raster_C.set_async(True)
        for _ in range(2):          # first call learns the capacity (it waits once), the second is the steady state
            raster_C.invalidate_geometry_cache()
            n0 = raster_C.async_status()["calls"]
            o_async, g_async = _render_and_grads(s, gpu_device, pair)
            assert raster_C.async_status()["calls"] > n0                   # the asynchronous entry point really ran
            for a, b in zip(o_sync + g_sync, o_async + g_async):
                assert torch.equal(a, b)
    finally:
        raster_C.set_async(prev)
    st = raster_C.async_status(block=True)
    assert st["overflows"] == [] and st["mean_instances"] > 0


def test_overflow_is_a_device_side_no_op_and_is_reported_late(gpu_device, monkeypatch):
    """Shrink the capacity policy so that the next forward cannot hold its instances: image = background, gradients = 0, the
    densification accumulators and (through the guarded Adam step) the parameters stay untouched; the report arrives at a later
    poll and the capacity grows so that the same view then renders normally."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.optim import Adam
    dev = gpu_device
    s = tiny_scene(P=4000, W=160, H=112, seed=8)
    prev = raster_C.set_async(True)
    prev_policy = raster_C.set_async_policy("speculative")                 # (the default policy renders again on the spot: next test)
    try:
        raster_C.invalidate_geometry_cache()
        ref_out, ref_grads = _render_and_grads(s, dev, pair=True)          # learns the true counts of this image size
        torch.cuda.synchronize()
        st = raster_C._async_state(dev)
        st.drain(block=True)
        key = (160, 112)
        true_R = st.hist[key][0]
        assert true_R > 200
        monkeypatch.setattr(raster_C, "_ASYNC_MIN_INSTANCES", 1)
        st.hist[key] = [true_R // 16, true_R // 16, 4]                     # pretend smaller scenes were all we had seen
        cap = st.caps(key)
        assert cap[0] < true_R
        rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
        t = lambda k: torch.nn.Parameter(s[k].to(dev).clone())
        m3, op, sc, rot, col = t("means3D"), t("opacities"), t("scales"), t("rotations"), t("colors_precomp")
        col2 = torch.nn.Parameter(s["colors_precomp"].flip(0).to(dev).clone())
        m2 = torch.zeros_like(m3, requires_grad=True)
        acc = (torch.zeros(4000, 1, device=dev), torch.zeros(4000, 1, device=dev), torch.zeros(4000, device=dev))
        opt = Adam([m3, op, sc, rot, col, col2], lr=1e-2)
        before = [p.detach().clone() for p in (m3, op, sc, rot, col, col2)]
        raster_C.invalidate_geometry_cache()
        n_over = len(st.overflows)
        color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=col2, scales=sc,
                                                        rotations=rot, densify_accum=acc)
        (color.sum() + depth.sum() + color2.sum() + (m3 ** 2).sum()).backward()     # the last term: a gradient that is NOT zero
        opt.step()
        torch.cuda.synchronize()
        bg = s["bg"].to(dev)
        assert torch.equal(color, bg[:, None, None].expand_as(color)) and torch.equal(color2, bg[:, None, None].expand_as(color2))
        assert float(depth.abs().max()) == 0.0
        assert torch.equal(radii, ref_out[3])                              # per-Gaussian geometry ran before the overflow was known
        for g_ in (op.grad, sc.grad, rot.grad, col.grad, col2.grad, m2.grad):
            assert float(g_.abs().max()) == 0.0
        assert float(m3.grad.abs().max()) > 0                              # the regulariser-like term still has its gradient ...
        for p, b in zip((m3, op, sc, rot, col, col2), before):
            assert torch.equal(p.detach(), b)                              # ... but the guarded optimizer step was dropped
        for a in acc:
            assert float(a.abs().max()) == 0.0                             # no densification bookkeeping
        assert int(raster_C.async_skip_flag(dev).item()) == 1
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            status = raster_C.async_status(block=True)                     # the late report
        assert len(status["overflows"]) == n_over + 1 and any("exceeded its arena" in str(x.message) for x in w)
        assert st.caps(key)[0] >= true_R                                   # the capacity follows the true counts
        # the same view again: normal render, normal step
        raster_C.invalidate_geometry_cache()
        opt.zero_grad(set_to_none=True)
        color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=col2, scales=sc,
                                                        rotations=rot, densify_accum=acc)
        assert torch.equal(color, ref_out[0]) and torch.equal(depth, ref_out[1]) and torch.equal(color2, ref_out[2])
        (color.sum() + depth.sum() + color2.sum()).backward()
        opt.step()
        torch.cuda.synchronize()
        assert int(raster_C.async_skip_flag(dev).item()) == 0
        assert not torch.equal(op.detach(), before[1]) and float(acc[1].sum()) > 0
    finally:
        raster_C.set_async(prev)
        raster_C.set_async_policy(prev_policy)
        raster_C._async_states.pop(dev.index or 0, None)                   # forget the doctored history
        raster_C.invalidate_geometry_cache()


def test_the_default_policy_renders_again_on_overflow_and_drops_nothing(gpu_device, monkeypatch):
    """through the drop-in boundary in its DEFAULT policy ("verified") the same doctored capacity costs
    nothing but a second issue of the forward: the host reads the verdict behind the counting kernels (s3g_raster_async.status_event),
    finds the overflow, renders again with the capacity the true counts ask for -- into the same outputs, before anything has read
    them.  Image, gradients, bookkeeping and optimizer step are those of an ordinary call; nothing is warned about, nothing dropped."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import raster_C
    from s3gaussian_amd.optim import Adam
    dev = gpu_device
    assert raster_C.POLICY == "verified"
    s = tiny_scene(P=4000, W=160, H=112, seed=8)
    prev = raster_C.set_async(True)
    try:
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()
        ref_out, ref_grads = _render_and_grads(s, dev, pair=True)
        st = raster_C._async_state(dev)
        st.drain(block=True)
        key = (160, 112)
        true_R = st.hist[key][0]
        monkeypatch.setattr(raster_C, "_ASYNC_MIN_INSTANCES", 1)
        for no_grad in (False, True):
            st.hist[key] = [true_R // 16, true_R // 16, 4]
            assert st.caps(key)[0] < true_R
            rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
            t = lambda k: torch.nn.Parameter(s[k].to(dev).clone())
            m3, op, sc, rot, col = t("means3D"), t("opacities"), t("scales"), t("rotations"), t("colors_precomp")
            col2 = torch.nn.Parameter(s["colors_precomp"].flip(0).to(dev).clone())
            m2 = torch.zeros_like(m3, requires_grad=True)
            acc = (torch.zeros(4000, 1, device=dev), torch.zeros(4000, 1, device=dev), torch.zeros(4000, device=dev))
            opt = Adam([m3, op, sc, rot, col, col2], lr=1e-2)
            before = op.detach().clone()
            raster_C.invalidate_geometry_cache()
            calls, reissued = st.seq, st.reissued
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                if no_grad:
                    with torch.no_grad():      # an evaluation render: the forward_only form is verified too
                        color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=col2,
                                                                        scales=sc, rotations=rot)
                else:
                    color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=m2, opacities=op, colors_a=col, colors_b=col2,
                                                                    scales=sc, rotations=rot, densify_accum=acc)
                    (color.sum() + depth.sum() + color2.sum()).backward()
                    opt.step()
                torch.cuda.synchronize()
                status = raster_C.async_status(block=True)
            assert torch.equal(color, ref_out[0]) and torch.equal(depth, ref_out[1]) and torch.equal(color2, ref_out[2])
            assert st.seq == calls + 2 and st.reissued == reissued + 1          # issued twice, the second time with room
            assert status["overflows"] == [] and not any("exceeded its arena" in str(x.message) for x in w)
            assert st.caps(key)[0] >= true_R
            if not no_grad:
                assert int(raster_C.async_skip_flag(dev).item()) == 0
                assert not torch.equal(op.detach(), before) and float(acc[1].sum()) > 0      # the step and the bookkeeping happened
    finally:
        raster_C.set_async(prev)
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()


def test_training_steps_are_enqueued_ahead_of_the_device(gpu_device):
    """With the synchronous forward the host is never more than one rasterizer call ahead; with the asynchronous one several
    status rows are still in flight while steps are being enqueued (no call inside an iteration waits for the device)."""
    from s3gaussian_amd import raster_C, synth
    from s3gaussian_amd.pipeline import GaussianParams, default_hyper, default_opt, training_step
    dev = gpu_device
    # BASELINE-size scene: the device needs ~7 ms per step, several times what the host needs to enqueue one even on a slow box
    scn = synth.street_scene(P=1_200_000, seed=0, width=1600, height=1066, n_frames=4)
    hyper, opt = default_hyper(), default_opt()
    pc = GaussianParams(3, hyper)
    gs = scn["gaussians"]
    pc.init_from_tensors(gs["xyz"], gs["log_scales"], gs["rotations_raw"], gs["opacity_logit"], gs["shs"], dev)
    pc._deformation.deformation_net.set_aabb(*scn["aabb"])
    pc.training_setup(opt)
    cams = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()} for c in scn["cameras"][:4]]
    H, W = cams[0]["image_height"], cams[0]["image_width"]
    gts = (torch.rand(3, H, W, device=dev), torch.rand(1, H, W, device=dev) * 50, torch.rand(3, H, W, device=dev))
    bg = scn["bg"].to(dev)
    prev = raster_C.set_async(True)
    prev_policy = raster_C.set_async_policy("speculative")      # "verified" (the default) reads each forward's verdict before going on
    try:
        for i in range(3):
            training_step(pc, cams[i % 4], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
        torch.cuda.synchronize()
        st = raster_C._async_state(dev)
        st.drain(block=True)
        max_pending = 0
        for i in range(8):
            training_step(pc, cams[i % 4], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
            max_pending = max(max_pending, len(st.pending))
        torch.cuda.synchronize()
        assert max_pending >= 2, max_pending
        assert raster_C.async_status(block=True)["overflows"] == []
        raster_C.set_async_policy("verified")
        for i in range(4):     # every forward's row has been read by the time its call returns: nothing is ever pending
            training_step(pc, cams[i % 4], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
            assert len(st.pending) == 0
        raster_C.set_async(False)
        loss, _ = training_step(pc, cams[0], *gts, hyper, opt, bg, stage="fine", densify_stats=True)
        assert torch.isfinite(loss)
    finally:
        raster_C.set_async(prev)
        raster_C.set_async_policy(prev_policy)


@pytest.mark.parametrize("pair", [False, True])
def test_forward_only_render_is_bit_identical_and_never_feeds_a_backward(gpu_device, pair):
    """A render under no_grad takes the forward_only form of the asynchronous forward (s3g_raster_async.forward_only: the
    instance -> position map of the backward gather is not built).  Same image / depth / radii bit for bit; and arenas cached
    from such a render must not serve a later render of the SAME geometry that does need gradients."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from s3gaussian_amd import raster_C
    dev = gpu_device
    s = tiny_scene(P=20_000, W=320, H=208, seed=4)
    assert raster_C.ASYNC and raster_C.FORWARD_ONLY
    raster_C.invalidate_geometry_cache()
    o_ref, g_ref = _render_and_grads(s, dev, pair)
    # (1) the same inputs under no_grad
    rast = GaussianRasterizer(raster_settings=settings_from(s, dev))
    k = lambda n: s[n].to(dev).clone()
    m3, op, sc, rot, col = k("means3D"), k("opacities"), k("scales"), k("rotations"), k("colors_precomp")
    col2 = s["colors_precomp"].flip(0).to(dev).clone()
    raster_C.invalidate_geometry_cache()
    with torch.no_grad():
        if pair:
            color, radii, depth, color2 = rast.forward_pair(means3D=m3, means2D=torch.zeros_like(m3), opacities=op, colors_a=col,
                                                            colors_b=col2, scales=sc, rotations=rot)
            outs = [color, depth, color2, radii]
        else:
            color, radii, depth = rast(means3D=m3, means2D=torch.zeros_like(m3), opacities=op, colors_precomp=col, scales=sc, rotations=rot)
            outs = [color, depth, radii]
    for a, b in zip(o_ref, outs):
        assert torch.equal(a, b)
    if not pair:
        # (2) the single-image node caches its geometry: the arenas of the no_grad render above are in the cache now and flagged
        assert raster_C._geom_cache is not None and raster_C._geom_cache_forward_only
        hits = raster_C._geom_cache_hits
        for x in (m3, op, sc, rot, col):
            x.requires_grad_(True)
        m2 = torch.zeros_like(m3, requires_grad=True)
        H, W = s["cam"]["image_height"], s["cam"]["image_width"]
        g = torch.Generator().manual_seed(5)
        gc, gd = (torch.randn(c, H, W, generator=g).to(dev) for c in (3, 1))
        color, radii, depth = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)
        assert raster_C._geom_cache_hits == hits and not raster_C._geom_cache_forward_only    # rendered afresh, with the map
        ((color * gc).sum() + (depth * gd).sum()).backward()
        for a, b in zip(g_ref, [x.grad for x in (m3, m2, op, sc, rot, col)]):
            assert torch.equal(a, b)
        # (3) and a no_grad render of that geometry may reuse the arenas that DO have the map
        with torch.no_grad():
            rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)
        assert raster_C._geom_cache_hits == hits + 1


def test_speculative_arenas_yield_to_a_crowded_device(gpu_device, monkeypatch):
    """the asynchronous forward sizes its arenas (and through them the backward's records) for 4 x the largest count seen,
    at least 16 M instances -- about 1.2 GB per render / backward pair.  When that does not fit half of the free device memory it
    tries 2 x without a floor, and when that does not fit either the call takes the SYNCHRONOUS forward (arenas sized for the true
    counts): same image, same gradients, no asynchronous call issued, and no stale overflow word left for the guarded optimizer."""
    from s3gaussian_amd import raster_C
    dev = gpu_device
    s = tiny_scene(P=20_000, W=320, H=208, seed=6)
    prev = raster_C.set_async(True)
    try:
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()
        ref_out, ref_grads = _render_and_grads(s, dev, pair=True)          # learns the counts; asynchronous from the second call on
        st = raster_C._async_state(dev)
        st.drain(block=True)
        calls = raster_C.async_status(dev, block=True)["calls"]
        real = torch.cuda.mem_get_info
        # (1) room for the tight capacity only: still asynchronous, smaller arenas
        need4 = 56 * st.caps((320, 208))[0]
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (need4 // 4, real(d)[1]))
        monkeypatch.setattr(torch.cuda, "memory_reserved", lambda d=None: 0)
        monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d=None: 0)
        st.fit_cache.clear()
        raster_C.invalidate_geometry_cache()
        out, grads = _render_and_grads(s, dev, pair=True)
        for a, b in zip(ref_out + ref_grads, out + grads):
            assert torch.equal(a, b)
        assert raster_C.async_status(dev, block=True)["calls"] == calls + 1
        # (2) no room at all: the synchronous forward
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (1 << 20, real(d)[1]))
        st.fit_cache.clear()
        raster_C.invalidate_geometry_cache()
        out, grads = _render_and_grads(s, dev, pair=True)
        for a, b in zip(ref_out + ref_grads, out + grads):
            assert torch.equal(a, b)
        # no asynchronous call was issued; the forward owns a skip word of its own (0: a synchronous forward cannot overflow), never
        # the word of an older forward
        assert raster_C.async_status(dev, block=True)["calls"] == calls + 1
        assert raster_C.async_skip_flag(dev) is st.fallback_word and int(st.fallback_word.item()) == 0
    finally:
        raster_C.set_async(prev)
        raster_C._async_states.pop(dev.index or 0, None)
        raster_C.invalidate_geometry_cache()
