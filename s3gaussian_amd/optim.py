"""Adam with the state layout and update rule of torch.optim.Adam (what the reference builds in
scene/gaussian_model.py:177-189 and what its densification code edits: `state[p]["exp_avg"]`, `["exp_avg_sq"]`,
`["step"]`), stepped by ONE HIP kernel launch for all parameters of all groups (include/s3g_optim.h)."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib

MAX_TENSORS = 64


class _AdamTensor(C.Structure):
    """struct s3g_adam_tensor (include/s3g_optim.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_size_t), ("step_size", C.c_float), ("inv_sqrt_bc2", C.c_float), ("eps", C.c_float),
                ("grad_scale", C.c_float)]


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class Adam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas, eps) without weight decay / amsgrad / maximize; `step()` runs on the MI355X
    library.  CPU parameters are refused (no CPU fallback on the product path); use torch.optim.Adam for those."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        self.grad_scale = 1.0   # every gradient is multiplied by this inside the kernel; the data-parallel wrapper sets
                                # 1 / world_size after a SUM all-reduce instead of spending a pass on the average
        self._entries = {}      # id(param) -> _Entry (launch record + what was checked once), see step()
        self._lib = None
        self.step_calls = 0     # number of step() calls that launched something (the replay loop notes it per iteration)
        self._journal = []      # [(step_calls after the call, [state dicts whose step count it advanced])], newest last, bounded

    @torch.no_grad()
    def rewind_to(self, step_calls: int) -> int:
        """The step() calls after the `step_calls`-th were dropped ON THE DEVICE (guarded step, skip word set: the asynchronous
        rasterizer's overflow in replay mode, raster_C.set_async_replay) and the iterations that issued them are about to be issued
        again: take the host-side step counts -- which feed the bias corrections -- back, so that the re-issued steps compute exactly
        what the dropped ones would have.  Per PARAMETER: each journal entry lists the states that call advanced (a parameter
        without a gradient in a dropped iteration, or stepped in the other phase of a two-phase data-parallel step, is not touched
        for it -- ADVICE r5).  The moments and parameters were never written by the dropped launches.  -> calls taken back."""
        n = 0
        while self._journal and self._journal[-1][0] > step_calls:
            _, states = self._journal.pop()
            for st in states:
                st["step"] -= 1.0
            n += 1
        if self.step_calls - n > step_calls:
            raise RuntimeError(f"optim.Adam.rewind_to({step_calls}): only the last {len(self._journal) + n} step() calls are journalled "
                               f"({self.step_calls} issued)")
        self.step_calls -= n
        return n

    def rewind(self, iterations: int) -> None:
        """rewind_to() for a loop that issues exactly one step() per iteration."""
        self.rewind_to(max(self.step_calls - int(iterations), 0))

    def _entry(self, group, p):
        """Everything about one parameter that does not change from step to step, checked once: the launch record with the
        parameter's own fields filled in, its state dict, its layout."""
        if not p.is_cuda:
            raise RuntimeError("s3gaussian_amd.optim.Adam: parameters must live on the GPU (no CPU fallback)")
        if p.dtype != torch.float32 or not _dense(p):
            raise RuntimeError("s3gaussian_amd.optim.Adam handles dense float32 parameters only")
        e = _Entry()
        e.p, e.stride, e.device = p, p.stride(), p.device
        e.contig = p.is_contiguous()
        e.rec = _AdamTensor(p.data_ptr(), None, None, None, p.numel(), 0.0, 0.0, 0.0, 1.0)
        e.p_ptr, e.m, e.v = p.data_ptr(), None, None
        e.st, e.state_obj, e.step_t, e.step_val, e.step_ver = None, None, None, 0.0, -1
        return e

    @torch.no_grad()
    def step(self, closure=None):
        """Host side: ~4 us per parameter before the launch (the launch records are kept from step to step; only the gradient pointer and
        the step-dependent scalars are written) -- on the zero-edit route the GPU sits idle behind train.py's `loss.item()` until this
        launch is out, so every microsecond in front of it is a microsecond of the iteration (profiles/r05_patched_iteration_trace.txt)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = self._lib
        if L is None:
            L = self._lib = _lib.lib()
            L.s3g_adam_step_guarded.restype = C.c_int
            L.s3g_adam_step_guarded.argtypes = [C.c_int, C.POINTER(_AdamTensor), C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        entries, f32, strided, gs = self._entries, torch.float32, torch.strided, float(self.grad_scale)
        state = self.state              # (load_state_dict installs a NEW dict: the per-entry shortcuts below are keyed on its identity)
        # 1. validate everything and fill the launch records; no state is touched before the launches are out: an exception must not
        #    leave some parameters with an advanced step count and others without.  Round 6: ~4 us per parameter (was ~8): the state
        #    dict, the step count as a Python float and the two bias corrections of a (betas, step) pair are looked up / computed once,
        #    not per parameter and step (profiles/r06_patched_host_profile.txt: 0.43 ms of the zero-edit route's iteration were this loop,
        #    with the GPU idle behind train.py's blocking reads)
        by_betas, keep, todo, stale, bias = {}, [], [], False, {}
        for group in self.param_groups:
            lr, eps = group["lr"], group["eps"]
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                e = entries.get(id(p))
                if e is None or e.p is not p or e.p_ptr != p.data_ptr():
                    e = entries[id(p)] = self._entry(group, p)
                    stale = True
                if g.layout is not strided:
                    raise RuntimeError("s3gaussian_amd.optim.Adam handles dense float32 parameters only")
                if g.dtype is not f32 or not (g.is_contiguous() if e.contig else g.stride() == e.stride):
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    keep.append(g)            # must outlive the launch call
                st = e.st
                if st is None or e.state_obj is not state or state.get(p) is not st:
                    st = e.st = state[p]
                    e.state_obj = state
                if len(st) == 0:   # same lazy initialisation as torch/optim/adam.py::_init_group
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m is not e.m or v is not e.v:          # first step, or densification / load_state_dict replaced the moments
                    for name in ("exp_avg", "exp_avg_sq"):
                        if st[name].stride() != e.stride:     # ... possibly with another layout
                            st[name] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[name])
                    m, v = e.m, e.v = st["exp_avg"], st["exp_avg_sq"]
                    e.rec.exp_avg, e.rec.exp_avg_sq = m.data_ptr(), v.data_ptr()
                t = st["step"]
                ver = getattr(t, "_version", -2)                       # (a checkpoint of an old torch may hold a plain number here)
                if t is not e.step_t or e.step_ver != ver:             # another step tensor, or someone else advanced / reset this one
                    e.step_t, e.step_val, e.step_ver = t, float(t), ver
                step = e.step_val + 1.0
                bc = bias.get((beta1, beta2, step))
                if bc is None:
                    bc = bias[(beta1, beta2, step)] = (1.0 - beta1 ** step, 1.0 / math.sqrt(1.0 - beta2 ** step))
                rec = e.rec
                rec.grad, rec.step_size, rec.inv_sqrt_bc2 = g.data_ptr(), lr / bc[0], bc[1]
                rec.eps, rec.grad_scale = eps, gs
                by_betas.setdefault((e.device, float(beta1), float(beta2)), []).append(rec)
                todo.append((e, st))
        if stale:
            # a parameter was new or replaced (densify / prune / reset_opacity put fresh nn.Parameters into the groups every 100
            # iterations, scene/gaussian_model.py:397-494): drop the records of the ones that left RIGHT AWAY.  A record holds its
            # parameter and both moments strongly -- 708 B per Gaussian and generation; kept "until there are many" (rounds 3-5) that was
            # up to ~20 GB of dead tensors at 1.2 M Gaussians, which the reference's torch.optim.Adam frees at once
            live = {id(p) for group in self.param_groups for p in group["params"]}
            for k in [k for k in entries if k not in live]:
                del entries[k]
        # 2. launch
        from . import raster_C
        for (dev, beta1, beta2), items in by_betas.items():
            # host-asynchronous rasterizer (raster_C.ASYNC): if the last forward on this device overflowed its speculative
            # arena -- it then rendered nothing and back-propagated zeros -- the device drops this step too; the host finds out
            # later without having waited (raster_C.async_status()).  None: no such forward, plain step.
            # The flag belongs to the most recent asynchronous forward on this device and applies to every optimizer step issued
            # before the next one (a two-phase data-parallel step calls step() twice per iteration).  A process that trains two
            # independent models on one device with interleaved forwards should switch the mechanism off (S3G_RASTER_ASYNC=0).
            flag = raster_C.async_skip_flag(dev)   # data parallel: dp.reduce_skip_flag() has all-reduced it in place
            with _lib.on_device(dev):
                stream = _lib.stream_ptr()
                for k in range(0, len(items), MAX_TENSORS):
                    chunk = items[k:k + MAX_TENSORS]
                    arr = (_AdamTensor * len(chunk))(*chunk)
                    _lib.check(L.s3g_adam_step_guarded(len(chunk), arr, beta1, beta2,
                                                       flag.data_ptr() if flag is not None else None, stream))
        # 3. the kernel is out: advance the step counts, and tell PyTorch that parameters and moments were written through raw
        #    pointers.  Version counters are what autograd's saved-tensor checks and the rasterizer's geometry cache
        #    (raster_C._geom_key) look at; without the bump a render of the SAME parameter tensors after this step could be served
        #    the previous step's binning.
        written = []
        for e, st in todo:
            t = st["step"]
            if isinstance(t, torch.Tensor):
                t += 1
                e.step_val, e.step_ver = e.step_val + 1.0, t._version
            else:
                st["step"] = t + 1
            written += (e.p, e.m, e.v)
        if todo:
            self.step_calls += 1
            self._journal.append((self.step_calls, [st for _, st in todo]))
            if len(self._journal) > 256:          # the host is never more than one status ring (64 forwards) ahead of the device
                del self._journal[:128]
            torch.autograd.graph.increment_version(written)     # one call for all of them: 6 us instead of 60
            raster_C.invalidate_geometry_cache()
        return loss


class _Entry:
    __slots__ = ("p", "stride", "device", "rec", "p_ptr", "m", "v", "contig", "st", "state_obj", "step_t", "step_val", "step_ver")


@torch.no_grad()
def densify_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                  viewspace_grad: torch.Tensor, radii: torch.Tensor, visible: torch.Tensor = None) -> None:
    """train.py:489-493 + scene/gaussian_model.py:693-695 in one pass (include/s3g_optim.h::s3g_densify_stats):
    `max_radii2D[vis] = max(., radii[vis]); xyz_gradient_accum[vis] += ||viewspace_grad[vis,:2]||; denom[vis] += 1`
    with vis = `visible` (bool [P]) or radii > 0.  Accumulators are the reference's tensors ([P,1], [P,1], [P] fp32),
    updated in place."""
    L = _lib.lib()
    P = radii.shape[0]
    for name, t_, n in (("xyz_gradient_accum", xyz_gradient_accum, P), ("denom", denom, P), ("max_radii2D", max_radii2D, P)):
        if not (t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == n):
            raise RuntimeError(f"densify_stats: {name} must be a contiguous float32 GPU tensor with {n} elements")
    g = viewspace_grad if (viewspace_grad.dtype == torch.float32 and viewspace_grad.stride(-1) == 1 and viewspace_grad.dim() == 2) \
        else viewspace_grad.float().contiguous()
    r = radii if radii.dtype == torch.int32 and radii.is_contiguous() else radii.to(torch.int32).contiguous()
    if visible is None:
        v = None
    elif visible.dtype == torch.bool and visible.is_contiguous():
        v = visible.view(torch.uint8)            # one byte per element, 0 / 1: the same storage, no conversion pass
    else:
        v = visible.to(torch.uint8).contiguous()
    from . import raster_C
    flag = raster_C.async_skip_flag(radii.device)     # the view's asynchronous forward overflowed -> no statistics
    with _lib.on_device(radii.device):
        _lib.check(L.s3g_densify_stats_guarded(P, g.data_ptr(), int(g.stride(0)), r.data_ptr(), v.data_ptr() if v is not None else None,
                                               xyz_gradient_accum.data_ptr(), denom.data_ptr(), max_radii2D.data_ptr(),
                                               flag.data_ptr() if flag is not None else None,
                                               _lib.stream_ptr()))
    for t_ in (xyz_gradient_accum, denom, max_radii2D):
        torch.autograd.graph.increment_version(t_)
