"""Adam with the state layout and update rule of torch.optim.Adam (what the reference builds in
scene/gaussian_model.py:177-189 and what its densification code edits: `state[p]["exp_avg"]`, `["exp_avg_sq"]`,
`["step"]`), stepped by ONE HIP kernel launch for all parameters of all groups (include/s3g_optim.h)."""
from __future__ import annotations

import ctypes as C

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

    @torch.no_grad()
    def rewind(self, iterations: int) -> None:
        """The last `iterations` iterations were dropped ON THE DEVICE (guarded step, skip word set: the asynchronous rasterizer's
        overflow in replay mode, raster_C.set_async_replay) and are about to be issued again: take the host-side step counts --
        which feed the bias corrections -- back, so that the re-issued steps compute exactly what the dropped ones would have.
        The moments and parameters were never touched by the dropped launches."""
        for st in self.state.values():
            if "step" in st:
                st["step"] -= float(iterations)
                if float(st["step"]) < 0:
                    st["step"].zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        L.s3g_adam_step_guarded.restype = C.c_int
        L.s3g_adam_step_guarded.argtypes = [C.c_int, C.POINTER(_AdamTensor), C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        # 1. validate everything before touching any state: an exception must not leave some parameters with an advanced
        #    step count and others without
        todo = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("s3gaussian_amd.optim.Adam: parameters must live on the GPU (no CPU fallback)")
                if p.dtype != torch.float32 or p.grad.is_sparse or not _dense(p):
                    raise RuntimeError("s3gaussian_amd.optim.Adam handles dense float32 parameters only")
                todo.append((group, p))
        # 2. lazy state, step counters, launch records
        by_betas = {}
        keep = []   # tensors created here must outlive the launch call
        for group, p in todo:
            beta1, beta2 = group["betas"]
            st = self.state[p]
            if len(st) == 0:   # same lazy initialisation as torch/optim/adam.py::_init_group
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            step = float(st["step"])
            g = p.grad
            if g.dtype != torch.float32 or g.stride() != p.stride():
                g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                keep.append(g)
            for name in ("exp_avg", "exp_avg_sq"):   # densification code may have replaced them with other layouts
                if st[name].stride() != p.stride():
                    st[name] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[name])
            bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
            by_betas.setdefault((p.device, float(beta1), float(beta2)), []).append(
                _AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                            group["lr"] / bc1, 1.0 / (bc2 ** 0.5), group["eps"], float(self.grad_scale)))
        from . import raster_C
        for (dev, beta1, beta2), items in by_betas.items():
            # host-asynchronous rasterizer (raster_C.ASYNC): if the last forward on this device overflowed its speculative
            # arena -- it then rendered nothing and back-propagated zeros -- the device drops this step too; the host finds out
            # later without having waited (raster_C.async_status()).  None: no such forward, plain step.
            # The flag belongs to the most recent asynchronous forward on this device and applies to every optimizer step issued
            # before the next one (a two-phase data-parallel step calls step() twice per iteration).  A process that trains two
            # independent models on one device with interleaved forwards should switch the mechanism off (S3G_RASTER_ASYNC=0).
            flag = raster_C.async_skip_flag(dev)   # data parallel: dp.reduce_skip_flag() has all-reduced it in place
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream().cuda_stream
                for k in range(0, len(items), MAX_TENSORS):
                    chunk = items[k:k + MAX_TENSORS]
                    arr = (_AdamTensor * len(chunk))(*chunk)
                    _lib.check(L.s3g_adam_step_guarded(len(chunk), arr, beta1, beta2,
                                                       flag.data_ptr() if flag is not None else None, stream))
        # 3. the kernel wrote the parameters (and moments) through raw pointers: tell PyTorch.  Version counters are what
        #    autograd's saved-tensor checks and the rasterizer's geometry cache (raster_C._geom_key) look at; without the
        #    bump a render of the SAME parameter tensors after this step could be served the previous step's binning.
        for _, p in todo:
            torch.autograd.graph.increment_version(p)
            st = self.state[p]
            torch.autograd.graph.increment_version(st["exp_avg"])
            torch.autograd.graph.increment_version(st["exp_avg_sq"])
        if todo:
            raster_C.invalidate_geometry_cache()
        return loss


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
    v = None if visible is None else visible.to(torch.uint8).contiguous()
    from . import raster_C
    flag = raster_C.async_skip_flag(radii.device)     # the view's asynchronous forward overflowed -> no statistics
    with torch.cuda.device(radii.device):
        _lib.check(L.s3g_densify_stats_guarded(P, g.data_ptr(), int(g.stride(0)), r.data_ptr(), v.data_ptr() if v is not None else None,
                                               xyz_gradient_accum.data_ptr(), denom.data_ptr(), max_radii2D.data_ptr(),
                                               flag.data_ptr() if flag is not None else None,
                                               torch.cuda.current_stream().cuda_stream))
    for t_ in (xyz_gradient_accum, denom, max_radii2D):
        torch.autograd.graph.increment_version(t_)
