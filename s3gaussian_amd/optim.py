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
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        L.s3g_adam_step.restype = C.c_int
        L.s3g_adam_step.argtypes = [C.c_int, C.POINTER(_AdamTensor), C.c_double, C.c_double, C.c_void_p]
        by_betas = {}
        keep = []   # tensors created here must outlive the launch call
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("s3gaussian_amd.optim.Adam: parameters must live on the GPU (no CPU fallback)")
                if p.dtype != torch.float32 or p.grad.is_sparse or not _dense(p):
                    raise RuntimeError("s3gaussian_amd.optim.Adam handles dense float32 parameters only")
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
        for (dev, beta1, beta2), items in by_betas.items():
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream().cuda_stream
                for k in range(0, len(items), MAX_TENSORS):
                    chunk = items[k:k + MAX_TENSORS]
                    arr = (_AdamTensor * len(chunk))(*chunk)
                    _lib.check(L.s3g_adam_step(len(chunk), arr, beta1, beta2, stream))
        return loss
