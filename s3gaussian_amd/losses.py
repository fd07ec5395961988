"""Loss kernels of the hot path on the MI355X.  `ssim(img1, img2)` has the signature and value of the reference's
utils/loss_utils.py::ssim (:66-96, window 11, size_average=True) for [C,H,W] or [1,C,H,W] images; the gradient flows
to img1 only (the rendered image), which is how train.py:416-418 uses it."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp = C.c_void_p
        L.s3g_ssim_forward.restype = C.c_int
        L.s3g_ssim_forward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.s3g_ssim_backward.restype = C.c_int
        L.s3g_ssim_backward.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
        _bound = True
    return L


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        if not img1.is_cuda:
            raise RuntimeError(f"ssim: images must live on the GPU (got {img1.device}); no CPU fallback")
        L = _bind()
        a = img1.detach().reshape(-1, img1.shape[-2], img1.shape[-1]).contiguous().float()
        b = img2.detach().reshape(-1, img2.shape[-2], img2.shape[-1]).contiguous().float()
        if a.shape != b.shape:
            raise RuntimeError("ssim: shape mismatch")
        Cn, H, W = a.shape
        total = torch.zeros((), dtype=torch.float64, device=a.device)
        maps = torch.empty((3, Cn, H, W), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            _lib.check(L.s3g_ssim_forward(Cn, H, W, a.data_ptr(), b.data_ptr(), total.data_ptr(), maps[0].data_ptr(),
                                          maps[1].data_ptr(), maps[2].data_ptr(), torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(a, b, maps)
        ctx.shape = img1.shape
        return (total / float(Cn * H * W)).float()

    @staticmethod
    def backward(ctx, g):
        a, b, maps = ctx.saved_tensors
        L = _bind()
        Cn, H, W = a.shape
        g = g.detach().reshape(1).contiguous().float()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(L.s3g_ssim_backward(Cn, H, W, a.data_ptr(), b.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(),
                                           maps[2].data_ptr(), g.data_ptr(), out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        return out.view(ctx.shape), None


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    if window_size != 11 or not size_average:
        raise NotImplementedError("only the reference's call signature ssim(img1, img2) is accelerated")
    return _SSIM.apply(img1, img2)


class _PlaneRegDesc(C.Structure):
    """struct s3g_plane_reg_desc (include/s3g_loss.h)."""
    _fields_ = [("plane", C.c_void_p), ("grad", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("w_smooth", C.c_float),
                ("w_l1", C.c_float)]


class _PlaneRegulation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *planes):
        L = _lib.lib()
        L.s3g_plane_regulation.restype = C.c_int
        L.s3g_plane_regulation.argtypes = [C.c_int, C.POINTER(_PlaneRegDesc), C.c_void_p, C.c_void_p]
        dev = planes[0].device
        if not planes[0].is_cuda:
            raise RuntimeError("plane regulation: planes must live on the GPU; no CPU fallback")
        grads = [torch.empty_like(p) for p in planes]  # preserves channels_last
        descs = (_PlaneRegDesc * len(planes))()
        for i, (p, g, (ws, wl)) in enumerate(zip(planes, grads, weights)):
            if p.dim() != 4 or p.shape[1] != 32 or not p.is_contiguous(memory_format=torch.channels_last):
                raise RuntimeError("plane regulation expects [1,32,H,W] channels_last planes")
            descs[i] = _PlaneRegDesc(p.data_ptr(), g.data_ptr(), p.shape[2], p.shape[3], ws, wl)
        value = torch.zeros((), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.s3g_plane_regulation(len(planes), descs, value.data_ptr(), torch.cuda.current_stream().cuda_stream))
        ctx.grads = grads
        return value.float()

    @staticmethod
    def backward(ctx, g):
        grads = ctx.grads
        torch._foreach_mul_(grads, g)
        return (None, *grads)


def plane_regulation(grids, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    """GaussianModel.compute_regulation (scene/gaussian_model.py:748-749) over HexPlaneField.grids, value and
    gradient in one fused pass."""
    planes, weights = [], []
    for level in grids:
        for i, p in enumerate(level):
            planes.append(p)
            weights.append((float(time_smoothness_weight), float(l1_time_planes_weight)) if i in (2, 4, 5)
                           else (float(plane_tv_weight), 0.0))
    return _PlaneRegulation.apply(tuple(weights), *planes)
