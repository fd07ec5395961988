"""distCUDA2 on the MI355X: mean squared distance to the 3 nearest neighbours (KNN/spatial.cu:15-26)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> float32 [P]; used once at scene creation (scene/gaussian_model.py:153)."""
    if not points.is_cuda:
        raise RuntimeError(f"points must live on the GPU (got {points.device}); distCUDA2 has no CPU fallback")
    L = _lib.lib()
    P = points.size(0)
    pts = points.contiguous()
    if pts.dtype != torch.float32:
        raise RuntimeError(f"points must be float32, got {pts.dtype}")
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P == 0:
        return means
    L.s3g_knn_workspace_bytes.restype = C.c_size_t
    L.s3g_knn_workspace_bytes.argtypes = [C.c_int]
    L.s3g_knn_mean_dist2.restype = C.c_int
    L.s3g_knn_mean_dist2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    work = torch.empty(L.s3g_knn_workspace_bytes(P), dtype=torch.uint8, device=points.device)
    with _lib.on_device(points.device):
        code = L.s3g_knn_mean_dist2(P, pts.data_ptr(), means.data_ptr(), work.data_ptr(),
                                    _lib.stream_ptr())
    _lib.check(code)
    return means
