"""Fused per-Gaussian render glue (activations + SH->RGB) on the MI355X; see include/s3g_glue.h."""
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
        L.s3g_glue_forward.restype = C.c_int
        L.s3g_glue_forward.argtypes = [C.c_int, C.c_int] + [vp] * 14
        L.s3g_glue_backward.restype = C.c_int
        L.s3g_glue_backward.argtypes = [C.c_int, C.c_int] + [vp] * 23
        _bound = True
    return L


def _p(t):
    return None if t is None else t.data_ptr()


class _Glue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, deg, f_dc, f_rest, dshs, xyz, campos, log_scales, rot_raw, opacity_logit, want_dshs_l1):
        if not xyz.is_cuda:
            raise RuntimeError(f"render glue: tensors must live on the GPU (got {xyz.device}); no CPU fallback")
        L = _bind()
        c = lambda t: None if t is None else t.detach().contiguous().float()
        f_dc, f_rest, dshs, xyz_c, campos_c = c(f_dc), c(f_rest), c(dshs), c(xyz), c(campos)
        ls, rr, ol = c(log_scales), c(rot_raw), c(opacity_logit)
        P, dev = xyz_c.shape[0], xyz_c.device
        if f_dc.shape != (P, 1, 3) or f_rest.shape != (P, 15, 3):
            raise RuntimeError("render glue expects f_dc [P,1,3] and f_rest [P,15,3] (sh_degree 3 storage)")
        colors = torch.empty((P, 3), dtype=torch.float32, device=dev)
        scales = torch.empty((P, 3), dtype=torch.float32, device=dev)
        rot = torch.empty((P, 4), dtype=torch.float32, device=dev)
        opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
        asked_l1 = bool(want_dshs_l1)
        want_dshs_l1 = asked_l1 and dshs is not None and P > 0
        abs_sum = torch.zeros(64 * 16, dtype=torch.float64, device=dev) if want_dshs_l1 else None   # S3G_SUM_DOUBLES
        with _lib.on_device(dev):
            _lib.check(L.s3g_glue_forward(P, int(deg), _p(f_dc), _p(f_rest), _p(dshs), _p(xyz_c), _p(campos_c), _p(ls), _p(rr),
                                          _p(ol), _p(colors), _p(scales), _p(rot), _p(opac), _p(abs_sum),
                                          _lib.stream_ptr()))
        ctx.deg = int(deg)
        ctx.has_dshs = dshs is not None
        ctx.save_for_backward(f_dc, f_rest, dshs if dshs is not None else torch.empty(0, device=dev), xyz_c, campos_c, rr,
                              colors, scales, rot, opac)
        ctx.want_dshs_l1 = want_dshs_l1
        if want_dshs_l1:
            dshs_l1 = (abs_sum.sum() / (48.0 * P)).float()
        elif asked_l1:
            dshs_l1 = torch.zeros((), dtype=torch.float32, device=dev)     # asked for, nothing to sum (no dshs / P == 0)
        else:
            dshs_l1 = torch.empty((), dtype=torch.float32, device=dev)     # not asked for: dropped by the wrapper, no fill launch
        return colors, scales, rot, opac, dshs_l1

    @staticmethod
    def backward(ctx, g_colors, g_scales, g_rot, g_opac, g_l1):
        f_dc, f_rest, dshs, xyz, campos, rr, colors, scales, rot, opac = ctx.saved_tensors
        L = _bind()
        P, dev = xyz.shape[0], xyz.device
        dshs = dshs if ctx.has_dshs else None
        c = lambda t: None if t is None else t.contiguous().float()
        g_colors, g_scales, g_rot, g_opac = c(g_colors), c(g_scales), c(g_rot), c(g_opac)
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        g_f_dc, g_f_rest, g_xyz, g_ls, g_rr, g_ol = e(P, 1, 3), e(P, 15, 3), e(P, 3), e(P, 3), e(P, 4), e(P, 1)
        g_dshs = e(P, 16, 3) if ctx.has_dshs else None
        g_l1 = g_l1.contiguous().float() if (ctx.want_dshs_l1 and g_l1 is not None) else None
        with _lib.on_device(dev):
            _lib.check(L.s3g_glue_backward(P, ctx.deg, _p(f_dc), _p(f_rest), _p(dshs), _p(xyz), _p(campos), _p(rr), _p(colors),
                                           _p(scales), _p(rot), _p(opac), _p(g_colors), _p(g_scales), _p(g_rot), _p(g_opac),
                                           _p(g_f_dc), _p(g_f_rest), _p(g_dshs), _p(g_xyz), _p(g_ls), _p(g_rr), _p(g_ol), _p(g_l1),
                                           _lib.stream_ptr()))
        return None, g_f_dc, g_f_rest, g_dshs, g_xyz, None, g_ls, g_rr, g_ol, None


def activations_and_colors(deg, f_dc, f_rest, dshs, xyz, campos, log_scales, rot_raw, opacity_logit, with_dshs_l1=False):
    """-> (colors_precomp [P,3], scales [P,3], rotations [P,4], opacity [P,1]); dshs may be None (coarse stage).
    with_dshs_l1=True appends mean|dshs| (the train.py:407-410 regulariser, differentiable) computed in the same pass."""
    out = _Glue.apply(deg, f_dc, f_rest, dshs, xyz, campos, log_scales, rot_raw, opacity_logit, with_dshs_l1)
    return out if with_dshs_l1 else out[:4]
