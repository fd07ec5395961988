"""oracle/ref_diff_raster_C.py -- TEST INFRASTRUCTURE ONLY.

The native module `diff_gaussian_rasterization._C` of the REFERENCE (RAST/ext.cpp:15-19, RAST/rasterize_points.h:18-68), rebuilt as
a ctypes binding of oracle/_ref/libref_raster*.so -- the reference's own forward.cu / backward.cu / rasterizer_impl.cu compiled for
gfx950 by oracle/build_ref.sh.  With this file standing where the pybind module stood, the reference's own Python wrapper
(`diff_gaussian_rasterization/__init__.py`), its own `gaussian_renderer.render`, `GaussianModel` and `train.py` run END TO END on
the MI355X with not one kernel of the product involved: that is the oracle trainer of the BASELINE-size PSNR-parity experiment
(tests/test_psnr_parity_cfg2_gpu.py, oracle/ref_py.py::load(rasterizer="reference")).

Same signatures, zero-filled outputs, return order and `numel() == 0 -> nullptr` convention as RAST/rasterize_points.cu:35-202.
S3G_REF_RASTER_FMA=0 selects the one-rounding build; the default is hipcc's default contraction, the analogue of nvcc's -fmad=true,
i.e. the arithmetic a user of the reference runs.  Never imported by the product.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_FMA = os.environ.get("S3G_REF_RASTER_FMA", "1") != "0"
_PATH = os.path.join(_HERE, "_ref", "libref_raster_fma.so" if _FMA else "libref_raster.so")
_RESIZE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_lib = None


class _Inputs(C.Structure):       # struct Inputs of oracle/ref_shim.cpp
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int), ("background", C.c_void_p),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p),
                ("scales", C.c_void_p), ("scale_modifier", C.c_float), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p), ("tan_fovx", C.c_float),
                ("tan_fovy", C.c_float)]


def _L():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} missing: run oracle/build_ref.sh where /root/reference exists")
        _lib = C.CDLL(_PATH)
        vp = C.c_void_p
        _lib.ref_forward_dev.restype = C.c_int
        _lib.ref_forward_dev.argtypes = [C.POINTER(_Inputs), _RESIZE, vp, _RESIZE, vp, _RESIZE, vp, vp, vp, vp, C.c_int, C.c_int]
        _lib.ref_backward_dev.restype = None
        _lib.ref_backward_dev.argtypes = [C.POINTER(_Inputs), C.c_int] + [vp] * 16 + [C.c_int]
        _lib.ref_mark_visible_dev.restype = None
        _lib.ref_mark_visible_dev.argtypes = [C.c_int, vp, vp, vp, vp]
    return _lib


def _p(t):
    return None if (t is None or t.numel() == 0) else t.data_ptr()


class _Arena:
    def __init__(self, device):
        self.t = torch.empty(0, dtype=torch.uint8, device=device)

        def cb(_user, n):
            self.t = torch.empty(int(n), dtype=torch.uint8, device=device)
            return self.t.data_ptr()

        self.cb = _RESIZE(cb)


def _inputs(P, D, M, W, H, bg, means3D, sh, colors, opacity, scales, scale_modifier, rotations, cov3D, view, proj, campos, tx, ty):
    return _Inputs(P, int(D), int(M), int(W), int(H), _p(bg), _p(means3D), _p(sh), _p(colors), _p(opacity), _p(scales),
                   float(scale_modifier), _p(rotations), _p(cov3D), _p(view), _p(proj), _p(campos), float(tx), float(ty))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug):
    """RasterizeGaussiansCUDA, RAST/rasterize_points.cu:35-117."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    out_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Arena(dev), _Arena(dev), _Arena(dev)
    rendered = 0
    if P != 0:
        keep = [t.contiguous() if t.numel() else t for t in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                                             viewmatrix, projmatrix, campos)]
        bg_, m3_, sh_, col_, op_, sc_, rot_, cov_, view_, proj_, cam_ = keep
        M = sh_.size(1) if sh_.numel() != 0 else 0
        inp = _inputs(P, degree, M, W, H, bg_, m3_, sh_, col_, op_, sc_, scale_modifier, rot_, cov_, view_, proj_, cam_, tan_fovx, tan_fovy)
        with torch.cuda.device(dev):
            rendered = _L().ref_forward_dev(C.byref(inp), geom.cb, None, binning.cb, None, img.cb, None, out_color.data_ptr(),
                                            out_depth.data_ptr(), radii.data_ptr(), int(bool(prefiltered)), int(bool(debug)))
    return rendered, out_color, out_depth, radii, geom.t, binning.t, img.t


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA, RAST/rasterize_points.cu:119-202: ten zero-filled arrays, eight returned."""
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_ddepths = z(P, 3), z(P, 3), z(P, 3), z(P, 1)
    dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = z(P, 2, 2), z(P, 1), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4)
    if P != 0:
        keep = [t.contiguous() if t.numel() else t for t in (background, means3D, sh, colors, scales, rotations, cov3D_precomp,
                                                             viewmatrix, projmatrix, campos, dL_dout_color, dL_dout_depth, radii)]
        bg_, m3_, sh_, col_, sc_, rot_, cov_, view_, proj_, cam_, gcol_, gdep_, radii_ = keep
        inp = _inputs(P, degree, M, W, H, bg_, m3_, sh_, col_, None, sc_, scale_modifier, rot_, cov_, view_, proj_, cam_, tan_fovx, tan_fovy)
        with torch.cuda.device(dev):
            _L().ref_backward_dev(C.byref(inp), int(R), radii_.data_ptr(), _p(geomBuffer), _p(binningBuffer), _p(imageBuffer),
                                  gcol_.data_ptr(), gdep_.data_ptr(), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(),
                                  dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_ddepths.data_ptr(), dL_dmeans3D.data_ptr(),
                                  dL_dcov3D.data_ptr(), _p(dL_dsh), dL_dscales.data_ptr(), dL_drotations.data_ptr(), int(bool(debug)))
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, RAST/rasterize_points.cu:204-223."""
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m, v, pr = means3D.contiguous(), viewmatrix.contiguous(), projmatrix.contiguous()
        with torch.cuda.device(means3D.device):
            _L().ref_mark_visible_dev(P, m.data_ptr(), v.data_ptr(), pr.data_ptr(), present.data_ptr())
    return present


def distCUDA2(points):
    """simple_knn._C.distCUDA2 (KNN/spatial.cu:15-26) on the reference's own SimpleKNN::knn build (oracle/_ref/libref_knn.so)."""
    from .ref_raster import ref_knn_mean_dist2
    return torch.from_numpy(ref_knn_mean_dist2(points.detach().cpu().numpy())).to(points.device)
