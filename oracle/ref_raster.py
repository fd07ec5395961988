"""ctypes front-end of oracle/_ref/libref_raster*.so: the REFERENCE rasterizer itself, compiled for gfx950.

TEST INFRASTRUCTURE ONLY (tests/ and tools/); the product never imports it.  `RefRaster` has the interface of
`oracle.oracle.RasterOracle` (numpy in, numpy out, same dict keys) so a parity test can run either checker, but what
executes underneath is the reference's own forward.cu / backward.cu / rasterizer_impl.cu (hipify-perl + hipcc, recipe in
oracle/build_ref.sh) on the GPU of the box: it needs a GPU, it is NOT a CPU baseline.

Two builds:  RefRaster()            -> -ffp-contract=off (one rounding per op; bit-comparable with the fp32 oracle)
             RefRaster(fma=True)    -> hipcc default contraction (what nvcc's default -fmad=true does to the reference)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import _make_structs

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")


def build(force: bool = False) -> None:
    """Runs oracle/build_ref.sh (a no-op where /root/reference is absent, i.e. on the GPU box)."""
    subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")] + (["-f"] if force else []))


def available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libref_raster.so"))


class RefRaster:
    dtype = np.dtype(np.float32)

    def __init__(self, fma: bool = False):
        path = os.path.join(REF_DIR, "libref_raster_fma.so" if fma else "libref_raster.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run oracle/build_ref.sh where /root/reference exists")
        self.lib = C.CDLL(path)
        assert self.lib.ref_sizeof_real() == 4
        self.Inputs, self.State = _make_structs(C.c_float)
        self.lib.ref_forward.restype = C.c_int
        self.lib.ref_backward.restype = None
        self.lib.ref_free.restype = None
        self.lib.ref_free.argtypes = [C.c_void_p]

    @staticmethod
    def _arr(a):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        return None if a.size == 0 else a

    @staticmethod
    def _ptr(a, ty=C.c_float):
        return None if a is None else a.ctypes.data_as(C.POINTER(ty))

    def forward(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height,
                image_width, sh_degree=0, scale_modifier=1.0, colors_precomp=None, shs=None, scales=None,
                rotations=None, cov3D_precomp=None):
        dt = np.float32
        means3D = self._arr(means3D)
        P = 0 if means3D is None else means3D.shape[0]
        H, W = int(image_height), int(image_width)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        keep = dict(bg=self._arr(bg), means3D=means3D, opacities=self._arr(opacities), shs=self._arr(shs),
                    colors_precomp=self._arr(colors_precomp), scales=self._arr(scales), rotations=self._arr(rotations),
                    cov3D_precomp=self._arr(cov3D_precomp), viewmatrix=self._arr(viewmatrix),
                    projmatrix=self._arr(projmatrix), campos=self._arr(campos))
        M = 0 if keep["shs"] is None else keep["shs"].shape[1]
        p = self._ptr
        inp = self.Inputs(P, int(sh_degree), M, W, H, p(keep["bg"]), p(means3D), p(keep["shs"]), p(keep["colors_precomp"]),
                          p(keep["opacities"]), p(keep["scales"]), scale_modifier, p(keep["rotations"]),
                          p(keep["cov3D_precomp"]), p(keep["viewmatrix"]), p(keep["projmatrix"]), p(keep["campos"]),
                          tanfovx, tanfovy)
        Pn = max(P, 1)
        sa = dict(depths=np.zeros(Pn, dt), clamped=np.zeros(Pn * 3, np.uint8), means2D=np.zeros((Pn, 2), dt),
                  cov3D=np.zeros((Pn, 6), dt), conic_opacity=np.zeros((Pn, 4), dt), rgb=np.zeros((Pn, 3), dt),
                  tiles_touched=np.zeros(Pn, np.uint32), point_offsets=np.zeros(Pn, np.uint32),
                  final_T=np.zeros(H * W, dt), n_contrib=np.zeros(H * W, np.uint32), ranges=np.zeros((gx * gy, 2), np.uint32))
        st = self.State(p(sa["depths"]), p(sa["clamped"], C.c_uint8), p(sa["means2D"]), p(sa["cov3D"]),
                        p(sa["conic_opacity"]), p(sa["rgb"]), p(sa["tiles_touched"], C.c_uint32),
                        p(sa["point_offsets"], C.c_uint32), p(sa["final_T"]), p(sa["n_contrib"], C.c_uint32),
                        p(sa["ranges"], C.c_uint32), None, None, 0)
        color = np.zeros((3, H, W), dt)
        depth = np.zeros((1, H, W), dt)
        radii = np.zeros(Pn, np.int32)
        handle = C.c_void_p()
        R = self.lib.ref_forward(C.byref(inp), C.byref(st), p(color), p(depth), p(radii, C.c_int), C.byref(handle))
        if R > 0:
            point_list = np.ctypeslib.as_array(st.point_list, shape=(R,)).copy()
            keys = np.ctypeslib.as_array(st.point_list_keys, shape=(R,)).copy()
        else:
            point_list, keys = np.zeros(0, np.uint32), np.zeros(0, np.uint64)
        self.lib.ref_free_binning(C.byref(st))
        sa.update(point_list=point_list, point_list_keys=keys)
        return dict(color=color, depth=depth, radii=radii[:P], num_rendered=R, state=sa, _keep=keep,
                    _handle=_Handle(self.lib, handle),
                    _cfg=dict(P=P, D=int(sh_degree), M=M, W=W, H=H, scale_modifier=scale_modifier, tanfovx=tanfovx,
                              tanfovy=tanfovy))

    def backward(self, fwd, dL_dout_color, dL_dout_depth):
        """Runs the reference backward on the forward's own device-side state (float atomics: run-to-run noise)."""
        dt = np.float32
        cfg = fwd["_cfg"]
        P, M, H, W = cfg["P"], cfg["M"], cfg["H"], cfg["W"]
        gc = np.ascontiguousarray(np.asarray(dL_dout_color, dt).reshape(3, H, W))
        gd = np.ascontiguousarray(np.asarray(dL_dout_depth, dt).reshape(H, W))
        Pn = max(P, 1)
        g = dict(dL_dmeans2D=np.zeros((Pn, 3), dt), dL_dconic=np.zeros((Pn, 2, 2), dt), dL_dopacity=np.zeros((Pn, 1), dt),
                 dL_dcolors=np.zeros((Pn, 3), dt), dL_ddepths=np.zeros((Pn, 1), dt), dL_dmeans3D=np.zeros((Pn, 3), dt),
                 dL_dcov3D=np.zeros((Pn, 6), dt), dL_dsh=np.zeros((Pn, max(M, 1), 3), dt), dL_dscales=np.zeros((Pn, 3), dt),
                 dL_drotations=np.zeros((Pn, 4), dt))
        p = self._ptr
        self.lib.ref_backward(fwd["_handle"].h, p(gc), p(gd), p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]),
                              p(g["dL_dcolors"]), p(g["dL_ddepths"]), p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]),
                              p(g["dL_dsh"]) if M else None, p(g["dL_dscales"]), p(g["dL_drotations"]))
        out = {k: v[:P] for k, v in g.items()}
        out["dL_dsh"] = out["dL_dsh"][:, :M]
        return out

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        m = self._arr(means3D)
        P = m.shape[0]
        out = np.zeros(P, np.uint8)
        v, pr = self._arr(viewmatrix), self._arr(projmatrix)
        self.lib.ref_mark_visible(P, self._ptr(m), self._ptr(v), self._ptr(pr), self._ptr(out, C.c_uint8))
        return out.astype(bool)


class _Handle:
    """Owns the device-side arenas of one reference forward until the dict that holds it dies."""

    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        if self.h:
            self.lib.ref_free(self.h)
            self.h = None


def knn_available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libref_knn.so"))


def ref_knn_mean_dist2(points: np.ndarray) -> np.ndarray:
    """The REFERENCE's own SimpleKNN::knn (simple_knn.cu:185-221, oracle/_ref/libref_knn.so via oracle/ref_knn_shim.cpp) on the
    GPU of the box: float32 [P,3] -> float32 [P] (what `distCUDA2` returns, spatial.cu:15-26)."""
    path = os.path.join(REF_DIR, "libref_knn.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run oracle/build_ref.sh where /root/reference exists")
    lib = C.CDLL(path)
    lib.ref_knn_mean_dist2.restype = C.c_int
    pts = np.ascontiguousarray(points, dtype=np.float32)
    out = np.zeros(pts.shape[0], np.float32)
    rc = lib.ref_knn_mean_dist2(int(pts.shape[0]), pts.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise RuntimeError(f"ref_knn_mean_dist2 failed ({rc})")
    return out
