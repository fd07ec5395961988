"""ctypes/numpy front-end of the CPU oracle (oracle/raster_oracle.c, oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (s3gaussian_amd/, diff_gaussian_rasterization/, simple_knn/) never imports it.

Argument names and layouts mirror the reference's `_C.rasterize_gaussians` /
`_C.rasterize_gaussians_backward` (RAST/rasterize_points.h:18-63) so tests read like reference calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile the oracle shared objects with gcc (building the checker is not using it)."""
    targets = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so", "liboracle_knn.so")]
    srcs = [os.path.join(_HERE, n) for n in ("raster_oracle.c", "knn_oracle.c", "Makefile")]
    newest_src = max(os.path.getmtime(s) for s in srcs)
    if force or not all(os.path.exists(t) and os.path.getmtime(t) >= newest_src for t in targets):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))




def _make_structs(real):
    rp = C.POINTER(real)

    class Inputs(C.Structure):
        _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
                    ("background", rp), ("means3D", rp), ("shs", rp), ("colors_precomp", rp), ("opacities", rp),
                    ("scales", rp), ("scale_modifier", real), ("rotations", rp), ("cov3D_precomp", rp),
                    ("viewmatrix", rp), ("projmatrix", rp), ("cam_pos", rp), ("tan_fovx", real), ("tan_fovy", real)]

    class State(C.Structure):
        _fields_ = [("depths", rp), ("clamped", C.POINTER(C.c_uint8)), ("means2D", rp), ("cov3D", rp),
                    ("conic_opacity", rp), ("rgb", rp), ("tiles_touched", C.POINTER(C.c_uint32)),
                    ("point_offsets", C.POINTER(C.c_uint32)), ("final_T", rp), ("n_contrib", C.POINTER(C.c_uint32)),
                    ("ranges", C.POINTER(C.c_uint32)), ("point_list_keys", C.POINTER(C.c_uint64)),
                    ("point_list", C.POINTER(C.c_uint32)), ("num_rendered", C.c_int)]

    return Inputs, State


class RasterOracle:
    """fp32 (default) or fp64 build of the rasterizer restatement."""

    def __init__(self, dtype=np.float32):
        build()
        self.dtype = np.dtype(dtype)
        if self.dtype == np.float32:
            self.real, name = C.c_float, "liboracle_f32.so"
        elif self.dtype == np.float64:
            self.real, name = C.c_double, "liboracle_f64.so"
        else:
            raise ValueError(dtype)
        self.lib = C.CDLL(os.path.join(_HERE, name))
        assert self.lib.orc_sizeof_real() == self.dtype.itemsize
        self.Inputs, self.State = _make_structs(self.real)
        self.lib.orc_forward.restype = C.c_int

    # -- helpers ---------------------------------------------------------------------------------------
    def _arr(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
        if a.size == 0:
            return None  # reference: numel()==0 tensor -> nullptr
        if shape is not None:
            a = a.reshape(shape)
        return a

    def _ptr(self, a, ty=None):
        if a is None:
            return None
        return a.ctypes.data_as(C.POINTER(ty or self.real))

    # -- forward ---------------------------------------------------------------------------------------
    def forward(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height,
                image_width, sh_degree=0, scale_modifier=1.0, colors_precomp=None, shs=None, scales=None,
                rotations=None, cov3D_precomp=None):
        dt = self.dtype
        means3D = self._arr(means3D)
        P = 0 if means3D is None else means3D.shape[0]
        H, W = int(image_height), int(image_width)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        keep = dict(bg=self._arr(bg), means3D=means3D, opacities=self._arr(opacities), shs=self._arr(shs),
                    colors_precomp=self._arr(colors_precomp), scales=self._arr(scales),
                    rotations=self._arr(rotations), cov3D_precomp=self._arr(cov3D_precomp),
                    viewmatrix=self._arr(viewmatrix), projmatrix=self._arr(projmatrix), campos=self._arr(campos))
        M = 0 if keep["shs"] is None else keep["shs"].shape[1]
        inp = self.Inputs(P, int(sh_degree), M, W, H, self._ptr(keep["bg"]), self._ptr(means3D), self._ptr(keep["shs"]),
                          self._ptr(keep["colors_precomp"]), self._ptr(keep["opacities"]), self._ptr(keep["scales"]),
                          scale_modifier, self._ptr(keep["rotations"]), self._ptr(keep["cov3D_precomp"]),
                          self._ptr(keep["viewmatrix"]), self._ptr(keep["projmatrix"]), self._ptr(keep["campos"]),
                          tanfovx, tanfovy)
        Pn = max(P, 1)
        st_arrays = dict(depths=np.zeros(Pn, dt), clamped=np.zeros(Pn * 3, np.uint8), means2D=np.zeros((Pn, 2), dt),
                         cov3D=np.zeros((Pn, 6), dt), conic_opacity=np.zeros((Pn, 4), dt), rgb=np.zeros((Pn, 3), dt),
                         tiles_touched=np.zeros(Pn, np.uint32), point_offsets=np.zeros(Pn, np.uint32),
                         final_T=np.zeros(H * W, dt), n_contrib=np.zeros(H * W, np.uint32),
                         ranges=np.zeros((gx * gy, 2), np.uint32))
        st = self.State(self._ptr(st_arrays["depths"]), self._ptr(st_arrays["clamped"], C.c_uint8),
                        self._ptr(st_arrays["means2D"]), self._ptr(st_arrays["cov3D"]),
                        self._ptr(st_arrays["conic_opacity"]), self._ptr(st_arrays["rgb"]),
                        self._ptr(st_arrays["tiles_touched"], C.c_uint32), self._ptr(st_arrays["point_offsets"], C.c_uint32),
                        self._ptr(st_arrays["final_T"]), self._ptr(st_arrays["n_contrib"], C.c_uint32),
                        self._ptr(st_arrays["ranges"], C.c_uint32), None, None, 0)
        color = np.zeros((3, H, W), dt)
        depth = np.zeros((1, H, W), dt)
        radii = np.zeros(Pn, np.int32)
        R = self.lib.orc_forward(C.byref(inp), C.byref(st), self._ptr(color), self._ptr(depth), self._ptr(radii, C.c_int))
        if R > 0:
            point_list = np.ctypeslib.as_array(st.point_list, shape=(R,)).copy()
            keys = np.ctypeslib.as_array(st.point_list_keys, shape=(R,)).copy()
        else:
            point_list = np.zeros(0, np.uint32)
            keys = np.zeros(0, np.uint64)
        self.lib.orc_free_binning(C.byref(st))
        st_arrays.update(point_list=point_list, point_list_keys=keys)
        return dict(color=color, depth=depth, radii=radii[:P], num_rendered=R, state=st_arrays, _keep=keep,
                    _cfg=dict(P=P, D=int(sh_degree), M=M, W=W, H=H, scale_modifier=scale_modifier, tanfovx=tanfovx,
                              tanfovy=tanfovy))

    # -- backward --------------------------------------------------------------------------------------
    def backward(self, fwd, dL_dout_color, dL_dout_depth):
        dt = self.dtype
        cfg, keep, sa = fwd["_cfg"], fwd["_keep"], fwd["state"]
        P, M, H, W = cfg["P"], cfg["M"], cfg["H"], cfg["W"]
        inp = self.Inputs(P, cfg["D"], M, W, H, self._ptr(keep["bg"]), self._ptr(keep["means3D"]), self._ptr(keep["shs"]),
                          self._ptr(keep["colors_precomp"]), self._ptr(keep["opacities"]), self._ptr(keep["scales"]),
                          cfg["scale_modifier"], self._ptr(keep["rotations"]), self._ptr(keep["cov3D_precomp"]),
                          self._ptr(keep["viewmatrix"]), self._ptr(keep["projmatrix"]), self._ptr(keep["campos"]),
                          cfg["tanfovx"], cfg["tanfovy"])
        pl = np.ascontiguousarray(sa["point_list"]) if sa["point_list"].size else np.zeros(1, np.uint32)
        pk = np.ascontiguousarray(sa["point_list_keys"]) if sa["point_list_keys"].size else np.zeros(1, np.uint64)
        st = self.State(self._ptr(sa["depths"]), self._ptr(sa["clamped"], C.c_uint8), self._ptr(sa["means2D"]),
                        self._ptr(sa["cov3D"]), self._ptr(sa["conic_opacity"]), self._ptr(sa["rgb"]),
                        self._ptr(sa["tiles_touched"], C.c_uint32), self._ptr(sa["point_offsets"], C.c_uint32),
                        self._ptr(sa["final_T"]), self._ptr(sa["n_contrib"], C.c_uint32), self._ptr(sa["ranges"], C.c_uint32),
                        self._ptr(pk, C.c_uint64), self._ptr(pl, C.c_uint32), fwd["num_rendered"])
        gc = np.ascontiguousarray(np.asarray(dL_dout_color, dt).reshape(3, H, W))
        gd = np.ascontiguousarray(np.asarray(dL_dout_depth, dt).reshape(H, W))
        Pn = max(P, 1)
        g = dict(dL_dmeans2D=np.zeros((Pn, 3), dt), dL_dconic=np.zeros((Pn, 2, 2), dt), dL_dopacity=np.zeros((Pn, 1), dt),
                 dL_dcolors=np.zeros((Pn, 3), dt), dL_ddepths=np.zeros((Pn, 1), dt), dL_dmeans3D=np.zeros((Pn, 3), dt),
                 dL_dcov3D=np.zeros((Pn, 6), dt), dL_dsh=np.zeros((Pn, max(M, 1), 3), dt), dL_dscales=np.zeros((Pn, 3), dt),
                 dL_drotations=np.zeros((Pn, 4), dt))
        radii = np.ascontiguousarray(np.concatenate([fwd["radii"], np.zeros(Pn - P, np.int32)]).astype(np.int32))
        self.lib.orc_backward(C.byref(inp), C.byref(st), self._ptr(radii, C.c_int), self._ptr(gc), self._ptr(gd),
                              self._ptr(g["dL_dmeans2D"]), self._ptr(g["dL_dconic"]), self._ptr(g["dL_dopacity"]),
                              self._ptr(g["dL_dcolors"]), self._ptr(g["dL_ddepths"]), self._ptr(g["dL_dmeans3D"]),
                              self._ptr(g["dL_dcov3D"]), self._ptr(g["dL_dsh"]), self._ptr(g["dL_dscales"]),
                              self._ptr(g["dL_drotations"]))
        out = {k: v[:P] for k, v in g.items()}
        out["dL_dsh"] = out["dL_dsh"][:, :M]
        return out

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        m = self._arr(means3D)
        P = m.shape[0]
        out = np.zeros(P, np.uint8)
        v, p = self._arr(viewmatrix), self._arr(projmatrix)
        self.lib.orc_mark_visible(P, self._ptr(m), self._ptr(v), self._ptr(p), self._ptr(out, C.c_uint8))
        return out.astype(bool)


def knn_mean_dist2(points) -> np.ndarray:
    """distCUDA2 restatement: mean squared distance to the 3 nearest neighbours, float32 [P]."""
    build()
    lib = C.CDLL(os.path.join(_HERE, "liboracle_knn.so"))
    pts = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 3))
    out = np.zeros(pts.shape[0], np.float32)
    lib.orc_knn_mean_dist2(pts.shape[0], pts.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
