"""ctypes binding of libs3g.so (the C ABI declared in include/*.h).

The product path has NO fallback: if the shared library is missing or was not built for this tree, importing
an op raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C s3gaussian_amd/csrc` (hipcc cross-compiles gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# S3G_LIB_PATH: load another build of the SAME ABI instead (kernel A/B experiments on the GPU box: tools/mkvariants.py);
# the default, and the only path the tests and bench.py use, is the in-tree library.
LIB_PATH = os.environ.get("S3G_LIB_PATH") or os.path.join(_HERE, "lib", "libs3g.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 13

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class RasterInputs(C.Structure):
    """struct s3g_raster_inputs (include/s3g_raster.h)."""
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("scale_modifier", C.c_float), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int), ("debug", C.c_int),
    ]


class DensifyAccum(C.Structure):
    """struct s3g_densify_accum (include/s3g_raster.h)."""
    _fields_ = [("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p), ("max_radii2D", C.c_void_p)]


class RasterAsync(C.Structure):
    """struct s3g_raster_async (include/s3g_raster.h)."""
    _fields_ = [("capacity_instances", C.c_uint32), ("capacity_slots", C.c_uint32), ("sort_lds_keys", C.c_uint32),
                ("long_lists", C.c_int), ("geometry_arena", C.c_void_p), ("binning_arena", C.c_void_p),
                ("image_arena", C.c_void_p), ("status_device", C.c_void_p), ("status_host", C.c_void_p),
                ("forward_only", C.c_int), ("sticky_device", C.c_void_p), ("status_event", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into s3gaussian_amd/lib/libs3g.so (in-tree, travels with gpurun)."""
    cmd = ["make", "-s", "-C", CSRC_DIR, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built. "
            "Run `make -C s3gaussian_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.s3g_last_error.restype = C.c_char_p
    L.s3g_abi_version.restype = C.c_int
    if L.s3g_abi_version() != ABI_VERSION:
        raise ImportError(f"libs3g.so ABI {L.s3g_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    vp = C.c_void_p
    L.s3g_raster_forward.restype = C.c_int
    L.s3g_raster_forward.argtypes = [C.POINTER(RasterInputs), RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, vp, vp, vp,
                                     C.POINTER(C.c_int), vp]
    L.s3g_raster_forward_reuse.restype = C.c_int
    L.s3g_raster_forward_reuse.argtypes = [C.POINTER(RasterInputs), C.c_int, vp, vp, vp, vp, vp, vp]
    L.s3g_raster_backward.restype = C.c_int
    L.s3g_raster_backward.argtypes = [C.POINTER(RasterInputs), C.c_int] + [vp] * 18
    L.s3g_raster_backward_workspace_bytes.restype = C.c_size_t
    L.s3g_raster_backward_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.s3g_raster_forward2.restype = C.c_int
    L.s3g_raster_forward2.argtypes = [C.POINTER(RasterInputs), vp, RESIZE_FN, vp, RESIZE_FN, vp, RESIZE_FN, vp, vp, vp, vp, vp,
                                      C.POINTER(C.c_int), vp]
    L.s3g_raster_set_exact_cull.restype = None
    L.s3g_raster_set_exact_cull.argtypes = [C.c_int]
    L.s3g_raster_get_exact_cull.restype = C.c_int
    L.s3g_raster_backward2.restype = C.c_int
    L.s3g_raster_backward2.argtypes = [C.POINTER(RasterInputs), vp, C.c_int] + [vp] * 19
    L.s3g_raster_backward2_workspace_bytes.restype = C.c_size_t
    L.s3g_raster_backward2_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.s3g_mark_visible.restype = C.c_int
    L.s3g_mark_visible.argtypes = [C.c_int, vp, vp, vp, vp, vp]
    L.s3g_raster_forward_decompose.restype = C.c_int
    L.s3g_raster_forward_decompose.argtypes = [C.POINTER(RasterInputs), C.c_int] + [vp] * 10
    L.s3g_raster_backward_accum.restype = C.c_int
    L.s3g_raster_backward_accum.argtypes = [C.POINTER(RasterInputs), C.c_int] + [vp] * 17 + [C.POINTER(DensifyAccum), vp]
    L.s3g_raster_backward2_accum.restype = C.c_int
    L.s3g_raster_backward2_accum.argtypes = [C.POINTER(RasterInputs), vp, C.c_int] + [vp] * 18 + [C.POINTER(DensifyAccum), vp]
    L.s3g_raster_arena_bytes.restype = C.c_int
    L.s3g_raster_arena_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32] + [C.POINTER(C.c_size_t)] * 3
    L.s3g_raster_forward_async.restype = C.c_int
    L.s3g_raster_forward_async.argtypes = [C.POINTER(RasterInputs), vp, C.POINTER(RasterAsync), vp, vp, vp, vp, vp]
    L.s3g_densify_stats.restype = C.c_int
    L.s3g_densify_stats.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.s3g_densify_stats_guarded.restype = C.c_int
    L.s3g_densify_stats_guarded.argtypes = [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


# ---- host-side shortcuts (round 6) -----------------------------------------------------------------------------------------------
# `torch.cuda.current_stream().cuda_stream` builds a Stream object and resolves the device three times: 17 us per call, thirteen calls
# per training iteration; `with torch.cuda.device(dev)` is 8 us more, twenty-four times -- 0.4 ms of host time per iteration, and on
# the zero-edit route (the reference's train.py waits for the device every iteration) host time in front of a launch is GPU idle time
# (profiles/r06_patched_host_profile.txt).  The raw-stream call below is what the Stream object wraps; both fall back to the public API.
def _bind_fast_paths():
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    cur = getattr(torch._C, "_cuda_getDevice", None)
    return torch, raw, cur


_torch = _raw_stream = _cur_device = None


def stream_ptr() -> int:
    """The raw HIP stream behind torch.cuda.current_stream() of the CURRENT device."""
    global _torch, _raw_stream, _cur_device
    if _torch is None:
        _torch, _raw_stream, _cur_device = _bind_fast_paths()
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return _torch.cuda.current_stream().cuda_stream


class _NullContext:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullContext()


def on_device(dev):
    """`with on_device(t.device):` -- torch.cuda.device(dev) unless `dev` already is the current device (then nothing at all)."""
    global _torch, _raw_stream, _cur_device
    if _torch is None:
        _torch, _raw_stream, _cur_device = _bind_fast_paths()
    if _cur_device is not None and getattr(dev, "index", None) is not None and _cur_device() == dev.index:
        return _NULL
    return _torch.cuda.device(dev)


def check(code: int) -> None:
    if code != 0:
        msg = lib().s3g_last_error().decode("utf-8", "replace")
        if code == 1:
            raise Exception(msg)  # reference raises plain Exception for bad argument combos
        raise RuntimeError(f"libs3g error {code}: {msg}")


EXPORTED_SYMBOLS = ["s3g_raster_forward", "s3g_raster_forward_reuse", "s3g_raster_forward_decompose", "s3g_raster_forward2", "s3g_raster_forward_async", "s3g_raster_arena_bytes", "s3g_adam_step_guarded", "s3g_raster_backward", "s3g_raster_backward_workspace_bytes", "s3g_raster_backward2", "s3g_raster_backward2_workspace_bytes", "s3g_hexplane_forward_workspace_bytes", "s3g_knn_workspace_bytes", "s3g_knn_mean_dist2", "s3g_hexplane_forward", "s3g_hexplane_backward", "s3g_hexplane_backward_algo", "s3g_hexplane_backward_workspace_bytes", "s3g_hexplane_backward_scratch_rows", "s3g_hexplane_debug_walk_mask", "s3g_hexplane_set_deterministic", "s3g_hexplane_get_deterministic", "s3g_hexplane_sort_state_words", "s3g_profile_enable", "s3g_profile_read", "s3g_ssim_forward", "s3g_ssim_backward", "s3g_plane_regulation", "s3g_scale_unless_one", "s3g_pixel_losses_forward", "s3g_pixel_losses_combine", "s3g_pixel_losses_backward", "s3g_glue_forward", "s3g_glue_backward", "s3g_deform_mlp_forward", "s3g_deform_mlp_backward", "s3g_deform_mlp_backward_ordered", "s3g_deform_mlp_wgrad_partial_bytes", "s3g_deform_mlp_stash_bytes", "s3g_deform_mlp_pack_bytes", "s3g_deform_mlp_set_arithmetic", "s3g_deform_mlp_get_arithmetic", "s3g_deform_infer", "s3g_deform_infer_split", "s3g_deform_infer_workspace_bytes", "s3g_adam_step", "s3g_densify_stats", "s3g_densify_stats_guarded", "s3g_raster_backward_accum", "s3g_raster_backward2_accum", "s3g_mark_visible", "s3g_raster_set_exact_cull", "s3g_raster_get_exact_cull", "s3g_raster_set_bin_band", "s3g_last_error", "s3g_abi_version"]
