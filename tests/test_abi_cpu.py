"""CPU-only: the C-ABI library loads, exports every symbol include/*.h declares, and the Python surface mirrors the
reference's names / signatures / error behaviour.  No compute calls (no GPU here)."""
import ctypes
import glob
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(s3g_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_library_exports_every_declared_symbol():
    from s3gaussian_amd import _lib
    L = _lib.lib()
    decl = _declared_functions()
    assert "s3g_raster_forward" in decl and "s3g_raster_backward" in decl
    for name in decl:
        assert hasattr(L, name), f"libs3g.so does not export {name}"
    assert L.s3g_abi_version() == _lib.ABI_VERSION
    assert set(_lib.EXPORTED_SYMBOLS) <= set(decl)


def test_raster_inputs_struct_matches_header_field_order():
    from s3gaussian_amd import _lib
    txt = open(os.path.join(ROOT, "include", "s3g_raster.h")).read()
    body = re.search(r"typedef struct s3g_raster_inputs \{(.*?)\} s3g_raster_inputs;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert fields == [f[0] for f in _lib.RasterInputs._fields_]


def _struct_fields(header, name):
    txt = open(os.path.join(ROOT, "include", header)).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = re.sub(r"\[[^\]]*\]", "", decl.strip())      # drop array extents
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    return fields


def test_other_structs_match_their_headers():
    """ctypes mirrors of the descriptor structs keep the headers' field order (and sizes where arrays are involved)."""
    from s3gaussian_amd import hexplane, losses, mlp
    assert _struct_fields("s3g_hexplane.h", "s3g_hexplane_desc") == [f[0] for f in hexplane._HexDesc._fields_]
    assert ctypes.sizeof(hexplane._HexDesc) == 4 + 8 * 4 * 4 + 4 + 8 * 6 * 8 + 6 * 4 + 4 + 4   # int, res, pad, planes, aabb, flag, pad
    assert _struct_fields("s3g_loss.h", "s3g_plane_reg_desc") == [f[0] for f in losses._PlaneRegDesc._fields_]
    assert _struct_fields("s3g_mlp.h", "s3g_mlp_params") == [f[0] for f in mlp._Params._fields_]
    from s3gaussian_amd import optim
    assert _struct_fields("s3g_optim.h", "s3g_adam_tensor") == [f[0] for f in optim._AdamTensor._fields_]
    assert ctypes.sizeof(optim._AdamTensor) == 56


def test_python_surface_matches_reference_names():
    import diff_gaussian_rasterization as d
    assert d.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(d.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert hasattr(d.GaussianRasterizer, "markVisible")
    for fn in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert callable(getattr(d._C, fn))
    from simple_knn._C import distCUDA2
    assert callable(distCUDA2)


def _settings():
    import diff_gaussian_rasterization as d
    return d.GaussianRasterizationSettings(image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                           scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                           campos=torch.zeros(3), prefiltered=False, debug=False)


def test_argument_validation_raises_like_reference():
    import diff_gaussian_rasterization as d
    r = d.GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4),
          cov3D_precomp=torch.zeros(4, 6))


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must fail loudly, never silently route to a slow path."""
    import diff_gaussian_rasterization as d
    from simple_knn._C import distCUDA2
    r = d.GaussianRasterizer(_settings())
    x = torch.rand(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        distCUDA2(x)


def test_product_package_never_imports_oracle():
    for root in ("s3gaussian_amd", "diff_gaussian_rasterization", "simple_knn"):
        for path in glob.glob(os.path.join(ROOT, root, "**", "*.py"), recursive=True):
            src = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path


def test_committed_bench_line_follows_the_contract():
    """profiles/rNN_bench_line.json is what `python bench.py` printed on the MI355X at the end of a round; the same validator
    runs on a LIVE bench.py invocation in tests/test_bench_gpu.py (the contract is tested on the program, not only on a file)."""
    import json
    from tests.util import validate_bench_line
    path = os.path.join(ROOT, "profiles", "r04_bench_line.json")
    if os.path.exists(path):
        d = json.loads(open(path).read().strip().splitlines()[-1])
        validate_bench_line(d, default_workload=True)
        assert d["config"]["raster_async"]["enabled"] is True and "host-asynchronous" in d["config"]["rasterizer_forward"]
        d["config"]["paths"]["something_else"] = {"ms_per_step": 1.0, "iters_per_s": 1000.0}     # an unknown path key does not pass
        with pytest.raises(AssertionError):
            validate_bench_line(d, default_workload=True)
    # round 3's lines (blend kernels priced as HBM kernels, render-loop launches mixed into the training bracket: fixed in round 4)
    for name in ("r03_bench_line.json", "r03_bench_line_split.json"):
        old = os.path.join(ROOT, "profiles", name)
        if os.path.exists(old):
            validate_bench_line(json.loads(open(old).read().strip().splitlines()[-1]), default_workload=True, round3_accounting=True)


def test_the_drivers_own_bench_records_validate():
    """BENCH_rNN.json is what the driver measured on its own box (its `parsed` is a trimmed copy of the printed line).  Round 3's
    validator rejected the driver's line (an ordering between two measured fast paths that the driver's box violated) and nobody
    noticed before the round ended: every record in the tree goes through the validator here."""
    import json
    from tests.util import validate_bench_line
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
        parsed = json.load(open(path)).get("parsed")
        if not isinstance(parsed, dict) or "value" not in parsed:
            continue
        if "roofline" not in parsed or "cpu_baseline" not in parsed or "path" not in parsed.get("config", {}):
            continue                                                      # round 1's line predates the contract's additions
        validate_bench_line(parsed, default_workload=True)
        seen += 1
    if not seen:
        pytest.skip("no driver record with a roofline object in the tree")


def test_mlp_arithmetic_switch_round_trips_without_a_gpu():
    """include/s3g_mlp.h::s3g_deform_mlp_set_arithmetic is a host-side, process-wide setting: default exact fp32, the two documented
    modes accepted, anything else refused with S3G_ERR_INVALID_ARG and the setting left alone."""
    import pytest
    from s3gaussian_amd import _lib, mlp
    L = _lib.lib()
    L.s3g_last_error.restype = __import__("ctypes").c_char_p
    # (the Python binding selects mlp.DEFAULT_ARITHMETIC -- "bf16x3" since round 6 -- when it first binds the library, whose own
    #  default is the exact chain)
    assert mlp.get_mlp_arithmetic() == mlp.DEFAULT_ARITHMETIC == "bf16x3" and L.s3g_deform_mlp_get_arithmetic() == 1
    try:
        mlp.set_mlp_arithmetic("f32")
        assert mlp.get_mlp_arithmetic() == "f32" and L.s3g_deform_mlp_get_arithmetic() == 0
        mlp.set_mlp_arithmetic("bf16x3")
        assert mlp.get_mlp_arithmetic() == "bf16x3" and L.s3g_deform_mlp_get_arithmetic() == 1
        assert L.s3g_deform_mlp_set_arithmetic(7) != 0 and b"arithmetic" in L.s3g_last_error()
        assert mlp.get_mlp_arithmetic() == "bf16x3"
        with pytest.raises(ValueError):
            mlp.set_mlp_arithmetic("bf16")
    finally:
        mlp.set_mlp_arithmetic(mlp.DEFAULT_ARITHMETIC)
    with pytest.raises(ValueError):          # the inference kernel's switch is validated before anything touches a device
        mlp.deform_infer(None, None, None, None, None, None, None, arithmetic="fp16")


def test_removed_hexplane_algorithm_is_refused_not_remapped():
    """ABI 12 removed the slab-free "walk" backward (S3G_HEX_WALK = 1): the entry point must say so instead of silently running
    something else under that id (argument validation only: nothing is launched, no GPU needed)."""
    import ctypes as C
    from s3gaussian_amd import _lib
    L = _lib.lib()
    L.s3g_hexplane_backward_algo.restype = C.c_int
    rc = L.s3g_hexplane_backward_algo(None, C.c_int(0), None, None, None, None, C.c_int(1), None, None, None, None, C.c_int(0), None)
    assert rc == 1 and b"removed in ABI 12" in L.s3g_last_error()
