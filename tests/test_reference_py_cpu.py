"""oracle/ref_py.py on the CPU: the archive of the reference's own Python files (oracle/_ref/reference_py.tar.gz) is what the
reference ships, byte for byte; it imports under the reference's own module names with stubs for the packages the image lacks; and
`patch_reference()` binds to the REAL modules.  (What those modules compute on the MI355X: tests/test_reference_py_gpu.py.)"""
import hashlib
import os
import tarfile

import pytest

from oracle import ref_py

needs_archive = pytest.mark.skipif(not ref_py.available(), reason="oracle/_ref/reference_py.tar.gz not built")


@pytest.mark.skipif(not ref_py.reference_present(), reason="/root/reference not present (GPU box)")
def test_archive_is_the_reference_byte_for_byte_and_nothing_else():
    assert ref_py.pack() == ref_py.ARCHIVE
    with tarfile.open(ref_py.ARCHIVE, "r:gz") as tar:
        names = tar.getnames()
        assert "train.py" in names and "gaussian_renderer/__init__.py" in names and "scene/gaussian_model.py" in names
        assert ref_py.RAST_WRAPPER in names
        assert all(n.endswith(".py") and (n.split("/")[0] in ref_py.TOP_LEVEL + ("train.py",) or n == ref_py.RAST_WRAPPER) for n in names)
        for n in names:
            want = hashlib.sha256(open(os.path.join(ref_py.REF, n), "rb").read()).hexdigest()
            assert hashlib.sha256(tar.extractfile(n).read()).hexdigest() == want, n
    # the archive is a built artefact: ignored by git, like the reference's compiled kernels next to it
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "oracle/_ref/" in open(os.path.join(root, ".gitignore")).read()


@needs_archive
def test_reference_modules_import_under_their_own_names_and_unload_again():
    import sys
    ref = ref_py.load(patch=False)
    try:
        assert ref.patched == {}
        assert ref.train.__file__.startswith(ref.root) and callable(ref.train.scene_reconstruction)
        assert ref.train.render is ref.gaussian_renderer.render
        assert ref.gaussian_model.GaussianModel.__module__ == "scene.gaussian_model"
        # the drop-in packages are what the reference's imports resolved to
        import diff_gaussian_rasterization
        import simple_knn._C
        assert ref.gaussian_renderer.GaussianRasterizer is diff_gaussian_rasterization.GaussianRasterizer
        assert ref.gaussian_model.distCUDA2 is simple_knn._C.distCUDA2
        args, dataset, hyper, opt, pipe = ref_py.default_arguments(ref)
        assert ref.train.args is args and hyper.feat_head is True and opt.densify_from_iter == 500 and pipe.convert_SHs_python is True
        assert hyper.kplanes_config["resolution"] == [64, 64, 64, 25] and hyper.multires == [1, 2, 4, 8]
    finally:
        ref_py.unload()
    assert "train" not in sys.modules and "scene.gaussian_model" not in sys.modules and not os.path.isdir(ref.root)


@needs_archive
def test_patch_reference_binds_to_the_real_modules_before_train_is_imported():
    from s3gaussian_amd import optim, patch
    ref = ref_py.load(patch=True)
    try:
        T = ref.train
        for name in ("render", "l1_loss", "l2_loss", "ssim", "compute_depth"):
            assert getattr(T, name).__module__ == "s3gaussian_amd.patch", name       # train.py's `from x import y` picked them up
        GM = ref.gaussian_model.GaussianModel
        assert GM.compute_regulation is patch.compute_regulation and GM.add_densification_stats is patch.add_densification_stats
        assert ref.gaussian_model.deform_network.__module__ == "s3gaussian_amd.deformation"
        assert set(ref.patched) >= {"gaussian_renderer.render", "GaussianModel.training_setup", "utils.loss_utils.ssim"}
        assert patch._PATCHED and "render" in patch._REFERENCE and patch._REFERENCE["render"].__module__ == "gaussian_renderer"
        assert optim.Adam.__mro__[1] is __import__("torch").optim.Adam
    finally:
        ref_py.unload()
    assert not patch._PATCHED and patch._REFERENCE == {}
    # and a plain load afterwards sees the reference's own functions again
    ref = ref_py.load(patch=False)
    try:
        assert ref.train.render.__module__ == "gaussian_renderer" and ref.train.ssim.__module__ == "utils.loss_utils"
    finally:
        ref_py.unload()


@needs_archive
def test_reference_rasterizer_route_resolves_to_the_reference_wrapper_and_restores_the_drop_ins():
    """load(rasterizer="reference"): `diff_gaussian_rasterization` is the REFERENCE's own Python wrapper (unchanged) over a `_C`
    that binds the reference's own kernels; unload() puts this repo's drop-in packages back under the same names."""
    import sys
    import diff_gaussian_rasterization as ours
    ref = ref_py.load(rasterizer="reference")
    try:
        import diff_gaussian_rasterization as theirs
        assert theirs is not ours and theirs.__file__.startswith(ref.root)
        want = open(os.path.join(ref.root, ref_py.RAST_WRAPPER), "rb").read()
        assert open(theirs.__file__, "rb").read() == want
        assert theirs._C.rasterize_gaussians.__module__ == "oracle.ref_diff_raster_C"
        assert ref.gaussian_renderer.GaussianRasterizer is theirs.GaussianRasterizer
        assert ref.gaussian_model.distCUDA2.__module__ == "oracle.ref_diff_raster_C"
    finally:
        ref_py.unload()
    import diff_gaussian_rasterization as again
    assert again is ours and sys.modules["diff_gaussian_rasterization"] is ours
