"""Zero-edit route (s3gaussian_amd/patch.py), CPU side:
  * the call sites the drop-in packages and the patch bind to are pinned against the REFERENCE'S OWN SOURCES with `ast` (skipped
    where /root/reference does not exist, i.e. on the GPU box): every keyword of every `GaussianRasterizationSettings(...)`,
    `rasterizer(...)`, `render(...)` call and the arity of `distCUDA2(...)` must bind to our signatures, and every name
    `patch_reference()` rebinds must exist in the reference with a compatible parameter list;
  * `patch_reference()` itself, on a throw-away package tree with the reference's module / attribute names."""
import ast
import inspect
import os
import sys
import textwrap

import pytest

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")), reason="/root/reference not present")


def _tree(rel):
    return ast.parse(open(os.path.join(REF, rel)).read())


def _calls(tree, name):
    """Every ast.Call whose callee is the bare name or attribute `name`."""
    out = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Call):
            f = n.func
            if (isinstance(f, ast.Name) and f.id == name) or (isinstance(f, ast.Attribute) and f.attr == name):
                out.append(n)
    return out


def _binds(call, sig, skip_first=False):
    params = list(sig.parameters.values())
    if skip_first:
        params = params[1:]
    names = [p.name for p in params]
    assert len(call.args) <= len(names), ast.dump(call)
    for kw in call.keywords:
        assert kw.arg in names, f"keyword {kw.arg!r} of the reference's call at line {call.lineno} does not bind to {names}"
    given = set(names[:len(call.args)]) | {kw.arg for kw in call.keywords}
    for p in params:
        if p.default is inspect.Parameter.empty and p.kind == p.POSITIONAL_OR_KEYWORD:
            assert p.name in given, f"required parameter {p.name!r} missing at line {call.lineno}"


@needs_ref
def test_rasterizer_call_sites_of_the_reference_bind_to_the_drop_in():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    tree = _tree("gaussian_renderer/__init__.py")
    settings = _calls(tree, "GaussianRasterizationSettings")
    assert len(settings) == 1
    fields = list(GaussianRasterizationSettings._fields)
    assert [kw.arg for kw in settings[0].keywords] == fields            # gaussian_renderer/__init__.py:44-57, same order
    ctor = _calls(tree, "GaussianRasterizer")
    assert len(ctor) == 1 and [kw.arg for kw in ctor[0].keywords] == ["raster_settings"]
    calls = _calls(tree, "rasterizer")
    assert len(calls) == 4                                               # RGB, feature, dynamic, static (:127-204)
    sig = inspect.signature(GaussianRasterizer.forward)
    for c in calls:
        _binds(c, sig, skip_first=True)
    imports = [n for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module == "diff_gaussian_rasterization"]
    assert sorted(a.name for n in imports for a in n.names) == ["GaussianRasterizationSettings", "GaussianRasterizer"]


@needs_ref
def test_distcuda2_call_sites_bind():
    from simple_knn._C import distCUDA2
    n = 0
    for rel in ("scene/gaussian_model.py", "scene/background_model.py"):
        tree = _tree(rel)
        assert any(isinstance(x, ast.ImportFrom) and x.module == "simple_knn._C" and x.names[0].name == "distCUDA2" for x in ast.walk(tree))
        for c in _calls(tree, "distCUDA2"):
            assert len(c.args) == 1 and not c.keywords
            n += 1
    assert n >= 2 and len(inspect.signature(distCUDA2).parameters) == 1


@needs_ref
def test_every_name_patch_reference_rebinds_exists_in_the_reference_with_a_compatible_signature():
    from s3gaussian_amd import patch

    def fn(tree, name, cls=None):
        scope = tree.body
        if cls is not None:
            scope = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
        return next(n for n in scope if isinstance(n, ast.FunctionDef) and n.name == name)

    def arg_names(f):
        return [a.arg for a in f.args.args]

    lu = _tree("utils/loss_utils.py")
    for name in ("l1_loss", "l2_loss", "ssim", "compute_depth"):
        ours = list(inspect.signature(getattr(patch, name)).parameters)
        assert arg_names(fn(lu, name)) == ours, name
    gm = _tree("scene/gaussian_model.py")
    for name in ("compute_regulation", "add_densification_stats"):
        assert arg_names(fn(gm, name, "GaussianModel")) == list(inspect.signature(getattr(patch, name)).parameters), name
    assert arg_names(fn(gm, "training_setup", "GaussianModel")) == ["self", "training_args"]
    assert any(isinstance(n, ast.ImportFrom) and n.module == "scene.deformation" and n.names[0].name == "deform_network" for n in ast.walk(gm))
    # the optimizer the patch rebuilds: torch.optim.Adam(l, lr=0.0, eps=1e-15) with eight named groups (scene/gaussian_model.py:177-189)
    adam = [c for c in _calls(fn(gm, "training_setup", "GaussianModel"), "Adam")]
    assert len(adam) == 1 and sorted(kw.arg for kw in adam[0].keywords) == ["eps", "lr"]
    sd = _tree("scene/deformation.py")
    assert any(isinstance(n, ast.ClassDef) and n.name == "deform_network" for n in sd.body)
    gr = _tree("gaussian_renderer/__init__.py")
    assert arg_names(fn(gr, "render")) == list(inspect.signature(patch.render).parameters)
    # train.py's own calls of the rebound names bind to the replacements
    tr = _tree("train.py")
    for c in _calls(tr, "render"):
        _binds(c, inspect.signature(patch.render))
    for name in ("l1_loss", "l2_loss", "ssim", "compute_depth"):
        cs = [c for c in _calls(tr, name) if isinstance(c.func, ast.Name)]
        assert cs, name
        for c in cs:
            _binds(c, inspect.signature(getattr(patch, name)))
    assert _calls(tr, "compute_regulation") and _calls(tr, "add_densification_stats")
    imported = {a.name for n in ast.walk(tr) if isinstance(n, ast.ImportFrom) and n.module == "utils.loss_utils" for a in n.names}
    assert {"l1_loss", "ssim", "l2_loss", "compute_depth"} <= imported       # bound by name at import: patch BEFORE importing train.py


def test_patch_reference_rebinds_a_package_tree_with_the_reference_layout(tmp_path, monkeypatch):
    files = {
        "scene/__init__.py": "",
        "scene/deformation.py": "class deform_network:\n    pass\n",
        "scene/gaussian_model.py": textwrap.dedent("""
            import torch
            from scene.deformation import deform_network
            class GaussianModel:
                def __init__(self):
                    self._xyz = torch.nn.Parameter(torch.zeros(4, 3))
                    self.seen = []
                def training_setup(self, training_args):
                    self.seen.append(training_args)
                    self.optimizer = torch.optim.Adam([{'params': [self._xyz], 'lr': 0.1, 'name': 'xyz'}], lr=0.0, eps=1e-15)
                def compute_regulation(self, a, b, c):
                    return 'reference'
                def add_densification_stats(self, g, f):
                    return 'reference'
            """),
        "utils/__init__.py": "",
        "utils/loss_utils.py": "def l1_loss(a, b):\n    return 'ref'\ndef l2_loss(a, b):\n    return 'ref'\n"
                               "def ssim(a, b, window_size=11, size_average=True):\n    return 'ref'\n"
                               "def compute_depth(t, a, b, max_depth=80):\n    return 'ref'\n",
        "gaussian_renderer/__init__.py": "def render(*a, **k):\n    return 'ref'\n",
    }
    for rel, src in files.items():
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(src)
    monkeypatch.syspath_prepend(str(tmp_path))
    for m in [k for k in sys.modules if k.split(".")[0] in ("scene", "utils", "gaussian_renderer")]:
        monkeypatch.delitem(sys.modules, m)
    from s3gaussian_amd import patch
    from s3gaussian_amd.deformation import deform_network as ours
    monkeypatch.setattr(patch, "_PATCHED", False)
    done = patch.patch_reference()
    import gaussian_renderer
    import scene.deformation
    import scene.gaussian_model as gm
    import utils.loss_utils as lu
    assert scene.deformation.deform_network is ours and gm.deform_network is ours
    assert gaussian_renderer.render is patch.render and lu.ssim is patch.ssim and lu.l1_loss is patch.l1_loss
    assert lu.compute_depth is patch.compute_depth and lu.l2_loss is patch.l2_loss
    assert gm.GaussianModel.compute_regulation is patch.compute_regulation
    assert gm.GaussianModel.add_densification_stats is patch.add_densification_stats
    g = gm.GaussianModel()
    g.training_setup("args")                       # CPU parameters: the reference's optimizer is kept (no CPU fallback to offer)
    assert g.seen == ["args"] and type(g.optimizer) is __import__("torch").optim.Adam
    assert len(done) >= 10 and patch.patch_reference() == {}          # idempotent
    monkeypatch.setattr(patch, "_PATCHED", False)
    for m in [k for k in sys.modules if k.split(".")[0] in ("scene", "utils", "gaussian_renderer")]:
        monkeypatch.delitem(sys.modules, m)


def test_the_dshs_stand_in_answers_only_abs_then_mean_and_is_an_ordinary_tensor_otherwise():
    """patch._L1Ready / _LazyAbs (the tensor render() hands out as render_pkg['dshs']): `torch.mean(torch.abs(x))` and `x.abs().mean()`
    return the fused scalar it was built with; everything else -- other reductions, arithmetic, indexing, in-place use, autograd --
    behaves like the plain tensor, with |x| computed the moment it is needed."""
    import torch
    from s3gaussian_amd.patch import _L1Ready, _LazyAbs
    x = torch.randn(40, 16, 3, requires_grad=True)
    y = x * 1.0
    fused = y.abs().mean() + 0.0          # stands for the glue's differentiable mean|dshs|
    t = _L1Ready.wrap(y, fused)
    assert isinstance(t, torch.Tensor) and t.shape == y.shape and t.dtype == y.dtype and t.requires_grad
    assert torch.mean(torch.abs(t)) is fused and t.abs().mean() is fused
    a = torch.abs(t)
    assert type(a) is _LazyAbs and a.shape == y.shape and a.device == y.device and "_s3g_value" not in a.__dict__     # nothing computed yet
    assert torch.equal(a.mean(dim=0), y.abs().mean(dim=0)) and "_s3g_value" in a.__dict__                               # now it is
    assert torch.equal(a.sum(), y.abs().sum()) and torch.equal(a[3], y.abs()[3]) and torch.equal(a.data, y.abs().data)
    assert torch.equal(torch.mean(a, dim=1), y.abs().mean(dim=1)) and torch.equal(a.mean(0, keepdim=True), y.abs().mean(0, keepdim=True))
    assert type(a + 1) is torch.Tensor and type(t * 2) is torch.Tensor and torch.equal(t * 2, y * 2) and torch.equal(t[1], y[1])
    assert torch.equal(t.detach(), y.detach()) and torch.equal(torch.cat([t, t]), torch.cat([y, y])) and torch.equal(t.abs().max(), y.abs().max())
    assert "nan" not in repr(a) and float(a.min()) >= 0.0
    # a consumer that gets at the object WITHOUT dispatching through __torch_function__ must fail, not read x's signed values:
    # the stand-in owns no storage
    with torch._C.DisableTorchFunctionSubclass():
        assert a.device.type == "meta" and a.untyped_storage().data_ptr() != y.untyped_storage().data_ptr()
        with pytest.raises((RuntimeError, NotImplementedError)):
            float(a.sum())
    (t.abs().mean() * 0.01 + (t * 2).sum() + torch.abs(t).sum()).backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = x2 * 1.0
    (y2.abs().mean() * 0.01 + (y2 * 2).sum() + y2.abs().sum()).backward()
    assert torch.allclose(x.grad, x2.grad, rtol=1e-6, atol=1e-7)
    # |x| taken, x changed in place, |x| used: the lazy form cannot give the old values any more -- it must say so, not return new ones
    z = torch.randn(5, 16, 3)
    held = torch.abs(_L1Ready.wrap(z, z.abs().mean()))
    z.mul_(2.0)
    with pytest.raises(RuntimeError, match="S3G_PATCH_FUSE_DSHS_L1"):
        held.sum()
