"""Zero-edit fast route: run the reference's own `train.py` / `render.py` UNCHANGED at (nearly) the speed of the fused path.

    python -m s3gaussian_amd.patch train.py -s <scene> --configs arguments/... [the reference's own arguments]

or, inside a launcher / `sitecustomize`:   `import s3gaussian_amd.patch as p; p.patch_reference()`   BEFORE `train.py` is imported
(it binds `render`, `ssim`, `l1_loss`, ... with `from x import y` at import time).

The drop-in packages (`diff_gaussian_rasterization`, `simple_knn`) alone already replace every CUDA kernel of the reference, but
its Python then still runs the deformation field, the activations / SH glue, SSIM, the plane regularisers and Adam as hundreds of
PyTorch launches per iteration (2.9 it/s at BASELINE cfg3).  `patch_reference()` rebinds, without touching a file of the reference:

  scene.deformation.deform_network (+ the name imported into scene.gaussian_model, scene/gaussian_model.py:22,55)
        -> s3gaussian_amd.deformation.deform_network      same parameter names / shapes: checkpoints interchange
  gaussian_renderer.render   (gaussian_renderer/__init__.py:23-210)
        -> render() below = s3gaussian_amd.pipeline.render on the reference's Camera / GaussianModel objects: fused sampler + MLP,
           one glue kernel each way, RGB + feature image as ONE two-image rasterizer node, same result-dict keys
  utils.loss_utils.{l1_loss, l2_loss, ssim, compute_depth}   (utils/loss_utils.py:21-96; train.py:395-425 calls them by name)
        -> one fused kernel pair each (include/s3g_loss.h)
  GaussianModel.compute_regulation   (scene/gaussian_model.py:710-749)   -> one pass over the 143 MB of planes
  GaussianModel.training_setup       (scene/gaussian_model.py:170-201)   -> the same call, then `self.optimizer` rebuilt as
           s3gaussian_amd.optim.Adam over the SAME param_groups (one launch per step, identical state layout, so
           cat_tensors_to_optimizer / prune_optimizer keep working)
  GaussianModel.add_densification_stats   (scene/gaussian_model.py:693-695)   -> one pass, no boolean-index host syncs

  render_pkg["dshs"]   -> a tensor that answers `torch.mean(torch.abs(.))` (train.py:407-410) with the sum the render glue already formed
           in its own pass (and folds the gradient into the glue's backward kernel): -0.5 ms of five passes over 230 MB; any other use
           sees the plain tensor

What train.py does inline (loss assembly one `loss +=` at a time, `loss.item()`, psnr, the max_radii2D update with boolean masks)
stays as it is; `bench.py` times exactly that iteration body as `config.paths.patched`.  Every replacement is value-checked against
the formulation it replaces (tests/test_patch_gpu.py) and the call sites it binds to are pinned against the reference's sources
with `ast` (tests/test_patch_cpu.py).
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict

import torch

from . import losses as _losses
from . import pipeline as _pipeline


# ---- replacements (usable directly; patch_reference() only rebinds names to them) --------------------------------------------------
def _cam_dict(cam) -> Dict:
    """The fields pipeline.render reads, from a reference Camera (scene/cameras.py:20-70) -- cached on the object."""
    if isinstance(cam, dict):
        return cam
    d = getattr(cam, "_s3g_cam", None)
    if d is None:
        d = dict(image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
                 tanfovy=math.tan(cam.FoVy * 0.5), viewmatrix=cam.world_view_transform.cuda(),
                 projmatrix=cam.full_proj_transform.cuda(), campos=cam.camera_center.cuda(), time=float(cam.time))
        try:
            cam._s3g_cam = d
        except Exception:
            pass
    return d


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, stage="fine",
           return_decomposition=False, return_dx=False, render_feat=False):
    """Signature of gaussian_renderer/__init__.py:23; `pc` is the reference's GaussianModel (same attribute names as
    pipeline.GaussianParams), `viewpoint_camera` its Camera."""
    # (pipe.compute_cov3D_python is honoured by pipeline.render itself -- covariance built in Python, handed over as cov3D_precomp --;
    #  until round 5 this function handed that switch back to the reference's own render(), which raises TypeError on it:
    #  torch.exp(None), gaussian_renderer/__init__.py:76-101.  One implementation for both entry points now)
    out = _pipeline.render(_cam_dict(viewpoint_camera), pc, pipe, bg_color, scaling_modifier, override_color, stage,
                           return_decomposition, return_dx, render_feat)
    reg = out.get("plane_reg") if isinstance(out, dict) else None
    if reg is not None:
        # fine stage under autograd: the plane regulariser was evaluated on the sampler's autograd node (its gradient seeds the
        # buffer the sampler's backward accumulates into).  train.py:412-413 asks for it by name a few lines later: the patched
        # compute_regulation below hands THIS value back instead of sweeping the 143 MB of planes a second time.
        hy = pc._deformation.deformation_net.args
        planes = pc._deformation.deformation_net.grid._planes()
        pc._s3g_plane_reg = ((float(hy.time_smoothness_weight), float(hy.l1_time_planes), float(hy.plane_tv_weight)),
                             tuple(p._version for p in planes), reg)
    if isinstance(out, dict) and out.get("dshs_l1") is not None and torch.is_tensor(out.get("dshs")) and FUSE_DSHS_L1:
        out["dshs"] = _L1Ready.wrap(out["dshs"], out["dshs_l1"])      # train.py:408-409 asks for mean|dshs| as abs -> mean
    return out


# ---- `torch.mean(torch.abs(render_pkg["dshs"]))` (train.py:407-410) without its five passes over the 230 MB of dshs ------------------
# train.py writes the regulariser out as abs -> mean on the [P,16,3] tensor: |x| materialised (read + write), reduced (read), and in
# backward a broadcast, a sign, a product and the accumulation into the gradient the renderer sends -- 0.5 ms of an 8.8-ms iteration at
# 1.2 M Gaussians (profiles/r05_patched_iteration_trace.txt), for a number the render glue has ALREADY summed in its own pass over dshs
# (glue.activations_and_colors(with_dshs_l1=True): differentiable, its gradient folded into the glue's backward kernel).  render() below
# hands dshs out as a tensor that recognises exactly that expression and answers it with the fused value; ANY other use -- another
# reduction, indexing, arithmetic, printing -- sees the ordinary tensor (|x| is computed the moment something else asks for it).
_SAFE_GETTERS = {"shape", "dtype", "device", "ndim", "requires_grad", "is_cuda", "layout", "is_sparse", "is_quantized", "is_meta", "names"}


def _standing_for(t):
    """The plain tensor an _L1Ready / _LazyAbs stands for (|x| is computed here, once, if it has not been asked for before)."""
    if isinstance(t, _LazyAbs):
        v = t.__dict__.get("_s3g_value")
        if v is None:
            src = t.__dict__["_s3g_src"]
            if src._version != t.__dict__["_s3g_version"]:      # loud, not wrong: abs() was taken BEFORE x changed in place
                raise RuntimeError("s3gaussian_amd.patch: render_pkg['dshs'] was modified in place between torch.abs(.) and the use of its "
                                   "result; set S3G_PATCH_FUSE_DSHS_L1=0 for this program")
            v = t.__dict__["_s3g_value"] = torch.abs(src)
        return v
    if isinstance(t, _L1Ready):
        return t.__dict__["_s3g_src"]
    return t


def _run_plain(func, args, kwargs):
    from torch.utils._pytree import tree_map
    with torch._C.DisableTorchFunctionSubclass():
        return func(*tree_map(_standing_for, tuple(args)), **tree_map(_standing_for, dict(kwargs or {})))


class _LazyAbs(torch.Tensor):
    """torch.abs(x) of an _L1Ready x, not computed yet.  `.mean()` / `torch.mean(.)` over everything -> the fused mean|x|."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        me = args[0] if args and isinstance(args[0], _LazyAbs) else None
        if me is not None and len(args) == 1 and not kwargs:
            if func in (torch.mean, torch.Tensor.mean):
                return me.__dict__["_s3g_l1"]
            getter = getattr(getattr(func, "__self__", None), "__name__", None)
            if getattr(func, "__name__", "") == "__get__" and getter in _SAFE_GETTERS:      # same for |x| as for x: no pass over the data
                with torch._C.DisableTorchFunctionSubclass():
                    return func(me.__dict__["_s3g_src"])
        return _run_plain(func, args, kwargs)


class _L1Ready(torch.Tensor):
    """x with mean|x| known (a differentiable scalar of the same graph).  `torch.abs(x)` / `x.abs()` -> _LazyAbs; all else: plain x."""

    @staticmethod
    def wrap(x: torch.Tensor, l1: torch.Tensor):
        t = x.as_subclass(_L1Ready)
        t.__dict__["_s3g_src"], t.__dict__["_s3g_l1"] = x, l1
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func in (torch.abs, torch.Tensor.abs) and len(args) == 1 and not kwargs and isinstance(args[0], _L1Ready):
            src, l1 = args[0].__dict__["_s3g_src"], args[0].__dict__["_s3g_l1"]
            with torch._C.DisableTorchFunctionSubclass():
                # the stand-in owns NO data: a meta tensor of x's shape and dtype.  Every consumer that dispatches through
                # __torch_function__ is handed |x| (or the fused mean) above; one that unwraps the object WITHOUT dispatching -- a
                # C++ custom op taking tensor lists, a functorch / compile path -- finds a meta tensor and fails on the spot,
                # where an alias of x's storage would have handed it the SIGNED values in silence
                lazy = torch.empty(src.shape, dtype=src.dtype, device="meta").as_subclass(_LazyAbs)
            lazy.__dict__["_s3g_src"], lazy.__dict__["_s3g_l1"], lazy.__dict__["_s3g_version"] = src, l1, src._version
            return lazy
        return _run_plain(func, args, kwargs)


def _as_image(t: torch.Tensor, channels: int):
    """[1,C,H,W] / [C,H,W] (/ [H,W] for one channel) -> [C,H,W], or None when the batch has more than one view."""
    if t.dim() == 4 and t.shape[0] == 1:
        t = t[0]
    if channels == 1 and t.dim() == 2:
        t = t[None]
    return t if (t.dim() == 3 and t.shape[0] == channels) else None


def l1_loss(network_output, gt):
    """utils/loss_utils.py:50-51.  One view ([1,3,H,W] or [3,H,W]): fused kernel; anything else: the reference's expression."""
    a, b = _as_image(network_output, 3), _as_image(gt, 3)
    if a is None or b is None or not network_output.is_cuda:
        return torch.abs(network_output - gt).mean()
    return _losses.pixel_terms(image=a, gt_image=b, w_l1=1.0)


def l2_loss(network_output, gt):
    """utils/loss_utils.py:53-54 (train.py:419-422 calls it on the [3,H,W] feature image)."""
    a, b = _as_image(network_output, 3), _as_image(gt, 3)
    if a is None or b is None or not network_output.is_cuda:
        return ((network_output - gt) ** 2).mean()
    return _losses.pixel_terms(feat=a, gt_feat=b, w_feat=1.0)


def compute_depth(loss_type, pred_depth, gt_depth, max_depth=80):
    """utils/loss_utils.py:24-45; only the "l2" branch train.py:411 uses is accelerated."""
    a, b = _as_image(pred_depth.squeeze(0) if pred_depth.dim() == 4 else pred_depth, 1), None
    if a is not None:
        b = _as_image(gt_depth.squeeze(0) if gt_depth.dim() == 4 else gt_depth, 1)
    if loss_type != "l2" or a is None or b is None or not pred_depth.is_cuda:
        return _REFERENCE["compute_depth"](loss_type, pred_depth, gt_depth, max_depth) if "compute_depth" in _REFERENCE else \
            _pipeline.compute_depth_l2(pred_depth, gt_depth, float(max_depth))
    return _losses.pixel_terms(depth=a, gt_depth=b, w_depth=1.0, max_depth=float(max_depth))


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:66-96."""
    if window_size != 11 or not size_average or not img1.is_cuda or (img1.dim() == 4 and img1.shape[0] != 1):
        return _REFERENCE["ssim"](img1, img2, window_size, size_average) if "ssim" in _REFERENCE else _pipeline.ssim(img1, img2, window_size)
    return _losses.ssim(img1, img2)


def compute_regulation(self, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    """GaussianModel.compute_regulation, scene/gaussian_model.py:748-749.  If render() has just evaluated the same expression on
    the sampler's node (same weights, planes untouched since), that value is returned -- once --; else one fused pass."""
    cached = self.__dict__.pop("_s3g_plane_reg", None)
    if cached is not None and torch.is_grad_enabled():
        weights, versions, reg = cached
        planes = self._deformation.deformation_net.grid._planes()
        if (weights == (float(time_smoothness_weight), float(l1_time_planes_weight), float(plane_tv_weight))
                and versions == tuple(p._version for p in planes) and reg.requires_grad):
            return reg
    return _losses.plane_regulation(self._deformation.deformation_net.grid.grids, time_smoothness_weight, l1_time_planes_weight,
                                    plane_tv_weight)


@torch.no_grad()
def add_densification_stats(self, viewspace_point_tensor, update_filter):
    """GaussianModel.add_densification_stats, scene/gaussian_model.py:693-695: `xyz_gradient_accum[f] += ||grad[f,:2]||`,
    `denom[f] += 1` in one pass (radii = 0 leaves max_radii2D, which train.py:491 updates itself, untouched)."""
    from .optim import densify_stats
    if not viewspace_point_tensor.is_cuda:
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1
        return
    key = (update_filter.shape[0], update_filter.device)
    zero_radii = _ZERO_RADII.get(key)       # (this call sits between train.py's loss.item() and the optimizer launch: the GPU is idle, the
    if zero_radii is None:                  #  host's time is the iteration's time -- no fill launch per call)
        _ZERO_RADII.clear()
        zero_radii = _ZERO_RADII[key] = torch.zeros(update_filter.shape[0], dtype=torch.int32, device=update_filter.device)
    densify_stats(self.xyz_gradient_accum, self.denom, self.max_radii2D, viewspace_point_tensor, zero_radii, update_filter)


def fused_optimizer_from(optimizer: torch.optim.Optimizer):
    """The optimizer GaussianModel.training_setup just built (torch.optim.Adam(l, lr=0.0, eps=1e-15), scene/gaussian_model.py:189)
    rebuilt as the one-launch Adam over the SAME groups (names, lrs and Parameter objects are kept)."""
    from .optim import Adam
    groups = [{k: v for k, v in g.items() if k in ("params", "lr", "name", "betas", "eps")} for g in optimizer.param_groups]
    d = optimizer.defaults
    return Adam(groups, lr=d.get("lr", 0.0), betas=d.get("betas", (0.9, 0.999)), eps=d.get("eps", 1e-15))


# S3G_PATCH_FUSE_DSHS_L1=0 switches the stand-in off for a whole run (the expression then runs as train.py spells it)
FUSE_DSHS_L1 = os.environ.get("S3G_PATCH_FUSE_DSHS_L1", "1") != "0"     # render(): hand dshs out as _L1Ready
_ZERO_RADII: Dict = {}    # (P, device) -> int32 zeros: add_densification_stats leaves max_radii2D to train.py:491
_REFERENCE: Dict = {}     # the reference's own callables, kept for the cases a replacement hands back
_PATCHED = False


def patch_reference(verbose: bool = False) -> Dict[str, str]:
    """Rebinds the names listed in the module docstring inside the ALREADY IMPORTABLE reference packages (`scene`, `utils`,
    `gaussian_renderer` on sys.path, i.e. the working directory is the reference checkout).  Idempotent.  Returns what was bound."""
    global _PATCHED
    import importlib
    done = {}
    if _PATCHED:
        return done
    from . import deformation as _deformation
    sd = importlib.import_module("scene.deformation")
    sd.deform_network = _deformation.deform_network
    done["scene.deformation.deform_network"] = "s3gaussian_amd.deformation.deform_network"
    gm = importlib.import_module("scene.gaussian_model")
    if hasattr(gm, "deform_network"):
        gm.deform_network = _deformation.deform_network
        done["scene.gaussian_model.deform_network"] = "s3gaussian_amd.deformation.deform_network"
    GM = gm.GaussianModel
    _REFERENCE["training_setup"] = GM.training_setup

    def training_setup(self, training_args):
        _REFERENCE["training_setup"](self, training_args)
        if self._xyz.is_cuda:
            self.optimizer = fused_optimizer_from(self.optimizer)

    GM.training_setup = training_setup
    GM.compute_regulation = compute_regulation
    GM.add_densification_stats = add_densification_stats
    done.update({"GaussianModel.training_setup": "reference + s3gaussian_amd.optim.Adam", "GaussianModel.compute_regulation":
                 "s3gaussian_amd.losses.plane_regulation", "GaussianModel.add_densification_stats": "s3gaussian_amd.optim.densify_stats"})
    lu = importlib.import_module("utils.loss_utils")
    for name, fn in (("l1_loss", l1_loss), ("l2_loss", l2_loss), ("ssim", ssim), ("compute_depth", compute_depth)):
        if hasattr(lu, name):
            _REFERENCE[name] = getattr(lu, name)
            setattr(lu, name, fn)
            done[f"utils.loss_utils.{name}"] = f"s3gaussian_amd.patch.{name}"
    gr = importlib.import_module("gaussian_renderer")
    _REFERENCE["render"] = gr.render
    gr.render = render
    done["gaussian_renderer.render"] = "s3gaussian_amd.patch.render"
    _PATCHED = True
    if verbose:
        for k, v in done.items():
            print(f"[s3gaussian_amd.patch] {k} -> {v}", file=sys.stderr)
    return done


def main(argv=None):
    """python -m s3gaussian_amd.patch <script.py> [its arguments]: patch, then run the reference script as __main__."""
    import os
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m s3gaussian_amd.patch train.py [arguments of the reference's train.py]")
    script = argv[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)) or ".")
    patch_reference(verbose=True)
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
