"""Python surface of the rasterizer: same public names, fields, call signatures and error behaviour as
RAST/diff_gaussian_rasterization/__init__.py:158-221 so that the reference's gaussian_renderer/__init__.py:18,44-57,
127-135 imports and calls it unchanged.  The top-level package `diff_gaussian_rasterization` re-exports these.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
from torch import nn

from . import raster_C as _C


class GaussianRasterizationSettings(NamedTuple):
    """12 fields, keyword-constructed by the caller (gaussian_renderer/__init__.py:44-57)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """debug=True: keep a host copy of every argument so a failing call can be dumped (reference :17-19, :83-90)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class _RasterizeGaussians(torch.autograd.Function):
    """autograd node; forward keeps the three private arenas alive for backward (reference :44-156)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                forward_only=False):
        rs = raster_settings
        native_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                       rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        if rs.debug:
            saved = _snapshot(native_args)
            try:
                out = _C.rasterize_gaussians(*native_args)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            # num_rendered only travels to the backward below as the key of the arenas: the call may run host-asynchronously
            out = _C.rasterize_gaussians(*native_args, allow_async=True, forward_only=forward_only)
        num_rendered, color, depth, radii, geom, binning, img = out
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, grad_out_depth):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
        if grad_out_depth is None:
            grad_out_depth = torch.zeros((1, rs.image_height, rs.image_width), device=means3D.device)
        native_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, sh,
                       rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning, img, rs.debug)
        if rs.debug:
            saved = _snapshot(native_args)
            try:
                grads = _C.rasterize_gaussians_backward(*native_args)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _C.rasterize_gaussians_backward(*native_args)
        g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot = grads
        # order of forward's inputs: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings
        return g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rot, g_cov3D, None, None


class _RasterizeGaussiansPair(torch.autograd.Function):
    """Two renders of one geometry (colours_a -> image + depth, colours_b -> second image) as ONE autograd node: one
    blend pass forward (s3g_raster_forward2) and one backward (s3g_raster_backward2) instead of two each, whose gradients
    autograd would then add."""

    @staticmethod
    def forward(ctx, means3D, means2D, colors_a, colors_b, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                densify_accum=None, forward_only=False):
        rs = raster_settings
        empty = torch.Tensor([])
        common = (opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                  rs.tanfovy, rs.image_height, rs.image_width, empty, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        num_rendered, color, depth, radii, geom, binning, img, color2 = _C.rasterize_gaussians(
            rs.bg, means3D, colors_a, *common, colors2=colors_b, allow_async=True, forward_only=forward_only)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.densify_accum = densify_accum
        ctx.save_for_backward(colors_a, colors_b, means3D, scales, rotations, cov3Ds_precomp, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, color2

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, grad_depth, grad_color2):
        rs = ctx.raster_settings
        colors_a, colors_b, means3D, scales, rotations, cov3Ds_precomp, radii, geom, binning, img = ctx.saved_tensors
        z = lambda c: torch.zeros((c, rs.image_height, rs.image_width), device=means3D.device)
        grad_color = z(3) if grad_color is None else grad_color
        grad_depth = z(1) if grad_depth is None else grad_depth
        grad_color2 = z(3) if grad_color2 is None else grad_color2
        (g_means2D, g_col_a, g_col_b, g_opac, g_means3D, g_cov3D, g_scales, g_rot) = _C.rasterize_gaussians_backward2(
            rs.bg, means3D, radii, colors_a, colors_b, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_depth, grad_color2, rs.campos, geom, ctx.num_rendered,
            binning, img, rs.debug, densify_accum=ctx.densify_accum)
        return g_means3D, g_means2D, g_col_a, g_col_b, g_opac, g_scales, g_rot, g_cov3D, None, None, None


def _no_backward(*tensors) -> bool:
    """True when autograd will record nothing for a node over these inputs (no_grad, or no input requires a gradient): the
    forward may then skip what only a backward would read (raster_C.rasterize_gaussians: forward_only)."""
    return not (torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors))


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    tensors = (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    return _RasterizeGaussians.apply(*tensors, raster_settings, _no_backward(*tensors))


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: Gaussians in front of the near cull plane (view z > 0.2)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])  # the reference's "not provided" sentinel: numel()==0 -> NULL at the C ABI
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)

    @torch.no_grad()
    def forward_decomposed(self, means3D, opacities, dynamic_mask, shs=None, colors_precomp=None, scales=None, rotations=None,
                           cov3D_precomp=None):
        """Extension for the evaluation path (gaussian_renderer/__init__.py:168-204, `return_decomposition=True`): the full
        render plus the dynamic-only and static-only renders, sharing ONE preprocess / binning / sort; the two subset images
        come from one extra blend pass and equal `self(...)` on the masked inputs bit for bit.  Inference only (no autograd).
        -> dict(render, radii, depth, render_d, depth_d, render_s, depth_s)"""
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = torch.Tensor([])
        n = lambda x: e if x is None else x
        P = means3D.shape[0]
        R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, means3D, n(colors_precomp), opacities, n(scales), n(rotations), rs.scale_modifier, n(cov3D_precomp),
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, n(shs), rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug, allow_async=True, forward_only=True)
        if P == 0:
            z3, z1 = torch.zeros_like(color), torch.zeros_like(depth)
            return dict(render=color, radii=radii, depth=depth, render_d=z3, depth_d=z1, render_s=z3.clone(), depth_s=z1.clone())
        cd, dd, cs, ds = _C.rasterize_decomposition(rs.bg, n(colors_precomp), dynamic_mask, rs.tanfovx, rs.tanfovy,
                                                    rs.image_height, rs.image_width, P, R, geom, binning, img, rs.debug)
        return dict(render=color, radii=radii, depth=depth, render_d=cd, depth_d=dd, render_s=cs, depth_s=ds)

    def forward_pair(self, means3D, means2D, opacities, colors_a, colors_b, scales=None, rotations=None, cov3D_precomp=None,
                     densify_accum=None):
        """Extension (not in the reference): render the SAME Gaussians with two sets of precomputed colours, e.g. RGB and the
        feature head's output (gaussian_renderer/__init__.py:127-166 does this with two calls).
        -> (image_a [3,H,W], radii [P], depth [1,H,W], image_b [3,H,W]); gradients equal those of the two separate calls.
        densify_accum = (xyz_gradient_accum, denom, max_radii2D): the backward of this node also performs the reference's
        densification bookkeeping (train.py:489-493) in its per-Gaussian pass -- valid because this node yields the whole
        viewspace gradient of the iteration."""
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if means3D.shape[0] == 0 or self.raster_settings.debug:
            # nothing to share / debug snapshots wanted: fall back to two ordinary nodes
            if densify_accum is not None:   # direct callers only: pipeline.render() decides before calling
                raise RuntimeError("densify_accum needs the fused two-image node (P > 0, debug off)")
            a, radii, depth = self.forward(means3D, means2D, opacities, colors_precomp=colors_a, scales=scales, rotations=rotations,
                                           cov3D_precomp=cov3D_precomp)
            b, _, _ = self.forward(means3D, means2D, opacities, colors_precomp=colors_b, scales=scales, rotations=rotations,
                                   cov3D_precomp=cov3D_precomp)
            return a, radii, depth, b
        empty = torch.Tensor([])
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        tensors = (means3D, means2D, colors_a, colors_b, opacities, scales, rotations, cov3D_precomp)
        return _RasterizeGaussiansPair.apply(*tensors, self.raster_settings, densify_accum, _no_backward(*tensors))
