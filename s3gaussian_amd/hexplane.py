"""HexPlane field on the MI355X: same module surface as the reference's scene/hexplane.py::HexPlaneField
(:109-183) -- `aabb`, `grids` (ModuleList of ParameterList, names `grids.{level}.{plane}`, shapes [1,32,H,W]),
`feat_dim`, `get_aabb`, `set_aabb`, `get_density`, `forward` -- so state_dicts / checkpoints interchange
(SURVEY.md 5.4).  The 24 grid_sample launches are replaced by one fused HIP kernel per direction
(include/s3g_hexplane.h).

Plane parameters are held in torch.channels_last memory format: logical shape stays [1,32,H,W] (what the
reference's optimizer groups, regularisers and checkpoints see) while the bytes are [H][W][32], which is the layout
the kernels read -- no shadow copy, no transposes.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib

MAX_LEVELS, CHANNELS = 8, 32
SORT_REFRESH = 16   # backward passes between two re-sorts of the points (see _HexPlaneSample.backward; 13 orders at 4 levels: 0.8 ms per re-sort)
SORT_REFRESH = int(os.environ.get("S3G_HEX_SORT_REFRESH", SORT_REFRESH))


def sort_state_words(levels: int) -> int:
    """32-bit words per point of the persistent `sort_state` (include/s3g_hexplane.h::s3g_hexplane_sort_state_words): one walk
    order per orientation and level, their compositions with the processing order, and the processing order (25 at 4 levels)."""
    L = _bind()
    return int(L.s3g_hexplane_sort_state_words(int(levels)))


# Backward algorithm (include/s3g_hexplane.h): "slab" (default, S3G_HEX_SLAB_DIV) = the per-point pass takes T = dL/dfeature *
# feature from the forward's saved output in one multiply, divides plane by plane, finishes dL/dxyz and writes ONE row per point and
# level; twelve sorted scatter walks read it back and divide by the sample they re-derive from the footprint they are accumulating.
# "slab_product" (S3G_HEX_SLAB) = the exact fallback that needs nothing from the forward: the same two passes with the per-point
# pass forming dL/d(sample) by the product rule.  (The slab-free "walk" of rounds 2-4 was removed in round 5.)
BACKWARD_MODE = os.environ.get("S3G_HEX_BACKWARD", "slab")
_ALGORITHM = {"slab_product": 0, "slab": 2}     # S3G_HEX_SLAB, S3G_HEX_SLAB_DIV (include/s3g_hexplane.h)


class _HexDesc(C.Structure):
    """struct s3g_hexplane_desc (include/s3g_hexplane.h)."""
    _fields_ = [("levels", C.c_int), ("res", (C.c_int * 4) * MAX_LEVELS), ("planes", (C.c_void_p * 6) * MAX_LEVELS),
                ("aabb_max", C.c_float * 3), ("aabb_min", C.c_float * 3), ("uniform_time", C.c_int)]


_PlanePtrs = (C.c_void_p * 6) * MAX_LEVELS
_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp = C.c_void_p
        L.s3g_hexplane_forward.restype = C.c_int
        L.s3g_hexplane_forward.argtypes = [C.POINTER(_HexDesc), C.c_int, vp, vp, vp, vp, vp, vp]
        L.s3g_hexplane_forward_workspace_bytes.restype = C.c_size_t
        L.s3g_hexplane_forward_workspace_bytes.argtypes = [C.POINTER(_HexDesc)]
        L.s3g_hexplane_backward.restype = C.c_int
        L.s3g_hexplane_backward.argtypes = [C.POINTER(_HexDesc), C.c_int, vp, vp, vp, vp, vp, C.POINTER(_PlanePtrs), vp, vp, C.c_int, vp]
        L.s3g_hexplane_backward_algo.restype = C.c_int
        L.s3g_hexplane_backward_algo.argtypes = [C.POINTER(_HexDesc), C.c_int, vp, vp, vp, vp, C.c_int, vp, C.POINTER(_PlanePtrs), vp, vp,
                                                 C.c_int, vp]
        L.s3g_hexplane_backward_workspace_bytes.restype = C.c_size_t
        L.s3g_hexplane_backward_workspace_bytes.argtypes = [C.POINTER(_HexDesc), C.c_int, C.c_int]
        L.s3g_scale_unless_one.restype = C.c_int
        L.s3g_scale_unless_one.argtypes = [vp, C.c_size_t, vp, vp]
        L.s3g_hexplane_sort_state_words.restype = C.c_int
        L.s3g_hexplane_sort_state_words.argtypes = [C.c_int]
        L.s3g_hexplane_set_deterministic.restype = None
        L.s3g_hexplane_set_deterministic.argtypes = [C.c_int]
        L.s3g_hexplane_get_deterministic.restype = C.c_int
        _bound = True
        if os.environ.get("S3G_HEX_DETERMINISTIC", "0") == "1":
            L.s3g_hexplane_set_deterministic(1)
    return L


def set_deterministic(on: bool) -> bool:
    """Deterministic mode of the HexPlane backward (include/s3g_hexplane.h::s3g_hexplane_set_deterministic; environment
    S3G_HEX_DETERMINISTIC=1): stable walk orders, run records instead of float atomics, a stencil gather in fixed order -- plane
    gradients bit-identical from run to run, for +2.0 ms per backward at 1.2 M points (the walk orders are then re-sorted, stably,
    on every backward).  Process-wide; returns the previous
    setting.  Cached walk orders of existing fields were sorted under the old setting: clear `field._order_cache` (or let them age
    out) before relying on bit-identity."""
    L = _bind()
    prev = bool(L.s3g_hexplane_get_deterministic())
    L.s3g_hexplane_set_deterministic(int(bool(on)))
    return prev


def _channels_last_ptr(p: torch.Tensor) -> int:
    if not p.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("hexplane planes must be in torch.channels_last memory format (see HexPlaneField)")
    return p.data_ptr()


def _make_desc(planes, resolutions, aabb_host, uniform_time=False) -> _HexDesc:
    d = _HexDesc()
    d.levels = len(resolutions)
    d.uniform_time = int(bool(uniform_time))
    for l, res in enumerate(resolutions):
        for k in range(4):
            d.res[l][k] = int(res[k])
        for i in range(6):
            d.planes[l][i] = _channels_last_ptr(planes[l * 6 + i])
    for k in range(3):
        d.aabb_max[k] = aabb_host[0][k]
        d.aabb_min[k] = aabb_host[1][k]
    return d


class _HexPlaneSample(torch.autograd.Function):
    """features[P, levels*32] = prod over the 6 planes of bilinear samples, concatenated over levels."""

    @staticmethod
    def forward(ctx, xyz, time, meta, *planes):
        resolutions, aabb_host, cache, uniform_time, reg_weights = meta
        if not xyz.is_cuda:
            raise RuntimeError(f"xyz must live on the GPU (got {xyz.device}); the HexPlane sampler has no CPU fallback")
        L = _bind()
        P = xyz.shape[0]
        xyz_c = xyz.detach().contiguous().float()
        t_c = time.detach().reshape(-1).contiguous().float()
        if t_c.numel() != P and not (uniform_time is True and t_c.numel() == 1):   # uniform time: one shared timestamp is enough
            raise RuntimeError("time must have one value per point")
        feat = torch.empty((P, len(resolutions) * CHANNELS), dtype=torch.float32, device=xyz.device)
        if uniform_time is None:   # one device reduction + host sync; callers that know (render()) pass the flag instead
            uniform_time = bool(P > 0 and (t_c == t_c[0]).all().item())
        uniform_time = bool(uniform_time) and P > 0
        d = _make_desc(planes, resolutions, aabb_host, uniform_time)
        nws = L.s3g_hexplane_forward_workspace_bytes(C.byref(d))
        ws = torch.empty(nws, dtype=torch.uint8, device=xyz.device) if nws else None
        order = cache.get("order") if cache is not None else None
        if order is not None and (order.numel() != P or order.device != xyz.device):
            order = None  # stale (densification changed P): fall back to identity order
        with _lib.on_device(xyz.device):
            _lib.check(L.s3g_hexplane_forward(C.byref(d), P, xyz_c.data_ptr(), t_c.data_ptr(), feat.data_ptr(),
                                              order.data_ptr() if order is not None else None,
                                              ws.data_ptr() if ws is not None else None,
                                              _lib.stream_ptr()))
        ctx.meta = (resolutions, aabb_host, cache, uniform_time)
        # the output is saved too: the backward divides dL/dfeature * feature by one re-derived sample instead of storing
        # dL/d(sample) for every plane-level (it stays alive anyway as the MLP's input)
        ctx.save_for_backward(xyz_c, t_c, feat, *planes)
        ctx.set_materialize_grads(False)   # an unused output must arrive as None, not as a [P,128] tensor of zeros
        ctx.reg_flat = None
        if reg_weights is None:
            return feat
        # plane regulariser (scene/gaussian_model.py:710-749) on the same node: its gradient becomes the INITIAL content of
        # the plane-gradient buffer the sampler's backward accumulates into (no separate fill, scale and 24 adds)
        from .losses import plane_regulation_into
        reg_flat = torch.empty(sum(p.numel() for p in planes), dtype=torch.float32, device=xyz.device)
        reg = plane_regulation_into(planes, reg_weights, _flat_plane_views(reg_flat, planes))
        ctx.reg_flat = reg_flat
        return feat, reg

    @staticmethod
    def backward(ctx, gfeat, g_reg=None):
        xyz_c, t_c, feat, *planes = ctx.saved_tensors
        resolutions, aabb_host, cache, uniform_time = ctx.meta
        L = _bind()
        P = xyz_c.shape[0]
        gxyz = torch.empty_like(xyz_c)
        need = [ctx.needs_input_grad[3 + i] for i in range(len(planes))]
        if ctx.reg_flat is not None:
            # every plane needs its gradient on this path (checked by the caller): regulariser gradient x upstream scalar
            if g_reg is not None:
                flat = ctx.reg_flat
                g32 = g_reg.detach().reshape(-1)[:1].contiguous().float()
                with _lib.on_device(flat.device):    # in place; a no-op decided on the device when the upstream gradient is 1
                    _lib.check(L.s3g_scale_unless_one(flat.data_ptr(), flat.numel(), g32.data_ptr(), _lib.stream_ptr()))
            else:
                flat = ctx.reg_flat.zero_()
            ctx.reg_flat = None
            gplanes = _flat_plane_views(flat, planes)
        else:
            # one zero fill for all plane gradients; each is a channels_last [1,C,H,W] view of the flat buffer
            flat = torch.zeros(sum(p.numel() for p, n in zip(planes, need) if n), dtype=torch.float32, device=xyz_c.device)
            gplanes = _flat_plane_views(flat, planes, need)
        if gfeat is None:   # only the regulariser was used downstream
            return (None, None, None, *gplanes)
        gfeat = gfeat.contiguous()
        ptrs = _PlanePtrs()
        for l in range(len(resolutions)):
            for i in range(6):
                g = gplanes[l * 6 + i]
                ptrs[l][i] = _channels_last_ptr(g) if g is not None else None
        d = _make_desc(planes, resolutions, aabb_host, uniform_time)
        if BACKWARD_MODE not in _ALGORITHM:
            raise RuntimeError(f"S3G_HEX_BACKWARD must be one of {sorted(_ALGORITHM)}, got {BACKWARD_MODE!r}")
        algorithm = _ALGORITHM[BACKWARD_MODE]
        work = torch.empty(L.s3g_hexplane_backward_workspace_bytes(C.byref(d), P, 0), dtype=torch.uint8,
                           device=xyz_c.device)
        # the spatial walk orders live in the field's cache and are refreshed every SORT_REFRESH backward passes (or when
        # P changes): they steer the walk, not the result, and the points move slowly between iterations
        state, reuse = None, 0
        if cache is not None:
            state = cache.get("sort_state")
            words = sort_state_words(len(resolutions))
            if state is None or state.numel() != words * P or state.device != xyz_c.device:
                state = torch.empty(words * P, dtype=torch.int32, device=xyz_c.device)
                cache["sort_state"], cache["sort_age"] = state, 0
            else:
                cache["sort_age"] = cache.get("sort_age", 0) + 1
                if cache["sort_age"] >= SORT_REFRESH:
                    cache["sort_age"] = 0
                else:
                    reuse = 1
        with _lib.on_device(xyz_c.device):
            _lib.check(L.s3g_hexplane_backward_algo(C.byref(d), P, xyz_c.data_ptr(), t_c.data_ptr(), gfeat.data_ptr(),
                                                    None if algorithm == 0 else feat.data_ptr(), algorithm,
                                                    gxyz.data_ptr(), C.byref(ptrs), work.data_ptr(),
                                                    state.data_ptr() if state is not None else None, reuse,
                                                    _lib.stream_ptr()))
        if cache is not None:
            cache["order"] = state[(words - 1) * P:words * P]  # 3-D blocked processing order for the forwards
        return (gxyz if ctx.needs_input_grad[0] else None, None, None, *gplanes)


def _flat_plane_views(flat, planes, need=None):
    """channels_last [1,C,H,W] views of one flat buffer, in plane order (None where need[i] is False)."""
    views, off = [], 0
    for i, p in enumerate(planes):
        if need is not None and not need[i]:
            views.append(None)
            continue
        _, Cn, Hn, Wn = p.shape
        views.append(flat[off:off + p.numel()].view(1, Hn, Wn, Cn).permute(0, 3, 1, 2))
        off += p.numel()
    return views


def hexplane_sample(xyz, time, planes, resolutions, aabb_host, cache=None, uniform_time=None, reg_weights=None):
    """uniform_time: True = the caller guarantees every point carries the same timestamp (fast path: the time planes are
    pre-interpolated into row tables), False = general per-point time, None = decide by looking at `time` (one host sync).
    reg_weights = (time_smoothness_weight, l1_time_planes_weight, plane_tv_weight): also return the plane regulariser of
    GaussianModel.compute_regulation as a second output of the same autograd node -> (features, regulation)."""
    if reg_weights is not None and not all(p.requires_grad for p in planes):
        reg_weights = None   # frozen planes: keep the two computations separate
        feat = _HexPlaneSample.apply(xyz, time, (tuple(tuple(r) for r in resolutions), aabb_host, cache, uniform_time, None), *planes)
        return feat, None
    return _HexPlaneSample.apply(xyz, time, (tuple(tuple(r) for r in resolutions), aabb_host, cache, uniform_time,
                                             tuple(float(w) for w in reg_weights) if reg_weights is not None else None), *planes)


class HexPlaneField(nn.Module):
    def __init__(self, bounds, planeconfig, multires) -> None:
        super().__init__()
        self.aabb = nn.Parameter(torch.tensor([[bounds] * 3, [-bounds] * 3], dtype=torch.float32), requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        if planeconfig["grid_dimensions"] != 2 or planeconfig["input_coordinate_dim"] != 4:
            raise NotImplementedError("only the 4D/2D-plane (HexPlane) configuration of the reference is supported")
        if planeconfig["output_coordinate_dim"] != CHANNELS:
            raise NotImplementedError(f"output_coordinate_dim must be {CHANNELS}")
        if len(multires) > MAX_LEVELS:
            raise NotImplementedError(f"at most {MAX_LEVELS} resolution levels")
        self.grids = nn.ModuleList()
        self.resolutions = []
        self.feat_dim = 0
        for res in multires:
            reso = [r * res for r in planeconfig["resolution"][:3]] + list(planeconfig["resolution"][3:])
            self.resolutions.append(tuple(reso))
            planes = nn.ParameterList()
            for comb in itertools.combinations(range(4), 2):
                p = torch.empty([1, CHANNELS] + [reso[c] for c in comb[::-1]])
                if 3 in comb:
                    nn.init.ones_(p)                  # time planes start at 1 (scene/hexplane.py:64-65)
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                planes.append(nn.Parameter(p.contiguous(memory_format=torch.channels_last)))
            self.grids.append(planes)
            self.feat_dim += CHANNELS
        self._aabb_host = None
        self._order_cache = {}  # spatial orders of the points left behind by the backward passes (see _HexPlaneSample)

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        self.aabb = nn.Parameter(torch.tensor([xyz_max, xyz_min], dtype=torch.float32, device=self.aabb.device),
                                 requires_grad=False)
        self._aabb_host = None

    def _apply(self, fn, *args, **kwargs):  # .to()/.cuda(): keep the planes channel-last on the new device
        out = super()._apply(fn, *args, **kwargs)
        for planes in self.grids:
            for p in planes:
                if not p.data.is_contiguous(memory_format=torch.channels_last):
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
        self._aabb_host = None
        return out

    def _host_aabb(self):
        key = (self.aabb.data_ptr(), self.aabb._version)
        if self._aabb_host is None or self._aabb_host[0] != key:
            a = self.aabb.detach().cpu().tolist()   # one tiny D2H, only when the aabb changed
            self._aabb_host = (key, (tuple(a[0]), tuple(a[1])))
        return self._aabb_host[1]

    def _planes(self):
        # (straight from the containers' dicts: iterating nn.ModuleList / nn.ParameterList goes through string indices -- 67 us for the 24
        #  planes, three times per training iteration; this is 3 us)
        return [p for planes in self.grids._modules.values() for p in planes._parameters.values()]

    def get_density(self, pts: torch.Tensor, timestamps: Optional[torch.Tensor] = None, uniform_time=None, reg_weights=None):
        pts = pts.reshape(-1, pts.shape[-1])
        return hexplane_sample(pts, timestamps, self._planes(), self.resolutions, self._host_aabb(), self._order_cache,
                               uniform_time, reg_weights)

    def forward(self, pts: torch.Tensor, timestamps: Optional[torch.Tensor] = None, uniform_time=None, reg_weights=None):
        """Reference signature (scene/hexplane.py:178-183) plus the optional hints of hexplane_sample; with reg_weights the
        result is (features, plane regulation)."""
        return self.get_density(pts, timestamps, uniform_time, reg_weights)
