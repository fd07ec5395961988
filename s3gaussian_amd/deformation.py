"""Deformation network on the MI355X with the module/parameter names of the reference's scene/deformation.py
(`deform_network` :179-235, `Deformation` :16-178) so checkpoints, `get_mlp_parameters` / `get_grid_parameters`
(name filter "grid") and the Adam groups of scene/gaussian_model.py:170-201 keep working.

Supported configuration = the reference defaults (arguments/__init__.py:202-236): grid_pe=0, no_grid=False,
static_mlp=False, empty_voxel=False, apply_rotation=False; the no_dx/no_ds/no_dr/no_do/no_dshs/feat_head switches
are honoured.  The HexPlane encoder is the fused HIP sampler (s3gaussian_amd/hexplane.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .hexplane import HexPlaneField
from .mlp import deform_mlp


def _head(W, out):
    return nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, out))


class Deformation(nn.Module):
    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=(), args=None):
        super().__init__()
        for flag in ("no_grid", "static_mlp", "empty_voxel", "apply_rotation"):
            if getattr(args, flag, False):
                raise NotImplementedError(f"{flag}=True is not on the accelerated path")
        if grid_pe != 0:
            raise NotImplementedError("grid_pe != 0 is not on the accelerated path")
        self.D, self.W, self.args, self.grid_pe = D, W, args, grid_pe
        self.input_ch, self.input_ch_time, self.skips = input_ch, input_ch_time, list(skips)
        self.no_grid = args.no_grid
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires)
        self.ratio = 0
        layers = [nn.Linear(self.grid.feat_dim, W)]
        for _ in range(D - 1):
            layers += [nn.ReLU(), nn.Linear(W, W)]
        self.feature_out = nn.Sequential(*layers)
        self.pos_deform = _head(W, 3)
        self.scales_deform = _head(W, 3)
        self.rotations_deform = _head(W, 4)
        self.opacity_deform = _head(W, 1)
        self.shs_deform = _head(W, 16 * 3)
        if args.feat_head:
            self.dino_head = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 3))

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_max, xyz_min)

    @property
    def get_empty_ratio(self):
        return self.ratio

    def query_time(self, rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb):
        return self.feature_out(self.grid(rays_pts_emb[:, :3], time_emb[:, :1]))

    def forward(self, rays_pts_emb, scales_emb=None, rotations_emb=None, opacity=None, shs_emb=None, time_feature=None,
                time_emb=None):
        if time_emb is None:
            raise NotImplementedError("forward_static needs static_mlp, which the reference defaults disable")
        return self.forward_dynamic(rays_pts_emb, scales_emb, rotations_emb, opacity, shs_emb, time_feature, time_emb)

    def _fused_ok(self):
        a = self.args
        return (self.D == 1 and self.W == 64 and self.grid.feat_dim == 128 and not a.no_dx and not a.no_dshs and a.no_ds
                and a.no_dr and a.no_do and a.feat_head)

    def deform_heads(self, xyz, time, uniform_time=None, reg_weights=None, need_feat=True):
        """(dx [P,3], dshs [P,16,3], feat [P,3]) only -- the part of forward_dynamic that is not a pass-through in the
        reference's default configuration.  Lets a caller that fuses `shs + dshs` downstream (pipeline.render) skip
        materialising the [P,16,3] sum."""
        reg = None
        if reg_weights is not None:   # plane regulariser evaluated on the sampler's autograd node (hexplane_sample)
            feats, reg = self.grid(xyz[:, :3], time[:, :1], uniform_time, reg_weights)
        else:
            feats = self.grid(xyz[:, :3], time[:, :1], uniform_time)
        # need_feat=False (honoured only when no backward follows): skip the feature head, `feat` is then None
        dx, dshs, feat = deform_mlp(feats, self.feature_out, self.pos_deform, self.shs_deform, self.dino_head, need_feat)
        out = (dx, dshs.reshape([xyz.shape[0], 16, 3]), feat)
        return out + (reg,) if reg_weights is not None else out

    def forward_dynamic(self, rays_pts_emb, scales_emb, rotations_emb, opacity_emb, shs_emb, time_feature, time_emb):
        a = self.args
        if self._fused_ok() and rays_pts_emb.is_cuda:
            # reference default configuration: HexPlane sampler -> one fused MFMA MLP kernel (include/s3g_mlp.h)
            feats = self.grid(rays_pts_emb[:, :3], time_emb[:, :1])
            dx, dshs, feat = deform_mlp(feats, self.feature_out, self.pos_deform, self.shs_deform, self.dino_head)
            dshs = dshs.reshape([shs_emb.shape[0], 16, 3])
            return (rays_pts_emb[:, :3] + dx, scales_emb[:, :3], rotations_emb[:, :4], opacity_emb[:, :1], shs_emb + dshs,
                    dx, feat, dshs)
        # other switch combinations: HexPlane sampler + library GEMMs
        hidden = self.query_time(rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb)
        dx = dshs = feat = None
        pts = rays_pts_emb[:, :3]
        if not a.no_dx:
            dx = self.pos_deform(hidden)
            pts = rays_pts_emb[:, :3] + dx            # mask == 1 in the default configuration (deformation.py:117)
        scales = scales_emb[:, :3] if a.no_ds else scales_emb[:, :3] + self.scales_deform(hidden)
        rotations = rotations_emb[:, :4] if a.no_dr else rotations_emb[:, :4] + self.rotations_deform(hidden)
        opacity = opacity_emb[:, :1] if a.no_do else opacity_emb[:, :1] + self.opacity_deform(hidden)
        shs = shs_emb
        if not a.no_dshs:
            dshs = self.shs_deform(hidden).reshape([shs_emb.shape[0], 16, 3])
            shs = shs_emb + dshs
        if a.feat_head:
            feat = self.dino_head(hidden)
        return pts, scales, rotations, opacity, shs, dx, feat, dshs

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


class deform_network(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.timenet = nn.Sequential(nn.Linear(2 * args.timebase_pe + 1, args.timenet_width), nn.ReLU(),
                                     nn.Linear(args.timenet_width, args.timenet_output))  # unused by the reference too
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, input_ch=3 + 3 * args.posebase_pe * 2,
                                           grid_pe=args.grid_pe, input_ch_time=args.timenet_output, args=args)
        for name, n in (("time_poc", args.timebase_pe), ("pos_poc", args.posebase_pe),
                        ("rotation_scaling_poc", args.scale_rotation_pe), ("opacity_poc", args.opacity_pe)):
            self.register_buffer(name, torch.FloatTensor([2 ** i for i in range(n)]))
        for m in self.modules():  # initialize_weights (deformation.py:237-243): xavier on weights, default biases
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.get_empty_ratio

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        return self.forward_dynamic(point, scales, rotations, opacity, shs, times_sel)

    def forward_dynamic(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        # The reference builds sin/cos embeddings of point/scales/rotations here (poc_fre, deformation.py:218-220) and
        # then reads only their first 3/4 columns (= the raw inputs): the embeddings are dead code and are skipped.
        return self.deformation_net(point, scales, rotations, opacity, shs, None, times_sel)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters() + list(self.timenet.parameters())

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()
