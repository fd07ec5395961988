"""Seeded synthetic scenes and pinhole cameras for tests and bench.py (no Waymo data in the sandbox).

Camera conventions replicate the reference exactly:
  * `world_view_transform` = (world->camera 4x4)^T, i.e. row-vector convention  (scene/cameras.py:59)
  * `projection_matrix`    = getProjectionMatrix(...)^T                          (scene/cameras.py:61, utils/graphics_utils.py:54-74)
  * `full_proj_transform`  = world_view_transform @ projection_matrix            (scene/cameras.py:63)
  * `camera_center`        = inverse(world_view_transform)[3,:3]                 (scene/cameras.py:64)
Camera frame is OpenCV: +x right, +y down, +z forward.  Scene generators follow SURVEY.md 8(d).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """utils/graphics_utils.py:54-74 (un-transposed)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(c2w_R: np.ndarray, cam_pos: np.ndarray, fovx: float, fovy: float, width: int, height: int,
                znear: float = 0.01, zfar: float = 100.0, time: float = 0.0) -> Dict:
    """c2w_R: 3x3 camera->world rotation (columns = camera axes in world), cam_pos: camera centre in world."""
    w2c = np.eye(4, dtype=np.float64)
    w2c[:3, :3] = c2w_R.T
    w2c[:3, 3] = -c2w_R.T @ cam_pos
    view = np.float32(w2c).T  # world_view_transform
    proj = projection_matrix(znear, zfar, fovx, fovy).T
    full = view @ proj
    center = np.linalg.inv(view)[3, :3]
    return dict(image_height=int(height), image_width=int(width), tanfovx=math.tan(fovx * 0.5),
                tanfovy=math.tan(fovy * 0.5), FoVx=fovx, FoVy=fovy,
                viewmatrix=torch.from_numpy(np.ascontiguousarray(view)),
                projmatrix=torch.from_numpy(np.ascontiguousarray(full.astype(np.float32))),
                campos=torch.from_numpy(np.ascontiguousarray(center.astype(np.float32))), time=float(time))


def look_at(eye, target, up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """Camera->world rotation for an OpenCV camera at `eye` looking at `target`."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)


def _base_gaussians(g: torch.Generator, P: int, xyz: torch.Tensor, log_scale_mu: float, log_scale_sigma: float,
                    sh_degree: int) -> Dict[str, torch.Tensor]:
    M = (sh_degree + 1) ** 2
    log_scales = math.log(log_scale_mu) + log_scale_sigma * torch.randn(P, 3, generator=g)
    quats = torch.randn(P, 4, generator=g)
    quats = quats / quats.norm(dim=1, keepdim=True)
    opacity_logit = 1.5 * torch.randn(P, 1, generator=g)
    shs = torch.cat([torch.randn(P, 1, 3, generator=g), 0.1 * torch.randn(P, M - 1, 3, generator=g)], dim=1)
    return dict(xyz=xyz.float().contiguous(), log_scales=log_scales, rotations_raw=quats, opacity_logit=opacity_logit,
                shs=shs.contiguous())


def cfg1_scene(P: int = 10_000, seed: int = 0, width: int = 400, height: int = 400, sh_degree: int = 3) -> Dict:
    """BASELINE config #1: random Gaussians in [-1,1]^3 placed 3-6 units in front of one 60-degree pinhole camera."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(P, 3, generator=g) * 2 - 1
    xyz[:, 2] = 3.0 + 3.0 * torch.rand(P, generator=g)  # camera looks along +z from the origin
    xyz[:, :2] *= 2.0
    gs = _base_gaussians(g, P, xyz, 0.02, 0.5, sh_degree)
    fov = math.radians(60.0)
    cam = make_camera(np.eye(3), np.zeros(3), fov, 2 * math.atan(math.tan(fov / 2) * height / width), width, height)
    return dict(gaussians=gs, cameras=[cam], sh_degree=sh_degree, bg=torch.zeros(3))


def street_scene(P: int = 1_200_000, seed: int = 0, width: int = 1600, height: int = 1066, n_frames: int = 50,
                 sh_degree: int = 3) -> Dict:
    """BASELINE configs #2/#3: street-like slab, x forward / y left / z up, 3 cameras x n_frames."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([-20.0, -20.0, -2.0])
    hi = torch.tensor([80.0, 20.0, 8.0])
    xyz = lo + (hi - lo) * torch.rand(P, 3, generator=g)
    gs = _base_gaussians(g, P, xyz, 0.05, 0.6, sh_degree)
    fx = 2080.0 * (width / 1920.0)
    fovx, fovy = 2 * math.atan(width / (2 * fx)), 2 * math.atan(height / (2 * fx))
    cams = []
    for k in range(n_frames):
        eye = np.array([0.5 * k, 0.0, 1.8])
        for yaw_deg in (0.0, 45.0, -45.0):
            yaw = math.radians(yaw_deg)
            fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0])
            cams.append(make_camera(look_at(eye, eye + fwd), eye, fovx, fovy, width, height,
                                    time=k / max(n_frames - 1, 1)))
    # hexplane AABB = scene bounds (dataset_readers.py:749-779 derives it from the frustum/lidar extent)
    return dict(gaussians=gs, cameras=cams, sh_degree=sh_degree, bg=torch.zeros(3), aabb=(hi.tolist(), lo.tolist()))
