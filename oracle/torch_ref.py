"""float64 PyTorch *autograd* restatement of the reference rasterizer forward (tiny scenes only).

TEST INFRASTRUCTURE ONLY.  Purpose: pin the hand-derived backward of oracle/raster_oracle.c (and through it
the HIP kernels) against automatic differentiation of the forward formulae, independent of any
hand-written gradient code.  Every Gaussian/pixel pair is materialised ([H*W, P] matrices), so keep
P <= ~2000 and images <= 64x64.

Follows RAST/cuda_rasterizer/forward.cu:74-379 and auxiliary.h:41-164.  Where the reference backward is
*not* the true derivative of its forward, the forward below is written with detach()/straight-through so
that autograd reproduces the reference's convention:
  * alpha = min(0.99, o*G): the reference back-propagates through the clamp as identity (backward.cu:525,571)
  * clamped t.x, t.y in computeCov2D: treated as constants w.r.t. t.z when clamped (backward.cu:175-176,262-264)
  * dconic/dcov uses 1/(det^2+1e-7) instead of 1/det^2 (backward.cu:203) -- NOT replicated; choose test
    scenes with det >> 1e-3 so the difference is < 1e-9 relative.
"""
from __future__ import annotations


import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_color(deg, shs, means, campos):
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = shs
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def rasterize(*, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
              sh_degree=0, scale_modifier=1.0, colors_precomp=None, shs=None, scales=None, rotations=None,
              cov3D_precomp=None, means2D=None):
    """Returns (color[3,H,W], radii[P], depth[1,H,W]).  `means2D` ([P,3], zeros) is the reference's dummy
    screen-space tensor: it is added in NDC x/y so that its autograd gradient equals dL_dmean2D."""
    dt = torch.float64
    H, W = int(image_height), int(image_width)
    P = means3D.shape[0]
    V, PM = viewmatrix.to(dt).reshape(16), projmatrix.to(dt).reshape(16)
    m = means3D
    mx, my, mz = m[:, 0], m[:, 1], m[:, 2]
    # transformPoint4x3 / 4x4 (column-major indexing of the flattened row-vector matrices)
    tx = V[0] * mx + V[4] * my + V[8] * mz + V[12]
    ty = V[1] * mx + V[5] * my + V[9] * mz + V[13]
    tz = V[2] * mx + V[6] * my + V[10] * mz + V[14]
    hx = PM[0] * mx + PM[4] * my + PM[8] * mz + PM[12]
    hy = PM[1] * mx + PM[5] * my + PM[9] * mz + PM[13]
    hw = PM[3] * mx + PM[7] * my + PM[11] * mz + PM[15]
    p_w = 1.0 / (hw + 0.0000001)
    ndc_x, ndc_y = hx * p_w, hy * p_w
    if means2D is not None:
        ndc_x = ndc_x + means2D[:, 0]
        ndc_y = ndc_y + means2D[:, 1]
    in_front = tz > 0.2
    # cov3D
    if cov3D_precomp is None:
        s = scale_modifier * scales
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        # glm column-major constructor: R[c][r]; as a math matrix Rm[r][c] = R[c][r]
        Rg = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)  # [P, c, r]
        Rm = Rg.transpose(1, 2)                      # math matrix
        Sm = torch.diag_embed(s)
        Mm = Sm @ Rm                                 # glm S*R in math terms
        Sigma = Mm.transpose(1, 2) @ Mm
        cov3 = Sigma
    else:
        c = cov3D_precomp
        cov3 = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1), torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                            torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], 1)
    # cov2D
    fy_, fx_ = H / (2.0 * tanfovy), W / (2.0 * tanfovx)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(in_front, tz, torch.ones_like(tz))
    txtz, tytz = tx / tzs, ty / tzs
    inx, iny = (txtz >= -limx) & (txtz <= limx), (tytz >= -limy) & (tytz <= limy)
    txc = torch.where(inx, txtz * tzs, (txtz.clamp(-limx, limx) * tzs).detach())
    tyc = torch.where(iny, tytz * tzs, (tytz.clamp(-limy, limy) * tzs).detach())
    zero = torch.zeros_like(tzs)
    # math matrix of glm J: Jm[r][c] = J[c][r]
    Jm = torch.stack([torch.stack([fx_ / tzs, zero, zero], -1), torch.stack([zero, fy_ / tzs, zero], -1),
                      torch.stack([-(fx_ * txc) / (tzs * tzs), -(fy_ * tyc) / (tzs * tzs), zero], -1)], 1)
    Wm = torch.stack([torch.stack([V[0], V[1], V[2]]), torch.stack([V[4], V[5], V[6]]),
                      torch.stack([V[8], V[9], V[10]])])  # math matrix of glm W (W[c][r]: c0=(v0,v4,v8))
    Tm = Wm.unsqueeze(0) @ Jm
    cov2 = Tm.transpose(1, 2) @ cov3.transpose(1, 2) @ Tm
    # glm cov[0][0], cov[0][1], cov[1][1] (symmetric, so index order is irrelevant)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c_ / det_s, -b / det_s, a / det_s], -1)
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2))).detach()
    pix_x = ((ndc_x + 1.0) * W - 1.0) * 0.5
    pix_y = ((ndc_y + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r_i = radius.to(torch.int64).to(dt)

    def _trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi).to(torch.int64)

    rmin_x = _trunc_clamp((pix_x.detach() - r_i) / 16, gx)
    rmin_y = _trunc_clamp((pix_y.detach() - r_i) / 16, gy)
    rmax_x = _trunc_clamp((pix_x.detach() + r_i + 15) / 16, gx)
    rmax_y = _trunc_clamp((pix_y.detach() + r_i + 15) / 16, gy)
    ok = ok & (((rmax_x - rmin_x) * (rmax_y - rmin_y)) != 0)
    radii = torch.where(ok, radius.to(torch.int32), torch.zeros_like(radius, dtype=torch.int32))
    # colours
    if colors_precomp is None:
        colors = _sh_color(sh_degree, shs, means3D, campos.to(dt))
    else:
        colors = colors_precomp
    depth_g = tz
    # order: (depth fp32 bits, index) -- positive floats order like their bit patterns
    order = torch.argsort(depth_g.detach().to(torch.float32), stable=True)
    order = order[ok[order]]
    # per pixel
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    tile_x, tile_y = xs // 16, ys // 16
    o = order
    member = ((tile_x[:, None] >= rmin_x[o][None]) & (tile_x[:, None] < rmax_x[o][None]) &
              (tile_y[:, None] >= rmin_y[o][None]) & (tile_y[:, None] < rmax_y[o][None]))
    dx = pix_x[o][None] - xs[:, None].to(dt)
    dy = pix_y[o][None] - ys[:, None].to(dt)
    cn = conic[o]
    power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
    G = torch.exp(torch.clamp_max(power, 0.0))
    raw = opacities.reshape(-1)[o][None] * G
    alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()  # straight-through min(0.99, .)
    active = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(active, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)                      # T after applying k (test_T when active)
    T_before = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    stop = active & (T_incl.detach() < 0.0001)
    stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0    # this and all later entries are not applied
    applied = active & ~stopped
    w = torch.where(applied, a_eff * T_before, torch.zeros_like(a_eff))
    one_m_applied = torch.where(applied, one_m, torch.ones_like(one_m))
    T_final = torch.prod(one_m_applied, dim=1)
    col = w @ colors[o]                                        # [HW,3]
    dep = w @ depth_g[o]
    color = (col + T_final[:, None] * bg.to(dt)[None]).t().reshape(3, H, W)
    depth = dep.reshape(1, H, W)
    return color, radii, depth
