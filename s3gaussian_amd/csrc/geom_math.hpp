// Per-Gaussian geometry math shared by the forward and backward per-Gaussian kernels (private header).
// All of it is compiled with -ffp-contract=off: one rounding per fp32 op, in the order the reference source
// writes it (glm 0.9.9.9 mat3 products expanded in glm's own summation order, type_mat3x3.inl:486-518), so the
// per-Gaussian integer outputs (radii, tile rectangles, instance counts) are bit-identical to the fp32 oracle.
#pragma once
#include "common.hpp"

namespace s3g {

struct M3 {
  float m[3][3];  // glm layout: m[column][row]
};
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
__device__ __forceinline__ M3 m3_T(const M3& A) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}

// Rotation matrix of an (un-normalised, like the reference) quaternion (r,x,y,z): forward.cu:127-138
__device__ __forceinline__ M3 quat_to_R(const float4 rot) {
  const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  M3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
  return R;
}

// forward.cu:118-152
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 rot, float* cov3D) {
  M3 S = {{{mod * scale.x, 0.f, 0.f}, {0.f, mod * scale.y, 0.f}, {0.f, 0.f, mod * scale.z}}};
  M3 Mm = m3_mul(S, quat_to_R(rot));
  M3 Sigma = m3_mul(m3_T(Mm), Mm);
  cov3D[0] = Sigma.m[0][0];
  cov3D[1] = Sigma.m[0][1];
  cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1];
  cov3D[4] = Sigma.m[1][2];
  cov3D[5] = Sigma.m[2][2];
}

// Everything computeCov2D needs in both directions (forward.cu:74-113, backward.cu:166-199).
struct Cov2DCtx {
  float3 t;  // view-space mean with x,y clamped to 1.3*tan(fov)
  float txtz, tytz, limx, limy;
  M3 W, T, Vrk, cov;  // cov is BEFORE the +0.3 low-pass
};
__device__ __forceinline__ Cov2DCtx cov2d_common(const float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                                 const float* cov3D, const float* __restrict__ V) {
  Cov2DCtx c;
  c.t = xform_4x3(mean, V);
  c.limx = 1.3f * tan_fovx;
  c.limy = 1.3f * tan_fovy;
  c.txtz = c.t.x / c.t.z;
  c.tytz = c.t.y / c.t.z;
  c.t.x = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * c.t.z;
  c.t.y = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * c.t.z;
  const float tz = c.t.z;
  M3 J = {{{fx / tz, 0.f, -(fx * c.t.x) / (tz * tz)}, {0.f, fy / tz, -(fy * c.t.y) / (tz * tz)}, {0.f, 0.f, 0.f}}};
  M3 W = {{{V[0], V[4], V[8]}, {V[1], V[5], V[9]}, {V[2], V[6], V[10]}}};
  M3 Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
  c.W = W;
  c.Vrk = Vrk;
  c.T = m3_mul(W, J);
  c.cov = m3_mul(m3_mul(m3_T(c.T), m3_T(Vrk)), c.T);
  return c;
}

// auxiliary.h:22-39
__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                   -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// Blend-kernel helpers shared by forward and backward so both take identical skip decisions.
constexpr float LOG2E = 1.4426950408889634f;

struct StagedGaussian {  // 48 B, three ds_read_b128
  float4 a;              // mean.x, mean.y, qa, qb       (q* = conic pre-scaled by -0.5*log2e / -log2e)
  float4 b;              // qc, opacity, depth, r
  float4 c;              // g, b, conic.x, conic.y       (un-scaled conic only used by the backward epilogue)
};

// Can this Gaussian put alpha >= 1/255 on ANY pixel of tile (tx, ty)?  The reference bins a Gaussian into every tile of
// the bounding square of its 3-sigma radius (auxiliary.h:46-56 getRect); an elongated Gaussian never touches most of them
// (half of all instances at cfg3), yet every pixel of those tiles evaluates it and throws the result away
// (forward.cu:330-341: `if (alpha < 1/255) continue`).  The test minimises the quadratic form q = a dx^2 + 2 b dx dy + c dy^2
// over the tile's rectangle of pixel centres (closed box, so at least as small as the minimum over the lattice): zero if
// the mean is inside, otherwise attained on the edge(s) of the box facing the mean; the tile is dropped only if
// 0.5 * q_min > ln(255 o) padded by 0.1 % + 1e-5 (alpha <= (1/255)(1 - 1e-5) everywhere), so rounding can never drop a
// pixel the exact test would keep.  o < 1/255 never passes (power <= 0); a degenerate conic keeps every tile.
struct TileCull {      // per-Gaussian part of the test, prepared once
  float a, b, c, inv_a, inv_c, budget, mx, my;
  int verdict;         // 0 = test each tile, 1 = keep every tile (degenerate conic), -1 = keep none (o < 1/255)
};
__device__ __forceinline__ TileCull tile_cull_prepare(float2 m, float4 co) {
  TileCull t;
  t.a = co.x; t.b = co.y; t.c = co.z; t.mx = m.x; t.my = m.y;
  const float o = co.w, det = t.a * t.c - t.b * t.b;
  const float L = __logf(255.0f * o);
  t.verdict = (!(det > 0.f) || !(t.a > 0.f) || !(t.c > 0.f) || !(o == o)) ? 1 : (L < 0.f ? -1 : 0);
  t.inv_a = 1.0f / t.a; t.inv_c = 1.0f / t.c;
  t.budget = 2.0f * (L * 1.001f + 1.0e-5f);  // keep a tile iff q_min <= budget
  return t;
}
// min over y in [lo, hi] of A X^2 + 2 B X y + C y^2
__device__ __forceinline__ float edge_min_q(float X, float A, float B, float C, float inv_C, float lo, float hi) {
  const float y = fminf(hi, fmaxf(lo, -(B * X) * inv_C));
  return A * X * X + 2.f * B * X * y + C * y * y;
}
__device__ __forceinline__ bool tile_can_contribute(const TileCull& t, int tx, int ty, int W, int H) {
  if (t.verdict != 0) return t.verdict > 0;
  // d = mean - pixel; pixel centres span [x0, x1] x [y0, y1], widened by 0.01 px
  const float x0 = (float)(tx * TILE_X) - 0.01f, x1 = (float)min(tx * TILE_X + TILE_X - 1, W - 1) + 0.01f;
  const float y0 = (float)(ty * TILE_Y) - 0.01f, y1 = (float)min(ty * TILE_Y + TILE_Y - 1, H - 1) + 0.01f;
  const float dx_lo = t.mx - x1, dx_hi = t.mx - x0, dy_lo = t.my - y1, dy_hi = t.my - y0;
  const bool in_x = dx_lo <= 0.f && dx_hi >= 0.f, in_y = dy_lo <= 0.f && dy_hi >= 0.f;
  if (in_x && in_y) return true;  // mean inside the tile
  // the nearer vertical and the nearer horizontal edge (the minimum over the box lies on one of them)
  const float X = in_x ? 0.f : (dx_lo > 0.f ? dx_lo : dx_hi), Y = in_y ? 0.f : (dy_lo > 0.f ? dy_lo : dy_hi);
  float q = 3.0e38f;
  if (!in_x) q = edge_min_q(X, t.a, t.b, t.c, t.inv_c, dy_lo, dy_hi);
  if (!in_y) q = fminf(q, edge_min_q(Y, t.c, t.b, t.a, t.inv_a, dx_lo, dx_hi));
  return q <= t.budget;
}

__device__ __forceinline__ float gaussian_exponent2(float dx, float dy, float qa, float qb, float qc) {
  float q = (qb * dx) * dy;
  q = __builtin_fmaf(qc * dy, dy, q);
  q = __builtin_fmaf(qa * dx, dx, q);
  return q;
}

}  // namespace s3g
