// Fused per-Gaussian render glue: activations + SH->RGB (forward and backward), one HBM pass each way.
// Reference: gaussian_renderer/__init__.py:99-115 + utils/sh_utils.py:57-112 = ~40 elementwise launches forward and
// ~80 backward over [P,16,3] tensors (230 MB each at 1.2 M Gaussians).  HBM-bound: ~420 B per Gaussian per direction.
#include "geom_math.hpp"

#include "../../include/s3g_glue.h"
#include "../../include/s3g_loss.h"  // S3G_SUM_* (slotted accumulators)

namespace s3g {

struct GlueArgs {
  int P, deg;
  const float *f_dc, *f_rest, *dshs, *xyz, *campos, *log_scales, *rot_raw, *opacity_logit;
  float *colors, *scales, *rot, *opacity;
  // backward only
  const float *g_colors, *g_scales, *g_rot, *g_opacity;
  float *g_f_dc, *g_f_rest, *g_dshs, *g_xyz, *g_log_scales, *g_rot_raw, *g_opacity_logit;
  // optional L1 regulariser on dshs (train.py:407-410: lambda_dshs * mean|dshs|), folded in because both kernels stream
  // dshs anyway: forward accumulates sum|dshs|, backward adds (*g_dshs_l1 / (48 P)) * sign(dshs)
  double* dshs_abs_sum;
  const float* g_dshs_l1;
};

__device__ __forceinline__ void load_sh(const GlueArgs& a, int p, float (&sh)[16][3]) {
#pragma unroll
  for (int c = 0; c < 3; c++) sh[0][c] = a.f_dc[3 * (size_t)p + c];
#pragma unroll
  for (int k = 1; k < 16; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) sh[k][c] = a.f_rest[(size_t)p * 45 + (k - 1) * 3 + c];
  if (a.dshs != nullptr) {
#pragma unroll
    for (int k = 0; k < 16; k++)
#pragma unroll
      for (int c = 0; c < 3; c++) sh[k][c] += a.dshs[(size_t)p * 48 + k * 3 + c];
  }
}

// thread = Gaussian: activations, then SH -> RGB from the summed coefficient rows sh (registers)
__device__ __forceinline__ void glue_forward_point(const GlueArgs& a, int p, const float (&sh)[16][3]) {
  // activations
#pragma unroll
  for (int k = 0; k < 3; k++) a.scales[3 * (size_t)p + k] = expf(a.log_scales[3 * (size_t)p + k]);
  const float4 q = reinterpret_cast<const float4*>(a.rot_raw)[p];
  const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);  // F.normalize eps
  reinterpret_cast<float4*>(a.rot)[p] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
  a.opacity[p] = 1.0f / (1.0f + expf(-a.opacity_logit[p]));
  // SH -> RGB (utils/sh_utils.py:57-112; same polynomial order as the rasterizer's in-kernel path)
  float dx = a.xyz[3 * (size_t)p] - a.campos[0], dy = a.xyz[3 * (size_t)p + 1] - a.campos[1], dz = a.xyz[3 * (size_t)p + 2] - a.campos[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / len, y = dy / len, z = dz / len;
  const int deg = a.deg;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float v = SH_C0 * sh[0][c];
    if (deg > 0) {
      v = v - SH_C1 * y * sh[1][c] + SH_C1 * z * sh[2][c] - SH_C1 * x * sh[3][c];
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        v = v + SH_C2[0] * xy * sh[4][c] + SH_C2[1] * yz * sh[5][c] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[6][c] +
            SH_C2[3] * xz * sh[7][c] + SH_C2[4] * (xx - yy) * sh[8][c];
        if (deg > 2) {
          v = v + SH_C3[0] * y * (3.0f * xx - yy) * sh[9][c] + SH_C3[1] * xy * z * sh[10][c] +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11][c] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12][c] +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13][c] + SH_C3[5] * z * (xx - yy) * sh[14][c] +
              SH_C3[6] * x * (xx - 3.0f * yy) * sh[15][c];
        }
      }
    }
    a.colors[3 * (size_t)p + c] = fmaxf(v + 0.5f, 0.f);
  }
}

// Forward with a deformation (dshs present).  Phase 1 (workgroup, coalesced): the block's 256 coefficient rows of dshs go through
// LDS as contiguous sweeps -- a wave instruction covers 256 consecutive bytes -- and pick up (f_dc | f_rest) on the way: element
// e = 48 q + r of the slice is f_rest[45 q + r - 3] = rest[e - 3 (q + 1)], contiguous but for a three-float step per row; the
// three f_dc columns follow in a short second sweep.  (Read per lane as 45- / 48-float rows, every load instruction touched 64
// different lines: 24 x dwordx4 per lane, 151 us for 0.57 GB -- the address path, not HBM.  Without dshs the 12 row loads that are
// left run at the HBM rate and that kernel is kept: glue_forward_rows_kernel.)  The |dshs| sum of the regulariser is taken from
// the same registers instead of a second sweep.  Phase 2 (thread = Gaussian): rows read back at an odd stride (conflict-free),
// same polynomial, and every coefficient is the same single addition f + d as before: colours are bit-identical.
constexpr int GLUE_ROW = 49;

template <bool FULL>   // FULL: all 256 rows of the block exist (compile-time trip counts: the loads of a sweep are issued back to back)
__device__ __forceinline__ float glue_stage_rows(const GlueArgs& a, int p0, int nrows, float* rows) {
  const float* dc = a.f_dc + (size_t)p0 * 3;
  const float* rest = a.f_rest + (size_t)p0 * 45;
  const float* d = a.dshs + (size_t)p0 * 48;
  const int n48 = FULL ? 256 * 48 : nrows * 48, n3 = FULL ? 256 * 3 : nrows * 3;
  // e = tid + 256 i; 256 = 5 * 48 + 16: (q, r) advance by (5, 16) with one carry; LDS word q * 49 + r, rest index e - 3 (q + 1)
  int r = threadIdx.x % 48, q = threadIdx.x / 48;
  int word = q * GLUE_ROW + r, src = (int)threadIdx.x - 3 * (q + 1);
  float s = 0.f;
#pragma unroll 16
  for (int e = threadIdx.x; e < n48; e += 256) {
    const float v = d[e];
    const float f = rest[src < 0 ? 0 : src];   // unconditional (no branch around the load): r < 3 reads a neighbour it does not use
    s += fabsf(v);
    rows[word] = r >= 3 ? f + v : v;
    r += 16; word += 5 * GLUE_ROW + 16; src += 256 - 15;
    if (r >= 48) { r -= 48; word += GLUE_ROW - 48; src -= 3; }
  }
  __syncthreads();   // the f_dc columns are added by other threads than the ones that stored dshs there
  for (int e = threadIdx.x; e < n3; e += 256) rows[(e / 3) * GLUE_ROW + e % 3] += dc[e];
  return s;
}

__global__ void __launch_bounds__(256) glue_forward_kernel(const GlueArgs a) {   // a.dshs != nullptr
  __shared__ float rows[256 * GLUE_ROW];
  __shared__ double part[4];
  const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x, nrows = min(256, a.P - p0);
  const float s = nrows == 256 ? glue_stage_rows<true>(a, p0, nrows, rows) : glue_stage_rows<false>(a, p0, nrows, rows);
  if (a.dshs_abs_sum != nullptr) {
    double t = (double)s;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (a.dshs_abs_sum != nullptr && threadIdx.x == 0)
    atomicAdd(&a.dshs_abs_sum[(blockIdx.x % S3G_SUM_SLOTS) * S3G_SUM_STRIDE], part[0] + part[1] + part[2] + part[3]);
  if (p >= a.P) return;
  float sh[16][3];
#pragma unroll
  for (int k = 0; k < 16; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) sh[k][c] = rows[threadIdx.x * GLUE_ROW + 3 * k + c];
  glue_forward_point(a, p, sh);
}

__global__ void __launch_bounds__(256) glue_forward_rows_kernel(const GlueArgs a) {   // no dshs (coarse stage): 12 row loads per lane
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.P) return;
  float sh[16][3];
  load_sh(a, p, sh);
  glue_forward_point(a, p, sh);
}

// Phase 1 (thread = Gaussian): everything but the SH coefficient gradients; the 16 basis values and the 3 colour gradients
// of the Gaussian go to LDS.  Phase 2 (workgroup, coalesced): g_f_dc / g_f_rest / g_dshs rows are the outer product
// basis[k] * dRGB[c] -- written as contiguous sweeps instead of 45- / 48-float rows per lane.
__device__ __forceinline__ void glue_backward_point(const GlueArgs& a, int p, float* stage);

__global__ void __launch_bounds__(256) glue_backward_kernel(const GlueArgs a) {
  __shared__ float stage[256 * 19];
  const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
  if (p < a.P) glue_backward_point(a, p, stage + threadIdx.x * 19);
  __syncthreads();
  const int n = min(256, a.P - p0);
  for (int e = threadIdx.x; e < n * 3; e += 256) a.g_f_dc[(size_t)p0 * 3 + e] = stage[(e / 3) * 19] * stage[(e / 3) * 19 + 16 + e % 3];
  for (int e = threadIdx.x; e < n * 45; e += 256) {
    const int q = e / 45, kc = e % 45 + 3;
    a.g_f_rest[(size_t)p0 * 45 + e] = stage[q * 19 + kc / 3] * stage[q * 19 + 16 + kc % 3];
  }
  if (a.g_dshs != nullptr) {
    const float l1 = a.g_dshs_l1 != nullptr ? *a.g_dshs_l1 / (48.0f * (float)a.P) : 0.f;
    for (int e = threadIdx.x; e < n * 48; e += 256) {
      const int q = e / 48, kc = e % 48;
      float v = stage[q * 19 + kc / 3] * stage[q * 19 + 16 + kc % 3];
      if (a.g_dshs_l1 != nullptr) {
        const float d = a.dshs[(size_t)p0 * 48 + e];
        v += d > 0.f ? l1 : (d < 0.f ? -l1 : 0.f);
      }
      a.g_dshs[(size_t)p0 * 48 + e] = v;
    }
  }
}

__device__ __forceinline__ void glue_backward_point(const GlueArgs& a, int p, float* stage) {
  // activations
#pragma unroll
  for (int k = 0; k < 3; k++)
    a.g_log_scales[3 * (size_t)p + k] = a.g_scales ? a.g_scales[3 * (size_t)p + k] * a.scales[3 * (size_t)p + k] : 0.f;
  {
    const float4 q = reinterpret_cast<const float4*>(a.rot_raw)[p];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.g_rot) {
      const float4 g = reinterpret_cast<const float4*>(a.g_rot)[p];
      if (nrm > 1e-12f) {  // d(x/n) = (g - y (y.g)) / n
        const float4 yv = reinterpret_cast<const float4*>(a.rot)[p];
        const float dot = yv.x * g.x + yv.y * g.y + yv.z * g.z + yv.w * g.w;
        gq = make_float4((g.x - yv.x * dot) / nrm, (g.y - yv.y * dot) / nrm, (g.z - yv.z * dot) / nrm, (g.w - yv.w * dot) / nrm);
      } else {
        gq = make_float4(g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f, g.w / 1e-12f);
      }
    }
    reinterpret_cast<float4*>(a.g_rot_raw)[p] = gq;
  }
  {
    const float o = a.opacity[p];
    a.g_opacity_logit[p] = a.g_opacity ? a.g_opacity[p] * o * (1.f - o) : 0.f;
  }
  // SH backward (same derivation as backward.cu:20-139, coefficients split over f_dc / f_rest / dshs)
  float sh[16][3];
  load_sh(a, p, sh);
  float dRGB[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float g = a.g_colors ? a.g_colors[3 * (size_t)p + c] : 0.f;
    dRGB[c] = a.colors[3 * (size_t)p + c] > 0.f ? g : 0.f;  // clamp_min(., 0): zero gradient where the clamp is active
  }
  const float ox = a.xyz[3 * (size_t)p] - a.campos[0], oy = a.xyz[3 * (size_t)p + 1] - a.campos[1], oz = a.xyz[3 * (size_t)p + 2] - a.campos[2];
  const float len = sqrtf(ox * ox + oy * oy + oz * oz);
  const float x = ox / len, y = oy / len, z = oz / len;
  const int deg = a.deg;
  float dsh[16], dRdx[3] = {0.f, 0.f, 0.f}, dRdy[3] = {0.f, 0.f, 0.f}, dRdz[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; k++) dsh[k] = 0.f;
  dsh[0] = SH_C0;
  if (deg > 0) {
    dsh[1] = -SH_C1 * y; dsh[2] = SH_C1 * z; dsh[3] = -SH_C1 * x;
#pragma unroll
    for (int c = 0; c < 3; c++) { dRdx[c] = -SH_C1 * sh[3][c]; dRdy[c] = -SH_C1 * sh[1][c]; dRdz[c] = SH_C1 * sh[2][c]; }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dsh[4] = SH_C2[0] * xy; dsh[5] = SH_C2[1] * yz; dsh[6] = SH_C2[2] * (2.f * zz - xx - yy);
      dsh[7] = SH_C2[3] * xz; dsh[8] = SH_C2[4] * (xx - yy);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        dRdx[c] += SH_C2[0] * y * sh[4][c] + SH_C2[2] * 2.f * -x * sh[6][c] + SH_C2[3] * z * sh[7][c] + SH_C2[4] * 2.f * x * sh[8][c];
        dRdy[c] += SH_C2[0] * x * sh[4][c] + SH_C2[1] * z * sh[5][c] + SH_C2[2] * 2.f * -y * sh[6][c] + SH_C2[4] * 2.f * -y * sh[8][c];
        dRdz[c] += SH_C2[1] * y * sh[5][c] + SH_C2[2] * 2.f * 2.f * z * sh[6][c] + SH_C2[3] * x * sh[7][c];
      }
      if (deg > 2) {
        dsh[9] = SH_C3[0] * y * (3.f * xx - yy); dsh[10] = SH_C3[1] * xy * z; dsh[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        dsh[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); dsh[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        dsh[14] = SH_C3[5] * z * (xx - yy); dsh[15] = SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          dRdx[c] += (SH_C3[0] * sh[9][c] * 3.f * 2.f * xy + SH_C3[1] * sh[10][c] * yz + SH_C3[2] * sh[11][c] * -2.f * xy +
                      SH_C3[3] * sh[12][c] * -3.f * 2.f * xz + SH_C3[4] * sh[13][c] * (-3.f * xx + 4.f * zz - yy) +
                      SH_C3[5] * sh[14][c] * 2.f * xz + SH_C3[6] * sh[15][c] * 3.f * (xx - yy));
          dRdy[c] += (SH_C3[0] * sh[9][c] * 3.f * (xx - yy) + SH_C3[1] * sh[10][c] * xz + SH_C3[2] * sh[11][c] * (-3.f * yy + 4.f * zz - xx) +
                      SH_C3[3] * sh[12][c] * -3.f * 2.f * yz + SH_C3[4] * sh[13][c] * -2.f * xy + SH_C3[5] * sh[14][c] * -2.f * yz +
                      SH_C3[6] * sh[15][c] * -3.f * 2.f * xy);
          dRdz[c] += (SH_C3[1] * sh[10][c] * xy + SH_C3[2] * sh[11][c] * 4.f * 2.f * yz + SH_C3[3] * sh[12][c] * 3.f * (2.f * zz - xx - yy) +
                      SH_C3[4] * sh[13][c] * 4.f * 2.f * xz + SH_C3[5] * sh[14][c] * (xx - yy));
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) stage[k] = dsh[k];
#pragma unroll
  for (int c = 0; c < 3; c++) stage[16 + c] = dRGB[c];
  const float ddx = dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2];
  const float ddy = dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2];
  const float ddz = dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2];
  const float sum2 = ox * ox + oy * oy + oz * oz;
  const float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);  // d normalize(v)/dv, auxiliary.h:107-117
  a.g_xyz[3 * (size_t)p + 0] = ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * inv;
  a.g_xyz[3 * (size_t)p + 1] = (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * inv;
  a.g_xyz[3 * (size_t)p + 2] = (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * inv;
}

}  // namespace s3g

using namespace s3g;

extern "C" int s3g_glue_forward(int P, int deg, const float* f_dc, const float* f_rest, const float* dshs, const float* xyz,
                                const float* campos, const float* log_scales, const float* rot_raw,
                                const float* opacity_logit, float* colors, float* scales, float* rot, float* opacity,
                                double* dshs_abs_sum, void* stream_) {
  if (P < 0 || deg < 0 || deg > 3 ||
      (P > 0 && (!f_dc || !f_rest || !xyz || !campos || !log_scales || !rot_raw || !opacity_logit || !colors || !scales || !rot || !opacity))) {
    set_error("s3g_glue_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  GlueArgs a;
  memset(&a, 0, sizeof a);
  a.P = P; a.deg = deg; a.f_dc = f_dc; a.f_rest = f_rest; a.dshs = dshs; a.xyz = xyz; a.campos = campos;
  a.log_scales = log_scales; a.rot_raw = rot_raw; a.opacity_logit = opacity_logit;
  a.colors = colors; a.scales = scales; a.rot = rot; a.opacity = opacity;
  a.dshs_abs_sum = dshs_abs_sum;
  if (dshs != nullptr) hipLaunchKernelGGL(glue_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  else hipLaunchKernelGGL(glue_forward_rows_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_glue_backward(int P, int deg, const float* f_dc, const float* f_rest, const float* dshs, const float* xyz,
                                 const float* campos, const float* rot_raw, const float* colors, const float* scales,
                                 const float* rot, const float* opacity, const float* g_colors, const float* g_scales,
                                 const float* g_rot, const float* g_opacity, float* g_f_dc, float* g_f_rest, float* g_dshs,
                                 float* g_xyz, float* g_log_scales, float* g_rot_raw, float* g_opacity_logit,
                                 const float* g_dshs_l1, void* stream_) {
  if (P < 0 || deg < 0 || deg > 3 ||
      (P > 0 && (!f_dc || !f_rest || !xyz || !campos || !rot_raw || !colors || !scales || !rot || !opacity || !g_f_dc ||
                 !g_f_rest || !g_xyz || !g_log_scales || !g_rot_raw || !g_opacity_logit))) {
    set_error("s3g_glue_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  GlueArgs a;
  memset(&a, 0, sizeof a);
  a.P = P; a.deg = deg; a.f_dc = f_dc; a.f_rest = f_rest; a.dshs = dshs; a.xyz = xyz; a.campos = campos; a.rot_raw = rot_raw;
  a.colors = const_cast<float*>(colors); a.scales = const_cast<float*>(scales); a.rot = const_cast<float*>(rot);
  a.opacity = const_cast<float*>(opacity);
  a.g_colors = g_colors; a.g_scales = g_scales; a.g_rot = g_rot; a.g_opacity = g_opacity;
  a.g_f_dc = g_f_dc; a.g_f_rest = g_f_rest; a.g_dshs = g_dshs; a.g_xyz = g_xyz; a.g_log_scales = g_log_scales;
  a.g_rot_raw = g_rot_raw; a.g_opacity_logit = g_opacity_logit;
  a.g_dshs_l1 = (dshs != nullptr && g_dshs != nullptr) ? g_dshs_l1 : nullptr;
  hipLaunchKernelGGL(glue_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
