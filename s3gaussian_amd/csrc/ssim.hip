// Fused SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward + backward for gfx950.
// Reference: utils/loss_utils.py:56-96 = 5 grouped conv2d + ~15 elementwise kernels forward, the same again backward
// (on MI355X MIOpen picks miopenSp3AsmConv for the grouped 11x11 convolutions: ~13 ms per step at 1066x1600).
// Here: one kernel per direction; a 16x16 pixel tile per workgroup, 26x26 halo tile in LDS, separable passes.
#include "common.hpp"

#include "../../include/s3g_loss.h"

namespace s3g {

constexpr int SS_T = 16, SS_R = 5, SS_H = SS_T + 2 * SS_R;  // tile, radius, halo tile edge (26)

struct SsimWindow {
  float g[11];
};
static SsimWindow make_window() {  // gaussian(11, 1.5), loss_utils.py:56-58, computed in fp32 like torch.Tensor([...])
  SsimWindow w;
  float s = 0.f;
  for (int x = 0; x < 11; x++) {
    w.g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5));
    s += w.g[x];
  }
  for (int x = 0; x < 11; x++) w.g[x] /= s;
  return w;
}

__global__ void __launch_bounds__(256) ssim_forward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, const SsimWindow win,
                                                           double* __restrict__ ssim_sum, float* __restrict__ m_mu1,
                                                           float* __restrict__ m_s11, float* __restrict__ m_s12) {
  __shared__ float t1[SS_H][SS_H + 1], t2[SS_H][SS_H + 1];
  __shared__ float hb[5][SS_H][SS_T + 1];
  __shared__ float red[4];
  const int c = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const size_t plane = (size_t)c * H * W;
  for (int i = tid; i < SS_H * SS_H; i += 256) {
    const int r = i / SS_H, q = i - r * SS_H;
    const int gy = y0 + r - SS_R, gx = x0 + q - SS_R;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    t1[r][q] = in ? img1[plane + (size_t)gy * W + gx] : 0.f;
    t2[r][q] = in ? img2[plane + (size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < SS_H * SS_T; i += 256) {  // horizontal pass
    const int r = i / SS_T, q = i - r * SS_T;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.g[k], u = t1[r][q + k], v = t2[r][q + k];
      a += w * u; b += w * v; aa += w * (u * u); bb += w * (v * v); ab += w * (u * v);
    }
    hb[0][r][q] = a; hb[1][r][q] = b; hb[2][r][q] = aa; hb[3][r][q] = bb; hb[4][r][q] = ab;
  }
  __syncthreads();
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {  // vertical pass
    const float w = win.g[k];
    mu1 += w * hb[0][ly + k][lx]; mu2 += w * hb[1][ly + k][lx]; e11 += w * hb[2][ly + k][lx];
    e22 += w * hb[3][ly + k][lx]; e12 += w * hb[4][ly + k][lx];
  }
  const int gx = x0 + lx, gy = y0 + ly;
  float val = 0.f;
  if (gx < W && gy < H) {
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s11 = e11 - mu1_sq, s22 = e22 - mu2_sq, s12 = e12 - mu12;
    const float a = 2.f * mu12 + C1, b = 2.f * s12 + C2, cc = mu1_sq + mu2_sq + C1, d = s11 + s22 + C2;
    const float inv = 1.f / (cc * d);
    val = (a * b) * inv;
    const float dm_ds11 = -val / d;                 // d map / d sigma1_sq
    const float dm_ds12 = 2.f * a * inv;            // d map / d sigma12
    const float dm_dmu1 = 2.f * mu2 * b * inv - 2.f * mu1 * val / cc + dm_ds11 * (-2.f * mu1) + dm_ds12 * (-mu2);
    const size_t o = plane + (size_t)gy * W + gx;
    m_mu1[o] = dm_dmu1; m_s11[o] = dm_ds11; m_s12[o] = dm_ds12;
  }
  for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
  if ((tid & 63) == 0) red[tid >> 6] = val;
  __syncthreads();
  if (tid == 0) atomicAdd(ssim_sum, (double)(red[0] + red[1] + red[2] + red[3]));
}

__global__ void __launch_bounds__(256) ssim_backward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                            const float* __restrict__ img2, const SsimWindow win,
                                                            const float* __restrict__ m_mu1, const float* __restrict__ m_s11,
                                                            const float* __restrict__ m_s12, const float* __restrict__ dL_dmean,
                                                            float* __restrict__ dL_dimg1) {
  __shared__ float t[3][SS_H][SS_H + 1];
  __shared__ float hb[3][SS_H][SS_T + 1];
  const int c = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const size_t plane = (size_t)c * H * W;
  for (int i = tid; i < SS_H * SS_H; i += 256) {
    const int r = i / SS_H, q = i - r * SS_H;
    const int gy = y0 + r - SS_R, gx = x0 + q - SS_R;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = plane + (size_t)gy * W + gx;
    t[0][r][q] = in ? m_mu1[o] : 0.f;
    t[1][r][q] = in ? m_s11[o] : 0.f;
    t[2][r][q] = in ? m_s12[o] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < SS_H * SS_T; i += 256) {
    const int r = i / SS_T, q = i - r * SS_T;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.g[k];
      a += w * t[0][r][q + k]; b += w * t[1][r][q + k]; d += w * t[2][r][q + k];
    }
    hb[0][r][q] = a; hb[1][r][q] = b; hb[2][r][q] = d;
  }
  __syncthreads();
  float A = 0.f, B = 0.f, D = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const float w = win.g[k];
    A += w * hb[0][ly + k][lx]; B += w * hb[1][ly + k][lx]; D += w * hb[2][ly + k][lx];
  }
  const int gx = x0 + lx, gy = y0 + ly;
  if (gx < W && gy < H) {
    const size_t o = plane + (size_t)gy * W + gx;
    const float scale = dL_dmean[0] / (float)((size_t)C * H * W);
    dL_dimg1[o] = scale * (A + 2.f * img1[o] * B + img2[o] * D);
  }
}

}  // namespace s3g

using namespace s3g;

extern "C" int s3g_ssim_forward(int C, int H, int W, const float* img1, const float* img2, double* ssim_sum, float* dm_dmu1,
                                float* dm_dsigma1_sq, float* dm_dsigma12, void* stream_) {
  if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_sum || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12) {
    set_error("s3g_ssim_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  static const SsimWindow win = make_window();
  dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
  hipLaunchKernelGGL(ssim_forward_kernel, grid, dim3(256), 0, (hipStream_t)stream_, C, H, W, img1, img2, win, ssim_sum,
                     dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                                 const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dmean,
                                 float* dL_dimg1, void* stream_) {
  if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dmean || !dL_dimg1) {
    set_error("s3g_ssim_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  static const SsimWindow win = make_window();
  dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
  hipLaunchKernelGGL(ssim_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream_, C, H, W, img1, img2, win, dm_dmu1,
                     dm_dsigma1_sq, dm_dsigma12, dL_dmean, dL_dimg1);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

// =========================================================================================================
// Fused HexPlane regulariser: value + gradient of scene/gaussian_model.py:710-749 in one pass over the planes.
// =========================================================================================================
namespace s3g {

struct PlaneRegArgs {
  s3g_plane_reg_desc pl[S3G_MAX_REG_PLANES];
  int first_block[S3G_MAX_REG_PLANES + 1];
  int nplanes;
  double* value;
};
constexpr int PR_ROWS = 16;  // rows of one (w, c) column handled per thread

__global__ void __launch_bounds__(256) plane_reg_kernel(const PlaneRegArgs a) {
  __shared__ float red[4];
  int pi = 0;
  while (pi + 1 < a.nplanes && (int)blockIdx.x >= a.first_block[pi + 1]) pi++;
  const s3g_plane_reg_desc d = a.pl[pi];
  const int cols = d.W * 32;                       // (w, c) columns, contiguous in memory
  const int col_blocks = (cols + 255) / 256;
  const int b = blockIdx.x - a.first_block[pi];
  const int col = (b % col_blocks) * 256 + threadIdx.x;
  const int h0 = (b / col_blocks) * PR_ROWS;
  float local = 0.f;
  if (col < cols) {
    const int H = d.H;
    const float cs = H > 2 ? d.w_smooth / ((float)(H - 2) * (float)cols) : 0.f;  // mean over C*(H-2)*W
    const float cl = d.w_l1 / ((float)H * (float)cols);
    const float* p = d.plane + col;
    auto at = [&](int h) { return (h >= 0 && h < H) ? p[(size_t)h * cols] : 0.f; };
    // second differences d2[j] = p[j+2] - 2 p[j+1] + p[j], valid for 0 <= j <= H-3
    auto d2 = [&](int j, float pj, float pj1, float pj2) { return (j >= 0 && j <= H - 3) ? (pj2 - 2.f * pj1 + pj) : 0.f; };
    float w[5];  // p[h-2 .. h+2]
    w[0] = at(h0 - 2); w[1] = at(h0 - 1); w[2] = at(h0); w[3] = at(h0 + 1); w[4] = at(h0 + 2);
    for (int h = h0; h < min(h0 + PR_ROWS, H); h++) {
      const float dm2 = d2(h - 2, w[0], w[1], w[2]), dm1 = d2(h - 1, w[1], w[2], w[3]), d0 = d2(h, w[2], w[3], w[4]);
      local += cs * d0 * d0;                       // each d2[h] is owned by row h
      float g = 2.f * cs * (dm2 - 2.f * dm1 + d0);
      if (d.w_l1 != 0.f) {
        const float x = 1.f - w[2];
        local += cl * fabsf(x);
        g += cl * (x > 0.f ? -1.f : (x < 0.f ? 1.f : 0.f));
      }
      d.grad[(size_t)h * cols + col] = g;
      w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = at(h + 3);
    }
  }
  for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(a.value, (double)(red[0] + red[1] + red[2] + red[3]));
}

}  // namespace s3g

extern "C" int s3g_plane_regulation(int nplanes, const s3g_plane_reg_desc* planes, double* value, void* stream_) {
  using namespace s3g;
  if (nplanes < 0 || nplanes > S3G_MAX_REG_PLANES || (nplanes > 0 && (!planes || !value))) {
    set_error("s3g_plane_regulation: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (nplanes == 0) return S3G_OK;
  PlaneRegArgs a;
  a.nplanes = nplanes;
  a.value = value;
  int blocks = 0;
  for (int i = 0; i < nplanes; i++) {
    if (!planes[i].plane || !planes[i].grad || planes[i].H < 1 || planes[i].W < 1) {
      set_error("s3g_plane_regulation: bad plane descriptor %d", i);
      return S3G_ERR_INVALID_ARG;
    }
    a.pl[i] = planes[i];
    a.first_block[i] = blocks;
    blocks += ((planes[i].W * 32 + 255) / 256) * ((planes[i].H + PR_ROWS - 1) / PR_ROWS);
  }
  a.first_block[nplanes] = blocks;
  hipLaunchKernelGGL(plane_reg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
