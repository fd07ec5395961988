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
  if (tid == 0) {
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    atomicAdd(&ssim_sum[(b % S3G_SUM_SLOTS) * S3G_SUM_STRIDE], (double)(red[0] + red[1] + red[2] + red[3]));
  }
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

namespace s3g {
// =========================================================================================================
// Per-pixel photometric terms of train.py:395-425 in one pass each way:
//   l1_loss(image, gt)            utils/loss_utils.py:50-51    mean |image - gt|
//   compute_depth("l2", pred, gt) utils/loss_utils.py:21-45    gt in (0.01, max_depth) selects pixels, both sides are
//                                                              clamp(x / max_depth, 0, 1), mean squared error
//   l2_loss(feat, gt_feat)        utils/loss_utils.py:53-54    mean (feat - gt)^2
// The reference spends ~45 launches here, among them a nonzero + gather for the boolean mask and a radix sort inside
// index_put's backward.
// =========================================================================================================
struct PixelLossArgs {
  int HW;
  const float *image, *gt_image, *depth, *gt_depth, *feat, *gt_feat;  // [3,HW] [3,HW] [HW] [HW] [3,HW] [3,HW]; pairs may be NULL
  float max_depth;
  double* sums;           // forward: 5 slotted accumulators ([1] l1 [2] depth sq. error [3] depth count [4] feat sq. error);
                          // backward: the 5 collapsed totals
  // backward
  const float* g;         // upstream gradient of the combined loss (device scalar)
  float w_l1, w_depth, w_feat;
  float *g_image, *g_depth, *g_feat;
  int accumulate_image;   // g_image already holds the SSIM gradient
};

__device__ __forceinline__ double block_sum(double v, double* part) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) pixel_loss_forward_kernel(const PixelLossArgs a) {
  __shared__ double part[4];
  float l1 = 0.f, dsq = 0.f, cnt = 0.f, fsq = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.HW; i += gridDim.x * 256) {
    if (a.image != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; c++) l1 += fabsf(a.image[(size_t)c * a.HW + i] - a.gt_image[(size_t)c * a.HW + i]);
    }
    if (a.depth != nullptr) {
      const float gd = a.gt_depth[i];
      if (gd > 0.01f && gd < a.max_depth) {
        const float cp = fminf(fmaxf(a.depth[i] / a.max_depth, 0.f), 1.f), cg = fminf(fmaxf(gd / a.max_depth, 0.f), 1.f);
        dsq += (cp - cg) * (cp - cg);
        cnt += 1.f;
      }
    }
    if (a.feat != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float d = a.feat[(size_t)c * a.HW + i] - a.gt_feat[(size_t)c * a.HW + i];
        fsq += d * d;
      }
    }
  }
  const double s1 = block_sum((double)l1, part), s2 = block_sum((double)dsq, part), s3 = block_sum((double)cnt, part),
               s4 = block_sum((double)fsq, part);
  if (threadIdx.x == 0) {
    double* slot = a.sums + (blockIdx.x % S3G_SUM_SLOTS) * S3G_SUM_STRIDE;
    if (a.image != nullptr) atomicAdd(&slot[1 * S3G_SUM_DOUBLES], s1);
    if (a.depth != nullptr) { atomicAdd(&slot[2 * S3G_SUM_DOUBLES], s2); atomicAdd(&slot[3 * S3G_SUM_DOUBLES], s3); }
    if (a.feat != nullptr) atomicAdd(&slot[4 * S3G_SUM_DOUBLES], s4);
  }
}

__global__ void __launch_bounds__(256) pixel_loss_backward_kernel(const PixelLossArgs a) {
  const float g = *a.g;
  const float k_l1 = g * a.w_l1 / (3.0f * (float)a.HW), k_feat = g * a.w_feat * 2.0f / (3.0f * (float)a.HW);
  // empty mask: the reference's mean over zero elements is NaN and so is its gradient; 0/0 reproduces that
  const float k_depth = a.g_depth != nullptr ? g * a.w_depth * 2.0f / ((float)a.sums[3] * a.max_depth) : 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.HW; i += gridDim.x * 256) {
    if (a.g_image != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float d = a.image[(size_t)c * a.HW + i] - a.gt_image[(size_t)c * a.HW + i];
        const float v = d > 0.f ? k_l1 : (d < 0.f ? -k_l1 : 0.f);
        float* dst = &a.g_image[(size_t)c * a.HW + i];
        *dst = a.accumulate_image ? *dst + v : v;
      }
    }
    if (a.g_depth != nullptr) {
      const float gd = a.gt_depth[i], x = a.depth[i] / a.max_depth;
      float v = 0.f;
      if (gd > 0.01f && gd < a.max_depth && x >= 0.f && x <= 1.f)  // clamp passes the gradient on its closed interval
        v = k_depth * (x - fminf(fmaxf(gd / a.max_depth, 0.f), 1.f));
      a.g_depth[i] = v;
    }
    if (a.g_feat != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        a.g_feat[(size_t)c * a.HW + i] = k_feat * (a.feat[(size_t)c * a.HW + i] - a.gt_feat[(size_t)c * a.HW + i]);
    }
  }
}

// totals[q] = sum of accumulator q; loss = w_l1 * T1 / N + w_depth * T2 / T3 + w_ssim * (1 - T0 / N) + w_feat * T4 / N  (N = 3 HW)
__global__ void __launch_bounds__(64) pixel_loss_combine_kernel(const double* __restrict__ sums, double* __restrict__ totals,
                                                                int HW, float w_l1, float w_depth, float w_ssim, float w_feat,
                                                                float* __restrict__ loss) {
  double T[5];
#pragma unroll
  for (int q = 0; q < 5; q++) {
    double v = sums[(size_t)q * S3G_SUM_DOUBLES + threadIdx.x * S3G_SUM_STRIDE];   // S3G_SUM_SLOTS == 64 == one wave
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    T[q] = v;
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int q = 0; q < 5; q++) totals[q] = T[q];
  const double N = 3.0 * (double)HW;
  double v = 0.0;
  if (w_l1 != 0.f) v += (double)w_l1 * T[1] / N;
  if (w_depth != 0.f) v += (double)w_depth * T[2] / T[3];
  if (w_ssim != 0.f) v += (double)w_ssim * (1.0 - T[0] / N);
  if (w_feat != 0.f) v += (double)w_feat * T[4] / N;
  *loss = (float)v;
}

// =========================================================================================================
// Fused HexPlane regulariser: value + gradient of scene/gaussian_model.py:710-749 in one pass over the planes.
// =========================================================================================================
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
  if (threadIdx.x == 0)
    atomicAdd(&a.value[(blockIdx.x % S3G_SUM_SLOTS) * S3G_SUM_STRIDE], (double)(red[0] + red[1] + red[2] + red[3]));
}

}  // namespace s3g

// x[i] *= *scale unless *scale == 1 (read on the device): x * 1.0f is x bit for bit, so the common case -- a loss that is
// back-propagated with the implicit seed of ones through a unit weight -- costs a launch instead of a read-modify-write of the array.
__global__ void __launch_bounds__(256) scale_unless_one_kernel(float* __restrict__ x, size_t n4, size_t n, const float* __restrict__ scale) {
  const float s = *scale;
  if (s == 1.0f) return;   // uniform
  float4* x4 = reinterpret_cast<float4*>(x);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = x4[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    x4[i] = v;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= s;
}

extern "C" int s3g_scale_unless_one(float* x, size_t n, const float* scale, void* stream_) {
  if (n > 0 && (!x || !scale)) {
    set_error("s3g_scale_unless_one: NULL argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (n == 0) return S3G_OK;
  const size_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  const size_t want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(scale_unless_one_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, n4, n, scale);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

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

extern "C" int s3g_pixel_losses_forward(int H, int W, const float* image, const float* gt_image, const float* depth,
                                        const float* gt_depth, const float* feat, const float* gt_feat, float max_depth,
                                        double* sums, void* stream_) {
  if (H <= 0 || W <= 0 || !sums || (image && !gt_image) || (depth && !gt_depth) || (feat && !gt_feat)) {
    set_error("s3g_pixel_losses_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  PixelLossArgs a;
  memset(&a, 0, sizeof a);
  a.HW = H * W; a.image = image; a.gt_image = gt_image; a.depth = depth; a.gt_depth = gt_depth; a.feat = feat;
  a.gt_feat = gt_feat; a.max_depth = max_depth; a.sums = sums;
  const int blocks = min((a.HW + 255) / 256, 1024);
  hipLaunchKernelGGL(pixel_loss_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_pixel_losses_combine(int H, int W, const double* sums, double* totals, float w_l1, float w_depth,
                                        float w_ssim, float w_feat, float* loss, void* stream_) {
  static_assert(S3G_SUM_SLOTS == 64, "the combine kernel reduces the slots with one wave");
  if (H <= 0 || W <= 0 || !sums || !totals || !loss) {
    set_error("s3g_pixel_losses_combine: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(pixel_loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, sums, totals, H * W, w_l1, w_depth,
                     w_ssim, w_feat, loss);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_pixel_losses_backward(int H, int W, const float* image, const float* gt_image, const float* depth,
                                         const float* gt_depth, const float* feat, const float* gt_feat, float max_depth,
                                         const double* sums, const float* g, float w_l1, float w_depth, float w_feat,
                                         float* g_image, int accumulate_image, float* g_depth, float* g_feat, void* stream_) {
  if (H <= 0 || W <= 0 || !sums || !g || (g_image && (!image || !gt_image)) || (g_depth && (!depth || !gt_depth)) ||
      (g_feat && (!feat || !gt_feat))) {
    set_error("s3g_pixel_losses_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  PixelLossArgs a;
  memset(&a, 0, sizeof a);
  a.HW = H * W; a.image = image; a.gt_image = gt_image; a.depth = depth; a.gt_depth = gt_depth; a.feat = feat;
  a.gt_feat = gt_feat; a.max_depth = max_depth; a.sums = const_cast<double*>(sums); a.g = g;
  a.w_l1 = w_l1; a.w_depth = w_depth; a.w_feat = w_feat;
  a.g_image = g_image; a.accumulate_image = accumulate_image; a.g_depth = g_depth; a.g_feat = g_feat;
  const int blocks = min((a.HW + 255) / 256, 4096);
  hipLaunchKernelGGL(pixel_loss_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
