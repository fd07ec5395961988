// MI355X-native multi-resolution HexPlane sampler: forward and backward, one fused pass each.
//
// Reference: scene/hexplane.py:73-106 runs 4 levels x 6 planes = 24 F.grid_sample launches, each materialising a
// [P,32] tensor, then 20 elementwise products and a concat; autograd replays the same 24 in backward.
//
// Here 32 consecutive lanes own the 32 channels of ONE point (a wave64 handles two points), so with the planes stored
// channel-last every texel fetch is one coalesced 128-byte line and the product over planes never leaves registers.
// Arithmetic follows torch's grid_sampler_2d (bilinear, border, align_corners=True) op for op; contraction is off.
#include "common.hpp"

#include "../../include/s3g_hexplane.h"

namespace s3g {

constexpr int HEXC = S3G_HEX_CHANNELS;

struct HexArgs {
  s3g_hexplane_desc d;
  float* gplanes[S3G_HEX_MAX_LEVELS][6];
  int P;
  const float* xyz;
  const float* time;
  const float* gfeat;
  float* feat;
  float* gxyz;
};

struct Tap {         // one bilinear footprint
  int o00, o01, o10, o11;  // texel offsets (in texels) of nw, ne, sw, se; -1 when out of bounds (weight is 0 then)
  float w00, w01, w10, w11;
  float x0f, x1f, y0f, y1f, ix, iy;
  float mx, my;      // d(ix)/d(u) incl. the border-clip gradient mask
};

// torch grid_sampler_unnormalize (align_corners) + clip_coordinates(_set_grad) + bilinear weights
__device__ __forceinline__ Tap make_tap(float ux, float uy, int W, int H) {
  Tap t;
  float ix = ((ux + 1.f) / 2.f) * (float)(W - 1);
  float iy = ((uy + 1.f) / 2.f) * (float)(H - 1);
  t.mx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : (float)(W - 1) / 2.f;
  t.my = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : (float)(H - 1) / 2.f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float x0 = floorf(ix), y0 = floorf(iy);
  const float x1 = x0 + 1.f, y1 = y0 + 1.f;
  t.ix = ix; t.iy = iy; t.x0f = x0; t.x1f = x1; t.y0f = y0; t.y1f = y1;
  t.w00 = (x1 - ix) * (y1 - iy);
  t.w01 = (ix - x0) * (y1 - iy);
  t.w10 = (x1 - ix) * (iy - y0);
  t.w11 = (ix - x0) * (iy - y0);
  const int xi0 = (int)x0, yi0 = (int)y0, xi1 = xi0 + 1, yi1 = yi0 + 1;
  const bool bx1 = xi1 < W, by1 = yi1 < H;  // xi0, yi0 are always in range after the clip
  t.o00 = yi0 * W + xi0;
  t.o01 = bx1 ? yi0 * W + xi1 : -1;
  t.o10 = by1 ? yi1 * W + xi0 : -1;
  t.o11 = (bx1 && by1) ? yi1 * W + xi1 : -1;
  return t;
}

__device__ __forceinline__ float fetch(const float* __restrict__ plane, int off, int c) {
  return off >= 0 ? plane[(size_t)off * HEXC + c] : 0.f;
}

__device__ __forceinline__ void point_coords(const HexArgs& a, int p, float* u) {
#pragma unroll
  for (int k = 0; k < 3; k++)
    u[k] = (a.xyz[3 * (size_t)p + k] - a.d.aabb_max[k]) * (2.0f / (a.d.aabb_min[k] - a.d.aabb_max[k])) - 1.0f;
  u[3] = a.time[p];
}

// coordinate pairs in itertools.combinations(range(4), 2) order
__device__ constexpr int PAIR0[6] = {0, 0, 0, 1, 1, 2};
__device__ constexpr int PAIR1[6] = {1, 2, 3, 2, 3, 3};

__global__ void __launch_bounds__(256) hexplane_forward_kernel(const HexArgs a) {
  const int c = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int F = a.d.levels * HEXC;
  for (int p = blockIdx.x * 8 + slot; p < a.P; p += gridDim.x * 8) {
    float u[4];
    point_coords(a, p, u);
    for (int l = 0; l < a.d.levels; l++) {
      float prod = 1.f;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
        const float* pl = a.d.planes[l][i];
        float s = fetch(pl, t.o00, c) * t.w00;
        s += fetch(pl, t.o01, c) * t.w01;
        s += fetch(pl, t.o10, c) * t.w10;
        s += fetch(pl, t.o11, c) * t.w11;
        prod = prod * s;
      }
      a.feat[(size_t)p * F + l * HEXC + c] = prod;
    }
  }
}

__global__ void __launch_bounds__(256) hexplane_backward_kernel(const HexArgs a) {
  const int c = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int F = a.d.levels * HEXC;
  for (int p0 = blockIdx.x * 8; p0 < a.P; p0 += gridDim.x * 8) {  // uniform trip count: shuffles below need all lanes
    const int p = p0 + slot;
    const bool live = p < a.P;
    float u[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) point_coords(a, p, u);
    float du[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < a.d.levels; l++) {
      Tap t[6];
      float v00[6], v01[6], v10[6], v11[6], s[6];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        t[i] = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
        const float* pl = a.d.planes[l][i];
        v00[i] = live ? fetch(pl, t[i].o00, c) : 0.f;
        v01[i] = live ? fetch(pl, t[i].o01, c) : 0.f;
        v10[i] = live ? fetch(pl, t[i].o10, c) : 0.f;
        v11[i] = live ? fetch(pl, t[i].o11, c) : 0.f;
        float acc = v00[i] * t[i].w00;
        acc += v01[i] * t[i].w01;
        acc += v10[i] * t[i].w10;
        acc += v11[i] * t[i].w11;
        s[i] = acc;
      }
      const float g = live ? a.gfeat[(size_t)p * F + l * HEXC + c] : 0.f;
      // product rule in the order autograd applies it to ((((1*s0)*s1)*s2)*s3)*s4)*s5: pre[i] = prod_{j<i} s_j, suffix by recursion
      float pre[6];
      pre[0] = 1.f;
#pragma unroll
      for (int i = 1; i < 6; i++) pre[i] = pre[i - 1] * s[i - 1];
      float gs = g;  // dL/d(prefix product through plane i)
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        const float gi = gs * pre[i];  // dL/ds_i
        gs = gs * s[i];
        if (live) {
          float* gp = a.gplanes[l][i];
          if (gp != nullptr) {
            if (t[i].o00 >= 0) atomicAdd(&gp[(size_t)t[i].o00 * HEXC + c], gi * t[i].w00);
            if (t[i].o01 >= 0) atomicAdd(&gp[(size_t)t[i].o01 * HEXC + c], gi * t[i].w01);
            if (t[i].o10 >= 0) atomicAdd(&gp[(size_t)t[i].o10 * HEXC + c], gi * t[i].w10);
            if (t[i].o11 >= 0) atomicAdd(&gp[(size_t)t[i].o11 * HEXC + c], gi * t[i].w11);
          }
          // torch grid_sampler_2d_backward: gix = -nw*(iy_se-iy) + ne*(iy_sw-iy) - sw*(iy-iy_ne) + se*(iy-iy_nw), ...
          const float gix = (-v00[i] * (t[i].y1f - t[i].iy) + v01[i] * (t[i].y1f - t[i].iy) - v10[i] * (t[i].iy - t[i].y0f) +
                             v11[i] * (t[i].iy - t[i].y0f)) * gi;
          const float giy = (-v00[i] * (t[i].x1f - t[i].ix) - v01[i] * (t[i].ix - t[i].x0f) + v10[i] * (t[i].x1f - t[i].ix) +
                             v11[i] * (t[i].ix - t[i].x0f)) * gi;
          if (PAIR0[i] < 3) du[PAIR0[i]] += t[i].mx * gix;
          if (PAIR1[i] < 3) du[PAIR1[i]] += t[i].my * giy;
        }
      }
    }
    // sum over the 32 channels (lanes of this half-wave), then undo the aabb normalisation
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = du[k];
      for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      du[k] = v;
    }
    if (live && c < 3) a.gxyz[3 * (size_t)p + c] = du[c] * (2.0f / (a.d.aabb_min[c] - a.d.aabb_max[c]));
  }
}

static int check_desc(const s3g_hexplane_desc* d) {
  if (!d || d->levels < 1 || d->levels > S3G_HEX_MAX_LEVELS) {
    set_error("hexplane: bad descriptor (levels)");
    return S3G_ERR_INVALID_ARG;
  }
  for (int l = 0; l < d->levels; l++) {
    for (int k = 0; k < 4; k++)
      if (d->res[l][k] < 2) {
        set_error("hexplane: resolution must be >= 2");
        return S3G_ERR_INVALID_ARG;
      }
    for (int i = 0; i < 6; i++)
      if (!d->planes[l][i]) {
        set_error("hexplane: NULL plane pointer");
        return S3G_ERR_INVALID_ARG;
      }
  }
  return S3G_OK;
}

}  // namespace s3g

using namespace s3g;

extern "C" int s3g_hexplane_forward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                    float* features, void* stream_) {
  if (int e = check_desc(d)) return e;
  if (P < 0 || (P > 0 && (!xyz || !time || !features))) {
    set_error("s3g_hexplane_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  HexArgs a;
  memset(&a, 0, sizeof a);
  a.d = *d; a.P = P; a.xyz = xyz; a.time = time; a.feat = features;
  const int blocks = min((P + 7) / 8, 256 * 16);
  hipLaunchKernelGGL(hexplane_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_hexplane_backward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                     const float* dL_dfeatures, float* dL_dxyz,
                                     float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* stream_) {
  if (int e = check_desc(d)) return e;
  if (P < 0 || (P > 0 && (!xyz || !time || !dL_dfeatures || !dL_dxyz || !dL_dplanes))) {
    set_error("s3g_hexplane_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  HexArgs a;
  memset(&a, 0, sizeof a);
  a.d = *d; a.P = P; a.xyz = xyz; a.time = time; a.gfeat = dL_dfeatures; a.gxyz = dL_dxyz;
  for (int l = 0; l < d->levels; l++)
    for (int i = 0; i < 6; i++) a.gplanes[l][i] = dL_dplanes[l][i];
  const int blocks = min((P + 7) / 8, 256 * 16);
  hipLaunchKernelGGL(hexplane_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
