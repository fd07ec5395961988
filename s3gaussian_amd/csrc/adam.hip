// Adam step for all parameters of the model in one launch (include/s3g_optim.h).  HBM-bound: 16 B read + 12 B written
// per element (p, g, m, v -> p, m, v); 106 M parameters at cfg3 = 2.98 GB per step.
#include "common.hpp"

#include "../../include/s3g_optim.h"

namespace s3g {

typedef float f4v __attribute__((ext_vector_type(4)));
constexpr bool ADAM_NONTEMPORAL = true;   // streaming loads / stores: 0.605 -> 0.534 ms (every element is touched exactly once per step)

struct AdamArgs {
  s3g_adam_tensor t[S3G_ADAM_MAX_TENSORS];
  float beta1, beta2, w1, w2;  // w_k = 1 - beta_k rounded from DOUBLE, like torch's python-side `1 - beta`
  const uint32_t* skip;        // optional device word: != 0 -> the whole step is a no-op (s3g_adam_step_guarded)
};

// grid = (blocks per tensor, tensors): a tensor is swept by its own row of workgroups with 16-byte accesses
__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a) {
  if (a.skip != nullptr && *a.skip != 0u) return;   // uniform: one scalar load
  const s3g_adam_tensor& t = a.t[blockIdx.y];
  const float b2 = a.beta2, w1 = a.w1, w2 = a.w2;
  const float step_size = t.step_size, isb = t.inv_sqrt_bc2, eps = t.eps, gs = t.grad_scale;
  // 16-byte accesses when all four arrays allow it (gradients that are views into a flat buffer may start anywhere)
  const bool vec = ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0);
  const size_t n4 = vec ? t.numel / 4 : 0;
  float4* p4 = reinterpret_cast<float4*>(t.param);
  const float4* g4 = reinterpret_cast<const float4*>(t.grad);
  float4* m4 = reinterpret_cast<float4*>(t.exp_avg);
  float4* v4 = reinterpret_cast<float4*>(t.exp_avg_sq);
  auto upd = [&](float& p, float g, float& m, float& v) {
    g = g * gs;
    m = m + (g - m) * w1;                      // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = b2 * v + w2 * g * g;                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * isb + eps;  // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);           // param.addcdiv_(exp_avg, denom, value = -step_size)
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 p, m, v, g;
    if (ADAM_NONTEMPORAL) {
      auto ld = [](const float4* q) { const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q)); return make_float4(x.x, x.y, x.z, x.w); };
      p = ld(p4 + i); m = ld(m4 + i); v = ld(v4 + i); g = ld(g4 + i);
    } else {
      p = p4[i]; m = m4[i]; v = v4[i]; g = g4[i];
    }
    upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
    if (ADAM_NONTEMPORAL) {
      auto st = [](float4* q, float4 x) { f4v y = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(y, reinterpret_cast<f4v*>(q)); };
      st(p4 + i, p); st(m4 + i, m); st(v4 + i, v);
    } else {
      p4[i] = p; m4[i] = m; v4[i] = v;
    }
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < t.numel; i += (size_t)gridDim.x * 256)
    upd(t.param[i], t.grad[i], t.exp_avg[i], t.exp_avg_sq[i]);  // the numel % 4 tail, or everything when unaligned
}

__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const float* __restrict__ g, int stride,
                                                            const int* __restrict__ radii, const unsigned char* __restrict__ visible,
                                                            float* __restrict__ accum, float* __restrict__ denom,
                                                            float* __restrict__ max_radii, const uint32_t* __restrict__ skip) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  if (skip && *skip != 0u) return;   // the view's asynchronous forward overflowed: its statistics do not exist (uniform load)
  const int r = radii[i];
  if (!(visible ? visible[i] != 0 : r > 0)) return;
  const float gx = g[(size_t)i * stride], gy = g[(size_t)i * stride + 1];
  accum[i] += sqrtf(gx * gx + gy * gy);
  denom[i] += 1.f;
  max_radii[i] = fmaxf(max_radii[i], (float)r);
}

}  // namespace s3g

using namespace s3g;

static int densify_stats_impl(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                              float* xyz_gradient_accum, float* denom, float* max_radii2D, const uint32_t* skip, void* stream_);

extern "C" int s3g_densify_stats(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                                 float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream_) {
  return densify_stats_impl(P, grad_xy, grad_stride, radii, visible, xyz_gradient_accum, denom, max_radii2D, nullptr, stream_);
}
extern "C" int s3g_densify_stats_guarded(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                                         float* xyz_gradient_accum, float* denom, float* max_radii2D, const uint32_t* skip_flag,
                                         void* stream_) {
  return densify_stats_impl(P, grad_xy, grad_stride, radii, visible, xyz_gradient_accum, denom, max_radii2D, skip_flag, stream_);
}

static int densify_stats_impl(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                              float* xyz_gradient_accum, float* denom, float* max_radii2D, const uint32_t* skip, void* stream_) {
  if (P < 0 || grad_stride < 2 || (P > 0 && (!grad_xy || !radii || !xyz_gradient_accum || !denom || !max_radii2D))) {
    set_error("s3g_densify_stats: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, grad_xy, grad_stride,
                     radii, visible, xyz_gradient_accum, denom, max_radii2D, skip);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

static int adam_step_impl(int n, const s3g_adam_tensor* tensors, double beta1, double beta2, const uint32_t* skip, void* stream_);

extern "C" int s3g_adam_step(int n, const s3g_adam_tensor* tensors, double beta1, double beta2, void* stream_) {
  return adam_step_impl(n, tensors, beta1, beta2, nullptr, stream_);
}
extern "C" int s3g_adam_step_guarded(int n, const s3g_adam_tensor* tensors, double beta1, double beta2,
                                     const uint32_t* skip_flag, void* stream_) {
  return adam_step_impl(n, tensors, beta1, beta2, skip_flag, stream_);
}

static int adam_step_impl(int n, const s3g_adam_tensor* tensors, double beta1, double beta2, const uint32_t* skip, void* stream_) {
  if (n < 0 || n > S3G_ADAM_MAX_TENSORS || (n > 0 && !tensors)) {
    set_error("s3g_adam_step: bad argument (at most %d tensors per call)", S3G_ADAM_MAX_TENSORS);
    return S3G_ERR_INVALID_ARG;
  }
  if (n == 0) return S3G_OK;
  AdamArgs a;
  memset(&a, 0, sizeof a);
  size_t largest = 0, total = 0;
  for (int k = 0; k < n; k++) {
    const s3g_adam_tensor& t = tensors[k];
    if (t.numel > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) {
      set_error("s3g_adam_step: NULL array in tensor %d", k);
      return S3G_ERR_INVALID_ARG;
    }
    a.t[k] = t;
    largest = t.numel > largest ? t.numel : largest;
    total += t.numel;
  }
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2);
  a.skip = skip;
  const size_t want = (largest / 4 + 255) / 256;
  const int bx = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  profile_begin(S3G_PROFILE_ADAM, (hipStream_t)stream_);
  hipLaunchKernelGGL(adam_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream_, a);
  profile_end(S3G_PROFILE_ADAM, (hipStream_t)stream_, (double)total, 0.0);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
