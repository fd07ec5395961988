/*
 * s3g_optim.h -- C ABI of the fused Adam step (libs3g.so).
 *
 *   s3g_adam_step  <- optimizer.step() of the reference's torch.optim.Adam(l, lr=0.0, eps=1e-15)
 *                     (/root/reference/scene/gaussian_model.py:177-189, stepped at train.py:521-522): the update of
 *                     torch/optim/adam.py (no weight decay, no amsgrad, not maximising)
 *                        m  = m + (g - m) (1 - beta1)            v = beta2 v + (1 - beta2) g g
 *                        p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)         bc_k = 1 - beta_k^step
 *                     for EVERY parameter of every group in ONE launch (the reference: ~10 foreach passes per group over
 *                     426 MB of state; PyTorch's fused variant: one launch per group and 4 GB chunk).
 */
#ifndef S3G_OPTIM_H
#define S3G_OPTIM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S3G_ADAM_MAX_TENSORS 64 /* per call; the binding splits longer lists */

typedef struct s3g_adam_tensor {
  float* param;         /* device, updated in place */
  const float* grad;    /* device, same element order as param */
  float* exp_avg;       /* device, updated in place */
  float* exp_avg_sq;    /* device, updated in place */
  size_t numel;
  float step_size;      /* lr / (1 - beta1^step)  */
  float inv_sqrt_bc2;   /* 1 / sqrt(1 - beta2^step) */
  float eps;
  float grad_scale;     /* g = grad * grad_scale (1 for plain Adam; 1/world_size folds the data-parallel average in) */
} s3g_adam_tensor;

/* All tensors fp32.  param/grad/exp_avg/exp_avg_sq of one entry must share one memory layout (the update is
 * elementwise over raw storage, so channels_last planes need no special case).  The betas are doubles because torch forms
 * 1 - beta in double before rounding to fp32 (1 - fp32(0.999) differs from fp32(0.001) by 1.3e-5 relative). */
int s3g_adam_step(int n, const s3g_adam_tensor* tensors /* host array */, double beta1, double beta2, void* stream);

/* The same step behind a device-side guard: if *skip_flag (device uint32, read by the kernel when it runs) is non-zero the
 * launch changes nothing.  Pairs with s3g_raster_forward_async (s3g_raster.h): a forward whose instance count exceeded its
 * speculative arena renders nothing and back-propagates zeros; with its `status_device` word as skip_flag the optimizer step
 * of that iteration is dropped as well, so the model is exactly what it was before the iteration and the view can be rendered
 * again once the host has noticed -- without the host ever waiting for the device inside an iteration.  skip_flag == NULL:
 * identical to s3g_adam_step. */
int s3g_adam_step_guarded(int n, const s3g_adam_tensor* tensors /* host array */, double beta1, double beta2,
                          const unsigned int* skip_flag /* device */, void* stream);

/* Densification bookkeeping of one iteration in one pass over [P] (train.py:489-493 +
 * scene/gaussian_model.py:693-695; the reference: ~8 indexing / norm / max launches):
 *     where visible[i]:  xyz_gradient_accum[i] += sqrt(gx^2 + gy^2);  denom[i] += 1;
 *                        max_radii2D[i] = max(max_radii2D[i], radii[i])
 * grad_xy: [P, grad_stride] fp32 (the viewspace gradient, columns 0 and 1 used); visible: uint8 [P] or NULL (then
 * radii[i] > 0, the reference's visibility_filter).  This is the stand-alone form for gradients that were summed over
 * several backward calls or all-reduced over ranks; s3g_raster_backward*_accum (s3g_raster.h) fuses the same update into
 * the rasterizer's per-Gaussian backward when one call produces the whole viewspace gradient. */
int s3g_densify_stats(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream);
/* The same behind a device word (the `status_device` word of the asynchronous rasterizer forward that rendered the view, or the
 * all-reduced verdict of a data-parallel batch): skip_flag != NULL and *skip_flag != 0 at run time -> the launch changes nothing
 * (the view rendered nothing; its radii say "visible" but no gradient exists).  skip_flag == NULL: identical to s3g_densify_stats. */
int s3g_densify_stats_guarded(int P, const float* grad_xy, int grad_stride, const int* radii, const unsigned char* visible,
                              float* xyz_gradient_accum, float* denom, float* max_radii2D, const uint32_t* skip_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif
