/*
 * s3g_loss.h -- C ABI of the fused SSIM loss kernels (libs3g.so).
 *
 *   s3g_ssim_forward / s3g_ssim_backward  <- ssim() / _ssim()  /root/reference/utils/loss_utils.py:66-96
 *       (create_window :60-64: 11x11 window = outer product of a sigma=1.5 Gaussian; five grouped F.conv2d with
 *        zero padding 5; C1 = 0.01^2, C2 = 0.03^2; mean over all channels and pixels) and its autograd.
 *
 * The window is separable, so each statistic is one horizontal + one vertical 11-tap pass through LDS; forward keeps
 * three per-pixel partial-derivative maps so backward is three more separable passes instead of replaying five
 * convolutions and ~25 elementwise kernels.  Only img1 (the rendered image) receives a gradient, like the reference's
 * call site train.py:416-418 (the ground truth needs none).
 */
#ifndef S3G_LOSS_H
#define S3G_LOSS_H
#ifdef __cplusplus
extern "C" {
#endif

/* Slotted sum accumulators.  Measured on MI355X: thousands of workgroups finishing with an atomicAdd on ONE address
 * serialise at ~10 ns each (20 100 SSIM tiles = 0.2 ms, more than the kernel's own work).  Every scalar these kernels
 * accumulate is therefore a device array of S3G_SUM_DOUBLES doubles: workgroup b adds into element
 * (b % S3G_SUM_SLOTS) * S3G_SUM_STRIDE (one 128-byte line per slot), all other elements stay as the caller left them.
 * The caller zero-fills the array; the value is the sum over the whole array. */
#ifndef S3G_SUM_SLOTS
#define S3G_SUM_SLOTS 64
#define S3G_SUM_STRIDE 16
#define S3G_SUM_DOUBLES (S3G_SUM_SLOTS * S3G_SUM_STRIDE)
#endif

/* img1, img2: [C,H,W] fp32 device.  ssim_sum: slotted accumulator (S3G_SUM_DOUBLES doubles, see above): sum of the SSIM
 * map, the loss value is sum(ssim_sum) / (C*H*W).  dm_dmu1, dm_dsigma1_sq, dm_dsigma12: [C,H,W] scratch kept for the backward. */
int s3g_ssim_forward(int C, int H, int W, const float* img1, const float* img2, double* ssim_sum, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* dL_dimg1[C,H,W] = (*dL_dmean / (C*H*W)) * d(sum of SSIM map)/d img1.  dL_dmean: device float (upstream gradient of
 * the mean SSIM). */
int s3g_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dmean, float* dL_dimg1,
                      void* stream);

/*
 *   s3g_plane_regulation  <- GaussianModel.compute_regulation = _plane_regulation + _time_regulation + _l1_regulation
 *                            (/root/reference/scene/gaussian_model.py:710-749) over compute_plane_smoothness
 *                            (/root/reference/scene/regulation.py:22-28), forward value AND gradient in one pass.
 * The reference launches ~25 elementwise/reduction kernels per plane and direction (24 planes -> ~600 launches per
 * iteration, several full sweeps over the 143 MB of planes); here every plane element is read once.
 *
 * Per plane (logical [1,C,H,W], channel-last bytes [H][W][C], C = 32):
 *   value += w_smooth * mean_{c,h<H-2,w} (p[h+2]-2p[h+1]+p[h])^2  +  w_l1 * mean |1 - p|
 *   grad   = d value / d p      (written, not accumulated)
 */
typedef struct s3g_plane_reg_desc {
  const float* plane;   /* device, channel-last */
  float* grad;          /* device, same layout, written */
  int H, W;
  float w_smooth, w_l1; /* plane_tv_weight or time_smoothness_weight ; l1_time_planes or 0 */
} s3g_plane_reg_desc;

#define S3G_MAX_REG_PLANES 48
/* value: slotted accumulator (S3G_SUM_DOUBLES doubles). */
int s3g_plane_regulation(int nplanes, const s3g_plane_reg_desc* planes, double* value, void* stream);

/* x[0..n) *= *scale (device fp32 scalar) -- unless *scale == 1.0f, which the kernel finds out on the device and then touches
 * nothing.  The plane regulariser's gradient (143 MB at the reference's resolutions) is written with the forward and has to be
 * multiplied by the upstream gradient of its scalar in the backward (autograd's `grad * upstream`): in a training step that
 * upstream is the implicit seed of ones times a unit weight, known only on the device; multiplying by it anyway was a 50 us
 * read-modify-write per iteration. */
int s3g_scale_unless_one(float* x, size_t n, const float* scale /* device */, void* stream);

/*
 *   s3g_pixel_losses_*  <- the per-pixel terms of the training loss, /root/reference/train.py:395-425:
 *       l1_loss(image, gt[:3])                 utils/loss_utils.py:50-51
 *       compute_depth("l2", depth, gt_depth)   utils/loss_utils.py:21-45 (normalize_depth :21-22, mask :32)
 *       l2_loss(feat, gt_feat)                 utils/loss_utils.py:53-54
 *   one read of the images forward, one read + one write backward (the reference: ~45 launches including a nonzero /
 *   gather for the mask and the radix sort inside index_put's backward).
 *
 * Images are [3,H,W], depths [H,W], fp32 device; each (x, gt_x) pair may be NULL to skip that term.
 * sums: FIVE slotted accumulators back to back (5 * S3G_SUM_DOUBLES doubles, caller zeroes): [0] is what
 * s3g_ssim_forward's ssim_sum is pointed at, [1] sum|image-gt|, [2] masked squared depth error, [3] mask count,
 * [4] sum (feat-gt)^2. */
int s3g_pixel_losses_forward(int H, int W, const float* image, const float* gt_image, const float* depth,
                             const float* gt_depth, const float* feat, const float* gt_feat, float max_depth,
                             double* sums, void* stream);

/* Collapses the five accumulators into totals[5] (device doubles, written) and
 * loss[0] = w_l1 * T1/N + w_depth * T2/T3 + w_ssim * (1 - T0/N) + w_feat * T4/N, N = 3*H*W;
 * a zero weight skips its term (an empty depth mask gives NaN like the reference's mean over no elements). */
int s3g_pixel_losses_combine(int H, int W, const double* sums, double* totals, float w_l1, float w_depth, float w_ssim,
                             float w_feat, float* loss, void* stream);

/* Gradients of  w_l1*L1 + w_depth*depth_l2 + w_feat*feat_l2  times the upstream device scalar *g.  g_image is written,
 * or added to when accumulate_image != 0 (it then already holds the SSIM term from s3g_ssim_backward); g_depth and
 * g_feat are written; any output may be NULL.  totals: the five doubles s3g_pixel_losses_combine wrote. */
int s3g_pixel_losses_backward(int H, int W, const float* image, const float* gt_image, const float* depth,
                              const float* gt_depth, const float* feat, const float* gt_feat, float max_depth,
                              const double* totals, const float* g, float w_l1, float w_depth, float w_feat,
                              float* g_image, int accumulate_image, float* g_depth, float* g_feat, void* stream);

#ifdef __cplusplus
}
#endif
#endif
