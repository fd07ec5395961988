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

/* img1, img2: [C,H,W] fp32 device.  ssim_sum: device DOUBLE, ACCUMULATED (caller zeroes): sum of the SSIM map, the loss
 * value is ssim_sum / (C*H*W).  dm_dmu1, dm_dsigma1_sq, dm_dsigma12: [C,H,W] scratch kept for the backward. */
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
/* value: device DOUBLE, ACCUMULATED (caller zeroes). */
int s3g_plane_regulation(int nplanes, const s3g_plane_reg_desc* planes, double* value, void* stream);

#ifdef __cplusplus
}
#endif
#endif
