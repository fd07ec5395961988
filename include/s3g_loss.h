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

/* img1, img2: [C,H,W] fp32 device.  ssim_sum: device float, ACCUMULATED (caller zeroes): sum of the SSIM map, the loss
 * value is ssim_sum / (C*H*W).  dm_dmu1, dm_dsigma1_sq, dm_dsigma12: [C,H,W] scratch kept for the backward. */
int s3g_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_sum, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* dL_dimg1[C,H,W] = (*dL_dmean / (C*H*W)) * d(sum of SSIM map)/d img1.  dL_dmean: device float (upstream gradient of
 * the mean SSIM). */
int s3g_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dmean, float* dL_dimg1,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
