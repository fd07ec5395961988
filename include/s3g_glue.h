/*
 * s3g_glue.h -- C ABI of the fused per-Gaussian render glue (libs3g.so).
 *
 *   s3g_glue_forward / s3g_glue_backward  <- the PyTorch glue of render()
 *       /root/reference/gaussian_renderer/__init__.py:99-115 :
 *         scales   = exp(_scaling)                       (scaling_activation,  scene/gaussian_model.py:44)
 *         rotation = normalize(_rotation)                (rotation_activation = F.normalize, :53)
 *         opacity  = sigmoid(_opacity)                   (opacity_activation, :49)
 *         shs      = cat(f_dc, f_rest) + dshs            (get_features :120-124, deformation.py:157-162)
 *         colors   = clamp_min(eval_sh(deg, shs^T, normalize(xyz - campos)) + 0.5, 0)   (utils/sh_utils.py:57-112)
 *       (directions use the UN-deformed xyz, like the reference, gaussian_renderer/__init__.py:110)
 *   and the ~80 elementwise kernels autograd replays for them, as ONE kernel per direction.
 */
#ifndef S3G_GLUE_H
#define S3G_GLUE_H
#ifdef __cplusplus
extern "C" {
#endif

/* All device fp32.  f_dc [P,1,3], f_rest [P,15,3], dshs [P,16,3] or NULL, xyz [P,3], campos [3],
 * log_scales [P,3], rot_raw [P,4], opacity_logit [P]  ->  colors [P,3], scales [P,3], rot [P,4], opacity [P].
 * dshs_abs_sum (slotted accumulator of S3G_SUM_DOUBLES doubles as in s3g_loss.h, caller zeroes, may be NULL): += sum |dshs| -- the numerator of the reference's
 * lambda_dshs * mean|dshs| regulariser (/root/reference/train.py:407-410), taken while dshs streams through anyway. */
int s3g_glue_forward(int P, int deg, const float* f_dc, const float* f_rest, const float* dshs, const float* xyz,
                     const float* campos, const float* log_scales, const float* rot_raw, const float* opacity_logit,
                     float* colors, float* scales, float* rot, float* opacity, double* dshs_abs_sum, void* stream);

/* Upstream g_colors [P,3], g_scales [P,3], g_rot [P,4], g_opacity [P] (any may be NULL = zero) ->
 * g_f_dc [P,1,3], g_f_rest [P,15,3], g_dshs [P,16,3] (may be NULL), g_xyz [P,3], g_log_scales [P,3], g_rot_raw [P,4],
 * g_opacity_logit [P]; all written.  `colors`, `scales`, `rot`, `opacity` are the forward outputs.
 * g_dshs_l1 (device float, may be NULL): upstream gradient of mean|dshs|; g_dshs += *g_dshs_l1 / (48 P) * sign(dshs). */
int s3g_glue_backward(int P, int deg, const float* f_dc, const float* f_rest, const float* dshs, const float* xyz,
                      const float* campos, const float* rot_raw, const float* colors, const float* scales,
                      const float* rot, const float* opacity, const float* g_colors, const float* g_scales,
                      const float* g_rot, const float* g_opacity, float* g_f_dc, float* g_f_rest, float* g_dshs,
                      float* g_xyz, float* g_log_scales, float* g_rot_raw, float* g_opacity_logit,
                      const float* g_dshs_l1, void* stream);

#ifdef __cplusplus
}
#endif
#endif
