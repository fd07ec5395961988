/*
 * s3g_mlp.h -- C ABI of the fused deformation-MLP kernels (fp32 MFMA on gfx950, libs3g.so).
 *
 *   s3g_deform_mlp_forward   <- Deformation.query_time's feature_out + the pos_deform / shs_deform / dino_head
 *                               Sequentials of forward_dynamic   (/root/reference/scene/deformation.py:53-76,78-94,108-166)
 *   s3g_deform_mlp_backward  <- their autograd (10 nn.Linear backward GEMM pairs + ReLU masks)
 *
 * Network (reference defaults, arguments/__init__.py:204-233: net_width 64, defor_depth 1, feat_head on):
 *   hidden = W0 x + b0                                   x: HexPlane features [P,128]
 *   dx     = P2 relu(P1 relu(hidden) + pb1) + pb2        [P,3]
 *   dshs   = S2 relu(S1 relu(hidden) + sb1) + sb2        [P,48]
 *   feat   = D2 relu(D1 relu(D0 hidden + db0) + db1) + db2   [P,3]   (dino_head has NO leading ReLU, deformation.py:70-76)
 * Weights are torch nn.Linear tensors as they are: weight [out,in] row-major, bias [out], device fp32.
 * Arithmetic: v_mfma_f32_32x32x2_f32 = exact fp32 fma chains (no reduced precision).
 */
#ifndef S3G_MLP_H
#define S3G_MLP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct s3g_mlp_params {  /* the same struct describes weights (const use) and their gradients */
  float *W0, *b0;   /* feature_out.0  [64,128], [64] */
  float *P1, *pb1;  /* pos_deform.1   [64,64],  [64] */
  float *P2, *pb2;  /* pos_deform.3   [3,64],   [3]  */
  float *S1, *sb1;  /* shs_deform.1   [64,64],  [64] */
  float *S2, *sb2;  /* shs_deform.3   [48,64],  [48] */
  float *D0, *db0;  /* dino_head.0    [64,64],  [64] */
  float *D1, *db1;  /* dino_head.2    [64,64],  [64] */
  float *D2, *db2;  /* dino_head.4    [3,64],   [3]  */
} s3g_mlp_params;

/* bytes of the stash written by forward and read by backward: the packed LDS images of the weights
 * (s3g_deform_mlp_pack_bytes) followed by 5 x [P,64] fp32 activations (hidden, pos1, shs1, dino1, dino2).  The
 * backward workspace (5 gradient signals) needs 5 * P * 64 * 4 bytes. */
size_t s3g_deform_mlp_stash_bytes(int P);
size_t s3g_deform_mlp_pack_bytes(void);

/* features [P,128] -> dx [P,3], dshs [P,48], feat [P,3].  `stash`: device scratch of s3g_deform_mlp_stash_bytes(P)
 * bytes when save_activations != 0 (a backward will follow), else at least s3g_deform_mlp_pack_bytes().
 * feat may be NULL when save_activations == 0: the feature (dino) head is then skipped -- a render that does not draw the
 * feature image (gaussian_renderer/__init__.py:153, render_feat=False) has no use for it. */
int s3g_deform_mlp_forward(const s3g_mlp_params* w, int P, const float* features, float* dx, float* dshs, float* feat,
                           float* stash, int save_activations, void* stream);

/* g_dx [P,3], g_dshs [P,48], g_feat [P,3] (upstream gradients) -> g_features [P,128] (written) and the parameter
 * gradients in `gw` (ACCUMULATED: the caller zero-fills them).  `stash` from the matching forward (weights must be
 * unchanged since); `workspace` of 5 * P * 64 * 4 bytes, uninitialised.
 * g_feat may be NULL = the feature output received no gradient (the feature image is not in the loss): the dino head's
 * backward is skipped and gw->D0..db2 are left untouched, like autograd leaving those parameters' .grad at None. */
int s3g_deform_mlp_backward(const s3g_mlp_params* w, int P, const float* features, const float* stash, const float* g_dx,
                            const float* g_dshs, const float* g_feat, float* g_features, const s3g_mlp_params* gw,
                            float* workspace, void* stream);

/* The same with an ORDERED flush of the weight gradients (ABI 13).  s3g_deform_mlp_backward lets the up to 256 workgroups of the
 * weight-gradient kernel add their partial [out][in] blocks onto `gw` with float atomics: the order of the additions, and with it
 * the last bits of every weight gradient, differs from run to run.  Here each workgroup stores its partial blocks into
 * `wgrad_partials` (device scratch of s3g_deform_mlp_wgrad_partial_bytes() bytes, uninitialised) and a second kernel adds them in
 * workgroup order: the weight gradients are bit-reproducible (tests/test_mlp_gpu.py), for 38 MB of scratch traffic and one launch. */
size_t s3g_deform_mlp_wgrad_partial_bytes(void);
int s3g_deform_mlp_backward_ordered(const s3g_mlp_params* w, int P, const float* features, const float* stash, const float* g_dx,
                                    const float* g_dshs, const float* g_feat, float* g_features, const s3g_mlp_params* gw,
                                    float* workspace, float* wgrad_partials, void* stream);

/* Arithmetic of the per-point GEMM chains of the two calls above (process-wide setting, default S3G_MLP_F32):
 *   S3G_MLP_F32     v_mfma_f32_32x32x2_f32, exact fp32 fma chains.
 *   S3G_MLP_BF16X3  v_mfma_f32_32x32x16_bf16 (the bf16 matrix pipe, 16 x the rate) with every fp32 operand -- weights, activations,
 *                   gradients -- split EXACTLY into three bf16 pieces and six piece products accumulated in fp32 per product: what
 *                   is dropped is <= 2^-23 of each product (fp32 accuracy; not bit-identical to S3G_MLP_F32).  The split is
 *                   exact for 2^-110 <= |x| < 3.39e38 (smaller numbers keep an absolute error below 2^-133; larger ones
 *                   become infinite like any bf16 conversion): tests/test_split_arith_cpu.py.
 *                   Round 5: the weight fragments of both chain kernels are split ONCE per call into 159 KiB LDS images
 *                   (mlp_forward_presplit_kernel / mlp_backward_presplit_kernel; P1 alone is split by the lanes that read it, the
 *                   3-row heads of the backward run on the exact fp32 MFMA): forward 0.71 -> 0.55 ms, backward 0.76 -> 0.63 ms at
 *                   1.2 M points on one box.  (The backward writes its transposed image into the slot the stash reserves for it;
 *                   forward and backward may run in different modes.)
 *   S3G_MLP_BF16X3_ONTHEFLY  the same arithmetic with every fragment split by the wave that uses it (rounds 3-4; bit-identical
 *                   results, 15-20 % slower): kept as the checker of the pre-split kernels.
 * The weight-gradient GEMMs (K = points) are the exact fp32 chain in every mode.  Returns S3G_ERR_INVALID_ARG for another mode. */
enum { S3G_MLP_F32 = 0, S3G_MLP_BF16X3 = 1, S3G_MLP_BF16X3_ONTHEFLY = 2 };
int s3g_deform_mlp_set_arithmetic(int mode);
int s3g_deform_mlp_get_arithmetic(void);

/* Inference only (no autograd): HexPlane sampler (+) feature_out + position / SH heads in ONE kernel -- what
 * deform_network.forward_dynamic computes under torch.no_grad() for a render that does not draw the feature image
 * (/root/reference/scene/deformation.py:108-166 with the default switches, called from gaussian_renderer/__init__.py:82-97).
 * xyz [P,3], time [P] -> dx [P,3], dshs [P,48]; the [P,128] feature array is never materialised.  Results are bit-identical to
 * s3g_hexplane_forward followed by s3g_deform_mlp_forward(feat = NULL).  The descriptor must have 4 levels (4 x 32 = the 128 inputs
 * of feature_out).  `workspace`: s3g_deform_infer_workspace_bytes(d) bytes, uninitialised.  proc_order as in s3g_hexplane_forward. */
#include "s3g_hexplane.h"
size_t s3g_deform_infer_workspace_bytes(const s3g_hexplane_desc* d);
int s3g_deform_infer(const s3g_hexplane_desc* d, const s3g_mlp_params* w, int P, const float* xyz, const float* time,
                     const unsigned int* proc_order, float* dx, float* dshs, void* workspace, void* stream);
/* The same call with the three GEMM layers on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16 x the fp32 MFMA rate): every fp32
 * operand -- weights and activations -- is the EXACT sum of three bf16 pieces and a product is accumulated (in fp32) as the six
 * piece products of weight >= 2^-16; what is dropped is <= 2^-23 of each product, i.e. fp32 accuracy (measured against fp64 beside
 * the exact path: tests/test_infer_gpu.py), but not bit-identical to s3g_deform_infer.  The sampler half is the same code. */
int s3g_deform_infer_split(const s3g_hexplane_desc* d, const s3g_mlp_params* w, int P, const float* xyz, const float* time,
                           const unsigned int* proc_order, float* dx, float* dshs, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
