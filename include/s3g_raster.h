/*
 * s3g_raster.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer (libs3g.so).
 *
 * Drop-in boundary for the reference's native rasterizer interface.  Each entry point replaces one
 * member of CudaRasterizer::Rasterizer (RAST = /root/reference/submodules/depth-diff-gaussian-rasterization):
 *
 *   s3g_raster_forward   <- Rasterizer::forward    RAST/cuda_rasterizer/rasterizer.h:36-59
 *                           (called by RasterizeGaussiansCUDA, RAST/rasterize_points.cu:35-117)
 *   s3g_raster_backward  <- Rasterizer::backward   RAST/cuda_rasterizer/rasterizer.h:61-87
 *                           (called by RasterizeGaussiansBackwardCUDA, RAST/rasterize_points.cu:119-202)
 *   s3g_mark_visible     <- Rasterizer::markVisible RAST/cuda_rasterizer/rasterizer.h:24-29
 *
 * Same argument meaning as the reference: all array arguments are DEVICE pointers to fp32 (unless
 * noted), a NULL pointer means "not provided" exactly like the reference's nullptr / numel()==0
 * tensors, and the three scratch arenas are obtained through resize callbacks (the reference's
 * std::function<char*(size_t)> geometryBuffer/binningBuffer/imageBuffer, rasterizer.h:37-39).  The
 * byte layout of the arenas is private to this library (forward writes them, backward of the SAME
 * build reads them); they must stay alive and unmodified between the two calls.
 *
 * Differences from the reference, on purpose:
 *   - every kernel is launched on the caller's `stream` (the reference uses the legacy default stream);
 *   - errors are returned as int codes (0 = ok) with a thread-local message, never thrown/trapped;
 *   - plain C: no torch, glm or std types cross the boundary.
 */
#ifndef S3G_RASTER_H
#define S3G_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3G_OK 0
#define S3G_ERR_INVALID_ARG 1   /* bad argument combination (reference: Python Exception / AT_ERROR) */
#define S3G_ERR_HIP 2           /* a HIP runtime call failed; see s3g_last_error() */
#define S3G_ERR_ALLOC 3         /* a resize callback returned NULL */
#define S3G_ERR_PREFILTERED 4   /* a Gaussian was culled although prefiltered=1 (reference: __trap, auxiliary.h:156-160) */

/* Resize callback: return a device pointer to at least `bytes` bytes, 128-byte aligned, owned by the caller. */
typedef void* (*s3g_resize_fn)(void* user, size_t bytes);

typedef struct s3g_raster_inputs {
  int P;                      /* number of Gaussians */
  int D;                      /* active SH degree 0..3 */
  int M;                      /* SH coefficients stored per Gaussian (0 when shs == NULL) */
  int width, height;          /* image size in pixels */
  const float* background;    /* [3] */
  const float* means3D;       /* [P,3] */
  const float* shs;           /* [P,M,3] or NULL (then colors_precomp != NULL) */
  const float* colors_precomp;/* [P,3]   or NULL */
  const float* opacities;     /* [P] */
  const float* scales;        /* [P,3] or NULL (then cov3D_precomp != NULL) */
  float scale_modifier;
  const float* rotations;     /* [P,4] (r,x,y,z), used un-normalised like the reference */
  const float* cov3D_precomp; /* [P,6] or NULL */
  const float* viewmatrix;    /* [16] world->view, row-vector convention (scene/cameras.py:59) */
  const float* projmatrix;    /* [16] full projection, row-vector convention (scene/cameras.py:63) */
  const float* cam_pos;       /* [3] */
  float tan_fovx, tan_fovy;
  int prefiltered;
  int debug;                  /* 1: synchronise + check after every kernel (reference CHECK_CUDA) */
} s3g_raster_inputs;

/* Forward.  out_color [3,H,W], out_depth [1,H,W], radii [P] int32: device, caller allocated (contents are
 * fully overwritten for P > 0).  *num_rendered receives the number of (Gaussian, tile) instances.
 * One host synchronisation on `stream` (to size the binning arena), like the reference (rasterizer_impl.cu:282). */
int s3g_raster_forward(const s3g_raster_inputs* in,
                       s3g_resize_fn geometry_buffer, void* geometry_user,
                       s3g_resize_fn binning_buffer, void* binning_user,
                       s3g_resize_fn image_buffer, void* image_user,
                       float* out_color, float* out_depth, int* radii,
                       int* num_rendered, void* stream /* hipStream_t */);

/* Forward of the SAME geometry (means, scales/rotations or cov3D, opacities, camera, image size) with other per-Gaussian
 * colours: reuses the arenas of a previous s3g_raster_forward (preprocess, binning and sort are skipped; radii and
 * num_rendered are those of that call).  Replaces the second Rasterizer::forward of an iteration
 * (gaussian_renderer/__init__.py:153-166 renders the feature image on the RGB pass's geometry).  Requires
 * in->colors_precomp.  The image arena's final_T / n_contrib are rewritten with identical values. */
int s3g_raster_forward_reuse(const s3g_raster_inputs* in, int R, const void* geometry_arena, const void* binning_arena,
                             void* image_arena, float* out_color, float* out_depth, void* stream);

/* Static / dynamic decomposition renders (gaussian_renderer/__init__.py:168-204: `render_d`/`depth_d` from the Gaussians
 * with is_dynamic != 0, `render_s`/`depth_s` from the others) out of the arenas of a previous s3g_raster_forward of ALL
 * Gaussians: one extra blend pass over the full sorted lists with one transmittance chain per class, instead of two more
 * complete rasterizations of boolean-masked copies.  Images are bit-identical to those (a subset's tile list is the full
 * list minus the other class, in the same order).  The arenas are only read.  Colours: in->colors_precomp, or the
 * forward's own SH colours when NULL.  is_dynamic: uint8 [P], device.  class_counts: device int64 [2] = number of static
 * and of dynamic Gaussians, or NULL: a class with count 0 renders as zeros WITHOUT background, like the reference's P == 0
 * early-out (read on the device: no host sync).  Outputs [3,H,W] / [1,H,W], fully written. */
int s3g_raster_forward_decompose(const s3g_raster_inputs* in, int R, const void* geometry_arena, const void* binning_arena,
                                 const void* image_arena, const uint8_t* is_dynamic, const long long* class_counts,
                                 float* out_color_d, float* out_depth_d,
                                 float* out_color_s, float* out_depth_s, void* stream);

/* Forward of TWO images from one geometry in one blend pass: colours in->colors_precomp -> out_color (+ out_depth) and
 * colors2 [P,3] -> out_color2 [3,H,W]; otherwise identical to s3g_raster_forward (same arenas, radii, num_rendered).
 * Replaces the pair of Rasterizer::forward calls of gaussian_renderer/__init__.py:127-166; pairs with
 * s3g_raster_backward2. */
int s3g_raster_forward2(const s3g_raster_inputs* in, const float* colors2,
                        s3g_resize_fn geometry_buffer, void* geometry_user,
                        s3g_resize_fn binning_buffer, void* binning_user,
                        s3g_resize_fn image_buffer, void* image_user,
                        float* out_color, float* out_depth, float* out_color2, int* radii,
                        int* num_rendered, void* stream /* hipStream_t */);

/* Host-asynchronous forward ("remove the host sync").  s3g_raster_forward / _forward2 wait for the
 * device once per call, exactly where the reference does (rasterizer_impl.cu:281-282), because the instance count R sizes the
 * binning arena.  Here the caller sizes the arenas BEFORE the call for a speculative capacity -- as many (tile, Gaussian)
 * instances and rect slots as it is prepared to hold, e.g. twice the largest count it has seen -- and the call only enqueues
 * kernels: the host can run any number of iterations ahead of the GPU.
 *   arenas            device, at least s3g_raster_arena_bytes(...) bytes each, 128-byte aligned, caller-owned;
 *   sort_lds_keys     estimate of the longest per-tile list (sizes the LDS buffer of the short-list sort launch; lists that
 *                     exceed it are sorted in global memory; 0 = 4096);
 *   long_lists        != 0: also launch the long-list sort pass (lists of more than 4096 instances);
 *   status_device     optional device word, written by EVERY call: bit 0 = the counts exceeded the capacity (instances >
 *                     capacity_instances, slots > capacity_slots, or a list > 4096 with long_lists == 0), bit 1 = a Gaussian was
 *                     culled although in->prefiltered was set.  On overflow the call renders the background only, every list is
 *                     empty, the matching backward returns zero gradients and skips the densification bookkeeping -- a
 *                     well-defined no-op that s3g_adam_step_guarded (s3g_optim.h) can be told to honour, never an
 *                     out-of-bounds write;
 *   status_host       optional PINNED host array of 8 words, filled by a copy enqueued behind the counting kernels:
 *                     [0] instances binned (0 on overflow) [1] longest list [2] error bits [3] slots [4] overflow
 *                     [5] true instance count [6] true slot count.  Valid once `stream` has passed this call (record an event
 *                     after it and poll); a caller that sees [4] != 0 raises its capacity and renders that view again.
 * The matching backward is the ordinary one with R = capacity_instances (the arena layout depends on it), workspace sized for
 * that R.  colors2 / out_color2: both NULL (one image) or both given (the two-image pass of s3g_raster_forward2). */
typedef struct s3g_raster_async {
  uint32_t capacity_instances;
  uint32_t capacity_slots;      /* >= capacity_instances (slots = rect areas before the exact cull) */
  uint32_t sort_lds_keys;
  int long_lists;
  void* geometry_arena;
  void* binning_arena;
  void* image_arena;
  uint32_t* status_device;
  uint32_t* status_host;
  int forward_only;             /* != 0: no backward will follow (inference): the instance -> list-position map that only the
                                 * backward gather reads is not built (no slot fill, no rect / offset gathers and no scattered
                                 * store per instance in the per-tile sort).  The arenas then serve s3g_raster_forward_reuse and
                                 * s3g_raster_forward_decompose, NOT s3g_raster_backward*. */
  uint32_t* sticky_device;      /* optional (NULL: off; ignored when forward_only != 0).  One device word shared by the calls of a
                                 * training loop: the call that overflows sets it, and every later call that is handed the same
                                 * word renders nothing either (status bit 0 set, status_host[7] = 1 "frozen by an earlier call")
                                 * until the host stores 0 into it.  With s3g_adam_step_guarded this freezes the model from the
                                 * overflowed iteration on; the host, which reads the status rows late, raises the capacity, clears
                                 * the word and re-issues the iterations from the overflowed one -- no (view, step) pair is dropped
                                 * or reordered w.r.t. the reference's synchronous loop (train.py:291-522). */
  void* status_event;           /* optional hipEvent_t (ABI 13), recorded on `stream` right behind the copy that fills status_host --
                                 * i.e. after the COUNTING kernels and before the sort / blend kernels of the same call.  A caller
                                 * that must know the verdict before it uses the image (the drop-in Python boundary in its default
                                 * "verified" policy: a render is never silently wrong) waits for THIS event: by then the row is
                                 * valid, while the device still has the rest of the forward to execute, so the wait costs the
                                 * device no idle time.  On [4] != 0 the caller issues the call again with the capacity the true
                                 * counts ask for, into the same output buffers, before anything has read them. */
} s3g_raster_async;
int s3g_raster_arena_bytes(int P, int width, int height, uint32_t capacity_instances, uint32_t capacity_slots,
                           size_t* geometry_bytes, size_t* binning_bytes, size_t* image_bytes);
int s3g_raster_forward_async(const s3g_raster_inputs* in, const float* colors2, const s3g_raster_async* async_,
                             float* out_color, float* out_depth, float* out_color2, int* radii, void* stream);

/* Backward.  `R` is the num_rendered returned by the matching forward; radii / arenas are the ones it filled.
 * `workspace`: device scratch of s3g_raster_backward_workspace_bytes(P, R) bytes (per-instance gradient records;
 * contents need no initialisation and are dead after the call).
 * dL_dpix [3,H,W], dL_dpix_depth [1,H,W].  Outputs (device): dL_dmean2D [P,3], dL_dopacity [P], dL_dcolor [P,3],
 * dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3] (may be NULL when M==0), dL_dscale [P,3], dL_drot [P,4]
 * (16-byte aligned); dL_dconic [P,2,2] (16-byte aligned) and dL_ddepth [P] are the reference's internal
 * intermediates and MAY BE NULL.  Unlike the reference (which accumulates with atomics into caller-zeroed arrays,
 * rasterize_points.cu:154-163) every element of every output is WRITTEN (zeros for culled Gaussians), so the
 * caller does not have to clear them, and the result is bit-reproducible run to run (no atomics). */
size_t s3g_raster_backward_workspace_bytes(int P, int R);
int s3g_raster_backward(const s3g_raster_inputs* in, int R, const int* radii,
                        const void* geometry_arena, const void* binning_arena, const void* image_arena,
                        void* workspace,
                        const float* dL_dpix, const float* dL_dpix_depth,
                        float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                        float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                        float* dL_dscale, float* dL_drot, void* stream);

/* Backward of TWO images blended from the same geometry -- the RGB+depth render (in->colors_precomp) and a second render
 * with colours `colors2` [P,3] through s3g_raster_forward_reuse (the feature image of
 * gaussian_renderer/__init__.py:153-166) -- in one pass instead of two Rasterizer::backward calls whose results autograd
 * then adds: the alpha test, exp2, transmittance recurrence and the six geometry sums are shared.  dL_dpix2 [3,H,W] is the
 * upstream gradient of the second image (its depth output is unused by the reference and gets none); dL_dcolor2 [P,3]
 * is written; every other output is the SUM over both images.  Requires in->colors_precomp (no SH path).
 * `workspace`: s3g_raster_backward2_workspace_bytes(P, R) bytes. */
size_t s3g_raster_backward2_workspace_bytes(int P, int R);
int s3g_raster_backward2(const s3g_raster_inputs* in, const float* colors2, int R, const int* radii,
                         const void* geometry_arena, const void* binning_arena, const void* image_arena,
                         void* workspace,
                         const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix2,
                         float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dcolor2,
                         float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot,
                         void* stream);

/* Backward + densification bookkeeping in the same per-Gaussian pass.  The reference follows every
 * backward with three PyTorch passes over [P] (train.py:489-493, scene/gaussian_model.py:693-695):
 *     max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);  xyz_gradient_accum[vis] += ||viewspace_grad[vis,:2]||;
 *     denom[vis] += 1                                          with vis = radii > 0.
 * The per-Gaussian backward kernel has dL_dmean2D and the radius in registers, so `dens` (all three arrays, device fp32
 * [P], caller-owned accumulators, read-modify-written for visible Gaussians only) makes it do the update itself.
 * Valid when this backward produces the WHOLE viewspace gradient of the iteration (one render, or the two-image pass);
 * with several backward calls per iteration or a data-parallel batch the norm must be taken after the sum: use
 * s3g_densify_stats (s3g_optim.h) on the summed gradient instead.  dens == NULL: identical to the plain entry points. */
typedef struct s3g_densify_accum {
  float* xyz_gradient_accum;  /* [P]   (the reference's [P,1]) */
  float* denom;               /* [P]   (the reference's [P,1]) */
  float* max_radii2D;         /* [P] */
} s3g_densify_accum;
int s3g_raster_backward_accum(const s3g_raster_inputs* in, int R, const int* radii,
                              const void* geometry_arena, const void* binning_arena, const void* image_arena,
                              void* workspace, const float* dL_dpix, const float* dL_dpix_depth,
                              float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                              float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                              float* dL_dscale, float* dL_drot, const s3g_densify_accum* dens, void* stream);
int s3g_raster_backward2_accum(const s3g_raster_inputs* in, const float* colors2, int R, const int* radii,
                               const void* geometry_arena, const void* binning_arena, const void* image_arena,
                               void* workspace, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix2,
                               float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                               float* dL_dcolor2, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                               float* dL_dscale, float* dL_drot, const s3g_densify_accum* dens, void* stream);

/* Exact (tile, Gaussian) culling at binning time, ON by default.  The reference bins a Gaussian into every tile of the
 * bounding square of its 3-sigma radius (auxiliary.h:46-56, rasterizer_impl.cu:88-115); tiles in which it cannot reach
 * alpha >= 1/255 on any pixel are evaluated and discarded pixel by pixel (forward.cu:330-341).  With culling on, such
 * (tile, Gaussian) instances are never created: images, depths, radii and every gradient are unchanged bit for bit, only
 * num_rendered and the private per-tile lists shrink (half the instances at BASELINE cfg3).  s3g_raster_set_exact_cull(0)
 * restores the reference's bounding-square binning (used by the tests that compare the lists with the oracle).
 * Process-wide; applies to forwards issued after the call. */
void s3g_raster_set_exact_cull(int on);
int s3g_raster_get_exact_cull(void);

/* Tile grids of more than 38 000 tiles (beyond 4K images) do not fit the binning pass's LDS histogram and are binned in
 * bands of consecutive tiles, one pair of launches per band; results are identical.  Testing hook: force a smaller band
 * (returns the previous size; tiles <= 0 restores the default) so the band path can be exercised on small images. */
int s3g_raster_set_bin_band(int tiles);

/* present[P] (uint8 0/1) = in_frustum (auxiliary.h:139-164: view-space z > 0.2). */
int s3g_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Optional in-library timing of the hot kernels with hipEvent pairs recorded on the launch stream (bench.py's roofline
 * leg).  s3g_profile_read sums and clears the recorded launches of one id (synchronising on their events) and returns
 * how many there were; the two totals are per-id work counts:
 *   blend kernels: (sorted instances R -- -1 per launch of the asynchronous forward, which does not know it --, pixels)   hexplane kernels: (points P, levels)   MLP kernels: (points P, 0)
 *   Adam: (parameters updated, 0).
 * MLP_WGRAD brackets the nine weight-gradient launches of one backward call. */
enum {
  S3G_PROFILE_BLEND_FORWARD = 0,
  S3G_PROFILE_BLEND_BACKWARD = 1,
  S3G_PROFILE_HEXPLANE_FORWARD = 2,
  S3G_PROFILE_HEXPLANE_BACKWARD_POINT = 3,
  S3G_PROFILE_HEXPLANE_SCATTER = 4,
  S3G_PROFILE_MLP_FORWARD = 5,
  S3G_PROFILE_MLP_BACKWARD = 6,
  S3G_PROFILE_MLP_WGRAD = 7,
  S3G_PROFILE_ADAM = 8,
  S3G_PROFILE_DEFORM_INFER = 9, /* s3g_deform_infer: HexPlane sampler (+) MLP heads, inference */
  S3G_PROFILE_IDS = 10
};
void s3g_profile_enable(int on);
int s3g_profile_read(int id, double* total_ms, double* total_instances, double* total_pixels);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* s3g_last_error(void);

/* Library/ABI version: bumped whenever the private arena layout or a signature changes. */
int s3g_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* S3G_RASTER_H */
