/*
 * s3g_hexplane.h -- C ABI of the MI355X-native multi-resolution HexPlane sampler (libs3g.so).
 *
 *   s3g_hexplane_forward   <- HexPlaneField.forward / get_density -> interpolate_ms_features -> 24x grid_sample_wrapper
 *                             (/root/reference/scene/hexplane.py:177-183, :151-175, :73-106, :21-46)
 *   s3g_hexplane_backward  <- the autograd backward of the same (torch grid_sampler_2d_backward x24 + product rule)
 *
 * One fused pass per direction instead of 24 grid_sample launches that each materialise a [P,32] tensor.
 *
 * Semantics (identical to the reference): pts = (xyz - aabb[0]) * (2 / (aabb[1] - aabb[0])) - 1 with aabb[0] the MAX
 * corner and aabb[1] the MIN corner (scene/hexplane.py:19-20,113-114); the time coordinate is used as given (NOT
 * normalised, scene/hexplane.py:164); each of the 6 planes of a level -- coordinate pairs (x,y) (x,z) (x,t) (y,z)
 * (y,t) (z,t) in itertools.combinations order -- is sampled bilinearly with align_corners=True and border padding;
 * the 6 samples are multiplied, the levels concatenated: features [P, levels*C].
 *
 * Plane memory layout: the reference parameter of plane (c0,c1) is a [1, C, res[c1], res[c0]] tensor; this library
 * reads it CHANNEL-LAST, i.e. as [res[c1]][res[c0]][C] floats -- exactly the bytes of the same tensor in
 * torch.channels_last memory format, so one texel (C = 32 floats = 128 B) is one cache line and one half-wave load.
 */
#ifndef S3G_HEXPLANE_H
#define S3G_HEXPLANE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S3G_HEX_MAX_LEVELS 8
#define S3G_HEX_CHANNELS 32 /* output_coordinate_dim of the reference config (arguments/__init__.py:219) */

typedef struct s3g_hexplane_desc {
  int levels;                                /* len(multires) */
  int res[S3G_HEX_MAX_LEVELS][4];            /* per level: resolution along x, y, z, t */
  const float* planes[S3G_HEX_MAX_LEVELS][6];/* device, channel-last [res[c1]][res[c0]][32] */
  float aabb_max[3], aabb_min[3];            /* aabb[0], aabb[1] */
  int uniform_time;                          /* != 0: the caller guarantees time[i] == time[0] for every point -- how
                                              * render() always calls the field (gaussian_renderer/__init__.py:58:
                                              * one camera timestamp repeated P times).  The three time planes of a level
                                              * are then pre-interpolated along t into 1-D row tables once per call and
                                              * sampled with 2 taps instead of 4.  Only time[0] is read then: the
                                              * `time` arrays below may hold a single element. */
} s3g_hexplane_desc;

/* features [P, levels*32].  xyz [P,3], time [P] (device fp32).  proc_order: optional (may be NULL) permutation of
 * [0,P) giving the order in which points are PROCESSED (results do not depend on it); passing the spatial order a
 * previous s3g_hexplane_backward returned makes neighbouring lanes fetch the same texels (L1/L2 hits). */
size_t s3g_hexplane_forward_workspace_bytes(const s3g_hexplane_desc* d);   /* 0 unless uniform_time (row tables) */
int s3g_hexplane_forward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time, float* features,
                         const unsigned int* proc_order, void* workspace /* may be NULL when the size above is 0 */,
                         void* stream);

/* dL_dfeatures [P, levels*32] -> dL_dxyz [P,3] (written) and dL_dplanes[l][i] (same layout as planes; ACCUMULATED,
 * the caller zero-fills them; a NULL entry skips that plane).  `workspace`: device scratch of
 * s3g_hexplane_backward_workspace_bytes(d, P, features != NULL) bytes (sort buffers, the row tables and their gradients when
 * uniform_time, + 128 B per point and level: one row T = dL/dfeature * feature; rounds 1-2: 3 KB per point), uninitialised. */
int s3g_hexplane_backward_scratch_rows(int levels);   /* 128-byte rows of scratch per point the per-point pass writes */
/* DIAGNOSTICS ONLY: which of the 3 * levels scatter walks of s3g_hexplane_backward run (bit orientation * levels + level; default all).
 * With any other mask the plane gradients are incomplete -- for timing the walks one by one (tools/hex_probe.py walks,
 * profiles/r06_hex_walks.txt).  Process-wide. */
void s3g_hexplane_debug_walk_mask(uint32_t mask);
/* Deterministic mode of s3g_hexplane_backward (process-wide, default off; round 6).  The default backward adds the scatter walks' partial
 * sums onto the plane gradients with float atomics, and its walk orders come from counting sorts whose placement step uses LDS atomics:
 * the ORDER of the additions -- and with it the last bits of every plane gradient -- differs from run to run.  With the mode on
 *   - the counting sorts place the points of a key in input order (ballot ranking instead of atomics: stable orders),
 *   - a walker STORES the sums of every finished footprint ("run") as a record -- per cell, or per segment for a cell that straddles
 *     segments -- instead of adding them with atomics,
 *   - one stencil pass adds, for every texel, the records of the four cells around it in a fixed order.
 * Plane gradients, dL/dxyz and the walk orders are then bit-identical between runs (tests/test_hexplane_gpu.py).  Requirements:
 * desc.uniform_time != 0 and spatial resolutions <= 512; s3g_hexplane_backward_workspace_bytes follows the setting (+ 0.55 GB of run
 * records at the reference's resolutions).  The exact fallback of (nearly) zero samples still uses atomics (a handful of addends). */
void s3g_hexplane_set_deterministic(int on);
int s3g_hexplane_get_deterministic(void);
/* 32-bit words per point of `sort_state` below: 2 x (walk orders) + 1.  Round 4: one walk order per orientation AND level,
 * 6 * levels + 1 words (25 at the reference's four levels); rounds 1-3 kept three orders (7 words). */
int s3g_hexplane_sort_state_words(int levels);
size_t s3g_hexplane_backward_workspace_bytes(const s3g_hexplane_desc* d, int P, int have_features);
/* Backward algorithms (s3g_hexplane_backward_algo; s3g_hexplane_backward selects SLAB for features == NULL, SLAB_DIV otherwise):
 *   S3G_HEX_SLAB      the exact fallback that needs nothing from the forward: a per-point pass forms dL/d(sample) by the product
 *                     rule, finishes dL/dxyz and stores ONE row per level, T = dL/dfeature * feature; sorted scatter walks divide
 *                     T by the sample they re-derive.
 *   S3G_HEX_WALK      (rounds 2-4: no per-point pass, 30 B of scratch per point, 1.4 x slower) REMOVED in ABI 12: the entry point
 *                     returns S3G_ERR_INVALID_ARG for it.
 *   S3G_HEX_SLAB_DIV  (the default: what s3gaussian_amd.hexplane uses) the two passes of SLAB, but the per-point pass takes
 *                     T = dL/dfeature * feature from the forward's output (`features`, required) and dL/d(sample_i) = T / sample_i
 *                     plane by plane, instead of keeping six samples live for the product rule: the forward's register count
 *                     and occupancy.  Same exact fallback for samples that cannot be divided by. */
#define S3G_HEX_SLAB 0
#define S3G_HEX_WALK 1
#define S3G_HEX_SLAB_DIV 2
int s3g_hexplane_backward_algo(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                               const float* dL_dfeatures, const float* features, int algorithm, float* dL_dxyz,
                               float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* workspace,
                               unsigned int* sort_state, int sort_reuse, void* stream);
int s3g_hexplane_backward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                          const float* dL_dfeatures,
                          const float* features /* [P, levels*32]: the OUTPUT of the matching s3g_hexplane_forward (same
                          planes, xyz, time): selects S3G_HEX_SLAB_DIV (T = dL/dfeature * feature in one multiply, every plane
                          divides by its own sample).  NULL selects S3G_HEX_SLAB (product rule; needs nothing from the forward).
                          Either way a per-point pass stores ONE row per level and finishes dL/dxyz, and the scatter walks divide
                          T by the sample they re-derive (128 B per point and level of scratch). */,
                          float* dL_dxyz, float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6],
                          void* workspace,
                          unsigned int* sort_state /* [s3g_hexplane_sort_state_words(levels) * P] device or NULL.  With
                          NW = 3 * levels walk orders -- order oi = orientation * levels + level sorts the points by that
                          level's own (major, minor) texel cells --: the orders [0, NW*P), for each of them the position of its
                          k-th point in the processing order [NW*P, 2*NW*P) (where that point's T rows are), and the 3-D
                          blocked processing order itself [2*NW*P, (2*NW+1)*P).  They only steer HOW the work is walked (texel
                          reuse, run-length combining), never the result, so a caller may keep them across iterations while the
                          points move slowly: sort_reuse != 0 = `sort_state` holds the orders of an earlier call with the same P and the
                          sorts are skipped; sort_reuse == 0 = they are recomputed and left there.  The last P words are the
                          blocked order, usable as proc_order of later forwards. */,
                          int sort_reuse, void* stream);

#ifdef __cplusplus
}
#endif
#endif
