/*
 * oracle/raster_oracle.c -- CPU restatement of the reference tile rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under s3gaussian_amd/ may import, link or
 * call this file; it is the checker the HIP path is compared against (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * PINNED BY THE REFERENCE ITSELF (round 2).  The reference ships no tests, golden vectors or CPU path for its
 * rasterizer, so the pin is the reference's own kernels: oracle/build_ref.sh hipifies
 * RAST = /root/reference/submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/{forward,backward,
 * rasterizer_impl}.cu in a temporary directory and builds oracle/_ref/libref_raster.so for gfx950 (test-only);
 * tests/test_raster_ref_gpu.py::test_cpu_oracle_is_pinned_by_the_reference_build runs this file against it on twelve
 * scenes up to BASELINE cfg3 size (1.2 M Gaussians, 1066x1600, R = 11.4 M): radii, tiles_touched, point_offsets,
 * num_rendered, tile ranges, sorted 64-bit keys, point_list, n_contrib EXACT; depths / means2D / cov3D / conic_opacity
 * / rgb / clamped BIT-EXACT (against the -ffp-contract=off build); colour <= 2.1e-5, final_T <= 4.8e-6; all ten
 * gradient arrays rel-L2 <= 3.3e-5 (the reference's own run-to-run atomics noise is 1.3e-5), numbers in
 * profiles/r02_parity_stats.jsonl.  This file is a line-by-line restatement of the algorithm in
 *   RAST/cuda_rasterizer/forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h, config.h, RAST/rasterize_points.cu
 * additionally cross-validated on CPU (tests/test_oracle_cpu.py) against
 *   - a float64 PyTorch autograd restatement of the forward (oracle/torch_ref.py)
 *     for every returned gradient (the fp64 build of this file agrees to 1e-9),
 *   - the reference's own utils/sh_utils.py::eval_sh for the SH path
 *     (golden vectors generated in-container, tests/golden/).
 *
 * Arithmetic: built with -ffp-contract=off so every float op rounds once, in
 * the order the reference source writes it (glm 0.9.9.9 mat3 products expanded
 * in glm's own summation order, type_mat3x3.inl:486-518).  nvcc contracts
 * a*b+c into fma where it likes, so bit-equality with a CUDA build is not
 * defined even for the reference against itself.
 *
 * REAL=float  (default)  -> liboracle_f32.so   the fp32 oracle
 * REAL=double            -> liboracle_f64.so   same code in fp64 (gradient checks)
 *
 * Gradient accumulation: the reference sums per-(pixel,Gaussian) terms with
 * float atomicAdd in non-deterministic order (backward.cu:550-587).  The oracle
 * accumulates those sums in double (OpenMP atomics over tiles; double sums of
 * fp32 terms are order-independent to ~1e-16 relative) and rounds once, i.e. the
 * order-independent limit of what the reference computes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define BLOCK_X 16 /* config.h:16 */
#define BLOCK_Y 16 /* config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3 /* config.h:15 */

/* auxiliary.h:22-39 */
static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554,  (real)-0.4570457994644658,
                              (real)0.3731763325901154,  (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rsqrt_(real x) { return (real)sqrt((double)x); } /* correctly rounded sqrt */
static inline real rexp_(real x) {
#if defined(ORACLE_F64)
  return exp(x);
#else
  return expf(x);
#endif
}

/* glm::mat3 is column major: m[c][r]. */
typedef struct { real m[3][3]; } mat3;

/* glm type_mat3x3.inl:486-518: Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
static mat3 mat3_mul(const mat3* A, const mat3* B) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A->m[0][r] * B->m[c][0] + A->m[1][r] * B->m[c][1] + A->m[2][r] * B->m[c][2];
  return R;
}
static mat3 mat3_T(const mat3* A) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.m[c][r] = A->m[r][c];
  return R;
}

/* auxiliary.h:58-77 */
static void transformPoint4x3(const real* p, const real* M, real* o) {
  o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
  o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
  o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
static void transformPoint4x4(const real* p, const real* M, real* o) {
  o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
  o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
  o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
  o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

/* auxiliary.h:41-44 -- the literals are double in the reference, so this is double arithmetic. */
static real ndc2Pix(real v, int S) { return (real)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56.  int max_radius, truncating float->int conversions, clamp to the tile grid. */
static void getRect(real px, real py, int max_radius, int gx, int gy, uint32_t* rmin_, uint32_t* rmax_) {
  int v;
  v = (int)((px - max_radius) / BLOCK_X); if (v < 0) v = 0; if (v > gx) v = gx; rmin_[0] = (uint32_t)v;
  v = (int)((py - max_radius) / BLOCK_Y); if (v < 0) v = 0; if (v > gy) v = gy; rmin_[1] = (uint32_t)v;
  v = (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X); if (v < 0) v = 0; if (v > gx) v = gx; rmax_[0] = (uint32_t)v;
  v = (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y); if (v < 0) v = 0; if (v > gy) v = gy; rmax_[1] = (uint32_t)v;
}

/* forward.cu:118-152.  Quaternion is used as given (normalisation commented out, :127). */
static void computeCov3D(const real* scale, real mod, const real* rot, real* cov3D) {
  mat3 S; memset(&S, 0, sizeof S);
  S.m[0][0] = mod * scale[0]; S.m[1][1] = mod * scale[1]; S.m[2][2] = mod * scale[2];
  real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
             {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
             {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}}};
  mat3 M = mat3_mul(&S, &R);
  mat3 Mt = mat3_T(&M);
  mat3 Sigma = mat3_mul(&Mt, &M);
  cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

/* Shared by forward.cu:74-113 and backward.cu:166-199: t (clamped), J, W, T, cov2D. */
typedef struct { real t[3]; real txtz, tytz, limx, limy; mat3 J, W, T, Vrk, cov; } cov2d_ctx;
static void cov2D_common(const real* mean, real fx, real fy, real tan_fovx, real tan_fovy, const real* cov3D,
                         const real* V, cov2d_ctx* c) {
  transformPoint4x3(mean, V, c->t);
  c->limx = (real)1.3 * tan_fovx; c->limy = (real)1.3 * tan_fovy;
  c->txtz = c->t[0] / c->t[2]; c->tytz = c->t[1] / c->t[2];
  c->t[0] = rmin(c->limx, rmax(-c->limx, c->txtz)) * c->t[2];
  c->t[1] = rmin(c->limy, rmax(-c->limy, c->tytz)) * c->t[2];
  real tz = c->t[2];
  mat3 J = {{{fx / tz, 0, -(fx * c->t[0]) / (tz * tz)}, {0, fy / tz, -(fy * c->t[1]) / (tz * tz)}, {0, 0, 0}}};
  mat3 W = {{{V[0], V[4], V[8]}, {V[1], V[5], V[9]}, {V[2], V[6], V[10]}}};
  mat3 Vrk = {{{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}}};
  c->J = J; c->W = W; c->Vrk = Vrk;
  c->T = mat3_mul(&c->W, &c->J);
  mat3 Tt = mat3_T(&c->T), Vt = mat3_T(&c->Vrk);
  mat3 tmp = mat3_mul(&Tt, &Vt);
  c->cov = mat3_mul(&tmp, &c->T);
}

/* forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const real* means, const real* campos, const real* shs,
                               uint8_t* clamped, real* out) {
  real dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
  real len = rsqrt_(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); /* glm::length = sqrt(dot) */
  dir[0] /= len; dir[1] /= len; dir[2] /= len;
  const real* sh = shs + (size_t)idx * max_coeffs * 3;
  real x = dir[0], y = dir[1], z = dir[2];
  for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
    real res = SH_C0 * SH(0);
    if (deg > 0) {
      res = res - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
      if (deg > 1) {
        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2 * zz - xx - yy) * SH(6) +
              SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3 * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                SH_C3[2] * y * (4 * zz - xx - yy) * SH(11) + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * SH(12) +
                SH_C3[4] * x * (4 * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                SH_C3[6] * x * (xx - 3 * yy) * SH(15);
        }
      }
    }
#undef SH
    res += (real)0.5;
    clamped[3 * idx + c] = (res < 0);
    out[c] = rmax(res, 0);
  }
}

/* ------------------------------------------------------------------------------------------------
 * All forward state the reference keeps in geomBuffer / binningBuffer / imgBuffer
 * (rasterizer_impl.h GeometryState / BinningState / ImageState), caller allocated.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  /* GeometryState, size P */
  real* depths;         /* [P] */
  uint8_t* clamped;     /* [P*3] */
  real* means2D;        /* [P*2] */
  real* cov3D;          /* [P*6] */
  real* conic_opacity;  /* [P*4] */
  real* rgb;            /* [P*3] */
  uint32_t* tiles_touched; /* [P] */
  uint32_t* point_offsets; /* [P] inclusive scan */
  /* ImageState, size W*H (ranges: tiles) */
  real* final_T;        /* [H*W]  (accum_alpha) */
  uint32_t* n_contrib;  /* [H*W] */
  uint32_t* ranges;     /* [tiles*2] */
  /* BinningState: allocated by orc_forward via malloc, size R; freed by orc_free_binning */
  uint64_t* point_list_keys; /* [R] sorted */
  uint32_t* point_list;      /* [R] sorted */
  int num_rendered;
} orc_state;

typedef struct {
  int P, D, M;
  int W, H;
  const real* background;
  const real* means3D;
  const real* shs;            /* NULL when colors_precomp given */
  const real* colors_precomp; /* NULL when shs given */
  const real* opacities;
  const real* scales;         /* NULL when cov3D_precomp given */
  real scale_modifier;
  const real* rotations;
  const real* cov3D_precomp;
  const real* viewmatrix;
  const real* projmatrix;
  const real* cam_pos;
  real tan_fovx, tan_fovy;
} orc_inputs;

/* forward.cu:155-256 */
static void preprocess_one(const orc_inputs* in, orc_state* st, int* radii, int idx, real fx, real fy, int gx, int gy) {
  radii[idx] = 0;
  st->tiles_touched[idx] = 0;
  const real* p_orig = in->means3D + 3 * idx;
  real p_view[3];
  transformPoint4x3(p_orig, in->viewmatrix, p_view); /* in_frustum, auxiliary.h:139-164 */
  if (p_view[2] <= (real)0.2) return;
  real p_hom[4];
  transformPoint4x4(p_orig, in->projmatrix, p_hom);
  real p_w = 1 / (p_hom[3] + (real)0.0000001);
  real p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
  const real* cov3D;
  if (in->cov3D_precomp) cov3D = in->cov3D_precomp + 6 * idx;
  else {
    computeCov3D(in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, st->cov3D + 6 * idx);
    cov3D = st->cov3D + 6 * idx;
  }
  cov2d_ctx c;
  cov2D_common(p_orig, fx, fy, in->tan_fovx, in->tan_fovy, cov3D, in->viewmatrix, &c);
  real cx = c.cov.m[0][0] + (real)0.3, cy = c.cov.m[0][1], cz = c.cov.m[1][1] + (real)0.3;
  real det = cx * cz - cy * cy;
  if (det == 0) return;
  real det_inv = 1 / det;
  real conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
  real mid = (real)0.5 * (cx + cz);
  real lambda1 = mid + rsqrt_(rmax((real)0.1, mid * mid - det));
  real lambda2 = mid - rsqrt_(rmax((real)0.1, mid * mid - det));
  real my_radius = (real)ceil((double)(3 * rsqrt_(rmax(lambda1, lambda2))));
  real pix[2] = {ndc2Pix(p_proj[0], in->W), ndc2Pix(p_proj[1], in->H)};
  uint32_t rmn[2], rmx[2];
  getRect(pix[0], pix[1], (int)my_radius, gx, gy, rmn, rmx);
  if ((rmx[0] - rmn[0]) * (rmx[1] - rmn[1]) == 0) return;
  if (!in->colors_precomp)
    computeColorFromSH(idx, in->D, in->M, in->means3D, in->cam_pos, in->shs, st->clamped, st->rgb + 3 * idx);
  st->depths[idx] = p_view[2];
  radii[idx] = (int)my_radius;
  st->means2D[2 * idx] = pix[0]; st->means2D[2 * idx + 1] = pix[1];
  st->conic_opacity[4 * idx + 0] = conic[0]; st->conic_opacity[4 * idx + 1] = conic[1];
  st->conic_opacity[4 * idx + 2] = conic[2]; st->conic_opacity[4 * idx + 3] = in->opacities[idx];
  st->tiles_touched[idx] = (rmx[1] - rmn[1]) * (rmx[0] - rmn[0]);
}

/* Stable LSD radix sort of (key,value) pairs on the low `bits` bits: same ordering contract as
 * cub::DeviceRadixSort::SortPairs(..., 0, 32+bit) at rasterizer_impl.cu:304-309 (stable, so ties in
 * (tile,depth) keep emission order = ascending Gaussian index). */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, size_t n, int bits) {
  if (n == 0) return;
  uint64_t* ktmp = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint32_t* vtmp = (uint32_t*)malloc(n * sizeof(uint32_t));
  uint64_t *ksrc = keys, *kdst = ktmp;
  uint32_t *vsrc = vals, *vdst = vtmp;
  for (int shift = 0; shift < bits; shift += 8) {
    size_t cnt[257]; memset(cnt, 0, sizeof cnt);
    for (size_t i = 0; i < n; i++) cnt[((ksrc[i] >> shift) & 255) + 1]++;
    for (int i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
    for (size_t i = 0; i < n; i++) { size_t d = cnt[(ksrc[i] >> shift) & 255]++; kdst[d] = ksrc[i]; vdst[d] = vsrc[i]; }
    uint64_t* tk = ksrc; ksrc = kdst; kdst = tk;
    uint32_t* tv = vsrc; vsrc = vdst; vdst = tv;
  }
  if (ksrc != keys) { memcpy(keys, ksrc, n * sizeof(uint64_t)); memcpy(vals, vsrc, n * sizeof(uint32_t)); }
  free(ktmp); free(vtmp);
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
  if (n >> msb) msb++;
  return msb;
}

/* forward.cu:261-379, one tile */
static void render_tile(const orc_inputs* in, const orc_state* st, const real* features, int tx, int ty, int gx,
                        real* out_color, real* out_depth) {
  const int W = in->W, H = in->H;
  uint32_t r0 = st->ranges[2 * (ty * gx + tx)], r1 = st->ranges[2 * (ty * gx + tx) + 1];
  for (int ly = 0; ly < BLOCK_Y; ly++)
    for (int lx = 0; lx < BLOCK_X; lx++) {
      int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
      if (px >= W || py >= H) continue;
      uint32_t pix_id = (uint32_t)W * py + px;
      real pixf[2] = {(real)px, (real)py};
      real T = 1, C[NUM_CHANNELS] = {0, 0, 0}, D = 0;
      uint32_t contributor = 0, last_contributor = 0;
      for (uint32_t k = r0; k < r1; k++) {
        contributor++;
        uint32_t id = st->point_list[k];
        real dx = st->means2D[2 * id] - pixf[0], dy = st->means2D[2 * id + 1] - pixf[1];
        const real* co = st->conic_opacity + 4 * id;
        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0) continue;
        real alpha = rmin((real)0.99, co[3] * rexp_(power));
        if (alpha < (real)1.0 / (real)255.0) continue;
        real test_T = T * (1 - alpha);
        if (test_T < (real)0.0001) break; /* done = true */
        for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
        D += st->depths[id] * alpha * T;
        T = test_T;
        last_contributor = contributor;
      }
      st->final_T[pix_id] = T;
      st->n_contrib[pix_id] = last_contributor;
      for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * in->background[ch];
      out_depth[pix_id] = D;
    }
}

/* CudaRasterizer::Rasterizer::forward, rasterizer_impl.cu:198-339.  out_color/out_depth/radii must be
 * zero-filled by the caller (rasterize_points.cu:68-70 does torch::full(0)).  Returns num_rendered. */
int orc_forward(const orc_inputs* in, orc_state* st, real* out_color, real* out_depth, int* radii) {
  const int P = in->P, W = in->W, H = in->H;
  const real fy = H / (2 * in->tan_fovy), fx = W / (2 * in->tan_fovx);
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  st->point_list_keys = NULL; st->point_list = NULL; st->num_rendered = 0;
  if (P == 0) return 0; /* rasterize_points.cu:82 */
  for (int i = 0; i < P; i++) preprocess_one(in, st, radii, i, fx, fy, gx, gy);
  uint32_t acc = 0; /* InclusiveSum, rasterizer_impl.cu:278 */
  for (int i = 0; i < P; i++) { acc += st->tiles_touched[i]; st->point_offsets[i] = acc; }
  int R = (int)acc;
  st->num_rendered = R;
  st->point_list_keys = (uint64_t*)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint64_t));
  st->point_list = (uint32_t*)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint32_t));
  /* duplicateWithKeys, rasterizer_impl.cu:70-111 */
  for (int idx = 0; idx < P; idx++) {
    if (radii[idx] > 0) {
      uint32_t off = idx == 0 ? 0 : st->point_offsets[idx - 1];
      uint32_t rmn[2], rmx[2];
      getRect(st->means2D[2 * idx], st->means2D[2 * idx + 1], radii[idx], gx, gy, rmn, rmx);
      float depth_f = (float)st->depths[idx]; /* key holds the fp32 bit pattern */
      uint32_t dbits; memcpy(&dbits, &depth_f, 4);
      for (uint32_t y = rmn[1]; y < rmx[1]; y++)
        for (uint32_t x = rmn[0]; x < rmx[0]; x++) {
          uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
          key <<= 32; key |= dbits;
          st->point_list_keys[off] = key; st->point_list[off] = (uint32_t)idx; off++;
        }
    }
  }
  int bit = (int)getHigherMsb((uint32_t)(gx * gy));
  radix_sort_pairs(st->point_list_keys, st->point_list, (size_t)R, 32 + bit);
  /* identifyTileRanges, rasterizer_impl.cu:116-138 (after cudaMemset 0, :311) */
  memset(st->ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
  for (int i = 0; i < R; i++) {
    uint32_t cur = (uint32_t)(st->point_list_keys[i] >> 32);
    if (i == 0) st->ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(st->point_list_keys[i - 1] >> 32);
      if (cur != prev) { st->ranges[2 * prev + 1] = (uint32_t)i; st->ranges[2 * cur] = (uint32_t)i; }
    }
    if (i == R - 1) st->ranges[2 * cur + 1] = (uint32_t)R;
  }
  const real* features = in->colors_precomp ? in->colors_precomp : st->rgb;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) render_tile(in, st, features, tx, ty, gx, out_color, out_depth);
  return R;
}

void orc_free_binning(orc_state* st) {
  free(st->point_list_keys); free(st->point_list);
  st->point_list_keys = NULL; st->point_list = NULL;
}

/* rasterizer_impl.cu:54-66,141-153 */
void orc_mark_visible(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present) {
  (void)projmatrix;
  for (int i = 0; i < P; i++) {
    real pv[3];
    transformPoint4x3(means3D + 3 * i, viewmatrix, pv);
    present[i] = !(pv[2] <= (real)0.2);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Backward
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double* dmean2D; /* [P*2] (.x,.y; .z of the reference float3 stays 0) */
  double* dconic;  /* [P*3] (.x,.y,.w of the reference float4; .z unused) */
  double* dopacity;
  double* dcolors; /* [P*3] */
  double* ddepths; /* [P] */
} bw_acc;

/* backward.cu:415-590, one tile.  Accumulates into acc (caller serialises tiles or uses omp critical). */
static void render_tile_bw(const orc_inputs* in, const orc_state* st, const real* colors, int tx, int ty, int gx,
                           const real* dL_dpixels, const real* dL_dpixel_depths, bw_acc* acc) {
  const int W = in->W, H = in->H;
  uint32_t r0 = st->ranges[2 * (ty * gx + tx)], r1 = st->ranges[2 * (ty * gx + tx) + 1];
  const int toDo0 = (int)(r1 - r0);
  const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);
  for (int ly = 0; ly < BLOCK_Y; ly++)
    for (int lx = 0; lx < BLOCK_X; lx++) {
      int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
      if (px >= W || py >= H) continue;
      uint32_t pix_id = (uint32_t)W * py + px;
      real pixf[2] = {(real)px, (real)py};
      const real T_final = st->final_T[pix_id];
      real T = T_final;
      uint32_t contributor = (uint32_t)toDo0;
      const uint32_t last_contributor = st->n_contrib[pix_id];
      real accum_rec[NUM_CHANNELS] = {0, 0, 0}, dL_dpixel[NUM_CHANNELS], accum_depth_rec = 0;
      for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
      real dL_dpixel_depth = dL_dpixel_depths[pix_id];
      real last_alpha = 0, last_color[NUM_CHANNELS] = {0, 0, 0}, last_depth = 0;
      for (int k = 0; k < toDo0; k++) { /* back to front: point_list[range.y - progress - 1] */
        contributor--;
        if (contributor >= last_contributor) continue;
        uint32_t id = st->point_list[r1 - 1 - (uint32_t)k];
        real dx = st->means2D[2 * id] - pixf[0], dy = st->means2D[2 * id + 1] - pixf[1];
        const real* co = st->conic_opacity + 4 * id;
        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0) continue;
        real G = rexp_(power);
        real alpha = rmin((real)0.99, co[3] * G);
        if (alpha < (real)1.0 / (real)255.0) continue;
        T = T / (1 - alpha);
        real dchannel_dcolor = alpha * T;
        real dL_dalpha = 0;
        for (int ch = 0; ch < NUM_CHANNELS; ch++) {
          real c = colors[id * NUM_CHANNELS + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          real dL_dchannel = dL_dpixel[ch];
          dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
#pragma omp atomic
          acc->dcolors[id * NUM_CHANNELS + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }
        real c_d = st->depths[id];
        accum_depth_rec = last_alpha * last_depth + (1 - last_alpha) * accum_depth_rec;
        last_depth = c_d;
        dL_dalpha += (c_d - accum_depth_rec) * dL_dpixel_depth;
#pragma omp atomic
        acc->ddepths[id] += (double)(dchannel_dcolor * dL_dpixel_depth);
        dL_dalpha *= T;
        last_alpha = alpha;
        real bg_dot_dpixel = 0;
        for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += in->background[i] * dL_dpixel[i];
        dL_dalpha += (-T_final / (1 - alpha)) * bg_dot_dpixel;
        real dL_dG = co[3] * dL_dalpha;
        real gdx = G * dx, gdy = G * dy;
        real dG_ddelx = -gdx * co[0] - gdy * co[1];
        real dG_ddely = -gdy * co[2] - gdx * co[1];
#pragma omp atomic
        acc->dmean2D[2 * id + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
#pragma omp atomic
        acc->dmean2D[2 * id + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
#pragma omp atomic
        acc->dconic[3 * id + 0] += (double)((real)-0.5 * gdx * dx * dL_dG);
#pragma omp atomic
        acc->dconic[3 * id + 1] += (double)((real)-0.5 * gdx * dy * dL_dG);
#pragma omp atomic
        acc->dconic[3 * id + 2] += (double)((real)-0.5 * gdy * dy * dL_dG);
#pragma omp atomic
        acc->dopacity[id] += (double)(G * dL_dalpha);
      }
    }
}

/* backward.cu:144-274 */
static void computeCov2D_bw(const orc_inputs* in, int idx, const real* cov3D, real fx, real fy, const real* dL_dconic4,
                            real* dL_dmean, real* dL_dcov) {
  cov2d_ctx c;
  cov2D_common(in->means3D + 3 * idx, fx, fy, in->tan_fovx, in->tan_fovy, cov3D, in->viewmatrix, &c);
  const real x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
  const real y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
  real dcx = dL_dconic4[0], dcy = dL_dconic4[1], dcz = dL_dconic4[3];
  real a = c.cov.m[0][0] + (real)0.3, b = c.cov.m[0][1], cc = c.cov.m[1][1] + (real)0.3;
  real denom = a * cc - b * b;
  real dL_da = 0, dL_db = 0, dL_dc = 0;
  real denom2inv = 1 / ((denom * denom) + (real)0.0000001);
  const mat3* T = &c.T; const mat3* Vrk = &c.Vrk; const mat3* Wm = &c.W;
  if (denom2inv != 0) {
    dL_da = denom2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcz);
    dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * cc) * dcx);
    dL_db = denom2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
    dL_dcov[0] = (T->m[0][0] * T->m[0][0] * dL_da + T->m[0][0] * T->m[1][0] * dL_db + T->m[1][0] * T->m[1][0] * dL_dc);
    dL_dcov[3] = (T->m[0][1] * T->m[0][1] * dL_da + T->m[0][1] * T->m[1][1] * dL_db + T->m[1][1] * T->m[1][1] * dL_dc);
    dL_dcov[5] = (T->m[0][2] * T->m[0][2] * dL_da + T->m[0][2] * T->m[1][2] * dL_db + T->m[1][2] * T->m[1][2] * dL_dc);
    dL_dcov[1] = 2 * T->m[0][0] * T->m[0][1] * dL_da + (T->m[0][0] * T->m[1][1] + T->m[0][1] * T->m[1][0]) * dL_db + 2 * T->m[1][0] * T->m[1][1] * dL_dc;
    dL_dcov[2] = 2 * T->m[0][0] * T->m[0][2] * dL_da + (T->m[0][0] * T->m[1][2] + T->m[0][2] * T->m[1][0]) * dL_db + 2 * T->m[1][0] * T->m[1][2] * dL_dc;
    dL_dcov[4] = 2 * T->m[0][2] * T->m[0][1] * dL_da + (T->m[0][1] * T->m[1][2] + T->m[0][2] * T->m[1][1]) * dL_db + 2 * T->m[1][1] * T->m[1][2] * dL_dc;
  } else {
    for (int i = 0; i < 6; i++) dL_dcov[i] = 0;
  }
#define TM(c_, r_) T->m[c_][r_]
#define VM(c_, r_) Vrk->m[c_][r_]
  real dL_dT00 = 2 * (TM(0,0) * VM(0,0) + TM(0,1) * VM(0,1) + TM(0,2) * VM(0,2)) * dL_da + (TM(1,0) * VM(0,0) + TM(1,1) * VM(0,1) + TM(1,2) * VM(0,2)) * dL_db;
  real dL_dT01 = 2 * (TM(0,0) * VM(1,0) + TM(0,1) * VM(1,1) + TM(0,2) * VM(1,2)) * dL_da + (TM(1,0) * VM(1,0) + TM(1,1) * VM(1,1) + TM(1,2) * VM(1,2)) * dL_db;
  real dL_dT02 = 2 * (TM(0,0) * VM(2,0) + TM(0,1) * VM(2,1) + TM(0,2) * VM(2,2)) * dL_da + (TM(1,0) * VM(2,0) + TM(1,1) * VM(2,1) + TM(1,2) * VM(2,2)) * dL_db;
  real dL_dT10 = 2 * (TM(1,0) * VM(0,0) + TM(1,1) * VM(0,1) + TM(1,2) * VM(0,2)) * dL_dc + (TM(0,0) * VM(0,0) + TM(0,1) * VM(0,1) + TM(0,2) * VM(0,2)) * dL_db;
  real dL_dT11 = 2 * (TM(1,0) * VM(1,0) + TM(1,1) * VM(1,1) + TM(1,2) * VM(1,2)) * dL_dc + (TM(0,0) * VM(1,0) + TM(0,1) * VM(1,1) + TM(0,2) * VM(1,2)) * dL_db;
  real dL_dT12 = 2 * (TM(1,0) * VM(2,0) + TM(1,1) * VM(2,1) + TM(1,2) * VM(2,2)) * dL_dc + (TM(0,0) * VM(2,0) + TM(0,1) * VM(2,1) + TM(0,2) * VM(2,2)) * dL_db;
#undef TM
#undef VM
  real dL_dJ00 = Wm->m[0][0] * dL_dT00 + Wm->m[0][1] * dL_dT01 + Wm->m[0][2] * dL_dT02;
  real dL_dJ02 = Wm->m[2][0] * dL_dT00 + Wm->m[2][1] * dL_dT01 + Wm->m[2][2] * dL_dT02;
  real dL_dJ11 = Wm->m[1][0] * dL_dT10 + Wm->m[1][1] * dL_dT11 + Wm->m[1][2] * dL_dT12;
  real dL_dJ12 = Wm->m[2][0] * dL_dT10 + Wm->m[2][1] * dL_dT11 + Wm->m[2][2] * dL_dT12;
  real tz = 1 / c.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
  real dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;
  real dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
  real dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * c.t[0]) * tz3 * dL_dJ02 + (2 * fy * c.t[1]) * tz3 * dL_dJ12;
  const real* V = in->viewmatrix; /* transformVec4x3Transpose, auxiliary.h:89-97 */
  dL_dmean[0] = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
  dL_dmean[1] = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
  dL_dmean[2] = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;
}

/* backward.cu:20-139 */
static void computeColorFromSH_bw(const orc_inputs* in, int idx, const uint8_t* clamped, const real* dL_dcolor,
                                  real* dL_dmeans, real* dL_dshs) {
  const int deg = in->D, max_coeffs = in->M;
  const real* means = in->means3D; const real* campos = in->cam_pos;
  real dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
  real len = rsqrt_(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  real x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
  const real* sh = in->shs + (size_t)idx * max_coeffs * 3;
  real* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
  real dL_dRGB[3];
  for (int c = 0; c < 3; c++) dL_dRGB[c] = dL_dcolor[3 * idx + c] * (clamped[3 * idx + c] ? 0 : 1);
  real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k) sh[(k) * 3 + c]
#define DSH(k, v) dL_dsh[(k) * 3 + c] = (v) * dL_dRGB[c]
  for (int c = 0; c < 3; c++) {
    DSH(0, SH_C0);
    if (deg > 0) {
      DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
      dRGBdx[c] = -SH_C1 * SH(3); dRGBdy[c] = -SH_C1 * SH(1); dRGBdz[c] = SH_C1 * SH(2);
      if (deg > 1) {
        real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2 * zz - xx - yy));
        DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
        dRGBdx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2 * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2 * x * SH(8);
        dRGBdy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2 * -y * SH(6) + SH_C2[4] * 2 * -y * SH(8);
        dRGBdz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2 * 2 * z * SH(6) + SH_C2[3] * x * SH(7);
        if (deg > 2) {
          DSH(9, SH_C3[0] * y * (3 * xx - yy)); DSH(10, SH_C3[1] * xy * z); DSH(11, SH_C3[2] * y * (4 * zz - xx - yy));
          DSH(12, SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy)); DSH(13, SH_C3[4] * x * (4 * zz - xx - yy));
          DSH(14, SH_C3[5] * z * (xx - yy)); DSH(15, SH_C3[6] * x * (xx - 3 * yy));
          dRGBdx[c] += (SH_C3[0] * SH(9) * 3 * 2 * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2 * xy +
                        SH_C3[3] * SH(12) * -3 * 2 * xz + SH_C3[4] * SH(13) * (-3 * xx + 4 * zz - yy) +
                        SH_C3[5] * SH(14) * 2 * xz + SH_C3[6] * SH(15) * 3 * (xx - yy));
          dRGBdy[c] += (SH_C3[0] * SH(9) * 3 * (xx - yy) + SH_C3[1] * SH(10) * xz + SH_C3[2] * SH(11) * (-3 * yy + 4 * zz - xx) +
                        SH_C3[3] * SH(12) * -3 * 2 * yz + SH_C3[4] * SH(13) * -2 * xy + SH_C3[5] * SH(14) * -2 * yz +
                        SH_C3[6] * SH(15) * -3 * 2 * xy);
          dRGBdz[c] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4 * 2 * yz + SH_C3[3] * SH(12) * 3 * (2 * zz - xx - yy) +
                        SH_C3[4] * SH(13) * 4 * 2 * xz + SH_C3[5] * SH(14) * (xx - yy));
        }
      }
    }
  }
#undef SH
#undef DSH
  real dL_ddir[3] = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                     dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                     dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
  /* dnormvdv, auxiliary.h:107-117 */
  const real* v = dir_orig; const real* dv = dL_ddir;
  real sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  real invsum32 = 1 / rsqrt_(sum2 * sum2 * sum2);
  dL_dmeans[3 * idx + 0] += ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  dL_dmeans[3 * idx + 1] += (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  dL_dmeans[3 * idx + 2] += (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* backward.cu:278-341 */
static void computeCov3D_bw(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds,
                            real* dL_dscales, real* dL_drots) {
  real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
             {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
             {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}}};
  mat3 S; memset(&S, 0, sizeof S);
  real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
  mat3 M = mat3_mul(&S, &R);
  const real* d = dL_dcov3Ds + 6 * idx;
  mat3 dL_dSigma = {{{d[0], (real)0.5 * d[1], (real)0.5 * d[2]}, {(real)0.5 * d[1], d[3], (real)0.5 * d[4]}, {(real)0.5 * d[2], (real)0.5 * d[4], d[5]}}};
  mat3 M2; for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2 * M.m[c][rr]; /* 2.0f * M */
  mat3 dL_dM = mat3_mul(&M2, &dL_dSigma);
  mat3 Rt = mat3_T(&R), dL_dMt = mat3_T(&dL_dM);
  real* ds = dL_dscales + 3 * idx;
  for (int k = 0; k < 3; k++) /* glm::dot(Rt[k], dL_dMt[k]) = x*x + y*y + z*z summed left to right */
    ds[k] = Rt.m[k][0] * dL_dMt.m[k][0] + Rt.m[k][1] * dL_dMt.m[k][1] + Rt.m[k][2] * dL_dMt.m[k][2];
  for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dL_dMt.m[k][rr] *= s[k];
#define D(c_, r_) dL_dMt.m[c_][r_]
  real* dq = dL_drots + 4 * idx;
  dq[0] = 2 * z * (D(0,1) - D(1,0)) + 2 * y * (D(2,0) - D(0,2)) + 2 * x * (D(1,2) - D(2,1));
  dq[1] = 2 * y * (D(1,0) + D(0,1)) + 2 * z * (D(2,0) + D(0,2)) + 2 * r * (D(1,2) - D(2,1)) - 4 * x * (D(2,2) + D(1,1));
  dq[2] = 2 * x * (D(1,0) + D(0,1)) + 2 * r * (D(2,0) - D(0,2)) + 2 * z * (D(1,2) + D(2,1)) - 4 * y * (D(2,2) + D(0,0));
  dq[3] = 2 * r * (D(0,1) - D(1,0)) + 2 * x * (D(2,0) + D(0,2)) + 2 * y * (D(1,2) + D(2,1)) - 4 * z * (D(1,1) + D(0,0));
#undef D
}

/* CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:343-444) + RasterizeGaussiansBackwardCUDA
 * (rasterize_points.cu:119-202).  All dL_* outputs must be zero-filled by the caller (:154-163).
 * Layouts as the reference: dL_dmean2D [P,3], dL_dconic [P,2,2] (.x,.y,.w used), dL_dopacity [P],
 * dL_dcolor [P,3], dL_ddepth [P], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4]. */
void orc_backward(const orc_inputs* in, const orc_state* st, const int* radii, const real* dL_dpix,
                  const real* dL_dpix_depth, real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor,
                  real* dL_ddepth, real* dL_dmean3D, real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot) {
  const int P = in->P, W = in->W, H = in->H;
  if (P == 0) return;
  const real fy = H / (2 * in->tan_fovy), fx = W / (2 * in->tan_fovx);
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const real* colors = in->colors_precomp ? in->colors_precomp : st->rgb;
  bw_acc acc;
  acc.dmean2D = (double*)calloc((size_t)P * 2, sizeof(double));
  acc.dconic = (double*)calloc((size_t)P * 3, sizeof(double));
  acc.dopacity = (double*)calloc((size_t)P, sizeof(double));
  acc.dcolors = (double*)calloc((size_t)P * 3, sizeof(double));
  acc.ddepths = (double*)calloc((size_t)P, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) render_tile_bw(in, st, colors, tx, ty, gx, dL_dpix, dL_dpix_depth, &acc);
  for (int i = 0; i < P; i++) {
    dL_dmean2D[3 * i] = (real)acc.dmean2D[2 * i]; dL_dmean2D[3 * i + 1] = (real)acc.dmean2D[2 * i + 1];
    dL_dconic[4 * i] = (real)acc.dconic[3 * i]; dL_dconic[4 * i + 1] = (real)acc.dconic[3 * i + 1];
    dL_dconic[4 * i + 3] = (real)acc.dconic[3 * i + 2];
    dL_dopacity[i] = (real)acc.dopacity[i];
    for (int c = 0; c < 3; c++) dL_dcolor[3 * i + c] = (real)acc.dcolors[3 * i + c];
    dL_ddepth[i] = (real)acc.ddepths[i];
  }
  free(acc.dmean2D); free(acc.dconic); free(acc.dopacity); free(acc.dcolors); free(acc.ddepths);
  /* BACKWARD::preprocess, backward.cu:592-658 */
  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    const real* cov3D = in->cov3D_precomp ? in->cov3D_precomp + 6 * idx : st->cov3D + 6 * idx;
    computeCov2D_bw(in, idx, cov3D, fx, fy, dL_dconic + 4 * idx, dL_dmean3D + 3 * idx, dL_dcov3D + 6 * idx);
  }
  for (int idx = 0; idx < P; idx++) { /* preprocessCUDA bwd, backward.cu:346-412 */
    if (!(radii[idx] > 0)) continue;
    const real* m = in->means3D + 3 * idx; const real* proj = in->projmatrix; const real* view = in->viewmatrix;
    real m_hom[4]; transformPoint4x4(m, proj, m_hom);
    real m_w = 1 / (m_hom[3] + (real)0.0000001);
    real mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
    real mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
    real g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
    real dm[3];
    dm[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dm[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dm[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    for (int k = 0; k < 3; k++) dL_dmean3D[3 * idx + k] += dm[k];
    real mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
    real dm2[3] = {(view[2] - view[3] * mul3) * dL_ddepth[idx], (view[6] - view[7] * mul3) * dL_ddepth[idx],
                   (view[10] - view[11] * mul3) * dL_ddepth[idx]};
    for (int k = 0; k < 3; k++) dL_dmean3D[3 * idx + k] += dm2[k];
    if (in->shs) computeColorFromSH_bw(in, idx, st->clamped, dL_dcolor, dL_dmean3D, dL_dsh);
    if (in->scales) computeCov3D_bw(idx, in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
  }
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
