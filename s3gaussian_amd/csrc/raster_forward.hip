// Forward half of the MI355X-native differentiable Gaussian rasterizer (gfx950, wave64).
//
// What it computes is the reference's CudaRasterizer::Rasterizer::forward
// (RAST/cuda_rasterizer/rasterizer_impl.cu:198-339); how it computes it is not:
//
//   reference (CUDA)                                   this file (CDNA4)
//   -------------------------------------------------  ------------------------------------------------------
//   preprocessCUDA  (forward.cu:155-256)               preprocess_kernel: same per-Gaussian math (bit-exact fp32,
//                                                      contraction off) + per-tile instance histogram
//   InclusiveSum over P + duplicateWithKeys + global   ATOMIC-FREE single-pass multisplit: device-scope atomics on
//   64-bit radix sort of all R instances + range scan   MI355X execute memory-side (~2-6 G/s measured), so instances are
//   (rasterizer_impl.cu:278-319)                        binned with per-workgroup histograms of ALL tiles held in LDS
//                                                      (27 KB for 6.7k tiles; 160 KB LDS allows ~38k tiles), a
//                                                      [workgroup][tile] offset table and LDS cursors; then a per-tile
//                                                      bitonic sort of (depth bits, index) keys staged in LDS.  Final
//                                                      order == the reference's stable (tile, depth) radix sort: ties
//                                                      in depth resolve by ascending Gaussian index.
//   renderCUDA (forward.cu:261-379)                    blend_forward_kernel: 16x16 tile = 4 wave64, Gaussian
//                                                      attributes (incl. colour+depth) staged in LDS, conic
//                                                      pre-scaled so alpha = o * exp2(q) is one v_exp_f32
#include "geom_math.hpp"

#include <stdarg.h>

#include <vector>

namespace s3g {

static thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// Exact (tile, Gaussian) culling at binning time (geom_math.hpp::tile_can_contribute); on by default, switchable so the
// instance lists can be compared bit for bit with the reference's bounding-square binning.
static bool g_exact_cull = true;
static int g_max_tiles_lds = MAX_TILES_LDS;  // band size of the binning histogram; lowered only by the tests (s3g_raster_set_bin_band)

// ---- in-library kernel timing -----------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; double instances, pixels; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof[S3G_PROFILE_IDS];
static hipEvent_t g_prof_pending[S3G_PROFILE_IDS];
void profile_begin(int id, hipStream_t stream) {
  if (!g_prof_on || id < 0 || id >= S3G_PROFILE_IDS) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, stream);
  g_prof_pending[id] = e;
}
void profile_end(int id, hipStream_t stream, double instances, double pixels) {
  if (!g_prof_on || id < 0 || id >= S3G_PROFILE_IDS) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, stream);
  g_prof[id].push_back(ProfRec{g_prof_pending[id], e, instances, pixels});
}

// =========================================================================================================
// 1. Per-Gaussian preprocess (EWA projection).  HBM-bound: 56 B read + ~68 B written per Gaussian.
//    Bit-exact with the fp32 oracle: contraction is disabled so every op rounds once, in source order
//    (glm mat3 products expanded in glm's summation order, type_mat3x3.inl:486-518).
// =========================================================================================================
// forward.cu:20-71
__device__ __forceinline__ float3 sh_to_rgb(int idx, int deg, int M, const float3 pos, const float3 campos,
                                            const float* __restrict__ shs, uint8_t* __restrict__ clamped) {
  float3 dir = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
  const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
  const float x = dir.x / len, y = dir.y / len, z = dir.z / len;
  const float* sh = shs + (size_t)idx * M * 3;
  float res[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k)*3 + c]
    float v = SH_C0 * SH(0);
    if (deg > 0) {
      v = v - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        v = v + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
            SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          v = v + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
              SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
        }
      }
    }
#undef SH
    v += 0.5f;
    clamped[3 * idx + c] = (v < 0.f);
    res[c] = fmaxf(v, 0.f);
  }
  return make_float3(res[0], res[1], res[2]);
}

constexpr int BIG_RECT = 32;  // rects of more tiles are walked by a whole wave in the binning kernels

struct PreprocessArgs {
  int P, D, M, W, H, gx, gy;
  int cull;  // exact (tile, Gaussian) culling: tile_mask is filled here for rects of <= BIG_RECT tiles
  const float* means3D;
  const float* scales;
  float scale_modifier;
  const float* rotations;
  const float* opacities;
  const float* shs;
  const float* cov3D_precomp;
  const float* colors_precomp;
  const float* viewmatrix;
  const float* projmatrix;
  const float* cam_pos;
  float tan_fovx, tan_fovy, focal_x, focal_y;
  int prefiltered;
  int* radii;
  GeomState g;
  uint32_t* ctrl;
};

__global__ void __launch_bounds__(256) preprocess_kernel(const PreprocessArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.P) return;
  a.radii[idx] = 0;
  a.g.rect[idx] = make_ushort4(0, 0, 0, 0);

  const float3 p = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
  const float3 p_view = xform_4x3(p, a.viewmatrix);
  if (p_view.z <= 0.2f) {  // in_frustum, auxiliary.h:154
    if (a.prefiltered) atomicOr(&a.ctrl[2], 1u);
    return;
  }
  const float4 p_hom = xform_4x4(p, a.projmatrix);
  const float p_w = 1.0f / (p_hom.w + 0.0000001f);
  const float2 p_proj = make_float2(p_hom.x * p_w, p_hom.y * p_w);

  float cov3D[6];
  if (a.cov3D_precomp != nullptr) {
#pragma unroll
    for (int k = 0; k < 6; k++) cov3D[k] = a.cov3D_precomp[6 * (size_t)idx + k];
  } else {
    const float3 s = make_float3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
    const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
    cov3d_from_scale_rot(s, a.scale_modifier, q, cov3D);
#pragma unroll
    for (int k = 0; k < 6; k++) a.g.cov3D[6 * (size_t)idx + k] = cov3D[k];
  }
  const Cov2DCtx cc = cov2d_common(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov3D, a.viewmatrix);
  const float3 cov = make_float3(cc.cov.m[0][0] + 0.3f, cc.cov.m[0][1], cc.cov.m[1][1] + 0.3f);
  const float det = cov.x * cov.z - cov.y * cov.y;
  if (det == 0.0f) return;
  const float det_inv = 1.f / det;
  const float3 conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
  const float mid = 0.5f * (cov.x + cov.z);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  // ndc2Pix (auxiliary.h:41-44) is double arithmetic in the reference (double literals)
  const float px = (float)((((double)p_proj.x + 1.0) * a.W - 1.0) * 0.5);
  const float py = (float)((((double)p_proj.y + 1.0) * a.H - 1.0) * 0.5);
  // getRect (auxiliary.h:46-56)
  const int r = (int)my_radius;
  const int rx0 = min(a.gx, max(0, (int)((px - r) / TILE_X)));
  const int ry0 = min(a.gy, max(0, (int)((py - r) / TILE_Y)));
  const int rx1 = min(a.gx, max(0, (int)((px + r + TILE_X - 1) / TILE_X)));
  const int ry1 = min(a.gy, max(0, (int)((py + r + TILE_Y - 1) / TILE_Y)));
  if ((rx1 - rx0) * (ry1 - ry0) == 0) return;

  if (a.colors_precomp == nullptr) {
    const float3 cp = make_float3(a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]);
    const float3 c = sh_to_rgb(idx, a.D, a.M, p, cp, a.shs, a.g.clamped);
    a.g.rgb[3 * (size_t)idx + 0] = c.x;
    a.g.rgb[3 * (size_t)idx + 1] = c.y;
    a.g.rgb[3 * (size_t)idx + 2] = c.z;
  }
  a.g.depths[idx] = p_view.z;
  a.radii[idx] = r;
  a.g.means2D[idx] = make_float2(px, py);
  a.g.conic_opacity[idx] = make_float4(conic.x, conic.y, conic.z, a.opacities[idx]);
  a.g.rect[idx] = make_ushort4((unsigned short)rx0, (unsigned short)ry0, (unsigned short)rx1, (unsigned short)ry1);
  // exact tile culling (geom_math.hpp::tile_can_contribute) evaluated HERE, one well-occupied thread per Gaussian; the two
  // binning passes (few fat workgroups, latency-bound) only replay the mask
  if (a.cull && (rx1 - rx0) * (ry1 - ry0) <= BIG_RECT) {
    const TileCull tc = tile_cull_prepare(make_float2(px, py), make_float4(conic.x, conic.y, conic.z, a.opacities[idx]));
    uint32_t mask = 0u, bit = 1u;
    for (int y = ry0; y < ry1; y++)
      for (int x = rx0; x < rx1; x++, bit <<= 1)
        if (tile_can_contribute(tc, x, y, a.W, a.H)) mask |= bit;
    a.g.tile_mask[idx] = mask;
  }
}

// =========================================================================================================
// 2. Atomic-free binning ("multisplit" of the R instances into tiles*).
//    bin_count : NB fat workgroups, each owns a contiguous chunk of Gaussians and histograms its instances over
//                ALL tiles in LDS (ds_add_u32), then stores its row of table[NB][tiles] + its chunk total.
//    bin_scan  : per tile, exclusive prefix over the NB workgroups (in place) and the tile total.
//    scan_tiles: exclusive scan over tiles -> ranges, R, longest list; exclusive scan of chunk totals.
//    bin_write : same walk as bin_count; LDS cursors start at ranges[t].x + table[wg][t]; every instance key is
//                stored at a private slot.  Also emits gauss_off[g] = exclusive scan of tiles_touched (Gaussian order),
//                the address of g's slots in the instance->position map used by the backward gather.
//    Rects wider than BIG_RECT tiles are walked by the whole wave instead of one lane.
//    Tile grids larger than the LDS histogram (MAX_TILES_LDS) are processed in BANDS of consecutive tiles: both walks are
//    launched once per band and only handle the instances whose tile lies in it (an 8K image is 4 bands).
//    (*) order inside a tile is arbitrary here; the per-tile sort fixes it.
// =========================================================================================================
#ifndef S3G_BIN_THREADS
#define S3G_BIN_THREADS 512
#endif
#ifndef S3G_BIN_PREFETCH
#define S3G_BIN_PREFETCH 1
#endif
// Both walks are chains of dependent global loads (rect -> tile mask / depth) in front of LDS work, run by 2 workgroups per CU
// (the histogram of ALL tiles lives in LDS: more workgroups would mean more table rows for bin_scan).  What hides the latency is
// (i) more waves per workgroup -- the histogram is shared, so threads are free -- and (ii) the next step's three loads requested
// before this step's walk; the block-wide scan of bin_write therefore synchronises on LDS only (an ordinary __syncthreads()
// would also wait for the prefetch).
constexpr int BIN_THREADS = S3G_BIN_THREADS, BIN_WAVES = BIN_THREADS / 64, BIN_SCRATCH = 2 * BIN_WAVES + 8;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BinArgs {
  int P, gx, tiles, chunk;         // chunk = Gaussians per workgroup
  int tile_lo, tile_n;             // the band of tiles this launch handles: [tile_lo, tile_lo + tile_n)
  const ushort4* rect;
  const float* depths;
  uint32_t* table;                 // [NB][tiles]
  uint32_t* chunk_total;           // [NB] instances emitted by each workgroup; after scan_tiles: exclusive prefix
  const uint2* ranges;             // bin_write only
  uint64_t* keys;                  // bin_write only
  uint32_t* gauss_off;             // bin_write only
  // exact (tile, Gaussian) culling (geom_math.hpp::tile_can_contribute); cull == 0: the reference's bounding square
  int cull, W, H;
  const float2* means2D;
  const float4* conic_opacity;
  const uint32_t* tile_mask;       // from preprocess_kernel (rects of <= BIG_RECT tiles)
  const uint32_t* ctrl;            // bin_write only: ctrl[4] != 0 = the speculative arena capacity was exceeded, write nothing
};

template <bool WRITE>
__global__ void __launch_bounds__(BIN_THREADS) bin_kernel(const BinArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [tiles] histogram / cursors, then BIN_SCRATCH words
  uint32_t* cell = lds;
  uint32_t* wsum = lds + a.tile_n;  // [2][BIN_WAVES] wave totals, alternating by step: ONE barrier per step
  if (WRITE && a.ctrl[4] != 0u) return;  // host-asynchronous forward: the instances do not fit the arena (see scan_tiles_kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* trow = a.table + (size_t)blockIdx.x * a.tiles;
  for (int i = tid; i < a.tile_n; i += BIN_THREADS) cell[i] = WRITE ? a.ranges[a.tile_lo + i].x + trow[a.tile_lo + i] : 0u;
  const bool first_band = a.tile_lo == 0;  // per-Gaussian outputs (gauss_off, chunk totals) are produced once
  uint32_t carry = WRITE ? a.chunk_total[blockIdx.x] : 0u;  // exclusive prefix of previous workgroups' instances
  uint32_t my_total = 0;
  __syncthreads();
  const int g0 = blockIdx.x * a.chunk, g1 = min(a.P, g0 + a.chunk);
  // what a step needs of its Gaussian: rect, the mask of tiles that survive the exact cull (meaningful for 0 < area <= BIG_RECT
  // only; whatever the word holds otherwise is not used) and the depth bits of the key
  ushort4 r_next = make_ushort4(0, 0, 0, 0);
  uint32_t mask_next = 0xffffffffu, depth_next = 0u;
  if (S3G_BIN_PREFETCH && g0 + tid < g1) {
    r_next = a.rect[g0 + tid];
    if (a.cull) mask_next = a.tile_mask[g0 + tid];
    if (WRITE) depth_next = __float_as_uint(a.depths[g0 + tid]);
  }
  int step = 0;
  for (int base = g0; base < g1; base += BIN_THREADS, step ^= 1) {
    const int g = base + tid;
    ushort4 r = make_ushort4(0, 0, 0, 0);
    uint32_t mask = 0xffffffffu, dbits = 0u;
    if (S3G_BIN_PREFETCH) {
      r = r_next; mask = mask_next; dbits = depth_next;
      const int gn = g + BIN_THREADS;
      r_next = make_ushort4(0, 0, 0, 0);
      if (gn < g1) {
        r_next = a.rect[gn];
        if (a.cull) mask_next = a.tile_mask[gn];
        if (WRITE) depth_next = __float_as_uint(a.depths[gn]);
      }
    } else if (g < g1) {
      r = a.rect[g];
    }
    const int w = (int)r.z - (int)r.x, h = (int)r.w - (int)r.y;
    const uint32_t area = (w > 0 && h > 0) ? (uint32_t)(w * h) : 0u;
    uint64_t key = 0;
    TileCull tc;
    tc.verdict = 1;
    if (a.cull && area > BIG_RECT) tc = tile_cull_prepare(a.means2D[g], a.conic_opacity[g]);  // small rects: mask replay
    if (WRITE) {
      // block-wide exclusive scan of area -> gauss_off (slots are counted per rect tile whether or not it survives)
      uint32_t incl = area;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += t;
      }
      uint32_t* ws = wsum + step * BIN_WAVES;
      if (lane == 63) ws[wave] = incl;
      lds_barrier();
      uint32_t wbase = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < BIN_WAVES; k++) {
        const uint32_t v = ws[k];
        if (k < wave) wbase += v;
        tot += v;
      }
      if (g < g1 && first_band) a.gauss_off[g] = carry + wbase + incl - area;
      carry += tot;
      if (area) key = ((uint64_t)(S3G_BIN_PREFETCH ? dbits : __float_as_uint(a.depths[g])) << 32) | (uint32_t)g;
    } else {
      my_total += area;
    }
    if (area != 0 && area <= BIG_RECT) {
      if (!S3G_BIN_PREFETCH) mask = a.cull ? a.tile_mask[g] : 0xffffffffu;  // computed by preprocess_kernel
      uint32_t bit = 1u;
      for (int y = r.y; y < r.w; y++)
        for (int x = r.x; x < r.z; x++, bit <<= 1) {
          if (!(mask & bit)) continue;
          const uint32_t tb = (uint32_t)(y * a.gx + x - a.tile_lo);
          if (tb >= (uint32_t)a.tile_n) continue;
          const uint32_t pos = atomicAdd(&cell[tb], 1u);
          if (WRITE) a.keys[pos] = key;
        }
    }
    uint64_t big = __ballot(area > BIG_RECT);
    while (big) {  // wave-uniform loop: all 64 lanes walk one large rect together
      const int src = __ffsll((unsigned long long)big) - 1;
      big &= big - 1;
      const int bx = __shfl((int)r.x, src), by = __shfl((int)r.y, src), bw = __shfl(w, src);
      const uint32_t barea = (uint32_t)__shfl((int)area, src);
      const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)key, src), khi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), src);
      TileCull bt;
      bt.a = __shfl(tc.a, src); bt.b = __shfl(tc.b, src); bt.c = __shfl(tc.c, src); bt.inv_a = __shfl(tc.inv_a, src);
      bt.inv_c = __shfl(tc.inv_c, src); bt.budget = __shfl(tc.budget, src); bt.mx = __shfl(tc.mx, src);
      bt.my = __shfl(tc.my, src); bt.verdict = __shfl(tc.verdict, src);
      for (uint32_t k = lane; k < barea; k += 64) {
        const int ty = by + (int)(k / (uint32_t)bw), tx = bx + (int)(k % (uint32_t)bw);
        const uint32_t tb = (uint32_t)(ty * a.gx + tx - a.tile_lo);
        if (tb >= (uint32_t)a.tile_n) continue;
        if (a.cull && !tile_can_contribute(bt, tx, ty, a.W, a.H)) continue;
        const uint32_t pos = atomicAdd(&cell[tb], 1u);
        if (WRITE) a.keys[pos] = ((uint64_t)khi << 32) | klo;
      }
    }
  }
  if (!WRITE) {
    __syncthreads();
    uint32_t* row = a.table + (size_t)blockIdx.x * a.tiles;
    for (int i = tid; i < a.tile_n; i += BIN_THREADS) row[a.tile_lo + i] = cell[i];
    for (int off = 32; off >= 1; off >>= 1) my_total += (uint32_t)__shfl_xor((int)my_total, off);
    if (lane == 0) wsum[wave] = my_total;
    __syncthreads();
    if (tid == 0 && first_band) {
      uint32_t tot = 0;
      for (int k = 0; k < BIN_WAVES; k++) tot += wsum[k];
      a.chunk_total[blockIdx.x] = tot;
    }
  }
}

// Exclusive prefix over the binning workgroups, per tile (coalesced across tiles).  A thread that walks all nb rows of its tile
// is a chain of nb / 32 dependent round trips on 27 workgroups (24 us at 6700 tiles, nb = 512: 1.1 TB/s); the rows are therefore
// split into SCAN_PARTS contiguous parts, one WAVE per part and 64 tiles per workgroup: every part sums its rows (32 independent
// loads in flight), the part sums meet in LDS, and a second sweep over the same rows (L2-resident by then) writes the prefixes.
#ifndef S3G_SCAN_PARTS
#define S3G_SCAN_PARTS 8
#endif
constexpr int SCAN_PARTS = S3G_SCAN_PARTS;
__global__ void __launch_bounds__(64 * SCAN_PARTS) bin_scan_kernel(int tiles, int nb, uint32_t* __restrict__ table,
                                                                    uint32_t* __restrict__ tile_count) {
  constexpr int INFLIGHT = 32;
  __shared__ uint32_t psum[SCAN_PARTS][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const bool live = t < tiles;
  const int rows = (nb + SCAN_PARTS - 1) / SCAN_PARTS, b0 = part * rows, b1 = min(nb, b0 + rows);
  uint32_t sum = 0;
  if (live)
    for (int b = b0; b < b1; b += INFLIGHT) {
      uint32_t v[INFLIGHT];
#pragma unroll
      for (int k = 0; k < INFLIGHT; k++) v[k] = (b + k < b1) ? table[(size_t)(b + k) * tiles + t] : 0u;
#pragma unroll
      for (int k = 0; k < INFLIGHT; k++) sum += v[k];
    }
  psum[part][lane] = sum;
  __syncthreads();
  if (!live) return;
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PARTS; k++) {
    const uint32_t x = psum[k][lane];
    if (k < part) run += x;
    total += x;
  }
  for (int b = b0; b < b1; b += INFLIGHT) {
    uint32_t v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) v[k] = (b + k < b1) ? table[(size_t)(b + k) * tiles + t] : 0u;
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) {
      if (b + k < b1) table[(size_t)(b + k) * tiles + t] = run;
      run += v[k];
    }
  }
  if (part == 0) tile_count[t] = total;
}

// Inclusive scan of one value per thread over the 1024 threads: shuffles inside a wave, the 16 wave totals through LDS (three
// barriers; a Hillis-Steele scan in LDS costs twenty, and scan_tiles_kernel is ONE workgroup on an otherwise idle device: 13 -> 4 us).
__device__ __forceinline__ uint32_t block_inclusive_scan_waves(uint32_t v, uint32_t* wtot, int tid, uint32_t* total) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();   // wtot may still be read from a previous call
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t x = wtot[k];
    if (k < wave) wbase += x;
    tot += x;
  }
  *total = tot;
  return wbase + incl;
}

// Exclusive scan over tiles: ranges[t] = [start, end); ctrl[0] = R, ctrl[1] = longest tile list, ctrl[3] = slots; also turns
// chunk_total[nb] into its exclusive prefix.  One 1024-thread workgroup; tiles is O(10^3..10^4), nb <= 1024.
// Host-asynchronous forward (s3g_raster_forward_async): the binning arena was sized BEFORE this kernel knew R.  cap_R != 0
// turns the capacity check on: if R > cap_R, S > cap_S or the longest list > cap_tile, ctrl[4] = 1, every range is emptied and
// R / S read as 0, so that every later kernel of the forward AND of the backward finds nothing to do (bin_write and the
// per-Gaussian backward also look at ctrl[4] themselves); the true counts stay in ctrl[5..6] for the host, which reads them
// late, without stalling.  *status (optional device word, written on every call): bit 0 = overflow, bit 1 = a Gaussian was
// culled although `prefiltered` was set.
// *sticky (optional device word, s3g_raster_async.sticky_device): set by the call that overflows and then honoured by every
// later call that is handed the same word -- they render nothing either (ctrl[4] = 1, ctrl[7] = 1 "because of an earlier call")
// until the host clears it.  With the guarded optimizer step this freezes the model from the overflowed iteration on, so that the
// host, which learns of the overflow a few iterations late, can raise the capacity, clear the word and RE-ISSUE the iterations
// from the overflowed one: the sequence of (view, optimizer step) pairs the model sees is then the reference's, none dropped.
__global__ void __launch_bounds__(1024) scan_tiles_kernel(int tiles, const uint32_t* __restrict__ tile_count,
                                                          uint2* __restrict__ ranges, uint32_t* __restrict__ ctrl,
                                                          int nb, uint32_t* __restrict__ chunk_total, uint32_t cap_R,
                                                          uint32_t cap_S, uint32_t cap_tile, uint32_t* __restrict__ status,
                                                          uint32_t* __restrict__ sticky) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t wmax[16];
  const int tid = threadIdx.x;
  // thread t owns the tiles [t * per, (t + 1) * per): a serial sum, ONE block scan of the 1024 sums, a serial pass for the ranges
  const int per = (tiles + 1023) / 1024, t0 = tid * per, t1 = min(tiles, t0 + per);
  uint32_t vmax = 0, mine = 0, total, carry;
  for (int i = t0; i < t1; i++) {
    const uint32_t v = tile_count[i];
    vmax = max(vmax, v);
    mine += v;
  }
  {
    uint32_t start = block_inclusive_scan_waves(mine, wtot, tid, &carry) - mine;   // carry = R
    for (int i = t0; i < t1; i++) {
      const uint32_t v = tile_count[i];
      ranges[i] = make_uint2(start, start + v);
      start += v;
    }
  }
  {
    const uint32_t v = tid < nb ? chunk_total[tid] : 0u;
    const uint32_t incl = block_inclusive_scan_waves(v, wtot, tid, &total);
    if (tid < nb) chunk_total[tid] = incl - v;
    if (tid == 0) ctrl[3] = total;  // S: slots = sum of rect areas (== R without culling)
  }
  for (int off = 32; off >= 1; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off));
  if ((tid & 63) == 0) wmax[tid >> 6] = vmax;
  __syncthreads();
  uint32_t m = 0;
  for (int w = 0; w < 16; w++) m = max(m, wmax[w]);
  const bool own = cap_R != 0u && (carry > cap_R || total > cap_S || m > cap_tile);   // `total` = S (last scan above)
  const bool frozen = cap_R != 0u && sticky != nullptr && *sticky != 0u;               // an EARLIER call overflowed (uniform load)
  const bool overflow = own || frozen;
  if (overflow)
    for (int i = tid; i < tiles; i += 1024) ranges[i] = make_uint2(0u, 0u);
  if (tid == 0) {
    ctrl[0] = overflow ? 0u : carry;
    ctrl[1] = m;
    if (overflow) ctrl[3] = 0u;
    ctrl[4] = overflow ? 1u : 0u;
    ctrl[5] = carry;
    ctrl[6] = total;
    ctrl[7] = (frozen && !own) ? 1u : 0u;
    if (sticky && own) *sticky = 1u;
    if (status) *status = (overflow ? 1u : 0u) | ((ctrl[2] & 1u) ? 2u : 0u);
  }
}

// slot_pos[0 .. S) = 0xffffffff ("tile culled") with S read on the device (the asynchronous forward does not know it).
__global__ void __launch_bounds__(256) fill_slots_kernel(uint32_t* __restrict__ slot_pos, const uint32_t* __restrict__ ctrl) {
  const uint32_t S = ctrl[3];
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < S; i += gridDim.x * 256u) slot_pos[i] = 0xffffffffu;
}

// =========================================================================================================
// 4. Per-tile sort of 64-bit (depth bits | index) keys: bitonic network with all comparators ascending
//    (first sub-step of each stage mirrors, i ^ (k-1)), so a non-power-of-two list is handled by skipping
//    comparators whose upper element is past the end (virtual +inf padding never moves).
//    Lists that fit stay in LDS; longer ones run the same network in place in global memory.
// =========================================================================================================
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_network(KeyPtr a, uint32_t n, uint32_t tid, uint32_t nthreads) {
  uint32_t lN = 0;
  while ((1u << lN) < n) lN++;
  const uint32_t half = (1u << lN) >> 1;  // comparators per step
  for (uint32_t lk = 1; lk <= lN; lk++) {
    {  // mirror step: element o of block b against element k-1-o
      const uint32_t k = 1u << lk, lhk = lk - 1, hk = 1u << lhk;
      for (uint32_t c = tid; c < half; c += nthreads) {
        const uint32_t b = c >> lhk, o = c & (hk - 1);
        const uint32_t i = (b << lk) + o, l = (b << lk) + (k - 1 - o);
        if (l < n) {
          const uint64_t x = a[i], y = a[l];
          if (x > y) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
    for (int lj = (int)lk - 2; lj >= 0; lj--) {  // half-cleaners, distance j = 2^lj
      const uint32_t j = 1u << lj;
      for (uint32_t c = tid; c < half; c += nthreads) {
        const uint32_t b = c >> lj, o = c & (j - 1);
        const uint32_t i = (b << (lj + 1)) + o, l = i + j;
        if (l < n) {
          const uint64_t x = a[i], y = a[l];
          if (x > y) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// ---- long lists (round 6): one bucket pass, then small sorts ---------------------------------------------------------------------
// The bitonic network moves every key through LDS log2(n) (log2(n) + 1) / 2 times: 78 steps at 4096 keys -- on a scene whose mean tile
// list is 1800 instances (bench.py's heavy_raster leg: R = 12 M) the per-tile sort was the most expensive kernel of the step (2 x 0.67
// ms).  A list longer than BUCKET_MIN keys is first split into SORT_BINS buckets by the leading bits of (depth bits - smallest depth bits
// of the list) -- monotone in the key, so the buckets are in order and only have to be sorted inside: LDS histogram, one scan, one
// scatter into a second LDS buffer (each an integer atomic per key: WHERE a key lands inside its bucket depends on their order, the
// sorted result does not -- the 64-bit keys of a list are all different) -- and then every key counts the smaller keys of its own
// bucket (<= BUCKET_RANK keys; a bucket holds one or two on average) and goes to bucket start + rank; the rare larger buckets take the
// network, by one wave (<= BUCKET_WAVE keys) or the workgroup.  About ten LDS operations per key instead of a hundred and fifty;
// the result is the same ascending list.
constexpr int SORT_BIN_BITS = 11;
constexpr uint32_t SORT_BINS = 1u << SORT_BIN_BITS, BUCKET_MIN = 512, BUCKET_RANK = 48, BUCKET_WAVE = 512, BUCKET_LIST = 256;
#ifndef S3G_SORT_THREADS_MID
#define S3G_SORT_THREADS_MID 512
#endif
#ifndef S3G_SORT_THREADS_LONG
#define S3G_SORT_THREADS_LONG 1024
#endif
#ifndef S3G_SORT_RANK_DIRECT
#define S3G_SORT_RANK_DIRECT 256
#endif
constexpr uint32_t RANK_DIRECT = S3G_SORT_RANK_DIRECT;   // lists of at most this many keys: one thread per key, rank by counting
constexpr uint32_t SORT_THREADS_MID = S3G_SORT_THREADS_MID, SORT_THREADS_LONG = S3G_SORT_THREADS_LONG;   // workgroup sizes of the two long-list launches
constexpr uint32_t BUCKET_LDS_EXTRA = SORT_BINS * 4 + BUCKET_LIST * 2 * 2;   // bytes behind the two key buffers: cursors, two bucket lists

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the network of bitonic_network run by ONE wave on a short list (no workgroup barrier: the list belongs to this wave)
__device__ __forceinline__ void bitonic_network_wave(uint64_t* a, uint32_t n, uint32_t lane) {
  uint32_t lN = 0;
  while ((1u << lN) < n) lN++;
  const uint32_t half = (1u << lN) >> 1;
  for (uint32_t lk = 1; lk <= lN; lk++) {
    {
      const uint32_t k = 1u << lk, lhk = lk - 1, hk = 1u << lhk;
      for (uint32_t c = lane; c < half; c += 64) {
        const uint32_t b = c >> lhk, o = c & (hk - 1);
        const uint32_t i = (b << lk) + o, l = (b << lk) + (k - 1 - o);
        if (l < n) {
          const uint64_t x = a[i], y = a[l];
          if (x > y) { a[i] = y; a[l] = x; }
        }
      }
      wave_lds_fence();
    }
    for (int lj = (int)lk - 2; lj >= 0; lj--) {
      const uint32_t j = 1u << lj;
      for (uint32_t c = lane; c < half; c += 64) {
        const uint32_t b = c >> lj, o = c & (j - 1);
        const uint32_t i = (b << (lj + 1)) + o, l = i + j;
        if (l < n) {
          const uint64_t x = a[i], y = a[l];
          if (x > y) { a[i] = y; a[l] = x; }
        }
      }
      wave_lds_fence();
    }
  }
}
// A[0, n) holds the list; returns with A[0, n) sorted (B: scratch of n keys).  cur: SORT_BINS words, lists: 2 x BUCKET_LIST uint16.
// nt = 256 ... 1024 threads (the long-list launches bring more waves: the passes below are chains of dependent LDS operations, and a
// workgroup that fills most of a CU's LDS is alone on it).  (n <= 7424 keys: at most n / (BUCKET_RANK + 1) < BUCKET_LIST buckets can be listed.)
__device__ __forceinline__ void bucket_sort_lds(uint64_t* __restrict__ A, uint64_t* __restrict__ B, uint32_t* __restrict__ cur,
                                                uint16_t* __restrict__ lists, uint32_t n, uint32_t tid, uint32_t nt) {
  __shared__ uint32_t red[32];
  __shared__ uint32_t nlist[2];
  const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = nt >> 6;
  uint32_t mn = 0xffffffffu, mx = 0u;
  for (uint32_t i = tid; i < n; i += nt) {
    const uint32_t h = (uint32_t)(A[i] >> 32);
    mn = min(mn, h); mx = max(mx, h);
  }
  for (int off = 32; off >= 1; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  }
  if (lane == 0) { red[wave] = mn; red[16 + wave] = mx; }
  if (tid < 2) nlist[tid] = 0u;
  for (uint32_t b = tid; b < SORT_BINS; b += nt) cur[b] = 0u;
  __syncthreads();
  mn = red[0]; mx = red[16];
  for (uint32_t w = 1; w < nwaves; w++) { mn = min(mn, red[w]); mx = max(mx, red[16 + w]); }
  const uint32_t range = mx - mn;
  const int sh = range ? max(0, 32 - (int)__clz(range) - SORT_BIN_BITS) : 0;     // (range >> sh) < SORT_BINS
  for (uint32_t i = tid; i < n; i += nt) atomicAdd(&cur[((uint32_t)(A[i] >> 32) - mn) >> sh], 1u);
  __syncthreads();
  {  // exclusive scan of the SORT_BINS counters: thread t < 256 owns PER consecutive bins (the other waves only keep the barriers company)
    constexpr int PER = SORT_BINS / 256;
    uint32_t c[PER], sum = 0;
    if (tid < 256) {
#pragma unroll
      for (int k = 0; k < PER; k++) { c[k] = cur[PER * tid + k]; sum += c[k]; }
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= (uint32_t)off) incl += t;
    }
    __syncthreads();               // red[] is read above
    if (lane == 63 && wave < 4) red[wave] = incl;
    __syncthreads();
    if (tid < 256) {
      uint32_t excl = incl - sum;
      for (uint32_t w = 0; w < wave; w++) excl += red[w];
#pragma unroll
      for (int k = 0; k < PER; k++) { cur[PER * tid + k] = excl; excl += c[k]; }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += nt) {
    const uint64_t k = A[i];
    B[atomicAdd(&cur[((uint32_t)(k >> 32) - mn) >> sh], 1u)] = k;
  }
  __syncthreads();
  // cur[b] is now the END of bucket b; it starts where bucket b - 1 ends.  Buckets of more than BUCKET_RANK keys (rare: > 30 x the mean)
  // are listed for a network sort; every other key finds its place by COUNTING the smaller keys of its own bucket -- one thread per
  // key, no thread waits for another, ~(2 + bucket size) LDS reads per key -- and goes back into A at bucket start + rank.
  for (uint32_t b = tid; b < SORT_BINS; b += nt) {
    const uint32_t s0 = b ? cur[b - 1] : 0u, m = cur[b] - s0;
    if (m > BUCKET_RANK) {
      const uint32_t which = m <= BUCKET_WAVE ? 0u : 1u;
      const uint32_t slot = atomicAdd(&nlist[which], 1u);
      if (slot < BUCKET_LIST) lists[which * BUCKET_LIST + slot] = (uint16_t)b;
    }
  }
  for (uint32_t i = tid; i < n; i += nt) {
    const uint64_t k = B[i];
    const uint32_t b = ((uint32_t)(k >> 32) - mn) >> sh;
    const uint32_t s0 = b ? cur[b - 1] : 0u, e0 = cur[b];
    if (e0 - s0 > BUCKET_RANK) continue;
    uint32_t rank = 0;
    for (uint32_t j = s0; j < e0; j++) rank += B[j] < k ? 1u : 0u;
    A[s0 + rank] = k;
  }
  __syncthreads();
  const uint32_t nw = nlist[0], ng = nlist[1];
  for (uint32_t q = wave; q < nw; q += nwaves) {      // one wave per medium bucket: sorted in B, copied to A
    const uint32_t b = lists[q];
    const uint32_t s0 = b ? cur[b - 1] : 0u, m = cur[b] - s0;
    bitonic_network_wave(B + s0, m, lane);
    for (uint32_t i = lane; i < m; i += 64) A[s0 + i] = B[s0 + i];
  }
  __syncthreads();
  for (uint32_t q = 0; q < ng; q++) {            // the workgroup on every large bucket (uniform loop; bitonic_network ends on a barrier)
    const uint32_t b = lists[BUCKET_LIST + q];
    const uint32_t s0 = b ? cur[b - 1] : 0u, m = cur[b] - s0;
    bitonic_network(B + s0, m, tid, nt);
    for (uint32_t i = tid; i < m; i += nt) A[s0 + i] = B[s0 + i];
  }
  __syncthreads();
}

// Tiles with lo < n <= hi are handled by this launch; lds_keys = capacity of the dynamic LDS buffer in keys; bucket_keys != 0: the
// buffer is laid out for bucket_sort_lds (two key buffers of bucket_keys keys + BUCKET_LDS_EXTRA bytes) and lists of more than
// BUCKET_MIN and at most bucket_keys keys take it.
// After sorting, slot_pos[gauss_off[g] + (tile's index inside g's rect)] = position of the instance in point_list:
// the instance -> position map that lets the backward gather per-instance gradients without atomics.
__device__ __forceinline__ void emit_instance(uint32_t pos, uint32_t g, int tx, int ty, const ushort4* __restrict__ rect,
                                              const uint32_t* __restrict__ gauss_off, uint32_t* __restrict__ point_list,
                                              uint32_t* __restrict__ slot_pos) {
  point_list[pos] = g;
  if (slot_pos == nullptr) return;   // forward-only render: nobody will gather through the map
  const ushort4 r = rect[g];
  const uint32_t local = (uint32_t)(ty - (int)r.y) * (uint32_t)((int)r.z - (int)r.x) + (uint32_t)(tx - (int)r.x);
  slot_pos[gauss_off[g] + local] = pos;
}

__global__ void __launch_bounds__(1024) sort_tiles_kernel(int tiles, int gx, const uint2* __restrict__ ranges,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list,
                                                         const ushort4* __restrict__ rect,
                                                         const uint32_t* __restrict__ gauss_off,
                                                         uint32_t* __restrict__ slot_pos,
                                                         uint32_t lo, uint32_t hi, uint32_t lds_keys, uint32_t bucket_keys) {
  extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
  const uint32_t t = xcd_swizzle(blockIdx.x, gridDim.x);
  if (t >= (uint32_t)tiles) return;
  const uint2 rg = ranges[t];
  const uint32_t n = rg.y - rg.x;
  if (n <= lo || n > hi) return;
  uint64_t* gk = keys + rg.x;
  const uint32_t tid = threadIdx.x, nt = blockDim.x;     // 256 threads for the short lists, 512 / 1024 for the launches of the long ones
  const int tx = (int)(t % (uint32_t)gx), ty = (int)(t / (uint32_t)gx);
  if (n > BUCKET_MIN && n <= bucket_keys) {
    uint64_t* B = skeys + bucket_keys;
    uint32_t* cur = reinterpret_cast<uint32_t*>(B + bucket_keys);
    for (uint32_t i = tid; i < n; i += nt) skeys[i] = gk[i];
    __syncthreads();
    bucket_sort_lds(skeys, B, cur, reinterpret_cast<uint16_t*>(cur + SORT_BINS), n, tid, nt);
    for (uint32_t i = tid; i < n; i += nt)
      emit_instance(rg.x + i, (uint32_t)skeys[i], tx, ty, rect, gauss_off, point_list, slot_pos);
  } else if (n <= RANK_DIRECT && n <= lds_keys && nt >= RANK_DIRECT) {
    // a short list: every thread counts the keys smaller than its own (all lanes read the same LDS address: a broadcast) and emits at
    // that rank -- one barrier, no network (36 steps at 256 keys).  Two keys per thread up to 512 keys: slower than the network
    // (profiles/r06_tile_sort_ab.txt).
    const uint64_t k = tid < n ? gk[tid] : ~0ull;
    if (tid < n) skeys[tid] = k;
    __syncthreads();
    if (tid < n) {
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) rank += skeys[j] < k ? 1u : 0u;
      emit_instance(rg.x + rank, (uint32_t)k, tx, ty, rect, gauss_off, point_list, slot_pos);
    }
  } else if (n <= lds_keys) {
    for (uint32_t i = tid; i < n; i += nt) skeys[i] = gk[i];
    __syncthreads();
    if (n > 1) bitonic_network(skeys, n, tid, nt);
    for (uint32_t i = tid; i < n; i += nt)
      emit_instance(rg.x + i, (uint32_t)skeys[i], tx, ty, rect, gauss_off, point_list, slot_pos);
  } else {
    bitonic_network((volatile uint64_t*)gk, n, tid, nt);  // same workgroup: coherent through its own L1 after barriers
    for (uint32_t i = tid; i < n; i += nt)
      emit_instance(rg.x + i, (uint32_t)gk[i], tx, ty, rect, gauss_off, point_list, slot_pos);
  }
}

// =========================================================================================================
// 5. Front-to-back alpha/depth blending, one 16x16 tile per workgroup (4 wave64), one pixel per lane.
//    Per batch of 256 Gaussians the workgroup gathers (mean2D, conic, opacity, rgb, depth) into LDS with one
//    coalesced index read + L2-resident attribute gathers; the inner loop then reads wave-uniform LDS
//    addresses (broadcast).  The conic is pre-scaled by -0.5*log2(e) / -log2(e) while staging so the
//    Gaussian weight is a bare v_exp_f32:  alpha = min(0.99, o * exp2(qa*dx*dx + qc*dy*dy + qb*dx*dy)).
// =========================================================================================================
//    NX = 3: a second image with other per-Gaussian colours (colors2 -> out_color2) is blended in the same pass; the
//    alpha test, exp2 and the transmittance recurrence are shared.
template <int NX>
__global__ void __launch_bounds__(256)
blend_forward_kernel(int W, int H, int gx, int tiles, const uint2* __restrict__ ranges,
                     const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
                     const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
                     const float* __restrict__ depths, const float* __restrict__ bg, float* __restrict__ final_T,
                     uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_hi, float* __restrict__ out_color,
                     float* __restrict__ out_depth, const float* __restrict__ colors2, float* __restrict__ out_color2) {
  __shared__ StagedGaussian sg[256];
  __shared__ float4 sg2[NX ? 256 : 1];
  __shared__ uint32_t wave_hi[4];
  const uint32_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= (uint32_t)tiles) return;
  const int tx = tile % gx, ty = tile / gx;
  const int tid = threadIdx.x;
  const int px = tx * TILE_X + (tid & 15), py = ty * TILE_Y + (tid >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 rg = ranges[tile];
  int todo = (int)(rg.y - rg.x);

  bool done = !inside;
  float T = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f;
  float C2r = 0.f, C2g = 0.f, C2b = 0.f;
  uint32_t contributor = 0, last_contributor = 0;

  for (uint32_t base = rg.x; base < rg.y; base += 256, todo -= 256) {
    if (__syncthreads_count(done) == 256) break;  // also protects sg[] reuse
    if (base + tid < rg.y) {
      const uint32_t id = point_list[base + tid];
      const float2 m = means2D[id];
      const float4 co = conic_opacity[id];
      StagedGaussian s;
      s.a = make_float4(m.x, m.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
      s.b = make_float4(-0.5f * LOG2E * co.z, co.w, depths[id], colors[3 * (size_t)id]);
      s.c = make_float4(colors[3 * (size_t)id + 1], colors[3 * (size_t)id + 2], co.x, co.y);
      sg[tid] = s;
      if (NX) sg2[tid] = make_float4(colors2[3 * (size_t)id], colors2[3 * (size_t)id + 1], colors2[3 * (size_t)id + 2], 0.f);
    }
    __syncthreads();
    const int cnt = min(256, todo);
    for (int j = 0; !done && j < cnt; j++) {
      contributor++;
      const float4 A = sg[j].a;
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float4 B = sg[j].b;
      const float q = gaussian_exponent2(dx, dy, A.z, A.w, B.x);
      if (q > 0.f) continue;
      const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(q));
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1.f - alpha);
      if (test_T < 0.0001f) {
        done = true;
        continue;
      }
      const float w = alpha * T;
      const float4 Cc = sg[j].c;
      Cr = __builtin_fmaf(B.w, w, Cr);
      Cg = __builtin_fmaf(Cc.x, w, Cg);
      Cb = __builtin_fmaf(Cc.y, w, Cb);
      D = __builtin_fmaf(B.z, w, D);
      if (NX) {
        const float4 C2 = sg2[j];
        C2r = __builtin_fmaf(C2.x, w, C2r);
        C2g = __builtin_fmaf(C2.y, w, C2g);
        C2b = __builtin_fmaf(C2.z, w, C2b);
      }
      T = test_T;
      last_contributor = contributor;
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * W + px, N = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last_contributor;
    out_color[pix] = Cr + T * bg[0];
    out_color[N + pix] = Cg + T * bg[1];
    out_color[2 * N + pix] = Cb + T * bg[2];
    out_depth[pix] = D;
    if (NX) {
      out_color2[pix] = C2r + T * bg[0];
      out_color2[N + pix] = C2g + T * bg[1];
      out_color2[2 * N + pix] = C2b + T * bg[2];
    }
  }
  // end (absolute list position) of the deepest contributor of the tile: the backward never looks behind it
  uint32_t m = inside ? last_contributor : 0u;
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  if ((tid & 63) == 0) wave_hi[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) tile_hi[tile] = rg.x + max(max(wave_hi[0], wave_hi[1]), max(wave_hi[2], wave_hi[3]));
}


// ---- static / dynamic decomposition renders of one geometry in ONE blend pass -------------------------
// gaussian_renderer/__init__.py:168-204 renders the Gaussians with max|dx| above / below the mean a second and third time
// (full preprocess + binning + sort + blend on boolean-masked copies of every input).  A subset's per-tile list is the full
// list with the other class removed -- same depth order, ties still by index -- so both subset images fall out of one walk
// over the FULL sorted lists with one transmittance chain per class: alpha is evaluated once per (pixel, Gaussian) and
// updates only the chain of the Gaussian's class.  Results are bit-identical to the two separate subset renders.
__global__ void __launch_bounds__(256)
blend_decompose_kernel(int W, int H, int gx, int tiles, const uint2* __restrict__ ranges,
                       const uint32_t* __restrict__ point_list, const float2* __restrict__ means2D,
                       const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
                       const float* __restrict__ depths, const float* __restrict__ bg, const uint8_t* __restrict__ cls,
                       const long long* __restrict__ class_counts /* [2]: static, dynamic; NULL = both non-empty */,
                       float* __restrict__ out_color_d, float* __restrict__ out_depth_d, float* __restrict__ out_color_s,
                       float* __restrict__ out_depth_s) {
  __shared__ StagedGaussian sg[256];
  __shared__ uint16_t sub[2][256];      // the batch's entries of each class, in list order
  __shared__ int wave_cnt[2][4];
  const uint32_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= (uint32_t)tiles) return;
  const int tx = tile % gx, ty = tile / gx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tx * TILE_X + (tid & 15), py = ty * TILE_Y + (tid >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 rg = ranges[tile];
  // one chain per class, each in its own scalar variables and its own loop over the class's sub-list of the batch: a chain
  // stops at ITS saturation, never pays for the other class's entries, and nothing is indexed by a run-time class (the first
  // version walked the full list once with T[k], C[k] selected per entry: 0.70 ms against 0.23 ms for the plain blend pass)
  bool done_s = !inside, done_d = !inside;
  float Ts = 1.f, Crs = 0.f, Cgs = 0.f, Cbs = 0.f, Ds = 0.f;
  float Td = 1.f, Crd = 0.f, Cgd = 0.f, Cbd = 0.f, Dd = 0.f;
  auto chain = [&](const uint16_t* __restrict__ list, int n, bool& done, float& T, float& Cr, float& Cg, float& Cb, float& D) {
    for (int j = 0; !done && j < n; j++) {
      const int e = list[j];
      const float4 A = sg[e].a;
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float4 B = sg[e].b;
      const float q = gaussian_exponent2(dx, dy, A.z, A.w, B.x);
      if (q > 0.f) continue;
      const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(q));
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1.f - alpha);
      if (test_T < 0.0001f) {
        done = true;
        continue;
      }
      const float w = alpha * T;
      const float4 Cc = sg[e].c;
      Cr = __builtin_fmaf(B.w, w, Cr);
      Cg = __builtin_fmaf(Cc.x, w, Cg);
      Cb = __builtin_fmaf(Cc.y, w, Cb);
      D = __builtin_fmaf(B.z, w, D);
      T = test_T;
    }
  };
  for (uint32_t base = rg.x; base < rg.y; base += 256) {
    if (__syncthreads_count(done_s && done_d) == 256) break;   // also protects sg[] / sub[] reuse
    const bool valid = base + tid < rg.y;
    int c = 0;
    if (valid) {
      const uint32_t id = point_list[base + tid];
      const float2 m = means2D[id];
      const float4 co = conic_opacity[id];
      StagedGaussian s;
      s.a = make_float4(m.x, m.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
      s.b = make_float4(-0.5f * LOG2E * co.z, co.w, depths[id], colors[3 * (size_t)id]);
      s.c = make_float4(colors[3 * (size_t)id + 1], colors[3 * (size_t)id + 2], co.x, co.y);
      sg[tid] = s;
      c = cls[id] ? 1 : 0;
    }
    // stable partition of the batch by class: ballots inside the wave, a 4-entry prefix across the waves
    const unsigned long long b1 = __ballot(valid && c == 1), b0 = __ballot(valid && c == 0);
    if (lane == 0) { wave_cnt[0][wave] = __popcll(b0); wave_cnt[1][wave] = __popcll(b1); }
    __syncthreads();
    int off0 = 0, off1 = 0, n0 = 0, n1 = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      if (w < wave) { off0 += wave_cnt[0][w]; off1 += wave_cnt[1][w]; }
      n0 += wave_cnt[0][w]; n1 += wave_cnt[1][w];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (valid) sub[c][(c ? off1 + __popcll(b1 & below) : off0 + __popcll(b0 & below))] = (uint16_t)tid;
    __syncthreads();
    chain(sub[0], n0, done_s, Ts, Crs, Cgs, Cbs, Ds);
    chain(sub[1], n1, done_d, Td, Crd, Cgd, Cbd, Dd);
  }
  const float T[2] = {Ts, Td}, Cr[2] = {Crs, Crd}, Cg[2] = {Cgs, Cgd}, Cb[2] = {Cbs, Cbd}, D[2] = {Ds, Dd};
  if (inside) {
    const size_t pix = (size_t)py * W + px, N = (size_t)H * W;
    // an EMPTY class renders as zeros WITHOUT background, like the reference's P == 0 early-out (rasterize_points.cu:81-116)
    const float ks = (class_counts && class_counts[0] == 0) ? 0.f : 1.f, kd = (class_counts && class_counts[1] == 0) ? 0.f : 1.f;
    out_color_s[pix] = ks * (Cr[0] + T[0] * bg[0]);
    out_color_s[N + pix] = ks * (Cg[0] + T[0] * bg[1]);
    out_color_s[2 * N + pix] = ks * (Cb[0] + T[0] * bg[2]);
    out_depth_s[pix] = D[0];
    out_color_d[pix] = kd * (Cr[1] + T[1] * bg[0]);
    out_color_d[N + pix] = kd * (Cg[1] + T[1] * bg[1]);
    out_color_d[2 * N + pix] = kd * (Cb[1] + T[1] * bg[2]);
    out_depth_d[pix] = D[1];
  }
}

__global__ void __launch_bounds__(256) check_frustum_kernel(int P, const float* __restrict__ means3D,
                                                            const float* __restrict__ viewmatrix,
                                                            uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
  present[idx] = xform_4x3(p, viewmatrix).z > 0.2f ? 1 : 0;
}

static inline uint32_t round_up8(uint32_t v) { return (v + 7u) & ~7u; }

}  // namespace s3g

using namespace s3g;

extern "C" const char* s3g_last_error(void) { return g_err; }
extern "C" int s3g_abi_version(void) { return 13; }

// as != NULL: the host-asynchronous variant (s3g_raster_forward_async) -- arenas are the caller's, sized for a speculative
// capacity, and nothing below waits for the device.
static int raster_forward_impl(const s3g_raster_inputs* in, const float* colors2, float* out_color2,
                               s3g_resize_fn geometry_buffer, void* geometry_user, s3g_resize_fn binning_buffer,
                               void* binning_user, s3g_resize_fn image_buffer, void* image_user, float* out_color,
                               float* out_depth, int* radii, int* num_rendered, void* stream_,
                               const s3g_raster_async* as = nullptr) {
  g_err[0] = 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || (!as && (!geometry_buffer || !binning_buffer || !image_buffer || !num_rendered))) {
    set_error("s3g_raster_forward: NULL argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (as && (!as->geometry_arena || !as->image_arena || !as->binning_arena || as->capacity_instances == 0 ||
             as->capacity_instances > 0x7fffffffu || as->capacity_slots < as->capacity_instances)) {
    set_error("s3g_raster_forward_async: needs the three arenas and 0 < capacity_instances <= capacity_slots");
    return S3G_ERR_INVALID_ARG;
  }
  if (num_rendered) *num_rendered = 0;
  const int P = in->P, W = in->width, H = in->height;
  if (P < 0 || W <= 0 || H <= 0) {
    set_error("s3g_raster_forward: bad sizes P=%d W=%d H=%d", P, W, H);
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;  // rasterize_points.cu:82: outputs keep the caller's zero fill
  if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) {
    set_error("Please provide excatly one of either SHs or precomputed colors!");
    return S3G_ERR_INVALID_ARG;
  }
  if (((in->scales == nullptr || in->rotations == nullptr) && in->cov3D_precomp == nullptr) ||
      ((in->scales != nullptr || in->rotations != nullptr) && in->cov3D_precomp != nullptr)) {
    set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return S3G_ERR_INVALID_ARG;
  }
  if (!in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix || !in->cam_pos || !in->background ||
      !out_color || !out_depth || !radii) {
    set_error("s3g_raster_forward: NULL array argument");
    return S3G_ERR_INVALID_ARG;
  }
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y, tiles = gx * gy;
  if (gx > 65535 || gy > 65535) {
    set_error("image too large for 16-bit tile coordinates");
    return S3G_ERR_INVALID_ARG;
  }
  const bool debug = in->debug != 0;

  // tile grids beyond the LDS histogram are binned in bands of consecutive tiles (see bin_kernel)
  const int band = g_max_tiles_lds < tiles ? g_max_tiles_lds : tiles;
  const int nb = bin_blocks(P), chunk = bin_chunk(P);

  size_t geom_bytes = 0, img_bytes = 0;
  GeomState::carve(nullptr, P, &geom_bytes);
  ImageState::carve(nullptr, (size_t)W * H, tiles, nb, &img_bytes);
  void* geom_p = as ? as->geometry_arena : geometry_buffer(geometry_user, geom_bytes);
  void* img_p = as ? as->image_arena : image_buffer(image_user, img_bytes);
  if (!geom_p || !img_p) {
    set_error("resize callback returned NULL");
    return S3G_ERR_ALLOC;
  }
  GeomState g = GeomState::carve(geom_p, P, nullptr);
  ImageState im = ImageState::carve(img_p, (size_t)W * H, tiles, nb, nullptr);

  S3G_HIP_CHECK(hipMemsetAsync(im.ctrl, 0, 8 * sizeof(uint32_t), stream));

  PreprocessArgs pa;
  pa.P = P; pa.D = in->D; pa.M = in->M; pa.W = W; pa.H = H; pa.gx = gx; pa.gy = gy;
  pa.cull = g_exact_cull ? 1 : 0;
  pa.means3D = in->means3D; pa.scales = in->scales; pa.scale_modifier = in->scale_modifier;
  pa.rotations = in->rotations; pa.opacities = in->opacities; pa.shs = in->shs;
  pa.cov3D_precomp = in->cov3D_precomp; pa.colors_precomp = in->colors_precomp;
  pa.viewmatrix = in->viewmatrix; pa.projmatrix = in->projmatrix; pa.cam_pos = in->cam_pos;
  pa.tan_fovx = in->tan_fovx; pa.tan_fovy = in->tan_fovy;
  pa.focal_y = H / (2.0f * in->tan_fovy); pa.focal_x = W / (2.0f * in->tan_fovx);
  pa.prefiltered = in->prefiltered; pa.radii = radii; pa.g = g; pa.ctrl = im.ctrl;
  hipLaunchKernelGGL(preprocess_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, pa);
  S3G_KERNEL_CHECK(stream, debug);

  // atomic-free binning, counting half
  const size_t bin_lds = ((size_t)band + BIN_SCRATCH) * sizeof(uint32_t);
  static std::atomic<uint64_t> bin_attr_set{0};
  if (device_needs_setup(bin_attr_set)) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)bin_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (MAX_TILES_LDS + BIN_SCRATCH) * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)bin_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (MAX_TILES_LDS + BIN_SCRATCH) * 4));
    device_setup_done(bin_attr_set);
  }
  BinArgs ba;
  ba.P = P; ba.gx = gx; ba.tiles = tiles; ba.chunk = chunk; ba.rect = g.rect; ba.depths = g.depths;
  ba.table = im.table; ba.chunk_total = im.chunk_total; ba.ranges = im.ranges; ba.keys = nullptr; ba.gauss_off = g.gauss_off;
  ba.cull = g_exact_cull ? 1 : 0; ba.W = W; ba.H = H; ba.means2D = g.means2D; ba.conic_opacity = g.conic_opacity;
  ba.tile_mask = g.tile_mask; ba.ctrl = im.ctrl;
  for (int lo = 0; lo < tiles; lo += band) {
    ba.tile_lo = lo; ba.tile_n = tiles - lo < band ? tiles - lo : band;
    hipLaunchKernelGGL(bin_kernel<false>, dim3(nb), dim3(BIN_THREADS), bin_lds, stream, ba);
    S3G_KERNEL_CHECK(stream, debug);
  }
  hipLaunchKernelGGL(bin_scan_kernel, dim3((tiles + 63) / 64), dim3(64 * SCAN_PARTS), 0, stream, tiles, nb, im.table, im.tile_count);
  S3G_KERNEL_CHECK(stream, debug);
  constexpr uint32_t SMALL = 4096, LARGE = 16384;  // per-tile sort: lists <= SMALL in <= 32 KiB of LDS, <= LARGE in 128 KiB
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, tiles, im.tile_count, im.ranges, im.ctrl, nb,
                     im.chunk_total, as ? as->capacity_instances : 0u, as ? as->capacity_slots : 0u,
                     as ? (as->long_lists ? 0xffffffffu : SMALL) : 0u, as ? as->status_device : nullptr,
                     (as && !as->forward_only) ? as->sticky_device : nullptr);
  S3G_KERNEL_CHECK(stream, debug);

  uint32_t R, max_tile, S;
  void* bin_p;
  if (!as) {
    // the one host sync of the forward (reference: rasterizer_impl.cu:282): R sizes the binning arena
    static thread_local uint32_t* h_ctrl = nullptr;
    if (!h_ctrl) S3G_HIP_CHECK(hipHostMalloc((void**)&h_ctrl, 8 * sizeof(uint32_t), hipHostMallocDefault));
    S3G_HIP_CHECK(hipMemcpyAsync(h_ctrl, im.ctrl, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    S3G_HIP_CHECK(hipStreamSynchronize(stream));
    R = h_ctrl[0]; max_tile = h_ctrl[1]; S = h_ctrl[3];
    if (h_ctrl[2] & 1u) {
      set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
      return S3G_ERR_PREFILTERED;
    }
    if (R > 0x7fffffffu) {
      set_error("too many Gaussian/tile instances (%u)", R);
      return S3G_ERR_INVALID_ARG;
    }
    *num_rendered = (int)R;
    size_t bin_bytes = 0;
    BinningState::carve(nullptr, R, S, &bin_bytes);
    bin_p = binning_buffer(binning_user, bin_bytes);
    if (!bin_p && R > 0) {
      set_error("resize callback returned NULL");
      return S3G_ERR_ALLOC;
    }
  } else {
    // host-asynchronous: the arena was sized for (capacity_instances, capacity_slots) before anything ran; the kernels below
    // read R / S / the ranges on the device and find nothing to do if scan_tiles_kernel saw the capacity exceeded.  The
    // control words travel to the caller's pinned buffer behind the kernels: whoever reads them must first know that the
    // stream has passed this point (an event recorded after this call).
    R = as->capacity_instances; S = as->capacity_slots;
    max_tile = as->sort_lds_keys ? as->sort_lds_keys : SMALL;
    bin_p = as->binning_arena;
    if (as->status_host)
      S3G_HIP_CHECK(hipMemcpyAsync(as->status_host, im.ctrl, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (as->status_event)   // the verdict is on its way: a caller that has to know it waits for THIS, not for the sort / blend below
      S3G_HIP_CHECK(hipEventRecord((hipEvent_t)as->status_event, stream));
    if (num_rendered) *num_rendered = (int)R;
  }
  BinningState b = BinningState::carve(bin_p, R, S, nullptr);
  const bool forward_only = as && as->forward_only != 0;
  uint32_t* const slot_map = forward_only ? nullptr : b.slot_pos;
  if (as && !forward_only) {
    hipLaunchKernelGGL(fill_slots_kernel, dim3(1024), dim3(256), 0, stream, b.slot_pos, (const uint32_t*)im.ctrl);
    S3G_KERNEL_CHECK(stream, debug);
  } else if (!as && S > 0) {
    S3G_HIP_CHECK(hipMemsetAsync(b.slot_pos, 0xff, (size_t)S * sizeof(uint32_t), stream));  // culled slots
  }

  const uint32_t tile_blocks = round_up8((uint32_t)tiles);
  if (R > 0) {
    ba.keys = b.keys;
    for (int lo = 0; lo < tiles; lo += band) {
      ba.tile_lo = lo; ba.tile_n = tiles - lo < band ? tiles - lo : band;
      hipLaunchKernelGGL(bin_kernel<true>, dim3(nb), dim3(BIN_THREADS), bin_lds, stream, ba);
      S3G_KERNEL_CHECK(stream, debug);
    }
    // short lists: <= 32 KiB of LDS per workgroup (5 workgroups/CU); long lists: up to 128 KiB, beyond that in global
    // (asynchronous: max_tile is the caller's estimate; a list longer than the LDS buffer is sorted in global memory)
    const uint32_t small_cap = max_tile < SMALL ? max_tile : SMALL;
    static std::atomic<uint64_t> attr_set{0};
    if (device_needs_setup(attr_set)) {
      S3G_HIP_CHECK(hipFuncSetAttribute((const void*)sort_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LARGE * 8));
      device_setup_done(attr_set);
    }
    // (a separate launch with a 2-8 KiB buffer for the lists of <= 256 / 512 / 1024 keys -- eight workgroups per CU instead of
    // five -- changes nothing: 136 / 127 / 122 us per frame against 123 with one launch, profiles/r04_sort.txt; the network's
    // LDS traffic bounds the kernel, not the workgroups in flight)
    // round 6: lists of (BUCKET_MIN, SMALL] keys in a launch of their own with the LDS layout of bucket_sort_lds (two key buffers +
    // cursors: <= 74 KiB, two workgroups per CU); the short lists keep their small buffer and their occupancy
    const uint32_t short_cap = small_cap < BUCKET_MIN ? small_cap : BUCKET_MIN;
    hipLaunchKernelGGL(sort_tiles_kernel, dim3(tile_blocks), dim3(256), (size_t)short_cap * 8, stream, tiles, gx,
                       im.ranges, b.keys, b.point_list, g.rect, g.gauss_off, slot_map, 0u, BUCKET_MIN, short_cap, 0u);
    S3G_KERNEL_CHECK(stream, debug);
    if (small_cap > BUCKET_MIN) {
      hipLaunchKernelGGL(sort_tiles_kernel, dim3(tile_blocks), dim3(SORT_THREADS_MID), (size_t)small_cap * 16 + BUCKET_LDS_EXTRA, stream, tiles, gx,
                         im.ranges, b.keys, b.point_list, g.rect, g.gauss_off, slot_map, BUCKET_MIN, SMALL, 2 * small_cap, small_cap);
      S3G_KERNEL_CHECK(stream, debug);
    }
    if (as ? as->long_lists != 0 : max_tile > SMALL) {
      // up to LARGE keys in 128 KiB: lists that fit twice (+ the cursors) take the bucket pass too, longer ones the network in LDS,
      // still longer ones the network in global memory
      const uint32_t large_cap = as ? LARGE : (max_tile < LARGE ? max_tile : LARGE);
      const uint32_t lds_bytes = large_cap * 8 > 2 * SMALL * 8 + BUCKET_LDS_EXTRA ? large_cap * 8 : 2 * SMALL * 8 + BUCKET_LDS_EXTRA;
      const uint32_t bucket_cap = (lds_bytes - BUCKET_LDS_EXTRA) / 16;
      hipLaunchKernelGGL(sort_tiles_kernel, dim3(tile_blocks), dim3(SORT_THREADS_LONG), (size_t)lds_bytes, stream, tiles, gx,
                         im.ranges, b.keys, b.point_list, g.rect, g.gauss_off, slot_map, SMALL, 0xffffffffu, lds_bytes / 8, bucket_cap);
      S3G_KERNEL_CHECK(stream, debug);
    }
  }
  const float* feat = in->colors_precomp ? in->colors_precomp : g.rgb;
  profile_begin(S3G_PROFILE_BLEND_FORWARD, stream);
  if (colors2 != nullptr)
    hipLaunchKernelGGL(blend_forward_kernel<3>, dim3(tile_blocks), dim3(256), 0, stream, W, H, gx, tiles, im.ranges,
                       b.point_list, g.means2D, g.conic_opacity, feat, g.depths, in->background, im.final_T, im.n_contrib,
                       im.tile_hi, out_color, out_depth, colors2, out_color2);
  else
    hipLaunchKernelGGL(blend_forward_kernel<0>, dim3(tile_blocks), dim3(256), 0, stream, W, H, gx, tiles, im.ranges,
                       b.point_list, g.means2D, g.conic_opacity, feat, g.depths, in->background, im.final_T, im.n_contrib,
                       im.tile_hi, out_color, out_depth, nullptr, nullptr);
  profile_end(S3G_PROFILE_BLEND_FORWARD, stream, as ? -1.0 : (double)R, (double)W * H);   // asynchronous: R is not known here
  S3G_KERNEL_CHECK(stream, debug);
  return S3G_OK;
}

extern "C" int s3g_raster_arena_bytes(int P, int width, int height, uint32_t capacity_instances, uint32_t capacity_slots,
                                      size_t* geometry_bytes, size_t* binning_bytes, size_t* image_bytes) {
  if (P < 0 || width <= 0 || height <= 0) {
    set_error("s3g_raster_arena_bytes: bad sizes");
    return S3G_ERR_INVALID_ARG;
  }
  const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
  size_t n = 0;
  GeomState::carve(nullptr, (size_t)P, &n);
  if (geometry_bytes) *geometry_bytes = n;
  BinningState::carve(nullptr, capacity_instances, capacity_slots, &n);
  if (binning_bytes) *binning_bytes = n;
  ImageState::carve(nullptr, (size_t)width * height, (size_t)gx * gy, bin_blocks(P), &n);
  if (image_bytes) *image_bytes = n;
  return S3G_OK;
}

extern "C" int s3g_raster_forward_async(const s3g_raster_inputs* in, const float* colors2, const s3g_raster_async* async_,
                                        float* out_color, float* out_depth, float* out_color2, int* radii, void* stream_) {
  if (!async_ || (colors2 != nullptr) != (out_color2 != nullptr) || (colors2 && (!in || !in->colors_precomp))) {
    set_error("s3g_raster_forward_async: needs the async descriptor; colors2 and out_color2 go together (with colors_precomp)");
    return S3G_ERR_INVALID_ARG;
  }
  return raster_forward_impl(in, colors2, out_color2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out_color, out_depth,
                             radii, nullptr, stream_, async_);
}

extern "C" int s3g_raster_forward(const s3g_raster_inputs* in, s3g_resize_fn geometry_buffer, void* geometry_user,
                                  s3g_resize_fn binning_buffer, void* binning_user, s3g_resize_fn image_buffer,
                                  void* image_user, float* out_color, float* out_depth, int* radii, int* num_rendered,
                                  void* stream_) {
  return raster_forward_impl(in, nullptr, nullptr, geometry_buffer, geometry_user, binning_buffer, binning_user,
                             image_buffer, image_user, out_color, out_depth, radii, num_rendered, stream_);
}

// Forward of two images from one geometry (colours in->colors_precomp -> out_color + out_depth, colors2 -> out_color2)
// with ONE blend pass; the arenas are those of an ordinary forward and feed s3g_raster_backward2.
extern "C" int s3g_raster_forward2(const s3g_raster_inputs* in, const float* colors2, s3g_resize_fn geometry_buffer,
                                   void* geometry_user, s3g_resize_fn binning_buffer, void* binning_user,
                                   s3g_resize_fn image_buffer, void* image_user, float* out_color, float* out_depth,
                                   float* out_color2, int* radii, int* num_rendered, void* stream_) {
  if (!in || !in->colors_precomp || !colors2 || !out_color2) {
    set_error("s3g_raster_forward2: needs colors_precomp, colors2 and out_color2");
    return S3G_ERR_INVALID_ARG;
  }
  return raster_forward_impl(in, colors2, out_color2, geometry_buffer, geometry_user, binning_buffer, binning_user,
                             image_buffer, image_user, out_color, out_depth, radii, num_rendered, stream_);
}

// Second (third, ...) render of the SAME geometry with different per-Gaussian colours (the reference renders RGB and
// then the feature image with identical means/scales/rotations/opacities, gaussian_renderer/__init__.py:127-166):
// everything up to the sorted per-tile lists is reused from the arenas of the first call; only the blend runs.
extern "C" int s3g_raster_forward_reuse(const s3g_raster_inputs* in, int R, const void* geometry_arena,
                                        const void* binning_arena, void* image_arena, float* out_color, float* out_depth,
                                        void* stream_) {
  g_err[0] = 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !in->colors_precomp || !in->background || !geometry_arena || !image_arena || (R > 0 && !binning_arena) ||
      !out_color || !out_depth || in->P <= 0) {
    set_error("s3g_raster_forward_reuse: bad argument (needs colors_precomp and the arenas of a previous forward)");
    return S3G_ERR_INVALID_ARG;
  }
  const int P = in->P, W = in->width, H = in->height;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y, tiles = gx * gy;
  GeomState g = GeomState::carve(const_cast<void*>(geometry_arena), P, nullptr);
  ImageState im = ImageState::carve(image_arena, (size_t)W * H, tiles, bin_blocks(P), nullptr);
  BinningState b = BinningState::carve(const_cast<void*>(binning_arena), (size_t)(R > 0 ? R : 0), 0, nullptr);
  const uint32_t tile_blocks = round_up8((uint32_t)tiles);
  profile_begin(S3G_PROFILE_BLEND_FORWARD, stream);
  hipLaunchKernelGGL(blend_forward_kernel<0>, dim3(tile_blocks), dim3(256), 0, stream, W, H, gx, tiles, im.ranges,
                     b.point_list, g.means2D, g.conic_opacity, in->colors_precomp, g.depths, in->background, im.final_T,
                     im.n_contrib, im.tile_hi, out_color, out_depth, nullptr, nullptr);
  profile_end(S3G_PROFILE_BLEND_FORWARD, stream, (double)R, (double)W * H);
  S3G_KERNEL_CHECK(stream, in->debug != 0);
  return S3G_OK;
}

extern "C" int s3g_raster_forward_decompose(const s3g_raster_inputs* in, int R, const void* geometry_arena,
                                            const void* binning_arena, const void* image_arena, const uint8_t* is_dynamic,
                                            const long long* class_counts, float* out_color_d, float* out_depth_d, float* out_color_s, float* out_depth_s,
                                            void* stream_) {
  g_err[0] = 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !in->background || !geometry_arena || !image_arena || (R > 0 && !binning_arena) || !is_dynamic ||
      !out_color_d || !out_depth_d || !out_color_s || !out_depth_s || in->P <= 0) {
    set_error("s3g_raster_forward_decompose: bad argument (needs the arenas of a previous forward and the class mask)");
    return S3G_ERR_INVALID_ARG;
  }
  const int P = in->P, W = in->width, H = in->height;
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y, tiles = gx * gy;
  GeomState g = GeomState::carve(const_cast<void*>(geometry_arena), P, nullptr);
  ImageState im = ImageState::carve(const_cast<void*>(image_arena), (size_t)W * H, tiles, bin_blocks(P), nullptr);
  BinningState b = BinningState::carve(const_cast<void*>(binning_arena), (size_t)(R > 0 ? R : 0), 0, nullptr);
  const float* color_ptr = in->colors_precomp ? in->colors_precomp : g.rgb;   // SH path: the forward's own colours
  hipLaunchKernelGGL(blend_decompose_kernel, dim3(round_up8((uint32_t)tiles)), dim3(256), 0, stream, W, H, gx, tiles,
                     im.ranges, b.point_list, g.means2D, g.conic_opacity, color_ptr, g.depths, in->background, is_dynamic,
                     class_counts, out_color_d, out_depth_d, out_color_s, out_depth_s);
  S3G_KERNEL_CHECK(stream, in->debug != 0);
  return S3G_OK;
}

extern "C" void s3g_raster_set_exact_cull(int on) { g_exact_cull = on != 0; }
extern "C" int s3g_raster_set_bin_band(int tiles) {  // testing hook: returns the previous band size; <= 0 restores the default
  const int prev = g_max_tiles_lds;
  g_max_tiles_lds = (tiles <= 0 || tiles > MAX_TILES_LDS) ? MAX_TILES_LDS : tiles;
  return prev;
}
extern "C" int s3g_raster_get_exact_cull(void) { return g_exact_cull ? 1 : 0; }

extern "C" void s3g_profile_enable(int on) { g_prof_on = on != 0; }

// Sums the recorded launches of kernel `id` (S3G_PROFILE_* in s3g_raster.h), synchronising on their events, then forgets
// them.  Returns the number of launches.
extern "C" int s3g_profile_read(int id, double* total_ms, double* total_instances, double* total_pixels) {
  if (id < 0 || id >= S3G_PROFILE_IDS) return 0;
  double ms = 0, inst = 0, pix = 0;
  int n = 0;
  for (ProfRec& r : g_prof[id]) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms += t; inst += r.instances; pix += r.pixels; n++;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof[id].clear();
  if (total_ms) *total_ms = ms;
  if (total_instances) *total_instances = inst;
  if (total_pixels) *total_pixels = pix;
  return n;
}

extern "C" int s3g_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                uint8_t* present, void* stream_) {
  g_err[0] = 0;
  (void)projmatrix;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
    set_error("s3g_mark_visible: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(check_frustum_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
