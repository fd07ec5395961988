// MI355X-native multi-resolution HexPlane sampler: forward and backward, one fused pass each.
//
// Reference: scene/hexplane.py:73-106 runs 4 levels x 6 planes = 24 F.grid_sample launches, each materialising a
// [P,32] tensor, then 20 elementwise products and a concat; autograd replays the same 24 in backward.
//
// Planes are stored channel-last, so one texel = 32 channels = one 128-byte line.  Forward and the per-point backward give
// a point to EIGHT lanes (4 channels each, one 16-byte load per texel): these kernels are VALU-bound on the bilinear-tap
// arithmetic, which every lane of a point repeats -- 8 copies instead of 32.  The product over planes never leaves
// registers.  Arithmetic follows torch's grid_sampler_2d (bilinear, border, align_corners=True) op for op; contraction
// is off.  When all points share one timestamp (desc.uniform_time) the three (axis, t) planes of a level are first
// collapsed to 1-D row tables (see "uniform time" below).
//
// Backward without an atomic storm.  A direct scatter is 96 line-coalesced float atomics per point; MI355X retires
// ~10 G such line-ops/s whatever the contention (tools/ubench/atomic_lines.hip), i.e. 11.5 ms at 1.2 M points.  So:
//   pass A  (point order)   re-gathers the taps, applies the product rule, writes dL/ds for all 24 plane-levels to a
//                           scratch slab G[24][P][32] (3 KB per point -- HBM is 288 GB) and finishes dL/dxyz;
//   sort    three 2-level counting sorts of the point indices by (major, minor) finest-level texel cell, one per plane
//           orientation, with LDS histograms (no global atomics, no library sort); the orders only steer the walk, so the
//           caller may keep them for several iterations (sort_state / sort_reuse);
//   pass B  (sorted order, one launch for the three orientations = 2 plane kinds x all levels each): a half-wave (32
//           lanes = the 32 channels) walks a run of spatially consecutive points keeping two bilinear footprints per
//           (level, plane) in registers and only issues atomics when a footprint is evicted -- consecutive points share
//           texels, so the 96 line-ops per point drop to ~7.  Taps are computed cooperatively (lane = point x tap) and
//           shared through LDS; index, coordinate and tap computation run one to two groups ahead of the accumulation.
#include "common.hpp"

#include "../../include/s3g_hexplane.h"

namespace s3g {

constexpr int HEXC = S3G_HEX_CHANNELS;

struct HexArgs {
  s3g_hexplane_desc d;
  float* gplanes[S3G_HEX_MAX_LEVELS][6];
  int P;
  const float* xyz;
  const float* time;
  const float* gfeat;
  float* feat;
  float* gxyz;
  const uint32_t* proc_order;  // optional: process point proc_order[i] at step i (spatially sorted -> texel reuse in L1/L2)
};

struct Tap {         // one bilinear footprint
  int o00, o01, o10, o11;  // texel offsets (in texels) of nw, ne, sw, se; -1 when out of bounds (weight is 0 then)
  float w00, w01, w10, w11;
  float x0f, x1f, y0f, y1f, ix, iy;
  float mx, my;      // d(ix)/d(u) incl. the border-clip gradient mask
};

// torch grid_sampler_unnormalize (align_corners) + clip_coordinates(_set_grad) + bilinear weights
__device__ __forceinline__ Tap make_tap(float ux, float uy, int W, int H) {
  Tap t;
  float ix = ((ux + 1.f) / 2.f) * (float)(W - 1);
  float iy = ((uy + 1.f) / 2.f) * (float)(H - 1);
  t.mx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : (float)(W - 1) / 2.f;
  t.my = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : (float)(H - 1) / 2.f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float x0 = floorf(ix), y0 = floorf(iy);
  const float x1 = x0 + 1.f, y1 = y0 + 1.f;
  t.ix = ix; t.iy = iy; t.x0f = x0; t.x1f = x1; t.y0f = y0; t.y1f = y1;
  t.w00 = (x1 - ix) * (y1 - iy);
  t.w01 = (ix - x0) * (y1 - iy);
  t.w10 = (x1 - ix) * (iy - y0);
  t.w11 = (ix - x0) * (iy - y0);
  const int xi0 = (int)x0, yi0 = (int)y0, xi1 = xi0 + 1, yi1 = yi0 + 1;
  const bool bx1 = xi1 < W, by1 = yi1 < H;  // xi0, yi0 are always in range after the clip
  t.o00 = yi0 * W + xi0;
  t.o01 = bx1 ? yi0 * W + xi1 : -1;
  t.o10 = by1 ? yi1 * W + xi0 : -1;
  t.o11 = (bx1 && by1) ? yi1 * W + xi1 : -1;
  return t;
}

__device__ __forceinline__ float fetch(const float* __restrict__ plane, int off, int c) {
  return off >= 0 ? plane[(size_t)off * HEXC + c] : 0.f;
}

__device__ __forceinline__ void point_coords(const HexArgs& a, int p, float* u) {
#pragma unroll
  for (int k = 0; k < 3; k++)
    u[k] = (a.xyz[3 * (size_t)p + k] - a.d.aabb_max[k]) * (2.0f / (a.d.aabb_min[k] - a.d.aabb_max[k])) - 1.0f;
  u[3] = a.time[p];
}

// coordinate pairs in itertools.combinations(range(4), 2) order
__device__ constexpr int PAIR0[6] = {0, 0, 0, 1, 1, 2};
__device__ constexpr int PAIR1[6] = {1, 2, 3, 2, 3, 3};

__device__ __forceinline__ float4 fetch4(const float* __restrict__ plane, int off, int c4) {
  return off >= 0 ? *reinterpret_cast<const float4*>(plane + (size_t)off * HEXC + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 operator*(float4 a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// EIGHT lanes own one point, four channels each (one 16-byte load per texel and lane; the 8 lanes read its 128-byte
// line): the kernel is VALU-bound on the tap arithmetic, which every lane of a point repeats -- 8 copies instead of 32.
__global__ void __launch_bounds__(256) hexplane_forward_kernel(const HexArgs a) {
  const int c4 = (threadIdx.x & 7) * 4, slot = threadIdx.x >> 3;
  const int F = a.d.levels * HEXC;
  for (int pi = blockIdx.x * 32 + slot; pi < a.P; pi += gridDim.x * 32) {
    const int p = a.proc_order ? (int)a.proc_order[pi] : pi;
    float u[4];
    point_coords(a, p, u);
    for (int l = 0; l < a.d.levels; l++) {
      float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
        const float* pl = a.d.planes[l][i];
        float4 s = fetch4(pl, t.o00, c4) * t.w00;
        s = s + fetch4(pl, t.o01, c4) * t.w01;
        s = s + fetch4(pl, t.o10, c4) * t.w10;
        s = s + fetch4(pl, t.o11, c4) * t.w11;
        prod = prod * s;
      }
      *reinterpret_cast<float4*>(a.feat + (size_t)p * F + l * HEXC + c4) = prod;
    }
  }
}

// ---- pass A: per point, dL/ds for every plane-level -> G, and dL/dxyz ----
// G layout: slab (orientation o, level l, kind q) = G + (((o * levels + l) * 2 + q) * P + rank_o[p]) * 32, i.e. every
// orientation's rows are stored in THAT orientation's sorted order, so pass B streams them sequentially.
__device__ constexpr int ORI_OF[6] = {0, 2, 0, 1, 1, 2};   // plane i -> orientation pass that scatters it
__device__ constexpr int KIND_OF[6] = {0, 0, 1, 0, 1, 1};  // 0 = spatial plane of the pass, 1 = its time plane

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Same lane mapping as the forward (8 lanes per point, 4 channels each).  Per plane only the sample s and its two
// coordinate derivatives are kept:  ds/dix = (ne - nw)(y1 - iy) + (se - sw)(iy - y0),  ds/diy = (sw - nw)(x1 - ix) +
// (se - ne)(ix - x0)  (the four terms of torch's grid_sampler_2d_backward, grouped).
__global__ void __launch_bounds__(256) hexplane_backward_point_kernel(const HexArgs a, float* __restrict__ G,
                                                                      const uint32_t* __restrict__ rank_all) {
  const int c4 = (threadIdx.x & 7) * 4, slot = threadIdx.x >> 3;
  const int F = a.d.levels * HEXC;
  const size_t PL = (size_t)a.P * HEXC;  // one slab of G
  for (int p0 = blockIdx.x * 32; p0 < a.P; p0 += gridDim.x * 32) {  // uniform trip count: shuffles below need all lanes
    const int pi = p0 + slot;
    const bool live = pi < a.P;
    const int p = live ? (a.proc_order ? (int)a.proc_order[pi] : pi) : 0;
    uint32_t rk[3] = {0u, 0u, 0u};
    if (live)
#pragma unroll
      for (int o = 0; o < 3; o++) rk[o] = rank_all[(size_t)o * a.P + p];
    float u[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) point_coords(a, p, u);
    float du[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < a.d.levels; l++) {
      float4 s[6], dX[6], dY[6];
      float mx[6], my[6];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
        const float* pl = a.d.planes[l][i];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v00 = live ? fetch4(pl, t.o00, c4) : z, v01 = live ? fetch4(pl, t.o01, c4) : z;
        const float4 v10 = live ? fetch4(pl, t.o10, c4) : z, v11 = live ? fetch4(pl, t.o11, c4) : z;
        float4 acc = v00 * t.w00;
        acc = acc + v01 * t.w01;
        acc = acc + v10 * t.w10;
        acc = acc + v11 * t.w11;
        s[i] = acc;
        dX[i] = (v01 - v00) * (t.y1f - t.iy) + (v11 - v10) * (t.iy - t.y0f);
        dY[i] = (v10 - v00) * (t.x1f - t.ix) + (v11 - v01) * (t.ix - t.x0f);
        mx[i] = t.mx;
        my[i] = t.my;
      }
      const float4 g = live ? *reinterpret_cast<const float4*>(a.gfeat + (size_t)p * F + l * HEXC + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      // product rule in the order autograd applies it to ((((1*s0)*s1)*s2)*s3)*s4)*s5: pre[i] = prod_{j<i} s_j, suffix by recursion
      float4 pre[6];
      pre[0] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
      for (int i = 1; i < 6; i++) pre[i] = pre[i - 1] * s[i - 1];
      float4 gs = g;  // dL/d(prefix product through plane i)
#pragma unroll
      for (int i = 5; i >= 0; i--) {
        const float4 gi = gs * pre[i];  // dL/ds_i
        gs = gs * s[i];
        if (live) {
          *reinterpret_cast<float4*>(G + (size_t)((ORI_OF[i] * a.d.levels + l) * 2 + KIND_OF[i]) * PL +
                                     (size_t)rk[ORI_OF[i]] * HEXC + c4) = gi;
          if (PAIR0[i] < 3) du[PAIR0[i]] += mx[i] * dot4(dX[i], gi);
          if (PAIR1[i] < 3) du[PAIR1[i]] += my[i] * dot4(dY[i], gi);
        }
      }
    }
    // sum over the 32 channels (the 8 lanes of this point), then undo the aabb normalisation
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = du[k];
      for (int off = 4; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      du[k] = v;
    }
    const int c = threadIdx.x & 7;
    if (live && c < 3) a.gxyz[3 * (size_t)p + c] = (c == 0 ? du[0] : (c == 1 ? du[1] : du[2])) * (2.0f / (a.d.aabb_min[c] - a.d.aabb_max[c]));
  }
}

// ---- sort: point indices ordered by (major cell, minor cell) on a 512 x 512 grid, per orientation ----
// orientation o: major axis MAJ[o], minor axis MIN_[o]; handles planes PLA[o] (spatial) and PLT[o] (the major axis vs time)
constexpr int SORT_BINS = 512, SORT_NB = 256;
__device__ constexpr int MAJ[3] = {0, 1, 2};
__device__ constexpr int MIN_[3] = {1, 2, 0};
__device__ constexpr int PLA[3] = {0, 3, 1};  // (x,y) (y,z) (x,z)
__device__ constexpr int PLT[3] = {2, 4, 5};  // (x,t) (y,t) (z,t)

// Sort cell of a point along `axis` = its texel column at the FINEST level, computed exactly like make_tap does, so
// all points of one cell share their four finest-level corner texels (a coarser cell grid would cut cells with texel
// boundaries -- align_corners grids of different levels do not nest -- and break the register run-length combining).
__device__ __forceinline__ int sort_cell(const HexArgs& a, int p, int axis) {
  const float u = (a.xyz[3 * (size_t)p + axis] - a.d.aabb_max[axis]) * (2.0f / (a.d.aabb_min[axis] - a.d.aabb_max[axis])) - 1.0f;
  const int W = a.d.res[a.d.levels - 1][axis];
  float ix = ((u + 1.f) / 2.f) * (float)(W - 1);
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  int cidx = (int)floorf(ix);
  if (W > SORT_BINS) cidx = (int)(((long long)cidx * SORT_BINS) / W);
  return min(SORT_BINS - 1, max(0, cidx));
}

struct SortWork {
  uint32_t* table;      // [3][SORT_NB][SORT_BINS]
  uint32_t* seg_start;  // [3][SORT_BINS + 1]
  uint32_t* tmp;        // [3][P]  indices grouped by major cell
  uint32_t* order;      // [3][P]  final order
  uint32_t* rank;       // [3][P]  inverse permutation: rank[o][order[o][k]] = k
};

template <bool WRITE>
__global__ void __launch_bounds__(256) hexsort_major_kernel(const HexArgs a, const SortWork w, int chunk) {
  __shared__ uint32_t cell[SORT_BINS];
  const int o = blockIdx.y;
  uint32_t* row = w.table + ((size_t)o * SORT_NB + blockIdx.x) * SORT_BINS;
  for (int i = threadIdx.x; i < SORT_BINS; i += 256) cell[i] = WRITE ? w.seg_start[o * (SORT_BINS + 1) + i] + row[i] : 0u;
  __syncthreads();
  const int g0 = blockIdx.x * chunk, g1 = min(a.P, g0 + chunk);
  for (int g = g0 + threadIdx.x; g < g1; g += 256) {
    const uint32_t pos = atomicAdd(&cell[sort_cell(a, g, MAJ[o])], 1u);
    if (WRITE) w.tmp[(size_t)o * a.P + pos] = (uint32_t)g;
  }
  if (!WRITE) {
    __syncthreads();
    for (int i = threadIdx.x; i < SORT_BINS; i += 256) row[i] = cell[i];
  }
}

// one workgroup per orientation: per-bin prefix over the SORT_NB workgroups, then exclusive scan of the bin totals
__global__ void __launch_bounds__(512) hexsort_scan_kernel(const SortWork w, int P) {
  __shared__ uint32_t tot[SORT_BINS];
  const int o = blockIdx.x, b = threadIdx.x;
  uint32_t* tab = w.table + (size_t)o * SORT_NB * SORT_BINS;
  uint32_t run = 0;
  for (int k = 0; k < SORT_NB; k++) {
    const uint32_t v = tab[(size_t)k * SORT_BINS + b];
    tab[(size_t)k * SORT_BINS + b] = run;
    run += v;
  }
  tot[b] = run;
  __syncthreads();
  if (b == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < SORT_BINS; i++) {
      w.seg_start[o * (SORT_BINS + 1) + i] = acc;
      acc += tot[i];
    }
    w.seg_start[o * (SORT_BINS + 1) + SORT_BINS] = acc;
  }
}

// one workgroup per (major bin, orientation): counting sort of the segment by minor cell
__global__ void __launch_bounds__(256) hexsort_minor_kernel(const HexArgs a, const SortWork w) {
  __shared__ uint32_t cnt[SORT_BINS];
  __shared__ uint32_t wsum[4];
  const int o = blockIdx.y, bin = blockIdx.x, tid = threadIdx.x;
  const uint32_t s0 = w.seg_start[o * (SORT_BINS + 1) + bin], s1 = w.seg_start[o * (SORT_BINS + 1) + bin + 1];
  if (s1 == s0) return;
  const uint32_t* tmp = w.tmp + (size_t)o * a.P;
  uint32_t* order = w.order + (size_t)o * a.P;
  for (int i = tid; i < SORT_BINS; i += 256) cnt[i] = 0u;
  __syncthreads();
  for (uint32_t k = s0 + tid; k < s1; k += 256) atomicAdd(&cnt[sort_cell(a, (int)tmp[k], MIN_[o])], 1u);
  __syncthreads();
  // exclusive scan of 512 counters: each thread owns two consecutive bins
  const uint32_t c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
  uint32_t incl = c0 + c1;
  const int lane = tid & 63, wave = tid >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = s0;
  for (int k = 0; k < wave; k++) base += wsum[k];
  const uint32_t excl = base + incl - (c0 + c1);
  __syncthreads();
  cnt[2 * tid] = excl;
  cnt[2 * tid + 1] = excl + c0;
  __syncthreads();
  for (uint32_t k = s0 + tid; k < s1; k += 256) {
    const uint32_t g = tmp[k];
    order[atomicAdd(&cnt[sort_cell(a, (int)g, MIN_[o])], 1u)] = g;
  }
}

__global__ void __launch_bounds__(256) hexsort_rank_kernel(int P, const uint32_t* __restrict__ order, uint32_t* __restrict__ rank) {
  const int k = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (k < P) rank[(size_t)o * P + order[(size_t)o * P + k]] = (uint32_t)k;
}

// ---- pass B: scatter in sorted order with register run-length combining ----
constexpr int SEG = 128;  // sorted points walked by one half-wave

// One bilinear footprint being accumulated in registers: key = texel offset of its nw corner (-1 = empty), flags bit0 =
// ne/se column in range, bit1 = sw/se row in range (the other three corners follow from key, flags and the plane width).
struct Foot {
  int key, flags;
  float a00, a01, a10, a11;
};
__device__ __forceinline__ void foot_flush(const Foot& f, float* __restrict__ gp, int W, int c) {
  if (f.key < 0) return;
  atomicAdd(&gp[(size_t)f.key * HEXC + c], f.a00);
  if (f.flags & 1) atomicAdd(&gp[(size_t)(f.key + 1) * HEXC + c], f.a01);
  if (f.flags & 2) atomicAdd(&gp[(size_t)(f.key + W) * HEXC + c], f.a10);
  if ((f.flags & 3) == 3) atomicAdd(&gp[(size_t)(f.key + W + 1) * HEXC + c], f.a11);
}
// Two-entry footprint cache.  align_corners grids of different levels do not nest, so inside one finest-level cell the
// points alternate between two (sometimes four) coarse footprints; remembering the previous one as well removes most of
// those flushes.
struct PackedTap {  // what the scatter needs of a Tap: 8 floats in LDS
  int key, flags;
  float w00, w01, w10, w11;
};
// Entries stay where they are (no MRU swap) and the hit path is BRANCH-FREE: both entries take an fma whose multiplicand is
// the gradient or 0.  The scatter kernel used to be instruction-bound on this function: the two half-waves of a wave walk
// different segments, so every data-dependent branch of the old hit-A / hit-B-swap / miss cascade ran both sides under
// complementary exec masks (~100 instructions per call, 32 calls per group of four points).  Only the miss -- about one
// call in four -- still branches: it flushes the entry that was NOT used last and installs the new footprint in its place.
struct Foot2 {
  int key0, key1, fl0, fl1, mru;
  float a0[4], a1[4];
};
__device__ __forceinline__ void foot2_init(Foot2& F) {
  F.key0 = F.key1 = -1;
  F.fl0 = F.fl1 = F.mru = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) F.a0[k] = F.a1[k] = 0.f;
}
__device__ __forceinline__ void foot2_flush_all(const Foot2& F, float* __restrict__ gp, int W, int c) {
  foot_flush(Foot{F.key0, F.fl0, F.a0[0], F.a0[1], F.a0[2], F.a0[3]}, gp, W, c);
  foot_flush(Foot{F.key1, F.fl1, F.a1[0], F.a1[1], F.a1[2], F.a1[3]}, gp, W, c);
}
__device__ __forceinline__ void foot2_add(Foot2& F, const PackedTap& t, float g, float* __restrict__ gp, int W, int c) {
  bool h0 = t.key == F.key0, h1 = t.key == F.key1;
  if (!(h0 || h1)) {  // miss (uniform inside the half-wave): evict the entry that is not the most recent one
    const bool v1 = F.mru == 0;
    foot_flush(Foot{v1 ? F.key1 : F.key0, v1 ? F.fl1 : F.fl0, v1 ? F.a1[0] : F.a0[0], v1 ? F.a1[1] : F.a0[1],
                    v1 ? F.a1[2] : F.a0[2], v1 ? F.a1[3] : F.a0[3]}, gp, W, c);
    F.key1 = v1 ? t.key : F.key1;  F.key0 = v1 ? F.key0 : t.key;
    F.fl1 = v1 ? t.flags : F.fl1;  F.fl0 = v1 ? F.fl0 : t.flags;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      F.a1[k] = v1 ? 0.f : F.a1[k];
      F.a0[k] = v1 ? F.a0[k] : 0.f;
    }
    h1 = v1;
    h0 = !v1;
  }
  const float g0 = h0 ? g : 0.f, g1 = h1 ? g : 0.f;
  F.a0[0] = __builtin_fmaf(g0, t.w00, F.a0[0]); F.a0[1] = __builtin_fmaf(g0, t.w01, F.a0[1]);
  F.a0[2] = __builtin_fmaf(g0, t.w10, F.a0[2]); F.a0[3] = __builtin_fmaf(g0, t.w11, F.a0[3]);
  F.a1[0] = __builtin_fmaf(g1, t.w00, F.a1[0]); F.a1[1] = __builtin_fmaf(g1, t.w01, F.a1[1]);
  F.a1[2] = __builtin_fmaf(g1, t.w10, F.a1[2]); F.a1[3] = __builtin_fmaf(g1, t.w11, F.a1[3]);
  F.mru = h1 ? 1 : 0;
}

// A half-wave (32 lanes = the 32 channels) walks SEG consecutive points of the sorted order.  The kernel used to be
// VALU-bound on make_tap, which all 32 lanes repeated for each of the 8 (level, plane) taps of a point; now the 32 lanes
// compute the 8 taps of FOUR points at once (lane = point q x tap j), park them in LDS, and every lane reads them back
// with broadcast loads while it accumulates its channel.
constexpr int SCATTER_LG = 2;          // levels handled per walk of a segment
constexpr int SCATTER_WG_PER_CU = 4;   // 4 waves per SIMD: the walk is bound by per-wave issue latency (IPC ~0.2), not by VALU throughput
constexpr int TAPF = 8;  // floats per packed tap in LDS (6 used; 32-byte slots keep the 16-byte reads aligned)
__global__ void __launch_bounds__(256, SCATTER_WG_PER_CU) hexplane_scatter_kernel(const HexArgs a, const float* __restrict__ G,
                                                               const uint32_t* __restrict__ order_all) {
  __shared__ __attribute__((aligned(16))) float tapbuf[8][2][4][8][TAPF];  // [half-wave][double buffer][point][tap]
  const int o = blockIdx.y;
  const int c = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int q = c >> 3, j = c & 7;  // tap-phase role: point q of the group of four, tap j = (level j >> 1, kind j & 1)
  const int seg = blockIdx.x * 8 + hw;
  const int k0 = seg * SEG, k1 = min(a.P, k0 + SEG);
  if (k0 >= a.P) return;  // whole half-waves drop out; the LDS traffic below is private to a half-wave (wave-ordered)
  const uint32_t* order = order_all + (size_t)o * a.P;
  const size_t PL = (size_t)a.P * HEXC;
  const int i0 = PLA[o], i1 = PLT[o];
  const int ip = (j & 1) ? i1 : i0;                   // the plane of this lane's tap
  const int axw = PAIR0[ip], axh = PAIR1[ip];
  constexpr int LG = SCATTER_LG;  // levels handled together: LG levels x 2 planes x 2 footprints live in registers
  for (int l0 = 0; l0 < a.d.levels; l0 += LG) {
    Foot2 ft[LG][2];
#pragma unroll
    for (int l = 0; l < LG; l++)
#pragma unroll
      for (int m = 0; m < 2; m++) foot2_init(ft[l][m]);
    const int lt = l0 + (j >> 1);                     // level of this lane's tap
    const bool tap_on = (j >> 1) < LG && lt < a.d.levels;
    const int Wt = tap_on ? a.d.res[lt][axw] : 2, Ht = tap_on ? a.d.res[lt][axh] : 2;
    // Three-stage software pipeline per lane role (point q of a group, tap j): the sorted index of group g+2, the
    // coordinates of group g+1 and the taps of group g+1 are produced while group g is accumulated, so neither the
    // index -> position load chain nor the tap arithmetic sits between a group's G loads and their use.
    auto load_index = [&](int kb) { return (int)order[min(kb + q, k1 - 1)]; };
    auto load_coords = [&](int p, float* u) { point_coords(a, p, u); };
    auto store_taps = [&](const float* u, int buf) {
      const Tap t = make_tap(u[axw], u[axh], Wt, Ht);
      float4 lo;
      lo.x = __int_as_float(t.o00);
      lo.y = __int_as_float((t.o01 >= 0 ? 1 : 0) | (t.o10 >= 0 ? 2 : 0));
      lo.z = t.w00;
      lo.w = t.w01;
      float* dst = &tapbuf[hw][buf][q][j][0];
      *reinterpret_cast<float4*>(dst) = lo;
      *reinterpret_cast<float2*>(dst + 4) = make_float2(t.w10, t.w11);
    };
    float un[4];                       // coordinates of the NEXT group's point
    {
      float u0[4];
      load_coords(load_index(k0), u0);
      store_taps(u0, 0);
    }
    load_coords(load_index(k0 + 4), un);
    int pnn = load_index(k0 + 8);      // index of the group after next
    int buf = 0;
    for (int kb = k0; kb < k1; kb += 4, buf ^= 1) {
      // 1. this group's G rows: 4 points x 8 rows requested at once (the walk is latency-bound, not bandwidth-bound)
      float g[4][LG][2];
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        const size_t k = (size_t)min(kb + qq, k1 - 1);
#pragma unroll
        for (int l = 0; l < LG; l++) {
          // unconditional (level clamped; slabs exist for every plane): a load behind a uniform branch costs two branch
          // instructions and splits the basic block the scheduler could have filled
          const int lv = min(l0 + l, a.d.levels - 1);
          g[qq][l][0] = G[(size_t)((o * a.d.levels + lv) * 2 + 0) * PL + k * HEXC + c];
          g[qq][l][1] = G[(size_t)((o * a.d.levels + lv) * 2 + 1) * PL + k * HEXC + c];
        }
      }
      // 2. the NEXT group's taps from coordinates loaded one iteration ago; then advance the two prefetch stages
      store_taps(un, buf ^ 1);
      load_coords(pnn, un);
      pnn = load_index(kb + 12);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // 3. accumulate
      const int nq = min(4, k1 - kb);
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        if (qq >= nq) break;
#pragma unroll
        for (int l = 0; l < LG; l++) {
          if (l0 + l >= a.d.levels) break;
#pragma unroll
          for (int m = 0; m < 2; m++) {
            float* gp = a.gplanes[l0 + l][m ? i1 : i0];
            if (gp == nullptr) continue;
            const float* src = &tapbuf[hw][buf][qq][l * 2 + m][0];
            const float4 lo = *reinterpret_cast<const float4*>(src);
            const float2 hi = *reinterpret_cast<const float2*>(src + 4);
            PackedTap t;
            t.key = __float_as_int(lo.x); t.flags = __float_as_int(lo.y);
            t.w00 = lo.z; t.w01 = lo.w; t.w10 = hi.x; t.w11 = hi.y;
            foot2_add(ft[l][m], t, g[qq][l][m], gp, a.d.res[l0 + l][PAIR0[m ? i1 : i0]], c);
          }
        }
      }
    }
#pragma unroll
    for (int l = 0; l < LG; l++) {
      if (l0 + l >= a.d.levels) break;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        float* gp = a.gplanes[l0 + l][m ? i1 : i0];
        if (gp == nullptr) continue;
        const int W = a.d.res[l0 + l][PAIR0[m ? i1 : i0]];
        foot2_flush_all(ft[l][m], gp, W, c);
      }
    }
  }
}

// ---- uniform time: the (axis, t) planes collapse to 1-D row tables ---------------------------------------------------
// When every point carries the same t, the t half of the bilinear footprint is the same for all of them:
//   R[x][c] = P[y0][x][c] * (y1 - iy) + P[y1][x][c] * (iy - y0)       (iy from time[0], exactly as make_tap computes it)
// The tables are handed to the SAME kernels as planes of height 1 (make_tap then yields iy = 0, weights (x1-ix, ix-x0, 0, 0)
// and two out-of-range taps), so a time-plane sample costs 2 L1-resident fetches instead of 4 gathers, its scatter is
// one-dimensional, and the table gradients are folded back into the two plane rows afterwards.
struct TimeRows {
  int W[S3G_HEX_MAX_LEVELS][3], H[S3G_HEX_MAX_LEVELS];
  const float* plane[S3G_HEX_MAX_LEVELS][3];
  float* gplane[S3G_HEX_MAX_LEVELS][3];
  float* table[S3G_HEX_MAX_LEVELS][3];
  float* gtable[S3G_HEX_MAX_LEVELS][3];
  const float* time;
};
__device__ __forceinline__ void time_rows(const float* time, int H, int& y0, int& y1, float& w0, float& w1) {
  float iy = ((time[0] + 1.f) / 2.f) * (float)(H - 1);
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float f0 = floorf(iy);
  y0 = (int)f0;
  y1 = y0 + 1 < H ? y0 + 1 : -1;
  w0 = (f0 + 1.f) - iy;
  w1 = iy - f0;
}
// grid = (row blocks, 3 planes, levels); BACKWARD: gplane rows += w * gtable (no other kernel touches these rows meanwhile)
template <bool BACKWARD>
__global__ void __launch_bounds__(256) hexplane_time_rows_kernel(const TimeRows r) {
  const int l = blockIdx.z, k = blockIdx.y, W = r.W[l][k];
  int y0, y1;
  float w0, w1;
  time_rows(r.time, r.H[l], y0, y1, w0, w1);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < W * HEXC; e += gridDim.x * 256) {
    if (!BACKWARD) {
      float v = r.plane[l][k][(size_t)y0 * W * HEXC + e] * w0;
      if (y1 >= 0) v += r.plane[l][k][(size_t)y1 * W * HEXC + e] * w1;
      r.table[l][k][e] = v;
    } else if (r.gplane[l][k] != nullptr) {
      const float g = r.gtable[l][k][e];
      r.gplane[l][k][(size_t)y0 * W * HEXC + e] += g * w0;
      if (y1 >= 0) r.gplane[l][k][(size_t)y1 * W * HEXC + e] += g * w1;
    }
  }
}
static size_t time_table_floats(const s3g_hexplane_desc* d) {
  size_t n = 0;
  for (int l = 0; l < d->levels; l++)
    for (int k = 0; k < 3; k++) n += (size_t)d->res[l][k] * HEXC;
  return n;
}
// Fills `r`, points the time planes of `a.d` at the tables (height 1) and launches the table build.
static void use_time_rows(HexArgs& a, TimeRows& r, float* tables, float* gtables, hipStream_t stream) {
  static const int TP[3] = {2, 4, 5};  // (x,t) (y,t) (z,t); their spatial axis is 0, 1, 2
  memset(&r, 0, sizeof r);
  r.time = a.time;
  size_t off = 0;
  int maxW = 0;
  for (int l = 0; l < a.d.levels; l++) {
    r.H[l] = a.d.res[l][3];
    for (int k = 0; k < 3; k++) {
      r.W[l][k] = a.d.res[l][k];
      maxW = max(maxW, r.W[l][k]);
      r.plane[l][k] = a.d.planes[l][TP[k]];
      r.gplane[l][k] = a.gplanes[l][TP[k]];
      r.table[l][k] = tables + off;
      r.gtable[l][k] = gtables ? gtables + off : nullptr;
      off += (size_t)r.W[l][k] * HEXC;
      a.d.planes[l][TP[k]] = r.table[l][k];
      if (gtables) a.gplanes[l][TP[k]] = r.gplane[l][k] ? r.gtable[l][k] : nullptr;
    }
    a.d.res[l][3] = 1;
  }
  hipLaunchKernelGGL(hexplane_time_rows_kernel<false>, dim3((maxW * HEXC + 255) / 256, 3, a.d.levels), dim3(256), 0, stream, r);
}

static int check_desc(const s3g_hexplane_desc* d) {
  if (!d || d->levels < 1 || d->levels > S3G_HEX_MAX_LEVELS) {
    set_error("hexplane: bad descriptor (levels)");
    return S3G_ERR_INVALID_ARG;
  }
  for (int l = 0; l < d->levels; l++) {
    for (int k = 0; k < 4; k++)
      if (d->res[l][k] < 2) {
        set_error("hexplane: resolution must be >= 2");
        return S3G_ERR_INVALID_ARG;
      }
    for (int i = 0; i < 6; i++)
      if (!d->planes[l][i]) {
        set_error("hexplane: NULL plane pointer");
        return S3G_ERR_INVALID_ARG;
      }
  }
  return S3G_OK;
}

}  // namespace s3g

using namespace s3g;

extern "C" size_t s3g_hexplane_forward_workspace_bytes(const s3g_hexplane_desc* d) {
  if (!d || d->levels < 1 || d->levels > S3G_HEX_MAX_LEVELS || !d->uniform_time) return 0;
  return time_table_floats(d) * sizeof(float);
}

extern "C" int s3g_hexplane_forward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                    float* features, const uint32_t* proc_order, void* workspace, void* stream_) {
  if (int e = check_desc(d)) return e;
  if (P < 0 || (P > 0 && (!xyz || !time || !features || (d->uniform_time && !workspace)))) {
    set_error("s3g_hexplane_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  HexArgs a;
  memset(&a, 0, sizeof a);
  a.d = *d; a.P = P; a.xyz = xyz; a.time = time; a.feat = features; a.proc_order = proc_order;
  TimeRows rows;
  if (d->uniform_time) use_time_rows(a, rows, (float*)workspace, nullptr, (hipStream_t)stream_);
  const int blocks = (P + 31) / 32;  // one group of 32 points per workgroup measured best (0.567 -> 0.535 ms vs a 4096 cap)
  profile_begin(S3G_PROFILE_HEXPLANE_FORWARD, (hipStream_t)stream_);
  hipLaunchKernelGGL(hexplane_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
  profile_end(S3G_PROFILE_HEXPLANE_FORWARD, (hipStream_t)stream_, (double)P, (double)d->levels);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" size_t s3g_hexplane_backward_workspace_bytes(const s3g_hexplane_desc* d, int P) {
  if (!d || d->levels < 1 || d->levels > S3G_HEX_MAX_LEVELS || P < 0) return 0;
  const int levels = d->levels;
  Carver c(nullptr);
  c.take<float>((size_t)levels * 6 * P * HEXC);
  if (d->uniform_time) c.take<float>(2 * time_table_floats(d));
  c.take<uint32_t>((size_t)3 * SORT_NB * SORT_BINS);
  c.take<uint32_t>((size_t)3 * (SORT_BINS + 1));
  c.take<uint32_t>((size_t)3 * P);
  c.take<uint32_t>((size_t)3 * P);
  c.take<uint32_t>((size_t)3 * P);
  return c.bytes();
}

extern "C" int s3g_hexplane_backward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                     const float* dL_dfeatures, float* dL_dxyz,
                                     float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* workspace,
                                     uint32_t* sort_state, int sort_reuse, void* stream_) {
  if (int e = check_desc(d)) return e;
  if (P < 0 || (P > 0 && (!xyz || !time || !dL_dfeatures || !dL_dxyz || !dL_dplanes || !workspace)) ||
      (sort_reuse && !sort_state)) {
    set_error("s3g_hexplane_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  hipStream_t stream = (hipStream_t)stream_;
  HexArgs a;
  memset(&a, 0, sizeof a);
  a.d = *d; a.P = P; a.xyz = xyz; a.time = time; a.gfeat = dL_dfeatures; a.gxyz = dL_dxyz;
  for (int l = 0; l < d->levels; l++)
    for (int i = 0; i < 6; i++) a.gplanes[l][i] = dL_dplanes[l][i];
  Carver c(workspace);
  float* G = c.take<float>((size_t)d->levels * 6 * P * HEXC);
  float* tables = d->uniform_time ? c.take<float>(2 * time_table_floats(d)) : nullptr;
  SortWork w;
  w.table = c.take<uint32_t>((size_t)3 * SORT_NB * SORT_BINS);
  w.seg_start = c.take<uint32_t>((size_t)3 * (SORT_BINS + 1));
  w.tmp = c.take<uint32_t>((size_t)3 * P);
  w.order = c.take<uint32_t>((size_t)3 * P);
  w.rank = c.take<uint32_t>((size_t)3 * P);
  if (sort_state) {  // caller-owned, persistent
    w.order = sort_state;
    w.rank = sort_state + (size_t)3 * P;
  }

  // 1. three spatial orders (2-level LDS counting sorts) and their inverse permutations
  if (!sort_reuse) {
    const int chunk = (((P + SORT_NB - 1) / SORT_NB + 255) / 256) * 256;
    hipLaunchKernelGGL(hexsort_major_kernel<false>, dim3(SORT_NB, 3), dim3(256), 0, stream, a, w, chunk);
    hipLaunchKernelGGL(hexsort_scan_kernel, dim3(3), dim3(512), 0, stream, w, P);
    hipLaunchKernelGGL(hexsort_major_kernel<true>, dim3(SORT_NB, 3), dim3(256), 0, stream, a, w, chunk);
    hipLaunchKernelGGL(hexsort_minor_kernel, dim3(SORT_BINS, 3), dim3(256), 0, stream, a, w);
    hipLaunchKernelGGL(hexsort_rank_kernel, dim3((P + 255) / 256, 3), dim3(256), 0, stream, P, w.order, w.rank);
    S3G_HIP_CHECK(hipGetLastError());
  }
  // 2. per-point pass, walking the points in (x,y) order so neighbouring half-waves share texels
  //    (the sorts above used the real resolutions; from here on the time planes are height-1 row tables if uniform_time)
  TimeRows rows;
  if (d->uniform_time) {
    const size_t nt = time_table_floats(d);
    S3G_HIP_CHECK(hipMemsetAsync(tables + nt, 0, nt * sizeof(float), stream));
    use_time_rows(a, rows, tables, tables + nt, stream);
  }
  a.proc_order = w.order;
  const int blocks = (P + 31) / 32;
  profile_begin(S3G_PROFILE_HEXPLANE_BACKWARD_POINT, stream);
  hipLaunchKernelGGL(hexplane_backward_point_kernel, dim3(blocks), dim3(256), 0, stream, a, G, w.rank);
  profile_end(S3G_PROFILE_HEXPLANE_BACKWARD_POINT, stream, (double)P, (double)d->levels);
  S3G_HIP_CHECK(hipGetLastError());
  const int nseg = (P + SEG - 1) / SEG;
  profile_begin(S3G_PROFILE_HEXPLANE_SCATTER, stream);
  hipLaunchKernelGGL(hexplane_scatter_kernel, dim3((nseg + 7) / 8, 3), dim3(256), 0, stream, a, G, w.order);
  profile_end(S3G_PROFILE_HEXPLANE_SCATTER, stream, (double)P, (double)d->levels);
  if (d->uniform_time) {
    int maxW = 0;
    for (int l = 0; l < d->levels; l++)
      for (int k = 0; k < 3; k++) maxW = max(maxW, d->res[l][k]);
    hipLaunchKernelGGL(hexplane_time_rows_kernel<true>, dim3((maxW * HEXC + 255) / 256, 3, d->levels), dim3(256), 0, stream, rows);
  }
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
