// MI355X-native multi-resolution HexPlane sampler: forward and backward, one fused pass each.
//
// Reference: scene/hexplane.py:73-106 runs 4 levels x 6 planes = 24 F.grid_sample launches, each materialising a
// [P,32] tensor, then 20 elementwise products and a concat; autograd replays the same 24 in backward.
//
// Planes are stored channel-last, so one texel = 32 channels = one 128-byte line.  Forward and the per-point backward give
// a point to EIGHT lanes (4 channels each, one 16-byte load per texel); the bilinear taps are computed once per point and
// shared through LDS; points are processed in a 3-D blocked order, groups dealt to the XCDs in contiguous eighths.  The
// product over planes never leaves registers.  Arithmetic follows torch's grid_sampler_2d (bilinear, border,
// align_corners=True) op for op; contraction is off.  When all points share one timestamp (desc.uniform_time) the three
// (axis, t) planes of a level are first collapsed to 1-D row tables (see "uniform time" below).
//
// Backward without an atomic storm.  A direct scatter is 96 line-coalesced float atomics per point; MI355X retires
// ~10 G such line-ops/s whatever the contention (tools/ubench/atomic_lines.hip), i.e. 11.5 ms at 1.2 M points.  So:
//   pass A  (blocked order) re-gathers the taps, finishes dL/dxyz and writes to the scratch G ONE row per level,
//                           T = dL/dfeature * feature (point-major, 512 B per point, streamed).  Round 4 (S3G_HEX_SLAB_DIV,
//                           hexplane_backward_pointdiv_kernel): T straight from the forward's saved output, dL/ds_i = T / s_i plane by
//                           plane at four waves per SIMD; rounds 1-3 (hexplane_backward_point_kernel): the product rule with six
//                           samples live at two waves per SIMD (rounds 1-2 also wrote dL/ds of all 24 plane-levels, 3 KB per point);
//   sort    2-level counting sorts of the point indices with LDS histograms (no global atomics, no library sort) -- round 4: ONE
//           ORDER PER (orientation, LEVEL), by (major, minor) texel cell of THAT level (thirteen sorts with the blocked processing
//           order; rounds 1-3: the finest level's cells only, four sorts); the orders only steer the walks, so the caller may keep
//           them for several iterations (sort_state / sort_reuse);
//   pass B  (sorted orders, one launch: blockIdx.y = orientation * levels + level): a half-wave (32 lanes = the 32 channels) walks
//           a run of consecutive points of ITS order keeping one bilinear footprint per plane kind (the spatial plane and the
//           (major, t) plane of the orientation) in registers -- sums AND texel values -- and only issues atomics when the footprint
//           changes (two instead of four when the walk just steps to the neighbouring footprint): the 96 line-ops per point
//           drop to ~2 (rounds 1-3, every level in the finest order with a two-entry cache: ~5-6).  Taps are computed
//           cooperatively (lane = point x tap) and shared through LDS; index, coordinate and tap computation run one to two groups
//           ahead of the accumulation.
#include <atomic>

#include "hexplane_dev.hpp"

namespace s3g {

// G slab contents.  0 (rounds 1-2): dL/d(sample) of all 24 plane-levels, 3 KB per point.  1 (round 3): ONE row per level,
// T = dL/dfeature * feature = g * prod_j s_j (the per-point pass has it for free at the end of its product rule); the scatter
// walk re-derives the one sample it is about to scatter -- the four texels (two row-table entries) of the footprint it is
// accumulating anyway, L1-resident in the sorted order -- and uses dL/ds_i = T / s_i.  512 B per point written instead of
// 3 KB, 1.5 KB read back instead of 3 KB.  A sample whose magnitude is not safely divisible (|s| <= 1e-18, or not finite) is
// left out by the walk and scattered EXACTLY (g * prod_{j != i} s_j, direct atomics) by the per-point pass, which has all six
// samples: both passes evaluate the same predicate on the same bits of s (same taps, same operation order, contraction off).
// (Rounds 1-2 stored dL/d(sample) for all 24 plane-levels -- 3 KB per point -- and rounds 1-3 walked every level in the finest
//  level's cell order, two levels per walk: both forms were compile-time switches until round 5 and are gone; DESIGN.md 6 / 10
//  keep their measurements.)  One walk order per orientation AND level: each (orientation, level) walk is monotone in its own
// cells (see sort_cell below).
#define S3G_POINT_PREFETCH 1
constexpr float TSLAB_SAFE = 1e-18f;
__device__ __forceinline__ bool tslab_divisible(float s) { return fabsf(s) > TSLAB_SAFE && fabsf(s) < __builtin_huge_valf(); }
// GATE of the exact fallback in the per-point passes: wider than the predicate itself.  The passes that DIVIDE decide with
// tslab_divisible on the sample they re-derive; the fallback (tslab_exact_scatter / exact_du) re-derives the sample the same way and
// applies the same predicate per channel.  The gate only has to make sure the fallback is ENTERED whenever some pass might refuse
// to divide: a sample within a factor of four of the threshold enters it even if this kernel's own evaluation sits on the safe side.
__device__ __forceinline__ bool tslab_near_unsafe(float s) { return !(fabsf(s) > 4.f * TSLAB_SAFE && fabsf(s) < __builtin_huge_valf()); }

template <bool UT>
__global__ void __launch_bounds__(256) hexplane_forward_kernel(const HexArgs a) {
  extern __shared__ float4 tapbuf[];   // [32 points][levels][TAP_SLOTS]
  const int j = threadIdx.x & 7, c4 = j * 4, slot = threadIdx.x >> 3;
  const int F = a.d.levels * HEXC;
  float4* taps = tapbuf + (size_t)slot * tap_stride(a.d.levels);
  for (int p0 = xcd_group(blockIdx.x, gridDim.x) * 32; p0 < a.P; p0 += gridDim.x * 32) {
    const int pi = p0 + slot;
    const bool live = pi < a.P;
    const int p = live ? (a.proc_order ? (int)a.proc_order[pi] : pi) : 0;
    float u[4];
    point_coords(a, p, u);
    wave_lds_sync();   // the previous point's taps have been read
    produce_taps(a, u, j, taps);
    wave_lds_sync();
    for (int l = 0; l < a.d.levels; l++) {
      float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        const float* pl = a.d.planes[l][i];
        float4 s;
        if (UT && IS_TIME_PLANE[i]) {
          const PointTap t = read_tap<true>(taps, l, i, W, H, c4);
          s = texel4(pl, t.off) * t.gx;
          s = s + texel4(pl, t.off + t.dx) * t.fx;
        } else {
          const PointTap t = read_tap<false>(taps, l, i, W, H, c4);
          s = texel4(pl, t.off) * (t.gx * t.gy);
          s = s + texel4(pl, t.off + t.dx) * (t.fx * t.gy);
          s = s + texel4(pl, t.off + t.dy) * (t.gx * t.fy);
          s = s + texel4(pl, t.off + t.dy + t.dx) * (t.fx * t.fy);
        }
        prod = prod * s;
      }
      if (live) {
        f4v v = {prod.x, prod.y, prod.z, prod.w};
        f4v* dst = reinterpret_cast<f4v*>(a.feat + (size_t)p * F + l * HEXC + c4);
        if (FEAT_NONTEMPORAL) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
  }
}

// ---- pass A: per point, dL/dxyz and the level's row T = dL/dfeature * feature -> G ----
// G layout (point-major): the `levels` rows of a point are contiguous, points in PROCESSING order, so pass A streams its stores:
// row l of processing position pi = G + (pi * levels + l) * 32; a scatter walk reads it by the position comp[oi][k] of its k-th point.


// Same lane mapping and tap sharing as the forward.  Per plane only the sample s and its two coordinate derivatives are kept:
// ds/dix = (ne - nw)(y1 - iy) + (se - sw)(iy - y0),  ds/diy = (sw - nw)(x1 - ix) + (se - ne)(ix - x0)  (the four terms of
// torch's grid_sampler_2d_backward, grouped).
//
// What bounds pass A (cfg3, 1.2 M points, 1.43 ms; PMC pass in profiles/r02_hexplane_sq_pmc.txt): its waves sit parked on
// s_waitcnt 63 % of their resident time and issue VALU 18 % of it (288 M wave-instructions = 0.5 ms of pure issue) -- at two
// waves per SIMD (172 registers: six samples and their derivatives have to be live for the product rule) nothing hides a
// memory round trip (1.07 ms with the G stores compiled out; issuing the next level's loads BEFORE this level's stores -- vmcnt
// is one in-order counter for loads and stores -- changed nothing: 1.437 vs 1.435 ms).  Everything tried against the latency
// made it slower or did nothing, because each costs registers and this kernel has none to give: a second texel register set prefetching the next level (persistent workgroups,
// next group's index / coordinates / taps prefetched as well): 1.86 ms at 256 VGPRs with spills; the same unrolled so that
// no set crosses a loop back-edge, next level's loads issued between samples() and this level's stores: 1.94 ms (285 VGPRs,
// or 256 with spills); launch_bounds for three waves: 1.69 ms (spills).  Without effect: halving the VALU work (shared taps),
// pointing every texel load at one hot line, the blocked order / XCD-contiguous groups (the forward gains 8 % from those).
// Point-major G (24 rows of a point contiguous, points in processing order -> streaming stores): 1.50 -> 1.43 ms; padding the
// tap slots against LDS bank conflicts: 1.42 -> 1.33 ms; more waves (16 lanes per point: 4 per SIMD) 1.73 ms.
// V = the channels one lane owns: f4v (8 lanes per point) or f2v (16 lanes per point: half the live registers per lane --
// the six samples and their derivatives -- hence twice the waves per SIMD to hide the round trips, for ~20 % more VALU work).
typedef float f2v_ __attribute__((ext_vector_type(2)));
template <typename V> struct vec_of;
template <> struct vec_of<f4v> { static constexpr int N = 4; };
template <> struct vec_of<f2v_> { static constexpr int N = 2; };
template <typename V> __device__ __forceinline__ V vsplat(float x);
template <> __device__ __forceinline__ f4v vsplat<f4v>(float x) { return f4v{x, x, x, x}; }
template <> __device__ __forceinline__ f2v_ vsplat<f2v_>(float x) { return f2v_{x, x}; }
__device__ __forceinline__ float vdot(f4v a, f4v b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float vdot(f2v_ a, f2v_ b) { return a.x * b.x + a.y * b.y; }
template <typename V>
__device__ __forceinline__ V texelv(const float* __restrict__ plane, uint32_t byte_off) {
  return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(plane) + byte_off);
}
template <typename V>
struct LevelIn {       // texels of one level's planes (uniform time: the three spatial planes only) + the dL/dfeature row
  V v[6][4];
  V g;
};
template <bool UT, typename V>
__device__ __forceinline__ void issue_level(const HexArgs& a, const float4* __restrict__ taps, int l, int c0, const float* __restrict__ grow, LevelIn<V>& in) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
    const float* pl = a.d.planes[l][i];
    if (UT && IS_TIME_PLANE[i]) {
      // row tables (a few hundred KB in total, L1 / L2 resident) are read where they are used
    } else {
      const PointTap t = read_tap<false>(taps, l, i, W, H, c0);
      in.v[i][0] = texelv<V>(pl, t.off);
      in.v[i][1] = texelv<V>(pl, t.off + t.dx);
      in.v[i][2] = texelv<V>(pl, t.off + t.dy);
      in.v[i][3] = texelv<V>(pl, t.off + t.dy + t.dx);
    }
  }
  const V* src = reinterpret_cast<const V*>(grow + l * HEXC);
  in.g = GFEAT_NONTEMPORAL ? __builtin_nontemporal_load(src) : *src;
}
// The arithmetic of one level in two halves:
//   samples()  texels -> s, ds/dix, ds/diy per plane (the texel registers are dead afterwards);
//   finish()   product rule -> six G rows (stored when `store`) and this level's share of dL/du.
template <typename V>
struct LevelS {
  V s[6], dX[6], dY[6];
  float mx[6], my[6];
};
template <bool UT, typename V>
__device__ __forceinline__ void samples_level(const HexArgs& a, const float4* __restrict__ taps, int l, int c0, const LevelIn<V>& in, LevelS<V>& S) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
    if (UT && IS_TIME_PLANE[i]) {
      const PointTap t = read_tap<true>(taps, l, i, W, H, c0);
      const float* pl = a.d.planes[l][i];
      const V v00 = texelv<V>(pl, t.off), v01 = texelv<V>(pl, t.off + t.dx);
      S.s[i] = v00 * t.gx;
      S.s[i] = S.s[i] + v01 * t.fx;
      S.dX[i] = v01 - v00;
      S.dY[i] = vsplat<V>(0.f);
      S.mx[i] = t.mx; S.my[i] = 0.f;
    } else {
      const PointTap t = read_tap<false>(taps, l, i, W, H, c0);
      const V v00 = in.v[i][0], v01 = in.v[i][1], v10 = in.v[i][2], v11 = in.v[i][3];
      V acc = v00 * (t.gx * t.gy);
      acc = acc + v01 * (t.fx * t.gy);
      acc = acc + v10 * (t.gx * t.fy);
      acc = acc + v11 * (t.fx * t.fy);
      S.s[i] = acc;
      // a corner that is out of range is the nw / ne / sw texel again: its difference terms are then multiplied by an
      // exactly-zero mask (mx or my) below, as the reference's are by the border clip
      S.dX[i] = (v01 - v00) * t.gy + (v11 - v10) * t.fy;
      S.dY[i] = (v10 - v00) * t.gx + (v11 - v01) * t.fx;
      S.mx[i] = t.mx; S.my[i] = t.my;
    }
  }
}
template <typename V> __device__ __forceinline__ float vget(V v, int k);
template <> __device__ __forceinline__ float vget<f4v>(f4v v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
template <> __device__ __forceinline__ float vget<f2v_>(f2v_ v, int k) { return k == 0 ? v.x : v.y; }
// one sample of plane i of level l at point coordinates u, channel c: make_tap + the four weighted texels in the order every
// kernel of this file uses (with uniform time the (axis, t) planes are height-1 row tables: res[l][3] == 1, iy == 0)
__device__ __forceinline__ float walk_sample(const HexArgs& a, int l, int i, const float* u, int c) {
  const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
  const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
  const float* pl = a.d.planes[l][i];
  float acc = fetch(pl, t.o00, c) * t.w00;
  acc = acc + fetch(pl, t.o01, c) * t.w01;
  acc = acc + fetch(pl, t.o10, c) * t.w10;
  acc = acc + fetch(pl, t.o11, c) * t.w11;
  return acc;
}
// The scatter walk divides T by the sample it re-derives; where it cannot (same predicate on the same bits) the exact gradient
// g * prod_{j != i} s_j is scattered here with the walk's own corner weights.  Runs for (nearly) zero or non-finite samples only,
// so it keeps nothing of the hot path's registers: everything is re-derived from the point's coordinates.
template <typename V>
__device__ __forceinline__ void tslab_exact_scatter(const HexArgs& a, int p, int l, int c0, V g, uint32_t badbits) {
  float u[4];
  point_coords(a, p, u);
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    float* gp = a.gplanes[l][i];
    if (!((badbits >> i) & 1u) || gp == nullptr) continue;
    const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
    const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
#pragma unroll 1
    for (int k = 0; k < vec_of<V>::N; k++) {
      const int c = c0 + k;
      if (tslab_divisible(walk_sample(a, l, i, u, c))) continue;
      float gk = vget<V>(g, k);
#pragma unroll 1
      for (int jj = 0; jj < 6; jj++)
        if (jj != i) gk *= walk_sample(a, l, jj, u, c);
      atomicAdd(gp + (size_t)t.o00 * HEXC + c, t.w00 * gk);
      if (t.o01 >= 0) atomicAdd(gp + (size_t)t.o01 * HEXC + c, t.w01 * gk);
      if (t.o10 >= 0) atomicAdd(gp + (size_t)t.o10 * HEXC + c, t.w10 * gk);
      if (t.o11 >= 0) atomicAdd(gp + (size_t)t.o11 * HEXC + c, t.w11 * gk);
    }
  }
}
template <bool UT, typename V>
__device__ __forceinline__ void finish_level(const HexArgs& a, int p, int l, int c0, const LevelS<V>& S, V g,
                                             bool store, float* __restrict__ G, size_t gbase, float* du) {
  // product rule in the order autograd applies it to ((((1*s0)*s1)*s2)*s3)*s4)*s5: pre[i] = prod_{j<i} s_j, suffix by recursion
  V pre[6];
  pre[0] = vsplat<V>(1.f);
#pragma unroll
  for (int i = 1; i < 6; i++) pre[i] = pre[i - 1] * S.s[i - 1];
  V gs = g;  // dL/d(prefix product through plane i)
  uint32_t badbits = 0;   // T-slab: planes with a sample the scatter walk cannot divide by
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    const V gi = gs * pre[i];  // dL/ds_i
    gs = gs * S.s[i];
    if (store) {
#pragma unroll
      for (int k = 0; k < vec_of<V>::N; k++) badbits |= tslab_near_unsafe(vget<V>(S.s[i], k)) ? (1u << i) : 0u;
      if (PAIR0[i] < 3) du[PAIR0[i]] += S.mx[i] * vdot(S.dX[i], gi);
      if (PAIR1[i] < 3) du[PAIR1[i]] += S.my[i] * vdot(S.dY[i], gi);
    }
  }
  if (store) {   // gs = g * s5 * s4 * ... * s0 = dL/dfeature * feature: the level's ONE row, gbase = position * levels rows
    V* trow = reinterpret_cast<V*>(G + gbase + (size_t)(l * HEXC + c0));
    if (G_NONTEMPORAL) __builtin_nontemporal_store(gs, trow);
    else *trow = gs;
    if (badbits) tslab_exact_scatter<V>(a, p, l, c0, g, badbits);   // rare: a sample that is (nearly) zero or not finite
  }
}

using PointV = f4v;    // channels per lane of pass A (f2v_: 106 VGPRs = 4 waves per SIMD, but 1.73 vs 1.53 ms)
template <bool UT, typename V, int LV>   // LV > 0: level count at compile time (unrolled: the per-level plane pointers and resolutions are fetched up front instead of four dependent scalar loads per level)
__global__ void __launch_bounds__(256) hexplane_backward_point_kernel(const HexArgs a, float* __restrict__ G) {
  constexpr int CPL = vec_of<V>::N, LPP = HEXC / CPL, PPW = 256 / LPP;   // channels per lane, lanes per point, points per workgroup
  extern __shared__ float4 tapbuf[];   // [PPW points][levels][TAP_SLOTS]
  const int j = threadIdx.x & (LPP - 1), c0 = j * CPL, slot = threadIdx.x / LPP;
  const int L = LV > 0 ? LV : a.d.levels;
  const int F = L * HEXC;
  float4* taps = tapbuf + (size_t)slot * tap_stride(L);
  for (int p0 = xcd_group(blockIdx.x, gridDim.x) * PPW; p0 < a.P; p0 += gridDim.x * PPW) {  // uniform trip count: shuffles below need all lanes
    const int pi = p0 + slot;
    const bool live = pi < a.P;
    const int p = live ? (a.proc_order ? (int)a.proc_order[pi] : pi) : 0;
    const size_t gbase = (size_t)pi * (size_t)(L * HEXC);   // point-major layout: the rows of this PROCESSING position
    float u[4];
    point_coords(a, p, u);
    wave_lds_sync();
    produce_taps(a, u, j, taps);
    wave_lds_sync();
    const float* grow = a.gfeat + (size_t)p * F + c0;
    float du[3] = {0.f, 0.f, 0.f};
    if constexpr (S3G_POINT_PREFETCH != 0 && LV == 4) {
      // T-slab: with the 24 G rows gone the kernel has registers to spare (188 of 256): the NEXT level's texels are requested
      // before this level's arithmetic, in two alternating register sets (fully unrolled: no set crosses a back-edge)
      LevelIn<V> X0, X1;
      issue_level<UT>(a, taps, 0, c0, grow, X0);
#pragma unroll
      for (int l = 0; l < 4; l += 2) {
        issue_level<UT>(a, taps, l + 1, c0, grow, X1);
        __builtin_amdgcn_sched_barrier(0);
        {
          LevelS<V> S;
          samples_level<UT>(a, taps, l, c0, X0, S);
          finish_level<UT>(a, p, l, c0, S, X0.g, live, G, gbase, du);
        }
        if (l + 2 < 4) issue_level<UT>(a, taps, l + 2, c0, grow, X0);
        __builtin_amdgcn_sched_barrier(0);
        {
          LevelS<V> S;
          samples_level<UT>(a, taps, l + 1, c0, X1, S);
          finish_level<UT>(a, p, l + 1, c0, S, X1.g, live, G, gbase, du);
        }
      }
    } else {
#pragma unroll LV > 0 ? LV : 1
    for (int l = 0; l < L; l++) {
      LevelIn<V> X;
      LevelS<V> S;
      issue_level<UT>(a, taps, l, c0, grow, X);
      samples_level<UT>(a, taps, l, c0, X, S);
      finish_level<UT>(a, p, l, c0, S, X.g, live, G, gbase, du);
    }
    }
    // sum over the 32 channels (the lanes of this point), then undo the aabb normalisation
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = du[k];
      for (int off = LPP / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      du[k] = v;
    }
    if (live && j < 3) a.gxyz[3 * (size_t)p + j] = (j == 0 ? du[0] : (j == 1 ? du[1] : du[2])) * (2.0f / (a.d.aabb_min[j] - a.d.aabb_max[j]));
  }
}

// ---- pass A, DIVISION form (round 4, `algorithm` S3G_HEX_SLAB_DIV: needs the forward's output `feat`) ----
// The product-rule kernel above keeps six samples and their twelve derivative vectors live per level (252 VGPRs with the next
// level's texels in flight: two waves per SIMD, parked on s_waitcnt 63 % of the time) only to form dL/ds_i = g * prod_{j != i} s_j.
// With the forward's own output f = prod_j s_j at hand the level's row is T = g * f in ONE multiply, and dL/ds_i = T / s_i needs
// nothing but plane i's own sample: the planes are processed one after the other like the forward does (texels, sample, two
// derivative vectors, two dot products -- then everything but three scalars is dead), at the forward's register count and
// occupancy, so that the 72 texel-line gathers per point hide behind other waves instead of behind nothing.  Same division and
// same safety predicate as the scatter walk (tv * rcp(s), |s| in (1e-18, inf)); a sample that fails it gets its EXACT
// g * prod_{j != i} s_j -- for the plane gradients through tslab_exact_scatter, for dL/dxyz through exact_du below -- re-derived
// from the coordinates on a path that costs the hot loop no registers.
__device__ __forceinline__ void exact_du(const HexArgs& a, int p, int l, int c0, f4v g, uint32_t badbits, float* du) {
  float u[4];
  point_coords(a, p, u);
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    if (!((badbits >> i) & 1u)) continue;
    const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
    const Tap t = make_tap(u[PAIR0[i]], u[PAIR1[i]], W, H);
    const float* pl = a.d.planes[l][i];
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
      const int c = c0 + k;
      if (!tslab_near_unsafe(walk_sample(a, l, i, u, c))) continue;     // that channel went through the division (same wide predicate)
      float gk = vget<f4v>(g, k);
#pragma unroll 1
      for (int jj = 0; jj < 6; jj++)
        if (jj != i) gk *= walk_sample(a, l, jj, u, c);
      const float v00 = fetch(pl, t.o00, c), v01 = fetch(pl, t.o01 >= 0 ? t.o01 : t.o00, c);
      const float v10 = fetch(pl, t.o10 >= 0 ? t.o10 : t.o00, c);
      const float v11 = fetch(pl, t.o11 >= 0 ? t.o11 : (t.o10 >= 0 ? t.o10 : (t.o01 >= 0 ? t.o01 : t.o00)), c);
      const float dX = (v01 - v00) * (t.y1f - t.iy) + (v11 - v10) * (t.iy - t.y0f);
      const float dY = (v10 - v00) * (t.x1f - t.ix) + (v11 - v01) * (t.ix - t.x0f);
      if (PAIR0[i] < 3) du[PAIR0[i]] += t.mx * dX * gk;
      if (PAIR1[i] < 3) du[PAIR1[i]] += t.my * dY * gk;
    }
  }
}

#ifndef S3G_HEX_POINTDIV_WAVES
#define S3G_HEX_POINTDIV_WAVES 4
#endif

template <bool UT>
__global__ void __launch_bounds__(256, S3G_HEX_POINTDIV_WAVES) hexplane_backward_pointdiv_kernel(const HexArgs a, const float* __restrict__ feat,
                                                                                                  float* __restrict__ G) {
  extern __shared__ float4 tapbuf[];   // [32 points][levels][TAP_SLOTS]
  const int j = threadIdx.x & 7, c0 = j * 4, slot = threadIdx.x >> 3;
  const int L = a.d.levels, F = L * HEXC;
  float4* taps = tapbuf + (size_t)slot * tap_stride(L);
  for (int p0 = xcd_group(blockIdx.x, gridDim.x) * 32; p0 < a.P; p0 += gridDim.x * 32) {  // uniform trip count: shuffles below need all lanes
    const int pi = p0 + slot;
    const bool live = pi < a.P;
    const int p = live ? (a.proc_order ? (int)a.proc_order[pi] : pi) : 0;
    const size_t gbase = (size_t)pi * (size_t)(L * HEXC);   // T rows of this PROCESSING position
    float u[4];
    point_coords(a, p, u);
    wave_lds_sync();
    produce_taps(a, u, j, taps);
    wave_lds_sync();
    const size_t row = (size_t)p * F + c0;
    float du[3] = {0.f, 0.f, 0.f};
    // (requesting the NEXT level's two rows a level ahead costs the eight registers that keep this kernel at four waves per SIMD:
    // 0.78 -> 1.12 ms with the spills, 0.88 ms at three waves -- measured, tools/variants/r04_pointdiv2.py)
    for (int l = 0; l < L; l++) {
      const f4v g = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.gfeat + row + l * HEXC));
      const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(feat + row + l * HEXC));
      const f4v T = g * f;
      if (live) __builtin_nontemporal_store(T, reinterpret_cast<f4v*>(G + gbase + (size_t)(l * HEXC + c0)));
      uint32_t badbits = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int W = a.d.res[l][PAIR0[i]], H = a.d.res[l][PAIR1[i]];
        const float* pl = a.d.planes[l][i];
        f4v sv, dX, dY;
        float mx, my;
        if (UT && IS_TIME_PLANE[i]) {
          const PointTap t = read_tap<true>(taps, l, i, W, H, c0);
          const f4v v00 = texelv<f4v>(pl, t.off), v01 = texelv<f4v>(pl, t.off + t.dx);
          sv = v00 * t.gx;
          sv = sv + v01 * t.fx;
          dX = v01 - v00;
          dY = vsplat<f4v>(0.f);
          mx = t.mx; my = 0.f;
        } else {
          const PointTap t = read_tap<false>(taps, l, i, W, H, c0);
          const f4v v00 = texelv<f4v>(pl, t.off), v01 = texelv<f4v>(pl, t.off + t.dx);
          const f4v v10 = texelv<f4v>(pl, t.off + t.dy), v11 = texelv<f4v>(pl, t.off + t.dy + t.dx);
          sv = v00 * (t.gx * t.gy);
          sv = sv + v01 * (t.fx * t.gy);
          sv = sv + v10 * (t.gx * t.fy);
          sv = sv + v11 * (t.fx * t.fy);
          dX = (v01 - v00) * t.gy + (v11 - v10) * t.fy;
          dY = (v10 - v00) * t.gx + (v11 - v01) * t.fx;
          mx = t.mx; my = t.my;
        }
        f4v gi;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float sk = vget<f4v>(sv, k);
          // ONE predicate per channel, the WIDE one: this pass divides only where |s| clears the threshold by a factor of four and
          // hands everything else to the exact fallback (exact_du applies the same wide predicate to the same bits; the plane
          // gradients' fallback, tslab_exact_scatter, applies the strict one the scatter walk uses).  A second, strict compare per
          // channel here cost the two registers that keep the kernel at four waves per SIMD: 0.76 -> 0.92 ms with the spills.
          const bool okk = !tslab_near_unsafe(sk);
          ok = ok && okk;
          const float q = okk ? vget<f4v>(T, k) * __builtin_amdgcn_rcpf(sk) : 0.f;
          if (k == 0) gi.x = q; else if (k == 1) gi.y = q; else if (k == 2) gi.z = q; else gi.w = q;
        }
        badbits |= ok ? 0u : (1u << i);
        if (PAIR0[i] < 3) du[PAIR0[i]] += mx * vdot(dX, gi);
        if (PAIR1[i] < 3) du[PAIR1[i]] += my * vdot(dY, gi);
      }
      if (badbits && live) {   // rare: a sample that is (nearly) zero or not finite
        tslab_exact_scatter<f4v>(a, p, l, c0, g, badbits);
        exact_du(a, p, l, c0, g, badbits, du);
      }
    }
    // sum over the 32 channels (the 8 lanes of this point), then undo the aabb normalisation
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = du[k];
      for (int off = 4; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      du[k] = v;
    }
    if (live && j < 3) a.gxyz[3 * (size_t)p + j] = (j == 0 ? du[0] : (j == 1 ? du[1] : du[2])) * (2.0f / (a.d.aabb_min[j] - a.d.aabb_max[j]));
  }
}

// ---- sort: point indices ordered by (major cell, minor cell) on a 512 x 512 grid, per orientation ----
// orientation o: major axis MAJ[o], minor axis MIN_[o]; handles planes PLA[o] (spatial) and PLT[o] (the major axis vs time)
constexpr int SORT_BINS = 512, SORT_NB = 256;
__device__ constexpr int MAJ[3] = {0, 1, 2};
__device__ constexpr int MIN_[3] = {1, 2, 0};
__device__ constexpr int PLA[3] = {0, 3, 1};  // (x,y) (y,z) (x,z)
__device__ constexpr int PLT[3] = {2, 4, 5};  // (x,t) (y,t) (z,t)

// Sort cell of a point along `axis` = its texel column at level `level`, computed exactly like make_tap does, so all points of
// one cell share their four corner texels at that level.
// Round 4: ONE ORDER PER (orientation, LEVEL).  Rounds 1-3 walked every level in the finest level's cell order: align_corners
// grids of different levels do not nest, a coarse footprint is then re-entered once per fine row that crosses it (eight times at
// level 0) and alternates at every cut -- which is what the two-entry footprint cache was for.  Counted on the bench's own point
// cloud with the cache modelled statement by statement (tools/sim/flush_orders.py): 5.9 M line-atomics per backward in the finest
// order against 2.4 M when every (orientation, level) is walked in ITS OWN cells' order (floor: 1.03 M distinct footprints x 2-4
// corners); measured before that: 7.9 M, i.e. ~0.8 ms of a 1.43 ms kernel at the 10 G line-ops/s the chip retires.
__device__ __forceinline__ int sort_cell(const HexArgs& a, int p, int axis, int level) {
  const float u = (a.xyz[3 * (size_t)p + axis] - a.d.aabb_max[axis]) * (2.0f / (a.d.aabb_min[axis] - a.d.aabb_max[axis])) - 1.0f;
  const int W = a.d.res[level][axis];
  float ix = ((u + 1.f) / 2.f) * (float)(W - 1);
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  int cidx = (int)floorf(ix);
  if (W > SORT_BINS) cidx = (int)(((long long)cidx * SORT_BINS) / W);
  return min(SORT_BINS - 1, max(0, cidx));
}

// The LAST order is the PROCESSING order of the per-point passes (forward, backward pass A): a two-level 3-D blocking -- major
// key = the 8 x 8 x 8 grid of blocks of the volume, minor key = the 8 x 8 x 8 sub-blocks of a block -- so that consecutive
// points are close in x, y AND z and all three spatial planes' texels stay in the L2 of the XCD that works on the block.
// (In an (x, y) order every tap of the (y, z) plane missed: 2.5 GB of 128-byte fetches per pass at 1.2 M points.)
// Order ids: oi = orientation * levels + level for the 3 * levels walk orders, oi = 3 * levels for the processing order.
static inline int n_walk_orders(int levels) { return 3 * levels; }
static inline int n_orders(int levels) { return n_walk_orders(levels) + 1; }
__device__ __forceinline__ int block_key(const HexArgs& a, int p, int shift) {
  int key = 0;
#pragma unroll
  for (int axis = 0; axis < 3; axis++) {
    const int Wc = min(a.d.res[a.d.levels - 1][axis], SORT_BINS);
    const int c = sort_cell(a, p, axis, a.d.levels - 1);
    key = key * 8 + (min(63, (c * 64) / Wc) >> shift & 7);
  }
  return key;
}
__device__ __forceinline__ int order_key(const HexArgs& a, int p, int oi, bool major) {
  const int nw = 3 * a.d.levels;
  if (oi >= nw) return block_key(a, p, major ? 3 : 0);
  const int o = oi / a.d.levels, level = oi % a.d.levels;
  return sort_cell(a, p, major ? MAJ[o] : MIN_[o], level);
}
__device__ __forceinline__ int major_key(const HexArgs& a, int p, int oi) { return order_key(a, p, oi, true); }
__device__ __forceinline__ int minor_key(const HexArgs& a, int p, int oi) { return order_key(a, p, oi, false); }

struct SortWork {
  uint32_t* table;      // [NO][SORT_NB][SORT_BINS]      NO = n_orders(levels), NW = n_walk_orders(levels) = NO - 1
  uint32_t* seg_start;  // [NO][SORT_BINS + 1]
  uint32_t* tmp;        // [NO][P]  indices grouped by major key
  uint32_t* order;      // [NW][P]  final orders of the walks: oi = orientation * levels + level
  uint32_t* comp;       // [NW][P]  comp[oi][k] = position of point order[oi][k] in the processing order (where its T rows are)
  uint32_t* proc;       // [P]      the last order: processing order of the per-point passes
  int nw;               // NW
};
__device__ __forceinline__ uint32_t* order_of(const SortWork& w, int o, int P) { return o < w.nw ? w.order + (size_t)o * P : w.proc; }

// STABLE placement (round 6, deterministic mode): the position of an element among the elements of its key must not depend on the
// order in which LDS atomics happen to execute.  One round = 256 consecutive elements.  Every wave ranks its lanes per key with one
// ballot per distinct key (registers only), the per-(wave, key) group sizes meet in LDS, and an element's position is
// base[key] + the groups of the earlier waves + its rank: the elements of a key keep their input order.  The bases advance by integer
// atomics (order-independent).  All four waves work in parallel: three barriers per round.  (The first version let the waves take
// turns, with the base read and written inside the ballot loop: 2.6 ms per re-sort against 0.8 ms for the unstable sort.)
// wcnt: [4][SORT_BINS] words of LDS, zero on entry, left zero.  Every thread of the workgroup calls this; inactive lanes pass active = false.
__device__ __forceinline__ uint32_t stable_claim(uint32_t* __restrict__ cell, uint32_t* __restrict__ wcnt, int key, bool active) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rank = 0, total = 0;
  bool leader = false;
  uint64_t remaining = __ballot(active);
  while (remaining) {   // uniform across the wave
    const int first = __ffsll((long long)remaining) - 1;
    const int k = __shfl(key, first);
    const bool mine = active && key == k;
    const uint64_t same = __ballot(mine);
    if (mine) {
      rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      total = (uint32_t)__popcll(same);
      leader = lane == first;
    }
    remaining &= ~same;
  }
  if (leader) wcnt[wave * SORT_BINS + key] = total;
  __syncthreads();
  uint32_t pos = 0;
  if (active) {
    pos = cell[key] + rank;
    for (int w = 0; w < wave; w++) pos += wcnt[w * SORT_BINS + key];
  }
  __syncthreads();
  if (leader) {
    atomicAdd(&cell[key], total);          // integer: the result does not depend on the order
    wcnt[wave * SORT_BINS + key] = 0u;
  }
  __syncthreads();
  return pos;
}

template <bool WRITE>
__global__ void __launch_bounds__(256) hexsort_major_kernel(const HexArgs a, const SortWork w, int chunk, int stable) {
  __shared__ uint32_t cell[SORT_BINS];
  __shared__ uint32_t wcnt[4 * SORT_BINS];
  if (WRITE && stable)
    for (int i = threadIdx.x; i < 4 * SORT_BINS; i += 256) wcnt[i] = 0u;
  const int o = blockIdx.y;
  uint32_t* row = w.table + ((size_t)o * SORT_NB + blockIdx.x) * SORT_BINS;
  for (int i = threadIdx.x; i < SORT_BINS; i += 256) cell[i] = WRITE ? w.seg_start[o * (SORT_BINS + 1) + i] + row[i] : 0u;
  __syncthreads();
  const int g0 = blockIdx.x * chunk, g1 = min(a.P, g0 + chunk);
  if (WRITE && stable) {
    for (int gb = g0; gb < g1; gb += 256) {      // uniform trip count: stable_claim synchronises the workgroup
      const int g = gb + threadIdx.x;
      const bool act = g < g1;
      const uint32_t pos = stable_claim(cell, wcnt, act ? major_key(a, g, o) : 0, act);
      if (act) w.tmp[(size_t)o * a.P + pos] = (uint32_t)g;
    }
    return;
  }
  for (int g = g0 + threadIdx.x; g < g1; g += 256) {
    const uint32_t pos = atomicAdd(&cell[major_key(a, g, o)], 1u);
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
__global__ void __launch_bounds__(256) hexsort_minor_kernel(const HexArgs a, const SortWork w, int stable) {
  __shared__ uint32_t cnt[SORT_BINS];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t wcnt[4 * SORT_BINS];
  const int o = blockIdx.y, bin = blockIdx.x, tid = threadIdx.x;
  const uint32_t s0 = w.seg_start[o * (SORT_BINS + 1) + bin], s1 = w.seg_start[o * (SORT_BINS + 1) + bin + 1];
  if (s1 == s0) return;
  const uint32_t* tmp = w.tmp + (size_t)o * a.P;
  uint32_t* order = order_of(w, o, a.P);
  for (int i = tid; i < SORT_BINS; i += 256) cnt[i] = 0u;
  __syncthreads();
  for (uint32_t k = s0 + tid; k < s1; k += 256) atomicAdd(&cnt[minor_key(a, (int)tmp[k], o)], 1u);
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
  if (stable) {
    for (int i = tid; i < 4 * SORT_BINS; i += 256) wcnt[i] = 0u;
    __syncthreads();
    for (uint32_t kb = s0; kb < s1; kb += 256) {   // uniform trip count (stable_claim synchronises); tmp is index-ascending per major bin
      const uint32_t k = kb + tid;
      const bool act = k < s1;
      const uint32_t g = act ? tmp[k] : 0u;
      const uint32_t pos = stable_claim(cnt, wcnt, act ? minor_key(a, (int)g, o) : 0, act);
      if (act) order[pos] = g;
    }
    return;
  }
  for (uint32_t k = s0 + tid; k < s1; k += 256) {
    const uint32_t g = tmp[k];
    order[atomicAdd(&cnt[minor_key(a, (int)g, o)], 1u)] = g;
  }
}

__global__ void __launch_bounds__(256) hexsort_rank_kernel(int P, const uint32_t* __restrict__ order, uint32_t* __restrict__ rank) {
  const int k = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (k < P) rank[(size_t)o * P + order[(size_t)o * P + k]] = (uint32_t)k;
}
// point-major G: where in the PROCESSING order is the k-th point of orientation o's order?  comp[o][k] = procrank[order[o][k]]
__global__ void __launch_bounds__(256) hexsort_compose_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ procrank,
                                                              uint32_t* __restrict__ comp) {
  const int k = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (k < P) comp[(size_t)o * P + k] = procrank[order[(size_t)o * P + k]];
}

// ---- pass B: scatter in sorted order with register run-length combining ----
// sorted points walked by one half-wave (HexArgs::seg_len): longer segments flush fewer footprints at their ends, shorter ones
// give more half-waves; 256 measured best at 1.2 M points (1.21 vs 1.30 ms), 128 below a million
static inline int segment_length(int P) { return P >= 1000000 ? 256 : 128; }

// One bilinear footprint being accumulated in registers by a walker lane (= one channel): key = texel offset of its nw corner
// (-1 = empty), flags bit0 = ne/se column in range, bit1 = sw/se row in range (the other three corners follow from key, flags and
// the plane width).
struct Foot {
  int key, flags;
  float a00, a01, a10, a11;
};
__device__ __forceinline__ void vatomic(char* base, uint32_t k, float v) { atomicAdd(reinterpret_cast<float*>(base + k), v); }
// Offsets are 32-bit BYTE offsets off a uniform base pointer (`base + zext(u32)` selects the scalar-base + VGPR-offset
// addressing mode: no 64-bit address arithmetic per atomic; a plane is at most 2^24 texels).  The corner tests stay
// branches on purpose: an unconditional atomic of an exact zero to a clamped address was measured 6x SLOWER for the whole
// pass -- every empty entry and every out-of-range corner then lands on the same few lines (texel 0 of each plane, the nw
// texel again), and same-address atomics serialise at ~10 ns each.  c = channel of the lane.
__device__ __forceinline__ void foot_flush(const Foot& f, float* __restrict__ gp, int W, int c) {
  if (f.key < 0) return;
  const uint32_t k = ((uint32_t)f.key * HEXC + (uint32_t)c) * 4u;
  const uint32_t dy = (uint32_t)W * (HEXC * 4u);
  char* base = reinterpret_cast<char*>(gp);
  vatomic(base, k, f.a00);
  if (f.flags & 1) vatomic(base, k + HEXC * 4u, f.a01);
  if (f.flags & 2) vatomic(base, k + dy, f.a10);
  if ((f.flags & 3) == 3) vatomic(base, k + dy + HEXC * 4u, f.a11);
}
struct PackedTap {  // what the scatter needs of a Tap: 8 floats in LDS
  int key, flags;
  float w00, w01, w10, w11;
};
// The walker's ONE remembered footprint (round 4; rounds 1-3 walked every level in the finest level's order and needed a two-entry
// cache with an MRU bit because foreign cell boundaries made the points alternate between two footprints -- removed in round 5,
// tools/sim/flush_orders.py still prices both).  A walk that is monotone in its OWN level's cells enters a footprint once.
//   hit   (three calls in four): no load at all -- the entry keeps the texel VALUES of its corners next to the partial sums; the
//         sample the footprint produced in the forward is re-derived from them and dL/ds = T / s is accumulated;
//   miss  (uniform inside the walker's lanes): the four (two) texels of the new footprint are loaded, and
//         evict  the old entry is flushed (up to 4 atomics) and restarts empty, or
//         shift  the new footprint is one row BELOW / one column RIGHT of the old one -- the usual step of a walk along the minor
//                axis: two of its texels are already being summed, only the row / column left behind is flushed (2 atomics
//                instead of 4 -- the walk is bound by the rate of atomic line-ops) and the other two sums move up.
// ROW = true: the plane is a height-1 row table (uniform time): only the nw / ne corners exist.
// An out-of-range corner has weight exactly 0 and takes the nw texel's value, like the per-point pass; a sample that is not safely
// divisible contributes nothing here (the per-point pass scattered it exactly: same predicate, same bits).
constexpr bool FOOT_SHIFT = true;
// Round 6: the corners live in PAIRS (nw, ne) / (sw, se) so that the hit path -- which three calls in four take, and on which the
// kernel is VALU-issue-bound (SQ_INSTS_VALU: 0.83 of its time in round 5) -- runs on packed fp32 instructions: the four products of
// the sample as two v_pk_mul_f32, the four accumulations as two v_pk_fma_f32.  Same values bit for bit: the products are rounded
// one by one and added in the order ((p00 + p01) + p10) + p11 exactly as before (the forward's order), an fma per corner as before.
struct Foot1 {
  int key;            // (texel index << 2 | corner flags), -1 = empty
  f2v_ a01, a23;      // partial sums of the footprint's corners (nw, ne) (sw, se)
  f2v_ v01, v23;      // texel values of the corners
};
__device__ __forceinline__ void foot1_init(Foot1& F) {
  F.key = -1;
  F.a01 = f2v_{0.f, 0.f};
  F.a23 = f2v_{0.f, 0.f};
}
struct PackedTap2 {   // one tap as the walker reads it back from LDS: 8 floats (32 bytes)
  int kf;             // texel index of the nw corner << 2 | flags (bit 0: ne / se column in range, bit 1: sw / se row in range)
  f2v_ w01, w23;      // bilinear weights of (nw, ne) (sw, se)
};
// ---- deterministic mode (round 6, opt-in: s3g_hexplane_set_deterministic) ---------------------------------------------------------
// Plane gradients that are bit-identical from run to run need (1) walk orders that do not depend on the timing of LDS atomics (the
// stable counting sorts above), (2) ONE writer per sum, and (3) a fixed order in which the sums of a texel are added.  The walk keeps
// its structure -- segments of seg_len sorted points per walker, one remembered footprint -- but a finished footprint ("run": the
// consecutive points of one cell inside one segment) is STORED, not added with atomics:
//   CELL[cell][corner][32]   the run that contains the cell's first point (cells are contiguous in the order: exactly one such run),
//   SEG[segment][corner][32] the first run of a segment when it continues a cell begun in an earlier segment (at most one per segment),
// and hexplane_stencil_kernel adds, for every texel, the four cells around it (nw of its own cell, ne of the cell to the left, sw of
// the cell above, se of the cell above-left), each as CELL + its SEG continuations in segment order.  The walkers also leave the index
// the stencil needs: cstart[cell] (written by whoever meets the cell's first point) and segcell[segment] (which cell a segment's first
// run continues: the walker compares its first point's cell with the cell of the point just before its segment).  No shift reuse here:
// every cell keeps its own four sums.  (A first version derived the cell extents in a kernel of its own -- two sort_cell evaluations
// per sorted position and walk through the order's indirection: 0.62 ms; the walk knows them for free.)
struct DetWalk {           // one (orientation, level) walk
  uint32_t* cstart;        // [cells] sorted position of the spatial cell's first point (cell id = plane texel index of its nw corner); ~0u: empty
  uint32_t* tstart;        // [Wmajor] the same for the 1-D cells of the (major, t) row table
  int* segcell;            // [segments] the cell the segment's FIRST run continues from an earlier segment, -1 if it opens its cell itself
  int* tsegcell;
  float* cell;             // [cells][4][32]
  float* seg;              // [segments][4][32]
  float* tcell;            // [Wmajor][2][32]
  float* tseg;             // [segments][2][32]
};
struct DetWork {
  DetWalk walk[3 * S3G_HEX_MAX_LEVELS];
};
template <bool ROW>
__device__ __forceinline__ void det_store_run(const Foot1& F, const DetWalk& dw, int seg, bool continuation, int c) {
  if (F.key < 0) return;
  const int cellid = F.key >> 2;
  constexpr int NC = ROW ? 2 : 4;
  float* rec = continuation ? (ROW ? dw.tseg : dw.seg) + (size_t)seg * (NC * HEXC) : (ROW ? dw.tcell : dw.cell) + (size_t)cellid * (NC * HEXC);
  rec[c] = F.a01.x;
  rec[HEXC + c] = F.a01.y;
  if (!ROW) {
    rec[2 * HEXC + c] = F.a23.x;
    rec[3 * HEXC + c] = F.a23.y;
  }
}
// the deterministic walker's tap: like foot1_add_t, but a finished footprint is stored as a run record and nothing is shifted.
// cont: the run being accumulated is the segment's first AND continues the cell of the point before the segment (prev_kf);
// kpos: sorted position of this point.
template <bool ROW>
__device__ __forceinline__ void foot1_add_det(Foot1& F, bool& cont, int prev_kf, int kpos, const PackedTap2& t, float tv, const DetWalk& dw, int seg,
                                              const float* __restrict__ pl, int W, int c) {
  const int tkf = t.kf;
  if (tkf != F.key) {
    const int tkey = tkf >> 2, tfl = tkf & 3;
    const float* px = pl + (size_t)tkey * HEXC;
    const float n0 = px[0], n1 = px[(tfl & 1) ? HEXC : 0];
    float n2 = 0.f, n3 = 0.f;
    if (!ROW) {
      n2 = px[(tfl & 2) ? (size_t)W * HEXC : 0];
      n3 = px[(tfl == 3) ? (size_t)W * HEXC + HEXC : 0];
    }
    const bool opening = F.key < 0;                 // the segment's first footprint
    if (!opening) det_store_run<ROW>(F, dw, seg, cont, c);
    cont = opening && tkf == prev_kf;
    if (c == 0) {
      if (opening) (ROW ? dw.tsegcell : dw.segcell)[seg] = cont ? tkey : -1;
      if (!cont) (ROW ? dw.tstart : dw.cstart)[tkey] = (uint32_t)kpos;       // this point is the first of its cell
    }
    F.a01 = f2v_{0.f, 0.f};
    F.v01 = f2v_{n0, n1};
    if (!ROW) {
      F.a23 = f2v_{0.f, 0.f};
      F.v23 = f2v_{n2, n3};
    }
    F.key = tkf;
  }
  const f2v_ p01 = F.v01 * t.w01;
  float sv = p01.x + p01.y;
  if (!ROW) {
    const f2v_ p23 = F.v23 * t.w23;
    sv = sv + p23.x;
    sv = sv + p23.y;
  }
  const float g = tslab_divisible(sv) ? tv * __builtin_amdgcn_rcpf(sv) : 0.f;
  const f2v_ gg = f2v_{g, g};
  F.a01 = __builtin_elementwise_fma(gg, t.w01, F.a01);
  if (!ROW) F.a23 = __builtin_elementwise_fma(gg, t.w23, F.a23);
}

template <bool ROW = false>
__device__ __forceinline__ void foot1_add_t(Foot1& F, const PackedTap2& t, float tv, float* __restrict__ gp,
                                            const float* __restrict__ pl /* plane values + channel */, int W, int c) {
  const int tkf = t.kf;
  if (tkf != F.key) {  // miss (uniform inside the walker's lanes)
    const int tkey = tkf >> 2, tfl = tkf & 3;
    const float* px = pl + (size_t)tkey * HEXC;
    const float n0 = px[0], n1 = px[(tfl & 1) ? HEXC : 0];
    float n2 = 0.f, n3 = 0.f;
    if (!ROW) {
      n2 = px[(tfl & 2) ? (size_t)W * HEXC : 0];
      n3 = px[(tfl == 3) ? (size_t)W * HEXC + HEXC : 0];
    }
    const int KF = F.key, K = KF >> 2, FL = KF & 3;
    const bool down = FOOT_SHIFT && !ROW && KF >= 0 && tkey == K + W;
    const bool right = FOOT_SHIFT && KF >= 0 && tkey == K + 1 && (FL & 1);
    const bool shift = down || right;
    const float A0 = F.a01.x, A1 = F.a01.y, A2 = ROW ? 0.f : F.a23.x, A3 = ROW ? 0.f : F.a23.y;
    if (KF >= 0) {
      const uint32_t k = ((uint32_t)K * HEXC + (uint32_t)c) * 4u;
      const uint32_t dy = (uint32_t)W * (HEXC * 4u);
      char* base = reinterpret_cast<char*>(gp);
      vatomic(base, k, A0);                                                    // nw leaves in every case
      if ((FL & 1) && !right) vatomic(base, k + HEXC * 4u, A1);               // ne stays when shifting right
      if (!ROW && (FL & 2) && !down) vatomic(base, k + dy, A2);               // sw stays when shifting down
      if (!ROW && (FL & 3) == 3 && !shift) vatomic(base, k + dy + HEXC * 4u, A3);
    }
    // new contents: shift down (nw, ne, sw, se) <- (sw, se, 0, 0); shift right <- (ne, 0, se, 0); evict <- 0
    F.a01 = f2v_{down ? A2 : (right ? A1 : 0.f), down ? A3 : 0.f};
    F.v01 = f2v_{n0, n1};
    if (!ROW) {
      F.a23 = f2v_{right ? A3 : 0.f, 0.f};
      F.v23 = f2v_{n2, n3};
    }
    F.key = tkf;
  }
  const f2v_ p01 = F.v01 * t.w01;
  float sv = p01.x + p01.y;
  if (!ROW) {
    const f2v_ p23 = F.v23 * t.w23;
    sv = sv + p23.x;
    sv = sv + p23.y;
  }
  const float g = tslab_divisible(sv) ? tv * __builtin_amdgcn_rcpf(sv) : 0.f;
  const f2v_ gg = f2v_{g, g};
  F.a01 = __builtin_elementwise_fma(gg, t.w01, F.a01);
  if (!ROW) F.a23 = __builtin_elementwise_fma(gg, t.w23, F.a23);
}
template <bool ROW = false>
__device__ __forceinline__ void foot1_flush_all(const Foot1& F, float* __restrict__ gp, int W, int c) {
  if (F.key < 0) return;
  foot_flush(Foot{F.key >> 2, ROW ? (F.key & 1) : (F.key & 3), F.a01.x, F.a01.y, ROW ? 0.f : F.a23.x, ROW ? 0.f : F.a23.y}, gp, W, c);
}

// A WALKER = 32 lanes (one per channel: a half-wave) walks seg_len consecutive points of ONE (orientation, level) order:
// blockIdx.y = orientation * levels + level.  The taps are computed by the walker's lanes for a whole GROUP of points at once
// (lane = point q x tap j), parked in LDS, and every lane reads them back with broadcast loads while it accumulates its channel.
// Round 6: groups of SIXTEEN points (rounds 1-5: four).  make_tap + the coordinate / index loads are ~60 wave-instructions whoever
// needs them; with 8 of a walker's 32 lanes busy they cost 15 per point, a quarter of everything the kernel issued -- with all 32
// lanes busy they cost 4.  The T rows are still requested four points at a time, one batch ahead of their use (their addresses
// come out of the same LDS records: the point's position in the processing order rides in the tap's spare slot).
// (Removed in round 5, measured slower in rounds 2-4: two levels per walk, two channels per lane with v_pk_fma -- twice the flush
// atomics, 2.07 vs 1.14 ms --, the two-entry footprint cache, 512- and 1024-point segments: DESIGN.md section 10.)
#ifndef S3G_HEX_SCATTER_WAVES
#define S3G_HEX_SCATTER_WAVES 6
#endif
constexpr int SCATTER_WG_PER_CU = S3G_HEX_SCATTER_WAVES;   // waves per SIMD the register budget is set for
constexpr int TAPF = 8;   // floats per packed tap in LDS: key, flags, w00, w01 | w10, w11, position of the point's T rows, -
constexpr int GRP = 16;   // points per tap group
__device__ __forceinline__ float load_g(const float* p) { return G_NONTEMPORAL_LOAD ? __builtin_nontemporal_load(p) : *p; }
template <bool UT, bool DET = false>   // UT: uniform time -- the (axis, t) planes are height-1 row tables; DET: deterministic mode (needs UT)
__global__ void __launch_bounds__(256, UT ? SCATTER_WG_PER_CU : SCATTER_WG_PER_CU - 1) hexplane_scatter_kernel(const HexArgs a, const float* __restrict__ G,
                                                               const uint32_t* __restrict__ order_all, const uint32_t* __restrict__ comp_all,
                                                               const DetWork detw) {
  constexpr int LANES = HEXC, WALKERS = 256 / LANES;
  constexpr int NTAP = 2;             // taps per point and walk: the orientation's spatial plane and its (major, t) plane
  static_assert(LANES == GRP * NTAP, "tap phase: one lane per (point of the group, tap)");
  __shared__ __attribute__((aligned(16))) float tapbuf[WALKERS][2][GRP][NTAP][TAPF];  // [walker][double buffer][point][tap]: 16 KiB
  const int oi = blockIdx.y;
  if (!((a.walk_mask >> oi) & 1u)) return;
  const int o = oi / a.d.levels, lv = oi % a.d.levels;
  const int c = threadIdx.x & (LANES - 1), hw = threadIdx.x / LANES;   // channel of this lane, walker of this half-wave
  const int q = c / NTAP, j = c % NTAP;  // tap-phase role: point q of the group, tap j (0 spatial, 1 time plane)
  const int seg = blockIdx.x * WALKERS + hw;
  const int k0 = seg * a.seg_len, k1 = min(a.P, k0 + a.seg_len);
  if (k0 >= a.P) return;  // whole walkers drop out; the LDS traffic below is private to a walker (wave-ordered)
  const uint32_t* order = order_all + (size_t)oi * a.P;
  const uint32_t* comp = comp_all + (size_t)oi * a.P;
  const size_t GP = (size_t)(a.d.levels * HEXC);   // point-major T rows: floats per point
  const int i0 = PLA[o], i1 = PLT[o];
  const int ip = j ? i1 : i0;                         // the plane of this lane's tap
  const int axw = PAIR0[ip], axh = PAIR1[ip];
  Foot1 f1[2];
  foot1_init(f1[0]);
  foot1_init(f1[1]);
  bool cont[2] = {false, false};      // deterministic mode: the open run is the segment's first and continues an earlier segment's cell
  int prev_kf[2] = {-2, -2};          // deterministic mode: hit key (texel << 2 | flags) of the point just before the segment, per tap
  const int Wt = a.d.res[lv][axw], Ht = a.d.res[lv][axh];
  // uniform per workgroup; read ONCE (indexed kernel-argument reads inside the loop were an s_load + s_waitcnt lgkmcnt(0) per tap,
  // i.e. every tap also waited for all of the wave's outstanding LDS reads)
  float* const gp0 = a.gplanes[lv][i0];
  float* const gp1 = a.gplanes[lv][i1];
  const float* const pl0 = a.d.planes[lv][i0] + c;
  const float* const pl1 = a.d.planes[lv][i1] + c;
  const int W0 = a.d.res[lv][PAIR0[i0]], W1 = a.d.res[lv][PAIR0[i1]];
  const float* Grow = G + (size_t)(lv * HEXC + c);    // this lane's column of every T row
  // Software pipeline per lane role (point q of a group, tap j): the sorted index (and T-row position) of group g+2, the
  // coordinates of group g+1 and the taps of group g+1 are produced while group g is accumulated, so neither the
  // index -> position load chain nor the tap arithmetic sits between a group's T loads and their use.
  auto slot_of = [&](int kb) { return min(kb + q, k1 - 1); };
  auto store_taps = [&](const float* u, uint32_t cpos, int buf) {
    const Tap t = make_tap(u[axw], u[axh], Wt, Ht);
    float* dst = &tapbuf[hw][buf][q][j][0];
    // slot 0: (texel index << 2 | corner flags) -- the word the hit test compares; a plane has at most 2^24 texels (check_desc)
    *reinterpret_cast<float4*>(dst) = make_float4(__int_as_float((t.o00 << 2) | (t.o01 >= 0 ? 1 : 0) | (t.o10 >= 0 ? 2 : 0)), 0.f, t.w00, t.w01);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(t.w10, t.w11, __uint_as_float(cpos), 0.f);
  };
  auto row_pos = [&](int buf, int qq) { return __float_as_uint(tapbuf[hw][buf][qq][0][6]); };
  if (DET && k0 > 0) {               // lane j of the walker (q == 0) evaluates tap j of the previous point; broadcast inside the half-wave
    float up[4];
    point_coords(a, (int)order[k0 - 1], up);
    const Tap tp = make_tap(up[axw], up[axh], Wt, Ht);
    const int kfp = (tp.o00 << 2) | (tp.o01 >= 0 ? 1 : 0) | (tp.o10 >= 0 ? 2 : 0);
    prev_kf[0] = __shfl(kfp, (int)(threadIdx.x & 32u));
    prev_kf[1] = __shfl(kfp, (int)(threadIdx.x & 32u) + 1);
  }
  float un[4];                       // coordinates of the NEXT group's point
  uint32_t cn;                       // ... and the position of its T rows
  {
    float u0[4];
    const int s0 = slot_of(k0);
    point_coords(a, (int)order[s0], u0);
    store_taps(u0, comp[s0], 0);
  }
  {
    const int s1 = slot_of(k0 + GRP);
    point_coords(a, (int)order[s1], un);
    cn = comp[s1];
  }
  int snn = slot_of(k0 + 2 * GRP);
  int pnn = (int)order[snn];         // index of the group after next
  uint32_t cnn = comp[snn];
  wave_lds_sync();
  // the first batch of T rows (four points; ONE row per point and level: T = dL/dfeature * feature -- both planes of the walk
  // divide it by their sample)
  float g[4], gn[4];
#pragma unroll
  for (int qq = 0; qq < 4; qq++) g[qq] = load_g(Grow + (size_t)row_pos(0, qq) * GP);
  int buf = 0;
  for (int kb = k0; kb < k1; kb += GRP, buf ^= 1) {
    // 1. the NEXT group's taps from coordinates loaded one iteration ago; then advance the two prefetch stages
    store_taps(un, cn, buf ^ 1);
    point_coords(a, pnn, un);
    cn = cnn;
    snn = slot_of(kb + 3 * GRP);
    pnn = (int)order[snn];
    cnn = comp[snn];
    wave_lds_sync();
    // 2. accumulate this group, four points at a time; each batch first requests the T rows of the batch after it
#pragma unroll 1     // (rolled on purpose: every copy of the body carries eight inlined miss paths)
    for (int sb = 0; sb < GRP / 4; sb++) {
#pragma unroll
      for (int qq = 0; qq < 4; qq++)
        gn[qq] = load_g(Grow + (size_t)row_pos(sb == GRP / 4 - 1 ? buf ^ 1 : buf, (4 * (sb + 1) + qq) % GRP) * GP);
      const int nq = k1 - kb - 4 * sb;       // points left from this batch on (<= 0: nothing)
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        if (qq >= nq) break;
#pragma unroll
        for (int m = 0; m < 2; m++) {
          float* gp = m ? gp1 : gp0;
          if (gp == nullptr) continue;
          const float* src = &tapbuf[hw][buf][4 * sb + qq][m][0];
          const float4 lo = *reinterpret_cast<const float4*>(src);
          PackedTap2 t;
          t.kf = __float_as_int(lo.x);
          t.w01 = f2v_{lo.z, lo.w};
          if (DET) {
            const int kpos = kb + 4 * sb + qq;
            if (m == 1) {
              foot1_add_det<true>(f1[1], cont[1], prev_kf[1], kpos, t, g[qq], detw.walk[oi], seg, pl1, W1, c);
            } else {
              const float2 hi = *reinterpret_cast<const float2*>(src + 4);
              t.w23 = f2v_{hi.x, hi.y};
              foot1_add_det<false>(f1[0], cont[0], prev_kf[0], kpos, t, g[qq], detw.walk[oi], seg, pl0, W0, c);
            }
          } else if (UT && m == 1) {
            foot1_add_t<true>(f1[m], t, g[qq], gp, pl1, W1, c);
          } else {
            const float2 hi = *reinterpret_cast<const float2*>(src + 4);
            t.w23 = f2v_{hi.x, hi.y};
            foot1_add_t<false>(f1[m], t, g[qq], gp, m ? pl1 : pl0, m ? W1 : W0, c);
          }
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; qq++) g[qq] = gn[qq];
    }
  }
#pragma unroll
  for (int m = 0; m < 2; m++) {
    float* gp = m ? gp1 : gp0;
    if (gp == nullptr) continue;
    if (DET) {
      if (m == 1) det_store_run<true>(f1[1], detw.walk[oi], seg, cont[1], c);
      else det_store_run<false>(f1[0], detw.walk[oi], seg, cont[0], c);
    } else if (UT && m == 1) foot1_flush_all<true>(f1[m], gp, W1, c);
    else foot1_flush_all<false>(f1[m], gp, m ? W1 : W0, c);
  }
}

// ---- deterministic mode: the stencil gather ---------------------------------------------------------------------------------------
// sum of one cell's run records for corner `corner`, lane = channel: CELL first, then the SEG continuations in segment order
template <int NC>
__device__ __forceinline__ float det_cell_sum(const float* __restrict__ cellrec, const float* __restrict__ segrec, const uint32_t* __restrict__ cs,
                                              const int* __restrict__ segcell, int cellid, int corner, int seg_len, int nseg, int c) {
  const uint32_t s0 = cs[cellid];
  if (s0 == 0xffffffffu) return 0.f;
  float acc = cellrec[(size_t)cellid * (NC * HEXC) + corner * HEXC + c];
  for (int sg = (int)(s0 / (uint32_t)seg_len) + 1; sg < nseg && segcell[sg] == cellid; sg++)
    acc += segrec[(size_t)sg * (NC * HEXC) + corner * HEXC + c];
  return acc;
}
// grid = (texel groups, walks): a half-wave (lane = channel) per texel of the walk's spatial plane; the row tables' 1-D stencil rides in
// the same launch (texels 0 .. Wmajor-1 of an extra "row" behind the plane).
__global__ void __launch_bounds__(256) hexplane_stencil_kernel(const HexArgs a, const DetWork detw) {
  const int oi = blockIdx.y, o = oi / a.d.levels, lv = oi % a.d.levels;
  if (!((a.walk_mask >> oi) & 1u)) return;
  const int c = threadIdx.x & (HEXC - 1);
  const int ip = PLA[o], it = PLT[o];
  const int Wx = a.d.res[lv][PAIR0[ip]], Wy = a.d.res[lv][PAIR1[ip]], Wm = a.d.res[lv][MAJ[o]];
  const int t = blockIdx.x * (256 / HEXC) + threadIdx.x / HEXC;
  const int sl = a.seg_len, nseg = (a.P + sl - 1) / sl;
  const DetWalk dw = detw.walk[oi];
  if (t < Wx * Wy) {
    float* gp = a.gplanes[lv][ip];
    if (gp == nullptr) return;
    const int x = t % Wx, y = t / Wx;
    // the four cells around the texel, in the fixed order nw (own cell), ne (left), sw (above), se (above-left).  Three rounds of
    // independent loads -- starts, then records + the next segment's link, then (rarely) continuation records -- instead of four
    // dependent chains one after the other: the pass is latency-bound (0.65 -> see profiles/r06_hex_deterministic.txt)
    const int cid[4] = {t, t - 1, t - Wx, t - Wx - 1};
    const bool ok[4] = {true, x > 0, y > 0, x > 0 && y > 0};
    uint32_t s0[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s0[k] = ok[k] ? dw.cstart[cid[k]] : 0xffffffffu;
    float v[4];
    int nxt[4], link[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool have = s0[k] != 0xffffffffu;
      v[k] = have ? dw.cell[(size_t)cid[k] * (4 * HEXC) + k * HEXC + c] : 0.f;
      nxt[k] = have ? (int)(s0[k] / (uint32_t)sl) + 1 : nseg;
      link[k] = nxt[k] < nseg ? dw.segcell[nxt[k]] : -1;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float part = v[k];
      int sg = nxt[k], lk = link[k];
      while (sg < nseg && lk == cid[k]) {          // the cell straddles segments: its continuation records, in segment order
        part += dw.seg[(size_t)sg * (4 * HEXC) + k * HEXC + c];
        sg++;
        lk = sg < nseg ? dw.segcell[sg] : -1;
      }
      acc += part;
    }
    gp[(size_t)t * HEXC + c] += acc;
  } else if (t < Wx * Wy + Wm) {
    float* gt = a.gplanes[lv][it];       // (uniform time: the row table's gradient, folded back into the plane rows afterwards)
    if (gt == nullptr) return;
    const int x = t - Wx * Wy;
    float acc = det_cell_sum<2>(dw.tcell, dw.tseg, dw.tstart, dw.tsegcell, x, 0, sl, nseg, c);
    if (x > 0) acc += det_cell_sum<2>(dw.tcell, dw.tseg, dw.tstart, dw.tsegcell, x - 1, 1, sl, nseg, c);
    gt[(size_t)x * HEXC + c] += acc;
  }
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
  const size_t lds = (size_t)32 * tap_stride(d->levels) * sizeof(float4);
  if (d->uniform_time) hipLaunchKernelGGL(hexplane_forward_kernel<true>, dim3(blocks), dim3(256), lds, (hipStream_t)stream_, a);
  else hipLaunchKernelGGL(hexplane_forward_kernel<false>, dim3(blocks), dim3(256), lds, (hipStream_t)stream_, a);
  profile_end(S3G_PROFILE_HEXPLANE_FORWARD, (hipStream_t)stream_, (double)P, (double)d->levels);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

static std::atomic<int> g_hex_deterministic{0};
static void carve_det(Carver& c, const s3g_hexplane_desc* d, int P, DetWork* dw, void** index_begin, size_t* index_bytes) {
  // deterministic mode: per-walk cell starts (0xff-filled every backward: they come first, contiguous), segment links and run records
  const int nseg = (P + segment_length(P) - 1) / segment_length(P);
  DetWork w;
  memset(&w, 0, sizeof w);
  const size_t b0 = (c.off + 127) & ~size_t(127);
  size_t b1 = b0;
  for (int pass = 0; pass < 2; pass++)
    for (int o = 0; o < 3; o++)
      for (int l = 0; l < d->levels; l++) {
        const int oi = o * d->levels + l;
        static const int PLA_H[3] = {0, 3, 1}, MAJ_H[3] = {0, 1, 2};
        const size_t cells = (size_t)d->res[l][PAIR0_HOST[PLA_H[o]]] * d->res[l][PAIR1_HOST[PLA_H[o]]], wm = (size_t)d->res[l][MAJ_H[o]];
        if (pass == 0) {
          w.walk[oi].cstart = c.take<uint32_t>(cells);
          w.walk[oi].tstart = c.take<uint32_t>(wm);
          b1 = c.off;
        } else {
          w.walk[oi].segcell = c.take<int>((size_t)nseg); w.walk[oi].tsegcell = c.take<int>((size_t)nseg);
          w.walk[oi].cell = c.take<float>(cells * 4 * HEXC); w.walk[oi].seg = c.take<float>((size_t)nseg * 4 * HEXC);
          w.walk[oi].tcell = c.take<float>(wm * 2 * HEXC); w.walk[oi].tseg = c.take<float>((size_t)nseg * 2 * HEXC);
        }
      }
  if (dw) *dw = w;
  if (index_begin) *index_begin = c.base ? c.base + b0 : nullptr;
  if (index_bytes) *index_bytes = b1 - b0;
}
static void carve_backward(Carver& c, const s3g_hexplane_desc* d, int P, float** G, float** tables, SortWork* w) {
  const size_t n = (size_t)P;
  float* g = c.take<float>((size_t)d->levels * n * HEXC);   // the T rows: one 128-byte row per point and level
  float* tb = d->uniform_time ? c.take<float>(2 * time_table_floats(d)) : nullptr;
  SortWork s;
  const size_t NO = (size_t)n_orders(d->levels), NW = (size_t)n_walk_orders(d->levels);
  s.nw = (int)NW;
  s.table = c.take<uint32_t>(NO * SORT_NB * SORT_BINS);
  s.seg_start = c.take<uint32_t>(NO * (SORT_BINS + 1));
  s.tmp = c.take<uint32_t>(NO * n);
  s.order = c.take<uint32_t>(NW * n);
  s.comp = c.take<uint32_t>(NW * n);
  s.proc = c.take<uint32_t>(n);
  if (G) *G = g;
  if (tables) *tables = tb;
  if (w) *w = s;
}

// 128-byte rows of scratch the default (slab) backward writes per point and level set: bench.py prices the implementation bytes
extern "C" int s3g_hexplane_backward_scratch_rows(int levels) { return levels; }

// Diagnostics only (tools/hex_probe.py walks): which of the 3 * levels scatter walks run -- bit orientation * levels + level.  With
// anything but all ones the plane gradients are INCOMPLETE; the setting is process-wide and meant for timing the walks one by one.
static std::atomic<uint32_t> g_walk_mask{0xffffffffu};
extern "C" void s3g_hexplane_debug_walk_mask(uint32_t mask) { g_walk_mask.store(mask, std::memory_order_relaxed); }

// Deterministic mode of the backward (process-wide; include/s3g_hexplane.h): stable walk orders, run records instead of atomics, a
// stencil gather in fixed order -- plane gradients bit-identical from run to run.  Needs uniform_time and resolutions <= 512; the
// workspace grows (s3g_hexplane_backward_workspace_bytes follows the setting).
extern "C" void s3g_hexplane_set_deterministic(int on) { g_hex_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }
extern "C" int s3g_hexplane_get_deterministic(void) { return g_hex_deterministic.load(std::memory_order_relaxed); }

// 32-bit words per point of the caller-kept `sort_state`: the walk orders, their compositions with the processing order, and the
// processing order itself (round 4: one walk order per orientation AND level, 6 * levels + 1; rounds 1-3: 7)
extern "C" int s3g_hexplane_sort_state_words(int levels) { return 2 * n_walk_orders(levels) + 1; }

extern "C" size_t s3g_hexplane_backward_workspace_bytes(const s3g_hexplane_desc* d, int P, int have_features) {
  if (!d || d->levels < 1 || d->levels > S3G_HEX_MAX_LEVELS || P < 0) return 0;
  Carver c(nullptr);
  (void)have_features;   // (until round 5 the slab-free "walk" algorithm had a different, smaller layout)
  carve_backward(c, d, P, nullptr, nullptr, nullptr);
  if (g_hex_deterministic.load(std::memory_order_relaxed)) carve_det(c, d, P, nullptr, nullptr, nullptr);
  return c.bytes();
}

static int hexplane_backward_impl(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                  const float* dL_dfeatures, const float* features, int algorithm, float* dL_dxyz,
                                  float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* workspace,
                                  uint32_t* sort_state, int sort_reuse, void* stream_);

extern "C" int s3g_hexplane_backward(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                     const float* dL_dfeatures, const float* features, float* dL_dxyz,
                                     float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* workspace,
                                     uint32_t* sort_state, int sort_reuse, void* stream_) {
  return hexplane_backward_impl(d, P, xyz, time, dL_dfeatures, features, features ? S3G_HEX_SLAB_DIV : S3G_HEX_SLAB, dL_dxyz,
                                dL_dplanes, workspace, sort_state, sort_reuse, stream_);
}

extern "C" int s3g_hexplane_backward_algo(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                          const float* dL_dfeatures, const float* features, int algorithm, float* dL_dxyz,
                                          float* const dL_dplanes[S3G_HEX_MAX_LEVELS][6], void* workspace,
                                          uint32_t* sort_state, int sort_reuse, void* stream_) {
  if (algorithm == S3G_HEX_WALK) {
    set_error("s3g_hexplane_backward_algo: the slab-free walk (algorithm %d) was removed in ABI 12; use S3G_HEX_SLAB_DIV or S3G_HEX_SLAB", algorithm);
    return S3G_ERR_INVALID_ARG;
  }
  if (algorithm != S3G_HEX_SLAB && algorithm != S3G_HEX_SLAB_DIV) {
    set_error("s3g_hexplane_backward_algo: unknown algorithm %d", algorithm);
    return S3G_ERR_INVALID_ARG;
  }
  if (algorithm != S3G_HEX_SLAB && !features && P > 0) {
    set_error("s3g_hexplane_backward_algo: this algorithm needs the forward's output `features`");
    return S3G_ERR_INVALID_ARG;
  }
  return hexplane_backward_impl(d, P, xyz, time, dL_dfeatures, features, algorithm, dL_dxyz, dL_dplanes, workspace, sort_state,
                                sort_reuse, stream_);
}

static int hexplane_backward_impl(const s3g_hexplane_desc* d, int P, const float* xyz, const float* time,
                                  const float* dL_dfeatures, const float* features, int algorithm, float* dL_dxyz,
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
  float *G, *tables;
  SortWork w;
  carve_backward(c, d, P, &G, &tables, &w);
  const int NO = n_orders(d->levels), NW = n_walk_orders(d->levels);
  const int det = g_hex_deterministic.load(std::memory_order_relaxed);
  DetWork detw;
  memset(&detw, 0, sizeof detw);
  void* det_index = nullptr;
  size_t det_index_bytes = 0;
  if (det) {
    if (!d->uniform_time) {
      set_error("s3g_hexplane_backward: the deterministic mode needs desc.uniform_time (the (axis, t) planes as row tables)");
      return S3G_ERR_INVALID_ARG;
    }
    for (int l = 0; l < d->levels; l++)
      for (int k = 0; k < 3; k++)
        if (d->res[l][k] > SORT_BINS) {
          set_error("s3g_hexplane_backward: the deterministic mode needs spatial resolutions <= %d (sort cells = texel cells)", SORT_BINS);
          return S3G_ERR_INVALID_ARG;
        }
    carve_det(c, d, P, &detw, &det_index, &det_index_bytes);
  }
  if (sort_state) {  // caller-owned, persistent: s3g_hexplane_sort_state_words(levels) * P words
    w.order = sort_state;
    w.comp = sort_state + (size_t)NW * P;
    w.proc = sort_state + (size_t)2 * NW * P;
  }

  // 1. three spatial orders (2-level LDS counting sorts); the legacy path also needs their inverse permutations.
  //    Deterministic mode: ALWAYS -- its run records rely on every cell being contiguous in the walk order, which only holds for orders
  //    sorted on the CURRENT coordinates (the default walk sums with atomics and is indifferent to a stale order).
  if (!sort_reuse || det) {
    const int chunk = (((P + SORT_NB - 1) / SORT_NB + 255) / 256) * 256;
    hipLaunchKernelGGL(hexsort_major_kernel<false>, dim3(SORT_NB, NO), dim3(256), 0, stream, a, w, chunk, det);
    hipLaunchKernelGGL(hexsort_scan_kernel, dim3(NO), dim3(512), 0, stream, w, P);
    hipLaunchKernelGGL(hexsort_major_kernel<true>, dim3(SORT_NB, NO), dim3(256), 0, stream, a, w, chunk, det);
    hipLaunchKernelGGL(hexsort_minor_kernel, dim3(SORT_BINS, NO), dim3(256), 0, stream, a, w, det);
    // comp[oi][k]; the inverse of the processing order goes through w.tmp (free after the sorts)
    hipLaunchKernelGGL(hexsort_rank_kernel, dim3((P + 255) / 256, 1), dim3(256), 0, stream, P, w.proc, w.tmp);
    hipLaunchKernelGGL(hexsort_compose_kernel, dim3((P + 255) / 256, NW), dim3(256), 0, stream, P, w.order, w.tmp, w.comp);
    S3G_HIP_CHECK(hipGetLastError());
  }
  //    (the sorts above used the real resolutions; from here on the time planes are height-1 row tables if uniform_time)
  TimeRows rows;
  if (d->uniform_time) {
    const size_t nt = time_table_floats(d);
    S3G_HIP_CHECK(hipMemsetAsync(tables + nt, 0, nt * sizeof(float), stream));
    use_time_rows(a, rows, tables, tables + nt, stream);
  }
  a.seg_len = segment_length(P);
  a.walk_mask = g_walk_mask.load(std::memory_order_relaxed);
  const int nseg = (P + a.seg_len - 1) / a.seg_len;
  {
    // 2. per-point pass (dL/dxyz; ONE row T = dL/dfeature * feature per point and level -> G), then the scatter walks reading it back
    a.proc_order = w.proc;
    profile_begin(S3G_PROFILE_HEXPLANE_BACKWARD_POINT, stream);
    constexpr int ppw = 256 / (HEXC / vec_of<PointV>::N);   // points per workgroup
    const int pblocks = (P + ppw - 1) / ppw;
    const size_t lds = (size_t)ppw * tap_stride(d->levels) * sizeof(float4);
    if (algorithm == S3G_HEX_SLAB_DIV) {
      const int dblocks = (P + 31) / 32;
      const size_t dlds = (size_t)32 * tap_stride(d->levels) * sizeof(float4);
      if (d->uniform_time) hipLaunchKernelGGL(hexplane_backward_pointdiv_kernel<true>, dim3(dblocks), dim3(256), dlds, stream, a, features, G);
      else hipLaunchKernelGGL(hexplane_backward_pointdiv_kernel<false>, dim3(dblocks), dim3(256), dlds, stream, a, features, G);
    } else if (d->uniform_time && d->levels == 4) hipLaunchKernelGGL((hexplane_backward_point_kernel<true, PointV, 4>), dim3(pblocks), dim3(256), lds, stream, a, G);
    else if (d->uniform_time) hipLaunchKernelGGL((hexplane_backward_point_kernel<true, PointV, 0>), dim3(pblocks), dim3(256), lds, stream, a, G);
    else hipLaunchKernelGGL((hexplane_backward_point_kernel<false, PointV, 0>), dim3(pblocks), dim3(256), lds, stream, a, G);
    profile_end(S3G_PROFILE_HEXPLANE_BACKWARD_POINT, stream, (double)P, (double)d->levels);
    S3G_HIP_CHECK(hipGetLastError());
    profile_begin(S3G_PROFILE_HEXPLANE_SCATTER, stream);
    constexpr int walkers = 256 / HEXC;
    if (det) {
      // run records + cell index by the walk (no atomics), then one stencil gather per texel
      S3G_HIP_CHECK(hipMemsetAsync(det_index, 0xff, det_index_bytes, stream));     // cstart / tstart: ~0u = empty cell
      hipLaunchKernelGGL((hexplane_scatter_kernel<true, true>), dim3((nseg + walkers - 1) / walkers, NW), dim3(256), 0, stream, a, G, w.order, w.comp, detw);
      int maxt = 0;
      for (int l = 0; l < d->levels; l++)
        for (int o = 0; o < 3; o++) {
          static const int PLA_H[3] = {0, 3, 1}, MAJ_H[3] = {0, 1, 2};
          maxt = max(maxt, d->res[l][PAIR0_HOST[PLA_H[o]]] * d->res[l][PAIR1_HOST[PLA_H[o]]] + d->res[l][MAJ_H[o]]);
        }
      hipLaunchKernelGGL(hexplane_stencil_kernel, dim3((maxt + walkers - 1) / walkers, NW), dim3(256), 0, stream, a, detw);
    } else if (d->uniform_time)
      hipLaunchKernelGGL((hexplane_scatter_kernel<true>), dim3((nseg + walkers - 1) / walkers, NW), dim3(256), 0, stream, a, G, w.order, w.comp, detw);
    else
      hipLaunchKernelGGL((hexplane_scatter_kernel<false>), dim3((nseg + walkers - 1) / walkers, NW), dim3(256), 0, stream, a, G, w.order, w.comp, detw);
    profile_end(S3G_PROFILE_HEXPLANE_SCATTER, stream, (double)P, (double)d->levels);
  }
  if (d->uniform_time) {
    int maxW = 0;
    for (int l = 0; l < d->levels; l++)
      for (int k = 0; k < 3; k++) maxW = max(maxW, d->res[l][k]);
    hipLaunchKernelGGL(hexplane_time_rows_kernel<true>, dim3((maxW * HEXC + 255) / 256, 3, d->levels), dim3(256), 0, stream, rows);
  }
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
