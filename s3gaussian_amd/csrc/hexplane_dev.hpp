// Device-side building blocks of the HexPlane sampler shared by hexplane.hip (forward / backward kernels) and mlp.hip (the fused
// inference kernel HexPlane (+) deformation MLP): tap construction and sharing through LDS, channel-last texel loads, the
// uniform-time row tables.  Private to libs3g.so (not part of the C ABI).
#pragma once
#include "common.hpp"

#include "../../include/s3g_hexplane.h"

namespace s3g {

constexpr int HEXC = S3G_HEX_CHANNELS;
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr bool G_NONTEMPORAL = true;         // streaming stores of the gradient slab: point pass 1.83 -> 1.60 ms
constexpr bool FEAT_NONTEMPORAL = true;      // forward's feature rows
constexpr bool GFEAT_NONTEMPORAL = true;     // point pass: dL/dfeature rows (read once)
constexpr bool G_NONTEMPORAL_LOAD = true;    // and streaming loads in the scatter: 1.28 -> 1.23 ms

// Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8) and every XCD has its own L2.  With the points in
// spatial order, giving XCD k the k-th CONTIGUOUS eighth of the groups keeps each texel line in one L2 instead of eight.
constexpr bool XCD_CONTIGUOUS = true;
__device__ __forceinline__ int xcd_group(int b, int nb) {
  if (!XCD_CONTIGUOUS) return b;
  const int per = nb >> 3;
  return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

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
  int seg_len;                 // scatter walks: sorted points per half-wave
  uint32_t walk_mask;          // scatter walks: bit oi = walk (orientation * levels + level) runs (diagnostics: time the walks one by one)
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
  u[3] = a.d.uniform_time ? a.time[0] : a.time[p];   // uniform time: `time` may hold a single element (s3g_hexplane.h)
}

// coordinate pairs in itertools.combinations(range(4), 2) order
__device__ constexpr int PAIR0[6] = {0, 0, 0, 1, 1, 2};
__device__ constexpr int PAIR1[6] = {1, 2, 3, 2, 3, 3};

__device__ __forceinline__ float4 operator*(float4 a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---- per-point passes (forward, backward pass A) ----
// EIGHT lanes own one point, four channels each (one 16-byte load per texel and lane; the 8 lanes read its 128-byte line).
// The r1 kernels spent most of their instructions on work every lane of a point repeated; this version issues half of them
// (forward 512 -> 268 VALU instructions per level, pass A 907 -> 585).  That did NOT make them faster by itself -- the forward is
// bound by the rate the L1 takes 128-byte lines plus its feature stores, pass A by memory latency at two waves per SIMD (SQ
// counters: DESIGN.md 6) -- but it is what the SIMDs no longer burn:
//   * the bilinear tap of a (level, plane) is computed ONCE per point -- lane j < 6 of the point's eight computes plane j of
//     every level -- and shared through LDS as 16 bytes (packed nw key + flags, ix - x0, iy - y0); the r1 kernels repeated
//     make_tap in all eight lanes (6 x ~45 instructions per level and lane);
//   * texel loads are branch-free: an out-of-range corner (only possible on the last column / row, where its weight is
//     exactly 0) reads the nw texel instead of selecting zeros behind an exec-mask branch (24 branches + 96 v_mov per level);
//   * texel addresses are a uniform base pointer + a 32-bit byte offset (planes are at most 2^24 texels: check_desc);
//   * uniform time: the (axis, t) row tables have two corners, not four (compile-time: template UT).
// x1 - ix is recomputed as 1 - (ix - x0): ix - x0 is exact (Sterbenz; x0 = 0 trivially), so both expressions are the correct
// rounding of the same real number -- bit-identical weights.
struct PointTap {  // as read back from LDS
  uint32_t off;    // byte offset of the nw texel's channel 0
  uint32_t dx, dy; // byte distance to the ne / sw texel, 0 when that corner is out of range
  float fx, fy, gx, gy;   // ix - x0, iy - y0, x1 - ix, y1 - iy
  float mx, my;    // d(ix)/d(u), d(iy)/d(u) incl. the border-clip mask
};
constexpr int TAP_SLOTS = 8;   // 16-byte tap slots per (point, level): 6 used
// float4 slots per point: one slot of padding, so that the points of a wave (who read the same tap index at once, each point's
// lanes the same address) start 4 banks apart instead of on the same bank (SQ_LDS_BANK_CONFLICT was 6 cycles per LDS instruction)
__host__ __device__ constexpr int tap_stride(int levels) { return levels * TAP_SLOTS + 1; }

// lane role j < 6: plane j of every level for the lane's point -> tapbuf[slot][l][j]
__device__ __forceinline__ void produce_taps(const HexArgs& a, const float* u, int j, float4* __restrict__ taps /* [levels][TAP_SLOTS] of this point */) {
  if (j >= 6) return;
  // plane j = axes (a0, a1) in itertools.combinations order: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3).  The lane's two resolutions
  // are picked out of the level's four by SHIFTS of two packed 64-bit values: an indexed res[] -- and a chain of selects, which
  // the compiler turns back into one -- goes through scratch memory, a serial memory round trip at the head of every group.
  const int a0 = j < 3 ? 0 : (j < 5 ? 1 : 2), a1 = j < 3 ? j + 1 : (j < 5 ? j - 1 : 3);
  const bool a0_is0 = j < 3, a0_is1 = j == 3 || j == 4;
  const bool a1_is1 = j == 0, a1_is2 = j == 1 || j == 3;
  // (the same for the coordinates: selects between loads of one private array are rewritten by the compiler into ONE load
  // with a selected index, i.e. the array is spilled to scratch and read back; the empty asm makes them plain registers)
  float u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
  asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
  const float ua = a0_is0 ? u0 : (a0_is1 ? u1 : u2);
  const float ub = a1_is1 ? u1 : (a1_is2 ? u2 : u3);
  for (int l = 0; l < a.d.levels; l++) {
    const uint64_t lo = (uint64_t)(uint32_t)a.d.res[l][0] | ((uint64_t)(uint32_t)a.d.res[l][1] << 32);
    const uint64_t hi = (uint64_t)(uint32_t)a.d.res[l][2] | ((uint64_t)(uint32_t)a.d.res[l][3] << 32);
    const int W = (int)(uint32_t)((a0 < 2 ? lo : hi) >> (32 * (a0 & 1)));
    const int H = (int)(uint32_t)((a1 < 2 ? lo : hi) >> (32 * (a1 & 1)));
    const Tap t = make_tap(ua, ub, W, H);
    const uint32_t flags = (t.o01 >= 0 ? 1u : 0u) | (t.o10 >= 0 ? 2u : 0u) | (t.mx != 0.f ? 4u : 0u) | (t.my != 0.f ? 8u : 0u);
    taps[l * TAP_SLOTS + j] = make_float4(__uint_as_float(((uint32_t)t.o00 << 4) | flags), t.ix - t.x0f, t.iy - t.y0f, 0.f);
  }
}
// one level only (the fused inference kernel keeps a single level's taps in LDS): lane role j < 6 -> taps[j]
__device__ __forceinline__ void produce_taps_level(const HexArgs& a, const float* u, int j, int l, float4* __restrict__ taps /* [TAP_SLOTS] of this point */) {
  if (j >= 6) return;
  const int a0 = j < 3 ? 0 : (j < 5 ? 1 : 2), a1 = j < 3 ? j + 1 : (j < 5 ? j - 1 : 3);
  const bool a0_is0 = j < 3, a0_is1 = j == 3 || j == 4;
  const bool a1_is1 = j == 0, a1_is2 = j == 1 || j == 3;
  float u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
  asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
  const float ua = a0_is0 ? u0 : (a0_is1 ? u1 : u2);
  const float ub = a1_is1 ? u1 : (a1_is2 ? u2 : u3);
  const uint64_t lo = (uint64_t)(uint32_t)a.d.res[l][0] | ((uint64_t)(uint32_t)a.d.res[l][1] << 32);
  const uint64_t hi = (uint64_t)(uint32_t)a.d.res[l][2] | ((uint64_t)(uint32_t)a.d.res[l][3] << 32);
  const int W = (int)(uint32_t)((a0 < 2 ? lo : hi) >> (32 * (a0 & 1)));
  const int H = (int)(uint32_t)((a1 < 2 ? lo : hi) >> (32 * (a1 & 1)));
  const Tap t = make_tap(ua, ub, W, H);
  const uint32_t flags = (t.o01 >= 0 ? 1u : 0u) | (t.o10 >= 0 ? 2u : 0u) | (t.mx != 0.f ? 4u : 0u) | (t.my != 0.f ? 8u : 0u);
  taps[j] = make_float4(__uint_as_float(((uint32_t)t.o00 << 4) | flags), t.ix - t.x0f, t.iy - t.y0f, 0.f);
}
template <bool ROW>
__device__ __forceinline__ PointTap read_tap(const float4* __restrict__ taps, int l, int i, int W, int H, int c4) {
  const float4 v = taps[l * TAP_SLOTS + i];
  const uint32_t pk = __float_as_uint(v.x);
  PointTap t;
  t.off = (pk >> 4) * (HEXC * 4u) + (uint32_t)c4 * 4u;
  t.dx = (pk & 1u) ? HEXC * 4u : 0u;
  t.dy = (!ROW && (pk & 2u)) ? (uint32_t)W * (HEXC * 4u) : 0u;
  t.fx = v.y; t.gx = 1.f - v.y;
  t.fy = ROW ? 0.f : v.z; t.gy = ROW ? 1.f : 1.f - v.z;
  t.mx = (pk & 4u) ? (float)(W - 1) / 2.f : 0.f;
  t.my = (pk & 8u) ? (float)(H - 1) / 2.f : 0.f;
  return t;
}
__device__ __forceinline__ float4 texel4(const float* __restrict__ plane, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(plane) + byte_off);
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ constexpr bool IS_TIME_PLANE[6] = {false, false, true, false, true, true};


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
static inline size_t time_table_floats(const s3g_hexplane_desc* d) {
  size_t n = 0;
  for (int l = 0; l < d->levels; l++)
    for (int k = 0; k < 3; k++) n += (size_t)d->res[l][k] * HEXC;
  return n;
}
// Fills `r`, points the time planes of `a.d` at the tables (height 1) and launches the table build.
static inline void use_time_rows(HexArgs& a, TimeRows& r, float* tables, float* gtables, hipStream_t stream) {
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

static constexpr int PAIR0_HOST[6] = {0, 0, 0, 1, 1, 2}, PAIR1_HOST[6] = {1, 2, 3, 2, 3, 3};
static inline int check_desc(const s3g_hexplane_desc* d) {
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
      if ((long long)d->res[l][PAIR0_HOST[i]] * d->res[l][PAIR1_HOST[i]] > (1ll << 24)) {
        set_error("hexplane: a plane has more than 2^24 texels (32-bit texel byte offsets)");
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
