// MI355X-native replacement of simple-knn (distCUDA2): mean squared distance to the 3 nearest neighbours.
//
// Same algorithm family as the reference (KNN/simple_knn.cu): order the points along a Morton curve, cut the order
// into boxes of 1024 consecutive points, keep an AABB per box, and for every point scan only the boxes whose AABB is
// not farther than its current 3rd-best distance.  The pruning is conservative, so the result is the exact 3-NN
// (the multiset of the three smallest distances does not depend on the order in which candidates are visited).
//
// CDNA4 re-design:
//   * ordering: the result does not depend on the order *inside* a Morton cell, so instead of a 4-pass radix sort of
//     30-bit codes the points are split into 32768 cells (top 15 Morton bits) with the same atomic-free LDS
//     multisplit as the rasterizer's binning: per-workgroup histograms of ALL cells live in LDS (128 KiB of the
//     160 KiB a CU has), one [workgroup][cell] table, LDS cursors.  No global atomics, no global sort.
//   * search: one workgroup = one box of 1024 points.  Box AABBs are wave-uniform (scalar loads); when any point of
//     the workgroup still needs a candidate box, its 1024 points are staged into LDS with one coalesced read and
//     scanned with broadcast LDS reads, instead of every thread chasing indices through global memory.
//   * min/max, offsets and the cell scan stay on the device: no host synchronisation (the reference has two).
#include "common.hpp"

#include <float.h>

#include "../../include/s3g_knn.h"

namespace s3g {

constexpr int KNN_BOX = 1024;         // reference BOX_SIZE, simple_knn.cu:12
constexpr int KNN_CELL_BITS = 15;
constexpr int KNN_CELLS = 1 << KNN_CELL_BITS;
constexpr int KNN_SPLIT_BLOCKS = 256; // one per CU: each holds a 128 KiB cell histogram in LDS

struct KnnWork {
  float* mm;             // [6] min xyz, max xyz
  float* partial;        // [KNN_SPLIT_BLOCKS][6]
  uint32_t* table;       // [KNN_SPLIT_BLOCKS][KNN_CELLS]
  uint32_t* cell_start;  // [KNN_CELLS]
  float4* sorted;        // [P] (x, y, z, bit-cast original index)
  float* boxes;          // [nboxes][6]
  static KnnWork carve(void* p, size_t P, size_t* bytes) {
    Carver c(p);
    KnnWork w;
    w.mm = c.take<float>(8);
    w.partial = c.take<float>(KNN_SPLIT_BLOCKS * 6);
    w.table = c.take<uint32_t>((size_t)KNN_SPLIT_BLOCKS * KNN_CELLS);
    w.cell_start = c.take<uint32_t>(KNN_CELLS);
    w.sorted = c.take<float4>(P);
    w.boxes = c.take<float>(((P + KNN_BOX - 1) / KNN_BOX) * 6);
    if (bytes) *bytes = c.bytes();
    return w;
  }
};

__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// Stage 1: per-workgroup min/max (n elements of 6-float records or raw points), stage 2: final.
__global__ void __launch_bounds__(256) knn_minmax_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial) {
  __shared__ float red[4][6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float v = pts[3 * (size_t)i + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    mn[k] = wave_min(mn[k]);
    mx[k] = wave_max(mx[k]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int k = 0; k < 3; k++) {
      red[wave][k] = mn[k];
      red[wave][3 + k] = mx[k];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}
__global__ void __launch_bounds__(64) knn_minmax_final_kernel(int nb, const float* __restrict__ partial, float* __restrict__ mm) {
  const int k = threadIdx.x;
  if (k >= 6) return;
  float v = partial[k];
  for (int b = 1; b < nb; b++) v = k < 3 ? fminf(v, partial[b * 6 + k]) : fmaxf(v, partial[b * 6 + k]);
  mm[k] = v;
}

// simple_knn.cu:45-61 (10 bits per axis); only the top 15 bits are used as the cell id.
__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__device__ __forceinline__ uint32_t morton_cell(float3 p, const float* __restrict__ mm) {
  uint32_t q[3];
  const float c[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float ext = mm[3 + k] - mm[k];
    float t = ext > 0.f ? (c[k] - mm[k]) / ext : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);
    q[k] = prep_morton((uint32_t)(t * 1023.0f));
  }
  return (q[0] | (q[1] << 1) | (q[2] << 2)) >> (30 - KNN_CELL_BITS);
}

// Atomic-free multisplit of the points into Morton cells (same scheme as the rasterizer's bin_kernel).
template <bool WRITE>
__global__ void __launch_bounds__(1024) knn_split_kernel(int P, int chunk, const float* __restrict__ pts,
                                                         const float* __restrict__ mm, uint32_t* __restrict__ table,
                                                         const uint32_t* __restrict__ cell_start, float4* __restrict__ sorted) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cell[];  // [KNN_CELLS]
  uint32_t* row = table + (size_t)blockIdx.x * KNN_CELLS;
  for (int i = threadIdx.x; i < KNN_CELLS; i += 1024) cell[i] = WRITE ? cell_start[i] + row[i] : 0u;
  __syncthreads();
  const int g0 = blockIdx.x * chunk, g1 = min(P, g0 + chunk);
  for (int g = g0 + threadIdx.x; g < g1; g += 1024) {
    const float3 p = make_float3(pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2]);
    const uint32_t pos = atomicAdd(&cell[morton_cell(p, mm)], 1u);
    if (WRITE) sorted[pos] = make_float4(p.x, p.y, p.z, __uint_as_float((uint32_t)g));
  }
  if (!WRITE) {
    __syncthreads();
    for (int i = threadIdx.x; i < KNN_CELLS; i += 1024) row[i] = cell[i];
  }
}

// Per cell: exclusive prefix over the split workgroups (in place) -> cell totals; then one workgroup scans the cells.
__global__ void __launch_bounds__(256) knn_cell_prefix_kernel(int nb, uint32_t* __restrict__ table, uint32_t* __restrict__ cell_total) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  uint32_t run = 0;
  for (int b = 0; b < nb; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (b + k < nb) ? table[(size_t)(b + k) * KNN_CELLS + c] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (b + k < nb) table[(size_t)(b + k) * KNN_CELLS + c] = run;
      run += v[k];
    }
  }
  cell_total[c] = run;
}
__global__ void __launch_bounds__(1024) knn_cell_scan_kernel(uint32_t* __restrict__ cell_start) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < KNN_CELLS; base += 1024) {
    const uint32_t v = cell_start[base + tid];
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int k = 0; k < 16; k++) {
      if (k < wave) wbase += wsum[k];
      tot += wsum[k];
    }
    const uint32_t carry = carry_s;
    cell_start[base + tid] = carry + wbase + incl - v;
    __syncthreads();
    if (tid == 0) carry_s = carry + tot;
    __syncthreads();
  }
}

// simple_knn.cu:78-117: AABB of each box of 1024 consecutive (cell-ordered) points.
__global__ void __launch_bounds__(1024) knn_box_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ boxes) {
  __shared__ float red[16][6];
  const int i = blockIdx.x * KNN_BOX + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
    const float4 p = sorted[i];
    mn[0] = mx[0] = p.x; mn[1] = mx[1] = p.y; mn[2] = mx[2] = p.z;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    mn[k] = wave_min(mn[k]);
    mx[k] = wave_max(mx[k]);
  }
  if (lane == 0)
    for (int k = 0; k < 3; k++) {
      red[wave][k] = mn[k];
      red[wave][3 + k] = mx[k];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < 16; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    boxes[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

// simple_knn.cu:129-145
__device__ __forceinline__ void update_3best(const float3 ref, const float4 pt, float* knn) {
  const float dx = pt.x - ref.x, dy = pt.y - ref.y, dz = pt.z - ref.z;
  float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (knn[j] > dist) {
      const float t = knn[j];
      knn[j] = dist;
      dist = t;
    }
}
// simple_knn.cu:119-128
__device__ __forceinline__ float dist_box_point(const float* __restrict__ b, const float3 p) {
  float dx = 0.f, dy = 0.f, dz = 0.f;
  if (p.x < b[0] || p.x > b[3]) dx = fminf(fabsf(p.x - b[0]), fabsf(p.x - b[3]));
  if (p.y < b[1] || p.y > b[4]) dy = fminf(fabsf(p.y - b[1]), fabsf(p.y - b[4]));
  if (p.z < b[2] || p.z > b[5]) dz = fminf(fabsf(p.z - b[2]), fabsf(p.z - b[5]));
  return dx * dx + dy * dy + dz * dz;
}

// simple_knn.cu:147-183, one workgroup per box; candidate boxes are staged through LDS for the whole workgroup.
__global__ void __launch_bounds__(1024) knn_search_kernel(int P, const float4* __restrict__ sorted,
                                                          const float* __restrict__ boxes, float* __restrict__ dists) {
  __shared__ float4 cand[KNN_BOX];
  const int idx = blockIdx.x * KNN_BOX + threadIdx.x;
  const bool live = idx < P;
  const float4 me4 = live ? sorted[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float3 me = make_float3(me4.x, me4.y, me4.z);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  if (live)
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
      if (i == idx) continue;
      update_3best(me, sorted[i], best);
    }
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  for (int b = 0; b < nboxes; b++) {
    const float d = dist_box_point(boxes + 6 * (size_t)b, me);
    const bool want = live && !(d > reject || d > best[2]);
    if (__syncthreads_or(want)) {  // (also the barrier protecting cand[] from the previous round)
      const int j = b * KNN_BOX + threadIdx.x;
      if (j < P) cand[threadIdx.x] = sorted[j];
      __syncthreads();
      if (want) {
        const int n = min(KNN_BOX, P - b * KNN_BOX);
        const int self = idx - b * KNN_BOX;
        for (int k = 0; k < n; k++) {
          if (k == self) continue;
          update_3best(me, cand[k], best);
        }
      }
    }
  }
  if (live) dists[__float_as_uint(me4.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace s3g

using namespace s3g;

extern "C" size_t s3g_knn_workspace_bytes(int P) {
  size_t bytes = 0;
  KnnWork::carve(nullptr, (size_t)(P > 0 ? P : 0), &bytes);
  return bytes;
}

extern "C" int s3g_knn_mean_dist2(int P, const float* points, float* meanDists, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || (P > 0 && (!points || !meanDists || !workspace))) {
    set_error("s3g_knn_mean_dist2: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  KnnWork w = KnnWork::carve(workspace, P, nullptr);
  const int nb = KNN_SPLIT_BLOCKS;
  const int chunk = (((P + nb - 1) / nb + 1023) / 1024) * 1024;
  static std::atomic<uint64_t> attr_set{0};
  if (device_needs_setup(attr_set)) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)knn_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KNN_CELLS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)knn_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KNN_CELLS * 4));
    device_setup_done(attr_set);
  }
  hipLaunchKernelGGL(knn_minmax_kernel, dim3(nb), dim3(256), 0, stream, P, points, w.partial);
  hipLaunchKernelGGL(knn_minmax_final_kernel, dim3(1), dim3(64), 0, stream, nb, w.partial, w.mm);
  hipLaunchKernelGGL(knn_split_kernel<false>, dim3(nb), dim3(1024), KNN_CELLS * 4, stream, P, chunk, points, w.mm, w.table,
                     w.cell_start, w.sorted);
  hipLaunchKernelGGL(knn_cell_prefix_kernel, dim3(KNN_CELLS / 256), dim3(256), 0, stream, nb, w.table, w.cell_start);
  hipLaunchKernelGGL(knn_cell_scan_kernel, dim3(1), dim3(1024), 0, stream, w.cell_start);
  hipLaunchKernelGGL(knn_split_kernel<true>, dim3(nb), dim3(1024), KNN_CELLS * 4, stream, P, chunk, points, w.mm, w.table,
                     w.cell_start, w.sorted);
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(knn_box_kernel, dim3(nboxes), dim3(1024), 0, stream, P, w.sorted, w.boxes);
  hipLaunchKernelGGL(knn_search_kernel, dim3(nboxes), dim3(1024), 0, stream, P, w.sorted, w.boxes, meanDists);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}
