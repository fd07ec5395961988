/*
 * oracle/knn_oracle.c -- CPU restatement of simple_knn (distCUDA2).
 *
 * TEST INFRASTRUCTURE ONLY (see raster_oracle.c header).  The reference has no tests of its own for this path; this
 * restatement is PINNED BY THE REFERENCE ITSELF since round 3: oracle/_ref/libref_knn.so is the reference's simple_knn.cu built
 * for gfx950 (oracle/build_ref.sh, host-pointer shim oracle/ref_knn_shim.cpp), compared on the GPU box up to the 1.2 M points of
 * cfg3 (tests/test_knn_gpu.py::test_knn_oracle_and_distcuda2_are_pinned_by_the_reference_build); and on the CPU against a
 * brute-force O(P^2) 3-NN, which the algorithm equals exactly because the box pruning is conservative (simple_knn.cu:168-181).
 *
 * Follows KNN = /root/reference/submodules/simple-knn/simple_knn.cu:
 *   :45-61  prepMorton / coord2Morton (10 bits per axis)
 *   :78-117 boxMinMax (AABB of BOX_SIZE=1024 consecutive Morton-sorted points)
 *   :119-145 distBoxPoint, updateKBest<3>
 *   :147-183 boxMeanDist (seed from +-3 Morton neighbours, then scan all boxes not farther than the
 *            seed's 3rd-best / current 3rd-best; output mean of the 3 smallest squared distances,
 *            self excluded by index, written at the point's original index)
 *   :185-221 SimpleKNN::knn (min/max reduce with init {0,0,0} -- so the AABB always contains the origin)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BOX_SIZE 1024

static uint32_t prepMorton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
static uint32_t coord2Morton(const float* c, const float* mn, const float* mx) {
  uint32_t x = prepMorton((uint32_t)(((c[0] - mn[0]) / (mx[0] - mn[0])) * ((1 << 10) - 1)));
  uint32_t y = prepMorton((uint32_t)(((c[1] - mn[1]) / (mx[1] - mn[1])) * ((1 << 10) - 1)));
  uint32_t z = prepMorton((uint32_t)(((c[2] - mn[2]) / (mx[2] - mn[2])) * ((1 << 10) - 1)));
  return x | (y << 1) | (z << 2);
}
typedef struct { float mn[3], mx[3]; } MinMax;

static float distBoxPoint(const MinMax* b, const float* p) {
  float d[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++)
    if (p[k] < b->mn[k] || p[k] > b->mx[k]) d[k] = fminf(fabsf(p[k] - b->mn[k]), fabsf(p[k] - b->mx[k]));
  return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}
static void updateKBest3(const float* ref, const float* pt, float* knn) {
  float d[3] = {pt[0] - ref[0], pt[1] - ref[1], pt[2] - ref[2]};
  float dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  for (int j = 0; j < 3; j++)
    if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
}

/* Stable LSD radix sort of (morton, index) on all 32 bits == cub::DeviceRadixSort::SortPairs (:213). */
static void sort_pairs_u32(uint32_t* keys, uint32_t* vals, size_t n) {
  uint32_t* k2 = (uint32_t*)malloc(n * 4); uint32_t* v2 = (uint32_t*)malloc(n * 4);
  uint32_t *ks = keys, *kd = k2, *vs = vals, *vd = v2;
  for (int shift = 0; shift < 32; shift += 8) {
    size_t cnt[257]; memset(cnt, 0, sizeof cnt);
    for (size_t i = 0; i < n; i++) cnt[((ks[i] >> shift) & 255) + 1]++;
    for (int i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
    for (size_t i = 0; i < n; i++) { size_t d = cnt[(ks[i] >> shift) & 255]++; kd[d] = ks[i]; vd[d] = vs[i]; }
    uint32_t* t = ks; ks = kd; kd = t; t = vs; vs = vd; vd = t;
  }
  /* 4 passes: result is back in keys/vals */
  free(k2); free(v2);
}

void orc_knn_mean_dist2(int P, const float* points, float* meanDists) {
  if (P <= 0) return;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; /* init = {0,0,0}, simple_knn.cu:191 */
  for (int i = 0; i < P; i++)
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], points[3 * i + k]); mx[k] = fmaxf(mx[k], points[3 * i + k]); }
  uint32_t* morton = (uint32_t*)malloc((size_t)P * 4);
  uint32_t* indices = (uint32_t*)malloc((size_t)P * 4);
  for (int i = 0; i < P; i++) { morton[i] = coord2Morton(points + 3 * i, mn, mx); indices[i] = (uint32_t)i; }
  sort_pairs_u32(morton, indices, (size_t)P);
  int num_boxes = (P + BOX_SIZE - 1) / BOX_SIZE;
  MinMax* boxes = (MinMax*)malloc((size_t)num_boxes * sizeof(MinMax));
  for (int b = 0; b < num_boxes; b++) {
    MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    int end = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
    for (int i = b * BOX_SIZE; i < end; i++) {
      const float* p = points + 3 * indices[i];
      for (int k = 0; k < 3; k++) { me.mn[k] = fminf(me.mn[k], p[k]); me.mx[k] = fmaxf(me.mx[k], p[k]); }
    }
    boxes[b] = me;
  }
#pragma omp parallel for schedule(dynamic, 256)
  for (int idx = 0; idx < P; idx++) {
    const float* point = points + 3 * indices[idx];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    int lo = idx - 3 > 0 ? idx - 3 : 0, hi = idx + 3 < P - 1 ? idx + 3 : P - 1;
    for (int i = lo; i <= hi; i++) {
      if (i == idx) continue;
      updateKBest3(point, points + 3 * indices[i], best);
    }
    float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int b = 0; b < num_boxes; b++) {
      float dist = distBoxPoint(&boxes[b], point);
      if (dist > reject || dist > best[2]) continue;
      int end = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
      for (int i = b * BOX_SIZE; i < end; i++) {
        if (i == idx) continue;
        updateKBest3(point, points + 3 * indices[i], best);
      }
    }
    meanDists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
  }
  free(morton); free(indices); free(boxes);
}
