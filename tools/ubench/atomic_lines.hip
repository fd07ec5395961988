// Microbenchmark: throughput of line-coalesced fp32 atomicAdd (32 lanes -> one 128 B texel) at random texels,
// the access pattern of a HexPlane backward scatter.  Decides whether the backward may use atomics at all.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void scatter(float* plane, const uint32_t* texel, int n, int reps) {
  const int lane = threadIdx.x & 31, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, ngrp = (gridDim.x * blockDim.x) >> 5;
  for (int i = grp; i < n; i += ngrp)
    for (int r = 0; r < reps; r++) atomicAdd(&plane[(size_t)texel[(i + r * 7919) % n] * 32 + lane], 1.0f);
}
__global__ void scatter_plain(float* plane, const uint32_t* texel, int n, int reps) {
  const int lane = threadIdx.x & 31, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, ngrp = (gridDim.x * blockDim.x) >> 5;
  for (int i = grp; i < n; i += ngrp)
    for (int r = 0; r < reps; r++) plane[(size_t)texel[(i + r * 7919) % n] * 32 + lane] += 1.0f;
}
int main() {
  const int n = 1200000, reps = 48;
  for (int texels : {4096, 16384, 65536, 262144}) {
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % texels); }
    uint32_t* d; float* p;
    hipMalloc(&d, n * 4); hipMalloc(&p, (size_t)texels * 128); hipMemset(p, 0, (size_t)texels * 128);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 2; variant++) {
      for (int it = 0; it < 2; it++) {
        hipEventRecord(a);
        if (variant == 0) scatter<<<2048, 256>>>(p, d, n, reps); else scatter_plain<<<2048, 256>>>(p, d, n, reps);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("texels=%7d %s: %.3f ms for %.1f M line-ops (%.2f G lines/s, %.2f G elem/s)\n", texels, variant ? "plain+=" : "atomic ", ms,
             n * (double)reps / 1e6, n * (double)reps / ms / 1e6, n * (double)reps * 32 / ms / 1e6);
    }
    hipFree(d); hipFree(p);
  }
  return 0;
}
