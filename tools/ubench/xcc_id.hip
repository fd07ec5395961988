// Which XCD does workgroup b land on?  Reads HW_REG_XCC_ID (gfx940+) in every workgroup of a 1-D launch and prints the
// mapping statistics: the XCD-aware swizzles in csrc/ assume XCD == blockIdx.x % 8.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/xcc_id.hip -o /tmp/xcc_id && /tmp/xcc_id
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

int main() {
  for (int n : {8, 64, 256, 4096, 6704}) {
    unsigned* d;
    hipMalloc(&d, n * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int match = 0;
    int hist[16] = {0};
    for (int b = 0; b < n; b++) {
      const unsigned x = h[b] & 0xf;  // XCC_ID field: bits [3:0]
      hist[x & 15]++;
      if ((int)x == b % 8) match++;
    }
    printf("blocks %5d: xcc == b %% 8 for %5d (%.1f%%); first 16 ids:", n, match, 100.0 * match / n);
    for (int b = 0; b < 16 && b < n; b++) printf(" %u", h[b] & 0xf);
    printf(" | per-xcc counts:");
    for (int x = 0; x < 8; x++) printf(" %d", hist[x]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
