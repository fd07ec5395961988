// What does rocprofv3's FETCH_SIZE report for the access shapes of the HexPlane kernels?  MI355X_MICROARCH.md calibrates the counter
// for ONE shape only -- wide coalesced streaming reads, 16 B per lane: FETCH_SIZE = half the bytes -- and says "other access widths ...
// are uncalibrated: calibrate on a known byte count in your own access pattern".  profiles/kernel_traffic.json doubles FETCH_SIZE for
// every kernel; the HexPlane forward / per-point backward gather 128-byte texels with 8 lanes x 16 B, the scatter walks gather
// 128-byte rows with 32 lanes x 4 B ("31 x algorithmic" for the scatter is an upper bound until this is known).
//
// Every variant reads each 128-byte line of a 4 GiB buffer (16 x the 256 MiB Infinity Cache) exactly once, so the HBM-side byte count
// is the buffer size whatever the order:
//   stream16     lane l of the grid reads 16 B at 16*l                       (the guide's calibrated shape)
//   gather16x8   8 lanes read one RANDOM line, 16 B each                     (hexplane_forward / pointdiv texel gathers)
//   gather4x32   32 lanes (a half-wave) read one RANDOM line, 4 B each       (hexplane_scatter T rows / footprint rows)
//   stream4      lane l reads 4 B at 4*l                                     (narrow but fully coalesced)
// Lines are visited in the order of a multiplicative permutation (i * ODD mod lines), so "random" costs no index array.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
//   cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fetch_calib
//   python tools/pmc_summary.py /tmp/fc        (FETCH_SIZE is in KB: factor = 4194304 KB / reported)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr uint64_t BYTES = 4ull << 30;
constexpr uint64_t LINES = BYTES / 128;          // 2^25
constexpr uint64_t ODD = 0x9E3779B1ull;          // odd: i -> i * ODD mod 2^25 is a permutation of the lines

__global__ void stream16(const float4* __restrict__ p, float* __restrict__ sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per lane, BYTES / 16 lanes
  const float4 v = p[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = 1.0f;
}

__global__ void stream4(const float* __restrict__ p, float* __restrict__ sink) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; k++) acc += p[i + (uint64_t)k * (BYTES / 16)];   // four coalesced dword sweeps, a quarter of the buffer each
  if (acc == 12345.678f) sink[0] = 1.0f;
}

__global__ void gather16x8(const float4* __restrict__ p, float* __restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t line = ((t >> 3) * ODD) & (LINES - 1);
  const float4 v = p[line * 8 + (t & 7)];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = 1.0f;
}

__global__ void gather4x32(const float* __restrict__ p, float* __restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // BYTES / 4 lanes
  const uint64_t line = ((t >> 5) * ODD) & (LINES - 1);
  const float v = p[line * 32 + (t & 31)];
  if (v == 12345.678f) sink[0] = 1.0f;
}

int main() {
  void* buf;
  float* sink;
  CHECK(hipMalloc(&buf, BYTES));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 0, BYTES));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(stream16, dim3(BYTES / 16 / 256), dim3(256), 0, 0, (const float4*)buf, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    printf("stream16   %8.3f ms  %7.1f GB/s\n", ms, BYTES / ms / 1e6);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(stream4, dim3(BYTES / 16 / 256), dim3(256), 0, 0, (const float*)buf, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    printf("stream4    %8.3f ms  %7.1f GB/s\n", ms, BYTES / ms / 1e6);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather16x8, dim3(BYTES / 16 / 256), dim3(256), 0, 0, (const float4*)buf, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    printf("gather16x8 %8.3f ms  %7.1f GB/s\n", ms, BYTES / ms / 1e6);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather4x32, dim3(BYTES / 4 / 256), dim3(256), 0, 0, (const float*)buf, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
    printf("gather4x32 %8.3f ms  %7.1f GB/s\n", ms, BYTES / ms / 1e6);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
