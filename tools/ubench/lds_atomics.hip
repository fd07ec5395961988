// Microbenchmark: cost of LDS atomics on gfx950 per wave64 instruction, with s_memtime around a loop of REPS instructions.
// One wave per workgroup (nothing else contends for the CU's LDS), then 8 waves per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics.hip -o tools/ubench/lds_atomics && tools/ubench/lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int REPS = 256;
enum Op { ADD_F32, ADD_U32, ADD_RTN_U32, PLAIN_RMW, PLAIN_WRITE };
enum Pat { DISTINCT, SAME, STRIDE32, PARTIAL8 };
template <int OP, int PAT>
__global__ void k(unsigned long long* out, float* sink) {
  __shared__ float lds[64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  int idx = PAT == DISTINCT ? lane : PAT == SAME ? 0 : PAT == STRIDE32 ? (lane * 32) % (64 * 64) : lane;
  idx += wave * 64 * 4 % (64 * 32);
  const bool on = PAT == PARTIAL8 ? lane < 8 : true;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int r = 0; r < REPS; r++) {
    if (on) {
      if (OP == ADD_F32) atomicAdd(&lds[idx], 1.0f);
      else if (OP == ADD_U32) atomicAdd(reinterpret_cast<uint32_t*>(&lds[idx]), 1u);
      else if (OP == ADD_RTN_U32) acc += atomicAdd(reinterpret_cast<uint32_t*>(&lds[idx]), 1u);
      else if (OP == PLAIN_RMW) lds[idx] += 1.0f;
      else lds[idx] = (float)r;
    }
    asm volatile("" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = lds[lane];
  __syncthreads();
  if (threadIdx.x == 0) sink[1] = lds[idx];
}
template <int OP, int PAT>
static void run(const char* name, unsigned long long* d, float* s) {
  for (int waves : {1, 8}) {
    unsigned long long h = 0;
    for (int it = 0; it < 2; it++) {
      hipLaunchKernelGGL((k<OP, PAT>), dim3(256), dim3(64 * waves), 0, 0, d, s);
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%-34s %d wave(s)/WG: %7.1f cycles per wave instruction\n", name, waves, (double)h / REPS);
  }
}
int main() {
  unsigned long long* d; float* s;
  hipMalloc(&d, 64); hipMalloc(&s, 64);
  run<ADD_F32, DISTINCT>("ds_add_f32      64 distinct", d, s);
  run<ADD_F32, STRIDE32>("ds_add_f32      stride 32 (1 bank)", d, s);
  run<ADD_F32, SAME>("ds_add_f32      same address", d, s);
  run<ADD_F32, PARTIAL8>("ds_add_f32      8 lanes active", d, s);
  run<ADD_U32, DISTINCT>("ds_add_u32      64 distinct", d, s);
  run<ADD_U32, SAME>("ds_add_u32      same address", d, s);
  run<ADD_RTN_U32, DISTINCT>("ds_add_rtn_u32  64 distinct", d, s);
  run<ADD_RTN_U32, SAME>("ds_add_rtn_u32  same address", d, s);
  run<ADD_RTN_U32, PARTIAL8>("ds_add_rtn_u32  8 lanes active", d, s);
  run<PLAIN_RMW, DISTINCT>("plain lds += (read, add, write)", d, s);
  run<PLAIN_WRITE, DISTINCT>("plain ds_write_b32", d, s);
  return 0;
}
