// Fused deformation MLP on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Reference: 10 nn.Linear + 7 ReLU modules (scene/deformation.py:53-76) = ~20 library GEMM/elementwise launches forward
// and ~40 backward on [P,128]/[P,64] activations; with P = 1.2 M the skinny GEMMs (N = 3..128) cost ~23 ms per
// iteration through hipBLASLt.  Here:
//   mlp_pack_kernel      builds the LDS image [in][out+1] of every layer once per call (odd stride -> conflict-free as
//                        MFMA A operand both straight and transposed).
//   mlp_forward_kernel   one pass: a wave owns a 32-point tile, activations live in LDS as [feature][point] (row
//                        stride 33 -> conflict-free as MFMA B operand AND for the transposing global loads/stores);
//                        while a layer's MFMAs run, the NEXT layer's weight slab streams global -> LDS through the DMA
//                        path (global_load_lds_dwordx4, no VGPR round trip) into the other half of a double buffer;
//                        the 5 hidden activations are stashed for the backward.
//   mlp_backward_kernel  the per-point chain (transposed-weight GEMMs + ReLU masks) -> g_features and 5 gradient signals.
//   mlp_wgrad_kernel     dW = sum_p g[p] (x) act[p] as an MFMA GEMM whose K dimension is the points (split over
//                        workgroups, accumulators stay in registers, one atomic flush per workgroup); biases alongside.
// HBM scratch is spent freely (2 x 1280 B per point): 3 GB of the 288 GB, ~1 ms of traffic for ~20 ms saved.
#include "common.hpp"

#include "../../include/s3g_mlp.h"

namespace s3g {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT = 32;    // points per wave tile (MFMA N)
constexpr int LDA = 33;   // activation row stride in LDS
constexpr int HID = 64;   // net_width
constexpr int FEAT = 128; // HexPlane feature width

// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains every outstanding GLOBAL store
// (s_waitcnt vmcnt(0)): the stash/output stores of a phase nobody in this kernel reads would stall all 4 waves at
// every one of the 9 phase boundaries.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// rows of the 32x32 accumulator held by (lane, reg): row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); col = lane & 31
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// acc[mb] (+)= W[mb*32.., :] * in   -- straight:  A(i,k) = W[out=m0+i][in=k] = wl[k*ld + m0 + i]
template <int MB, bool RELU_IN>
__device__ __forceinline__ void gemm_straight(const float* wl, int ld, const float* in, int K, f32x16 (&acc)[MB], int lane) {
  const int i = lane & 31, kk = lane >> 5;
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 2) {
    float b = in[(k0 + kk) * LDA + i];
    if (RELU_IN) b = fmaxf(b, 0.f);
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
      const float a = wl[(k0 + kk) * ld + mb * 32 + i];
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mb], 0, 0, 0);
    }
  }
}
// acc[mb] += W^T[mb*32.., :] * in  -- transposed: A(i,k) = W[out=k][in=m0+i] = wl[(m0+i)*ld + k]; K = (padded) out dim
template <int MB>
__device__ __forceinline__ void gemm_transposed(const float* wl, int ld, const float* in, int K, f32x16 (&acc)[MB], int lane) {
  const int i = lane & 31, kk = lane >> 5;
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 2) {
    const float b = in[(k0 + kk) * LDA + i];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
      const float a = wl[(mb * 32 + i) * ld + k0 + kk];
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mb], 0, 0, 0);
    }
  }
}

template <int MB>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MB], const float* __restrict__ bias, int out, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = mb * 32 + acc_row(r, lane);
      acc[mb][r] = (bias != nullptr && row < out) ? bias[row] : 0.f;
    }
}
template <int MB, bool RELU>
__device__ __forceinline__ void acc_store(const f32x16 (&acc)[MB], float* out, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = acc[mb][r];
      out[(mb * 32 + acc_row(r, lane)) * LDA + (lane & 31)] = RELU ? fmaxf(v, 0.f) : v;
    }
}
// out = acc masked by (mask_buf > 0)
template <int MB>
__device__ __forceinline__ void acc_store_masked(const f32x16 (&acc)[MB], const float* mask, float* out, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int o = (mb * 32 + acc_row(r, lane)) * LDA + (lane & 31);
      out[o] = mask[o] > 0.f ? acc[mb][r] : 0.f;
    }
}

// Tile <-> global transposing copies.  Global is point-major [P][WIDTH]; LDS is [ROWS >= WIDTH][LDA] (zero padded).
// Loads are issued in batches of 16 per lane before the LDS writes so their latencies overlap.
template <int WIDTH, int ROWS, bool RELU = false>
__device__ __forceinline__ void tile_load(float* buf, const float* __restrict__ g, int p0, int npts, int lane) {
  constexpr int PER = MT * ROWS / 64, BATCH = PER < 16 ? PER : 16;
  for (int c = 0; c < PER; c += BATCH) {
    float v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; u++) {
      const int e = (c + u) * 64 + lane, pt = e / ROWS, f = e % ROWS;  // e = point * ROWS + feature
      float x = (pt < npts && f < WIDTH) ? g[(size_t)(p0 + pt) * WIDTH + f] : 0.f;
      v[u] = RELU ? fmaxf(x, 0.f) : x;
    }
#pragma unroll
    for (int u = 0; u < BATCH; u++) {
      const int e = (c + u) * 64 + lane, pt = e / ROWS, f = e % ROWS;
      buf[f * LDA + pt] = v[u];
    }
  }
}
template <int WIDTH>
__device__ __forceinline__ void tile_store(const float* buf, float* __restrict__ g, int p0, int npts, int lane) {
  constexpr int PER = (MT * WIDTH + 63) / 64;
#pragma unroll 8
  for (int u = 0; u < PER; u++) {
    const int e = u * 64 + lane, pt = e / WIDTH, f = e % WIDTH;
    if (pt < npts) g[(size_t)(p0 + pt) * WIDTH + f] = buf[f * LDA + pt];
  }
}

// ---- weight slabs: LDS images built once per call in global memory, streamed into LDS by the DMA engine -----------
// Slab k is the [in][out+1] image of one layer (feature_out is cut in two K halves), padded to SLAB floats = 17 KiB so a
// slab is 17 global_load_lds_dwordx4 wave-instructions (1 KiB each).  Forward order 0..8; the backward walks 8..0 with
// the two W0 halves last.
constexpr int SLAB = 17 * 256;  // floats
constexpr int NSLAB = 9;        // W0[:, :64] | W0[:, 64:] | P1 | S1 | P2 | S2 | D0 | D1 | D2
constexpr int PACK_FLOATS = NSLAB * SLAB + 8 * 64;  // + the 8 bias vectors zero padded to 64

__global__ void __launch_bounds__(256) mlp_pack_kernel(const s3g_mlp_params w, float* __restrict__ packed) {
  const int k = blockIdx.x, tid = threadIdx.x;
  float* dst = packed + (size_t)k * SLAB;
  if (k == NSLAB) {  // biases
    float* bl = packed + (size_t)NSLAB * SLAB;
    const float* src[8] = {w.b0, w.pb1, w.sb1, w.pb2, w.sb2, w.db0, w.db1, w.db2};
    const int n[8] = {64, 64, 64, 3, 48, 64, 64, 3};
    for (int e = tid; e < 8 * 64; e += 256) bl[e] = (e & 63) < n[e >> 6] ? src[e >> 6][e & 63] : 0.f;
    return;
  }
  const float* W = k <= 1 ? w.W0 : k == 2 ? w.P1 : k == 3 ? w.S1 : k == 4 ? w.P2 : k == 5 ? w.S2 : k == 6 ? w.D0 : k == 7 ? w.D1 : w.D2;
  const int out = (k == 4 || k == 8) ? 3 : (k == 5 ? 48 : 64), outpad = (k == 4 || k == 8) ? 32 : 64;
  const int in = k <= 1 ? FEAT : HID, in0 = k == 1 ? 64 : 0, ld = outpad + 1;
  for (int e = tid; e < SLAB; e += 256) dst[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < outpad * 64; e += 256) {
    const int o = e / 64, i = e % 64;
    dst[i * ld + o] = o < out ? W[(size_t)o * in + in0 + i] : 0.f;
  }
}

// Asynchronous global -> LDS copy of one slab by the 4 waves of the workgroup (no VGPR round trip).
__device__ __forceinline__ void dma_slab(float* lds_dst, const float* __restrict__ gsrc, int wave, int lane) {
  for (int c = wave; c < SLAB / 256; c += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + c * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds_dst + c * 256), 16, 0, 0);
}
// End of a phase: the slab streamed during the phase has landed (vmcnt(0)), everyone is done with the current one.
#define PHASE_END()                                      \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
  lds_barrier();                                         \
  cur ^= 1

// accumulator -> global [P][WIDTH] directly (rows = output index, cols = points of the tile)
template <int WIDTH, int MB>
__device__ __forceinline__ void acc_store_global(const f32x16 (&acc)[MB], float* __restrict__ g, int p0, int npts, int lane) {
  const int col = lane & 31;
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = mb * 32 + acc_row(r, lane);
      if (row < WIDTH && col < npts) g[(size_t)(p0 + col) * WIDTH + row] = acc[mb][r];
    }
}

struct MlpFwdArgs {
  int P;
  const float* x;
  const float* packed;
  float *dx, *dshs, *feat, *stash;
};

// LDS: two weight slabs (double buffered by DMA) | biases | per wave: X[128][33] | H[64][33]
constexpr int WAVE_LDS = (FEAT + HID) * LDA;
constexpr int MLP_LDS_FLOATS = 2 * SLAB + 8 * 64 + 4 * WAVE_LDS;

__global__ void __launch_bounds__(256) mlp_forward_kernel(const MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* bl = lds + 2 * SLAB;  // biases: b0 | pb1 | sb1 | pb2 | sb2 | db0 | db1 | db2
  float* X = bl + 8 * 64 + wave * WAVE_LDS;
  float* H = X + FEAT * LDA;
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;  // one stash plane
  int cur = 0;
  for (int e = tid; e < 8 * 64; e += 256) bl[e] = a.packed[(size_t)NSLAB * SLAB + e];
  {
    const int tile = blockIdx.x * 4 + wave, p0 = tile * MT;
    tile_load<FEAT, FEAT>(X, a.x, p0, tile < ntiles ? min(MT, a.P - p0) : 0, lane);
  }
  dma_slab(lds, a.packed, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
#define WCUR (lds + cur * SLAB)
#define WNEXT(k) dma_slab(lds + (cur ^ 1) * SLAB, a.packed + (size_t)(k) * SLAB, wave, lane)
  for (int t0 = blockIdx.x * 4; t0 < ntiles; t0 += gridDim.x * 4) {  // uniform trip count per workgroup
    const int tile = t0 + wave;
    const int p0 = tile * MT;
    const int npts = tile < ntiles ? min(MT, a.P - p0) : 0;
    f32x16 acc[2];
    // ---- hidden = W0 x + b0 (two K halves) ----
    WNEXT(1);
    acc_init_bias<2>(acc, bl + 0 * 64, HID, lane);
    gemm_straight<2, false>(WCUR, 65, X, 64, acc, lane);
    PHASE_END();
    WNEXT(2);
    gemm_straight<2, false>(WCUR, 65, X + 64 * LDA, 64, acc, lane);
    acc_store<2, false>(acc, H, lane);
    if (a.stash) tile_store<HID>(H, a.stash + 0 * PS, p0, npts, lane);
    PHASE_END();
    // ---- pos1 = relu(P1 relu(hidden) + pb1) -> X[0:64] ----
    WNEXT(3);
    acc_init_bias<2>(acc, bl + 1 * 64, HID, lane);
    gemm_straight<2, true>(WCUR, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X, lane);
    if (a.stash) tile_store<HID>(X, a.stash + 1 * PS, p0, npts, lane);
    PHASE_END();
    // ---- shs1 = relu(S1 relu(hidden) + sb1) -> X[64:128] ----
    WNEXT(4);
    acc_init_bias<2>(acc, bl + 2 * 64, HID, lane);
    gemm_straight<2, true>(WCUR, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X + 64 * LDA, lane);
    if (a.stash) tile_store<HID>(X + 64 * LDA, a.stash + 2 * PS, p0, npts, lane);
    PHASE_END();
    // ---- dx = P2 pos1 + pb2 ----
    WNEXT(5);
    {
      f32x16 o[1];
      acc_init_bias<1>(o, bl + 3 * 64, 3, lane);
      gemm_straight<1, false>(WCUR, 33, X, HID, o, lane);
      acc_store_global<3, 1>(o, a.dx, p0, npts, lane);
    }
    PHASE_END();
    // ---- dshs = S2 shs1 + sb2 ----
    WNEXT(6);
    acc_init_bias<2>(acc, bl + 4 * 64, 48, lane);
    gemm_straight<2, false>(WCUR, 65, X + 64 * LDA, HID, acc, lane);
    acc_store_global<48, 2>(acc, a.dshs, p0, npts, lane);
    PHASE_END();
    // ---- dino1 = relu(D0 hidden + db0) -> X[0:64] ----
    WNEXT(7);
    acc_init_bias<2>(acc, bl + 5 * 64, HID, lane);
    gemm_straight<2, false>(WCUR, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X, lane);
    if (a.stash) tile_store<HID>(X, a.stash + 3 * PS, p0, npts, lane);
    PHASE_END();
    // ---- dino2 = relu(D1 dino1 + db1) -> X[64:128] ----
    WNEXT(8);
    acc_init_bias<2>(acc, bl + 6 * 64, HID, lane);
    gemm_straight<2, false>(WCUR, 65, X, HID, acc, lane);
    acc_store<2, true>(acc, X + 64 * LDA, lane);
    if (a.stash) tile_store<HID>(X + 64 * LDA, a.stash + 4 * PS, p0, npts, lane);
    PHASE_END();
    // ---- feat = D2 dino2 + db2 ; W0's first half and the NEXT tile's features stream in meanwhile ----
    WNEXT(0);
    {
      f32x16 o[1];
      acc_init_bias<1>(o, bl + 7 * 64, 3, lane);
      gemm_straight<1, false>(WCUR, 33, X + 64 * LDA, HID, o, lane);
      acc_store_global<3, 1>(o, a.feat, p0, npts, lane);
      const int ntile = tile + gridDim.x * 4, np0 = ntile * MT;
      tile_load<FEAT, FEAT>(X, a.x, np0, ntile < ntiles ? min(MT, a.P - np0) : 0, lane);  // X is dead from here on
    }
    PHASE_END();
  }
}

struct MlpBwdArgs {
  int P;
  const float* packed;
  const float *stash, *g_dx, *g_dshs, *g_feat;
  float *g_x, *ws;
};

// Per-point backward chain.  LDS per wave: U0[64][33] | U1[64][33] | H[64][33]  (same footprint as forward)
__global__ void __launch_bounds__(256) mlp_backward_kernel(const MlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* U0 = lds + 2 * SLAB + 8 * 64 + wave * WAVE_LDS;
  float* U1 = U0 + HID * LDA;
  float* H = U1 + HID * LDA;
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;
  int cur = 0;
  dma_slab(lds, a.packed + (size_t)8 * SLAB, wave, lane);  // D2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  for (int t0 = blockIdx.x * 4; t0 < ntiles; t0 += gridDim.x * 4) {
    const int tile = t0 + wave;
    const int p0 = tile * MT;
    const int npts = tile < ntiles ? min(MT, a.P - p0) : 0;
    f32x16 ghid[2], acc[2];
    acc_init_bias<2>(ghid, nullptr, 0, lane);
    tile_load<HID, HID>(H, a.stash + 0 * PS, p0, npts, lane);            // hidden (raw)
    tile_load<HID, HID>(U1, a.stash + 4 * PS, p0, npts, lane);           // dino2
    tile_load<3, 32>(U0, a.g_feat, p0, npts, lane);                       // g_feat in rows 0..2, rows 3..31 zero
    // ================= dino head =================
    WNEXT(7);                                                             // D1
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 33, U0, 32, acc, lane);                      // D2^T g_feat
    acc_store_masked<2>(acc, U1, U1, lane);                               // * [dino2 > 0] -> g wrt dino2 pre-activation
    tile_store<HID>(U1, a.ws + 0 * PS, p0, npts, lane);
    tile_load<HID, HID>(U0, a.stash + 3 * PS, p0, npts, lane);           // dino1
    PHASE_END();
    WNEXT(6);                                                             // D0
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U1, HID, acc, lane);                     // D1^T g_d2
    acc_store_masked<2>(acc, U0, U0, lane);                               // * [dino1 > 0]
    tile_store<HID>(U0, a.ws + 1 * PS, p0, npts, lane);
    PHASE_END();
    WNEXT(4);                                                             // P2
    gemm_transposed<2>(WCUR, 65, U0, HID, ghid, lane);                    // ghid += D0^T g_d1 (dino input is raw hidden: no mask)
    tile_load<HID, HID>(U1, a.stash + 1 * PS, p0, npts, lane);           // pos1
    tile_load<3, 32>(U0, a.g_dx, p0, npts, lane);
    PHASE_END();
    // ================= pos head =================
    WNEXT(2);                                                             // P1
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 33, U0, 32, acc, lane);                      // P2^T g_dx
    acc_store_masked<2>(acc, U1, U1, lane);                               // g wrt pos1 pre-activation
    tile_store<HID>(U1, a.ws + 2 * PS, p0, npts, lane);
    PHASE_END();
    WNEXT(5);                                                             // S2
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U1, HID, acc, lane);                     // P1^T g_pos1 (gradient wrt relu(hidden))
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (H[(mb * 32 + acc_row(r, lane)) * LDA + (lane & 31)] > 0.f) ghid[mb][r] += acc[mb][r];
    tile_load<HID, HID>(U1, a.stash + 2 * PS, p0, npts, lane);           // shs1
    tile_load<48, 64>(U0, a.g_dshs, p0, npts, lane);                      // g_dshs, rows 48..63 zero
    PHASE_END();
    // ================= shs head =================
    WNEXT(3);                                                             // S1
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U0, 64, acc, lane);                      // S2^T g_dshs
    acc_store_masked<2>(acc, U1, U1, lane);
    tile_store<HID>(U1, a.ws + 3 * PS, p0, npts, lane);
    PHASE_END();
    WNEXT(0);                                                             // W0[:, :64]
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U1, HID, acc, lane);                     // S1^T g_shs1
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (H[(mb * 32 + acc_row(r, lane)) * LDA + (lane & 31)] > 0.f) ghid[mb][r] += acc[mb][r];
    acc_store<2, false>(ghid, U1, lane);                                  // total gradient wrt hidden
    tile_store<HID>(U1, a.ws + 4 * PS, p0, npts, lane);
    PHASE_END();
    // ================= feature_out: g_x[:, half] = W0[:, half]^T ghid =================
    WNEXT(1);                                                             // W0[:, 64:]
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U1, HID, acc, lane);
    acc_store<2, false>(acc, U0, lane);
#pragma unroll 8
    for (int pt = 0; pt < MT; pt++)
      if (pt < npts) a.g_x[(size_t)(p0 + pt) * FEAT + lane] = U0[lane * LDA + pt];
    PHASE_END();
    WNEXT(8);                                                             // D2 for the next tile
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(WCUR, 65, U1, HID, acc, lane);
    acc_store<2, false>(acc, U0, lane);
#pragma unroll 8
    for (int pt = 0; pt < MT; pt++)
      if (pt < npts) a.g_x[(size_t)(p0 + pt) * FEAT + 64 + lane] = U0[lane * LDA + pt];
    PHASE_END();
  }
}
#undef PHASE_END
#undef WCUR
#undef WNEXT

// dW[o][i] += sum_p G[p][o] * A[p][i];  db[o] += sum_p G[p][o].   K dimension = points, streamed from HBM:
// the next tile's G and A rows are prefetched into registers while the current tile's MFMAs run.
struct WgradArgs {
  const float* G;  // [P][GW]
  const float* A;  // [P][AW]
  float* dW;       // [GW][AW]
  float* db;       // [GW]
  int P;
};
template <int GW, int AW>
struct WgradCfg {
  static constexpr int MB = GW > 32 ? 2 : 1, NB = AW / 32, GROWS = MB * 32;
  static constexpr int GPER = MT * GROWS / 64, APER = MT * AW / 64;
};

template <int GW, int AW, bool RELU_A>
__global__ void __launch_bounds__(256) mlp_wgrad_kernel(const WgradArgs a) {
  using Cf = WgradCfg<GW, AW>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Gl = lds + wave * (Cf::GROWS + AW) * LDA;  // [GROWS][33], rows >= GW zero
  float* Al = Gl + Cf::GROWS * LDA;                  // [AW][33]
  f32x16 acc[Cf::MB][Cf::NB];
#pragma unroll
  for (int m = 0; m < Cf::MB; m++)
#pragma unroll
    for (int n = 0; n < Cf::NB; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
  float bsum = 0.f;  // lane o accumulates db[o]
  const int ntiles = (a.P + MT - 1) / MT;
  const int i = lane & 31, kk = lane >> 5;
  float gv[Cf::GPER], av[Cf::APER];
  auto prefetch = [&](int tile) {
    const int p0 = tile * MT, npts = min(MT, a.P - p0);
#pragma unroll
    for (int u = 0; u < Cf::GPER; u++) {
      const int e = u * 64 + lane, pt = e / Cf::GROWS, f = e % Cf::GROWS;
      gv[u] = (pt < npts && f < GW) ? a.G[(size_t)(p0 + pt) * GW + f] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < Cf::APER; u++) {
      const int e = u * 64 + lane, pt = e / AW, f = e % AW;
      const float x = pt < npts ? a.A[(size_t)(p0 + pt) * AW + f] : 0.f;
      av[u] = RELU_A ? fmaxf(x, 0.f) : x;
    }
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) prefetch(tile);
  for (; tile < ntiles; tile += gridDim.x * 4) {
#pragma unroll
    for (int u = 0; u < Cf::GPER; u++) {
      const int e = u * 64 + lane, pt = e / Cf::GROWS, f = e % Cf::GROWS;
      Gl[f * LDA + pt] = gv[u];
    }
#pragma unroll
    for (int u = 0; u < Cf::APER; u++) {
      const int e = u * 64 + lane, pt = e / AW, f = e % AW;
      Al[f * LDA + pt] = av[u];
    }
    const int next = tile + gridDim.x * 4;
    if (next < ntiles) prefetch(next);  // global loads fly while the MFMAs below run
    // one wave owns Gl/Al: LDS ops of a wave execute in order; the fences only pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int k0 = 0; k0 < MT; k0 += 2) {  // K = points
      float bv[Cf::NB];
#pragma unroll
      for (int n = 0; n < Cf::NB; n++) bv[n] = Al[(n * 32 + i) * LDA + k0 + kk];  // B(k = point, j = in)
#pragma unroll
      for (int m = 0; m < Cf::MB; m++) {
        const float avv = Gl[(m * 32 + i) * LDA + k0 + kk];  // A(i = out, k = point)
#pragma unroll
        for (int n = 0; n < Cf::NB; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv, bv[n], acc[m][n], 0, 0, 0);
      }
    }
    if (lane < Cf::GROWS) {
      float s0 = 0.f;
#pragma unroll 8
      for (int j = 0; j < MT; j++) s0 += Gl[lane * LDA + j];
      bsum += s0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // flush: acc[m][n][r] is dW[m*32 + row][n*32 + col]
#pragma unroll
  for (int m = 0; m < Cf::MB; m++)
#pragma unroll
    for (int n = 0; n < Cf::NB; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = m * 32 + acc_row(r, lane);
        if (o < GW) atomicAdd(&a.dW[(size_t)o * AW + n * 32 + (lane & 31)], acc[m][n][r]);
      }
  if (lane < GW && a.db != nullptr) atomicAdd(&a.db[lane], bsum);
}

template <int GW, int AW, bool RELU_A>
static int launch_wgrad(const float* G, const float* A, float* dW, float* db, int P, hipStream_t stream) {
  using Cf = WgradCfg<GW, AW>;
  constexpr int lds_bytes = 4 * (Cf::GROWS + AW) * LDA * 4;
  static bool attr = false;
  if (!attr) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_wgrad_kernel<GW, AW, RELU_A>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr = true;
  }
  WgradArgs a{G, A, dW, db, P};
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + 3) / 4, 512);
  hipLaunchKernelGGL((mlp_wgrad_kernel<GW, AW, RELU_A>), dim3(blocks), dim3(256), lds_bytes, stream, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

}  // namespace s3g

using namespace s3g;

// stash = [packed weight slabs + biases (PACK_FLOATS)] [5 x P x 64 activations]
extern "C" size_t s3g_deform_mlp_stash_bytes(int P) {
  return ((size_t)PACK_FLOATS + (size_t)5 * (size_t)(P > 0 ? P : 0) * HID) * sizeof(float);
}
extern "C" size_t s3g_deform_mlp_pack_bytes(void) { return (size_t)PACK_FLOATS * sizeof(float); }

static int mlp_set_attrs() {
  static bool done = false;
  if (!done) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    done = true;
  }
  return S3G_OK;
}

extern "C" int s3g_deform_mlp_forward(const s3g_mlp_params* w, int P, const float* features, float* dx, float* dshs,
                                      float* feat, float* stash, int save_activations, void* stream_) {
  if (!w || P < 0 || (P > 0 && (!features || !dx || !dshs || !feat || !stash))) {
    set_error("s3g_deform_mlp_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(NSLAB + 1), dim3(256), 0, stream, *w, stash);
  MlpFwdArgs a;
  a.P = P; a.x = features; a.packed = stash; a.dx = dx; a.dshs = dshs; a.feat = feat;
  a.stash = save_activations ? stash + PACK_FLOATS : nullptr;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + 3) / 4, 256);
  hipLaunchKernelGGL(mlp_forward_kernel, dim3(blocks), dim3(256), MLP_LDS_FLOATS * 4, stream, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_deform_mlp_backward(const s3g_mlp_params* w, int P, const float* features, const float* stash_,
                                       const float* g_dx, const float* g_dshs, const float* g_feat, float* g_features,
                                       const s3g_mlp_params* gw, float* workspace, void* stream_) {
  if (!w || !gw || P < 0 || (P > 0 && (!features || !stash_ || !g_dx || !g_dshs || !g_feat || !g_features || !workspace))) {
    set_error("s3g_deform_mlp_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const float* stash = stash_ + PACK_FLOATS;  // activations; the packed weight slabs of the forward sit in front
  MlpBwdArgs b;
  b.P = P; b.packed = stash_; b.stash = stash; b.g_dx = g_dx; b.g_dshs = g_dshs; b.g_feat = g_feat; b.g_x = g_features; b.ws = workspace;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + 3) / 4, 256);
  hipLaunchKernelGGL(mlp_backward_kernel, dim3(blocks), dim3(256), MLP_LDS_FLOATS * 4, stream, b);
  S3G_HIP_CHECK(hipGetLastError());
  const size_t PS = (size_t)P * HID;
  if (int e = launch_wgrad<3, 64, false>(g_feat, stash + 4 * PS, gw->D2, gw->db2, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, false>(workspace + 0 * PS, stash + 3 * PS, gw->D1, gw->db1, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, false>(workspace + 1 * PS, stash + 0 * PS, gw->D0, gw->db0, P, stream)) return e;
  if (int e = launch_wgrad<3, 64, false>(g_dx, stash + 1 * PS, gw->P2, gw->pb2, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, true>(workspace + 2 * PS, stash + 0 * PS, gw->P1, gw->pb1, P, stream)) return e;
  if (int e = launch_wgrad<48, 64, false>(g_dshs, stash + 2 * PS, gw->S2, gw->sb2, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, true>(workspace + 3 * PS, stash + 0 * PS, gw->S1, gw->sb1, P, stream)) return e;
  if (int e = launch_wgrad<64, 128, false>(workspace + 4 * PS, features, gw->W0, gw->b0, P, stream)) return e;
  return S3G_OK;
}
