// Fused deformation MLP on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Reference: 10 nn.Linear + 7 ReLU modules (scene/deformation.py:53-76) = ~20 library GEMM/elementwise launches forward
// and ~40 backward on [P,128]/[P,64] activations; with P = 1.2 M the skinny GEMMs (N = 3..128) cost ~23 ms per
// iteration through hipBLASLt.  Here:
//   mlp_forward_kernel   one pass: a wave owns a 32-point tile, activations live in LDS as [feature][point] (row
//                        stride 33 -> conflict-free as MFMA B operand AND for the transposing global loads/stores),
//                        the layer's weights are staged in LDS as [in][out+1] (odd stride -> conflict-free as A operand
//                        both straight and transposed); the 5 hidden activations are stashed for the backward.
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
constexpr int WREGION = 64 * 65;  // floats: largest weight slab staged at once ([64 in][64 out + 1])

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

// Stage a [OUT,IN] row-major weight slab (input columns [in0, in0+NIN)) into LDS as wl[(i-in0)*(OUTPAD+1) + o],
// zero for o >= OUT.  Coalesced global reads along `in`, conflict-free LDS writes (odd row stride).  All loads of a
// thread are issued before the first LDS write so their latencies overlap.
template <int OUT, int IN, int OUTPAD, int NIN>
__device__ __forceinline__ void stage_weight(float* wl, const float* __restrict__ Wg, int in0, int tid) {
  constexpr int LD = OUTPAD + 1, TOTAL = OUTPAD * NIN, PER = (TOTAL + 255) / 256;
  float v[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int e = u * 256 + tid, o = e / NIN, i = e % NIN;
    v[u] = (e < TOTAL && o < OUT) ? Wg[(size_t)o * IN + in0 + i] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int e = u * 256 + tid, o = e / NIN, i = e % NIN;
    if (e < TOTAL) wl[i * LD + o] = v[u];
  }
}

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

struct MlpFwdArgs {
  s3g_mlp_params w;
  int P;
  const float* x;
  float *dx, *dshs, *feat, *stash;
};

// LDS: [WREGION] weights | per wave: X[128][33] | H[64][33] | T[64][33]
constexpr int WAVE_LDS = (FEAT + HID + HID) * LDA;
constexpr int MLP_LDS_FLOATS = WREGION + 4 * WAVE_LDS;

__global__ void __launch_bounds__(256) mlp_forward_kernel(const MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* X = lds + WREGION + wave * WAVE_LDS;
  float* H = X + FEAT * LDA;
  float* T = H + HID * LDA;
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;  // one stash plane
  for (int t0 = blockIdx.x * 4; t0 < ntiles; t0 += gridDim.x * 4) {  // uniform trip count per workgroup
    const int tile = t0 + wave;
    const int p0 = tile * MT;
    const int npts = tile < ntiles ? min(MT, a.P - p0) : 0;
    tile_load<FEAT, FEAT>(X, a.x, p0, npts, lane);
    f32x16 acc[2];
    // ---- hidden = W0 x + b0 (two K halves of W0 staged in turn) ----
    acc_init_bias<2>(acc, a.w.b0, HID, lane);
    for (int half = 0; half < 2; half++) {
      lds_barrier();
      stage_weight<HID, FEAT, HID, 64>(wl, a.w.W0, half * 64, tid);
      lds_barrier();
      gemm_straight<2, false>(wl, 65, X + half * 64 * LDA, 64, acc, lane);
    }
    acc_store<2, false>(acc, H, lane);
    if (a.stash) tile_store<HID>(H, a.stash + 0 * PS, p0, npts, lane);
    // ---- pos1 = relu(P1 relu(hidden) + pb1) -> X[0:64] ----
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.P1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, a.w.pb1, HID, lane);
    gemm_straight<2, true>(wl, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X, lane);
    if (a.stash) tile_store<HID>(X, a.stash + 1 * PS, p0, npts, lane);
    // ---- shs1 = relu(S1 relu(hidden) + sb1) -> X[64:128] ----
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.S1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, a.w.sb1, HID, lane);
    gemm_straight<2, true>(wl, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X + 64 * LDA, lane);
    if (a.stash) tile_store<HID>(X + 64 * LDA, a.stash + 2 * PS, p0, npts, lane);
    // ---- dx = P2 pos1 + pb2 ----
    lds_barrier();
    stage_weight<3, HID, 32, HID>(wl, a.w.P2, 0, tid);
    lds_barrier();
    {
      f32x16 o[1];
      acc_init_bias<1>(o, a.w.pb2, 3, lane);
      gemm_straight<1, false>(wl, 33, X, HID, o, lane);
      acc_store<1, false>(o, T, lane);
      tile_store<3>(T, a.dx, p0, npts, lane);
    }
    // ---- dshs = S2 shs1 + sb2 ----
    lds_barrier();
    stage_weight<48, HID, 64, HID>(wl, a.w.S2, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, a.w.sb2, 48, lane);
    gemm_straight<2, false>(wl, 65, X + 64 * LDA, HID, acc, lane);
    acc_store<2, false>(acc, T, lane);
    tile_store<48>(T, a.dshs, p0, npts, lane);
    // ---- dino1 = relu(D0 hidden + db0) -> X[0:64] ----
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.D0, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, a.w.db0, HID, lane);
    gemm_straight<2, false>(wl, 65, H, HID, acc, lane);
    acc_store<2, true>(acc, X, lane);
    if (a.stash) tile_store<HID>(X, a.stash + 3 * PS, p0, npts, lane);
    // ---- dino2 = relu(D1 dino1 + db1) -> X[64:128] ----
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.D1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, a.w.db1, HID, lane);
    gemm_straight<2, false>(wl, 65, X, HID, acc, lane);
    acc_store<2, true>(acc, X + 64 * LDA, lane);
    if (a.stash) tile_store<HID>(X + 64 * LDA, a.stash + 4 * PS, p0, npts, lane);
    // ---- feat = D2 dino2 + db2 ----
    lds_barrier();
    stage_weight<3, HID, 32, HID>(wl, a.w.D2, 0, tid);
    lds_barrier();
    {
      f32x16 o[1];
      acc_init_bias<1>(o, a.w.db2, 3, lane);
      gemm_straight<1, false>(wl, 33, X + 64 * LDA, HID, o, lane);
      acc_store<1, false>(o, T, lane);
      tile_store<3>(T, a.feat, p0, npts, lane);
    }
  }
}

struct MlpBwdArgs {
  s3g_mlp_params w;
  int P;
  const float *stash, *g_dx, *g_dshs, *g_feat;
  float *g_x, *ws;
};

// Per-point backward chain.  LDS per wave: U0[64][33] | U1[64][33] | H[64][33] | T[64][33]  (same footprint as forward)
__global__ void __launch_bounds__(256) mlp_backward_kernel(const MlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* U0 = lds + WREGION + wave * WAVE_LDS;
  float* U1 = U0 + HID * LDA;
  float* H = U1 + HID * LDA;
  float* T = H + HID * LDA;
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;
  for (int t0 = blockIdx.x * 4; t0 < ntiles; t0 += gridDim.x * 4) {
    const int tile = t0 + wave;
    const int p0 = tile * MT;
    const int npts = tile < ntiles ? min(MT, a.P - p0) : 0;
    f32x16 ghid[2], acc[2];
    acc_init_bias<2>(ghid, nullptr, 0, lane);
    tile_load<HID, HID>(H, a.stash + 0 * PS, p0, npts, lane);            // hidden (raw)
    // ================= dino head =================
    tile_load<HID, HID>(U1, a.stash + 4 * PS, p0, npts, lane);           // dino2
    tile_load<3, 32>(T, a.g_feat, p0, npts, lane);                        // g_feat, rows 3..31 zero
    lds_barrier();
    stage_weight<3, HID, 32, HID>(wl, a.w.D2, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 33, T, 32, acc, lane);                         // D2^T g_feat
    acc_store_masked<2>(acc, U1, U1, lane);                               // (.) * [dino2 > 0]  -> g wrt dino2 pre-activation
    tile_store<HID>(U1, a.ws + 0 * PS, p0, npts, lane);
    tile_load<HID, HID>(U0, a.stash + 3 * PS, p0, npts, lane);           // dino1
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.D1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 65, U1, HID, acc, lane);                       // D1^T g_d2
    acc_store_masked<2>(acc, U0, U0, lane);                               // * [dino1 > 0]
    tile_store<HID>(U0, a.ws + 1 * PS, p0, npts, lane);
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.D0, 0, tid);
    lds_barrier();
    gemm_transposed<2>(wl, 65, U0, HID, ghid, lane);                      // ghid += D0^T g_d1   (no mask: dino input is raw hidden)
    // ================= pos head =================
    tile_load<HID, HID>(U1, a.stash + 1 * PS, p0, npts, lane);           // pos1
    tile_load<3, 32>(T, a.g_dx, p0, npts, lane);
    lds_barrier();
    stage_weight<3, HID, 32, HID>(wl, a.w.P2, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 33, T, 32, acc, lane);
    acc_store_masked<2>(acc, U1, U1, lane);                               // g wrt pos1 pre-activation
    tile_store<HID>(U1, a.ws + 2 * PS, p0, npts, lane);
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.P1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 65, U1, HID, acc, lane);                       // P1^T g_pos1  (gradient wrt relu(hidden))
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (H[(mb * 32 + acc_row(r, lane)) * LDA + (lane & 31)] > 0.f) ghid[mb][r] += acc[mb][r];
    // ================= shs head =================
    tile_load<HID, HID>(U1, a.stash + 2 * PS, p0, npts, lane);           // shs1
    tile_load<48, 64>(U0, a.g_dshs, p0, npts, lane);                      // g_dshs, rows 48..63 zero
    lds_barrier();
    stage_weight<48, HID, 64, HID>(wl, a.w.S2, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 65, U0, 64, acc, lane);
    acc_store_masked<2>(acc, U1, U1, lane);
    tile_store<HID>(U1, a.ws + 3 * PS, p0, npts, lane);
    lds_barrier();
    stage_weight<HID, HID, HID, HID>(wl, a.w.S1, 0, tid);
    lds_barrier();
    acc_init_bias<2>(acc, nullptr, 0, lane);
    gemm_transposed<2>(wl, 65, U1, HID, acc, lane);
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (H[(mb * 32 + acc_row(r, lane)) * LDA + (lane & 31)] > 0.f) ghid[mb][r] += acc[mb][r];
    // ================= feature_out =================
    acc_store<2, false>(ghid, T, lane);
    tile_store<HID>(T, a.ws + 4 * PS, p0, npts, lane);
    for (int half = 0; half < 2; half++) {                                // g_x[:, half*64 : half*64+64] = W0[:, half]^T ghid
      lds_barrier();
      stage_weight<HID, FEAT, HID, 64>(wl, a.w.W0, half * 64, tid);
      lds_barrier();
      acc_init_bias<2>(acc, nullptr, 0, lane);
      gemm_transposed<2>(wl, 65, T, HID, acc, lane);
      acc_store<2, false>(acc, U0, lane);
#pragma unroll 8
      for (int pt = 0; pt < MT; pt++)
        if (pt < npts) a.g_x[(size_t)(p0 + pt) * FEAT + half * 64 + lane] = U0[lane * LDA + pt];
    }
  }
}

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

extern "C" size_t s3g_deform_mlp_stash_bytes(int P) { return (size_t)5 * (size_t)(P > 0 ? P : 0) * HID * sizeof(float); }

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
                                      float* feat, float* stash, void* stream_) {
  if (!w || P < 0 || (P > 0 && (!features || !dx || !dshs || !feat))) {
    set_error("s3g_deform_mlp_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  MlpFwdArgs a;
  a.w = *w; a.P = P; a.x = features; a.dx = dx; a.dshs = dshs; a.feat = feat; a.stash = stash;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + 3) / 4, 256);
  hipLaunchKernelGGL(mlp_forward_kernel, dim3(blocks), dim3(256), MLP_LDS_FLOATS * 4, (hipStream_t)stream_, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_deform_mlp_backward(const s3g_mlp_params* w, int P, const float* features, const float* stash,
                                       const float* g_dx, const float* g_dshs, const float* g_feat, float* g_features,
                                       const s3g_mlp_params* gw, float* workspace, void* stream_) {
  if (!w || !gw || P < 0 || (P > 0 && (!features || !stash || !g_dx || !g_dshs || !g_feat || !g_features || !workspace))) {
    set_error("s3g_deform_mlp_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  hipStream_t stream = (hipStream_t)stream_;
  MlpBwdArgs b;
  b.w = *w; b.P = P; b.stash = stash; b.g_dx = g_dx; b.g_dshs = g_dshs; b.g_feat = g_feat; b.g_x = g_features; b.ws = workspace;
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
