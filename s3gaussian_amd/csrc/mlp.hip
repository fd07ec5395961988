// Fused deformation MLP on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Reference: 10 nn.Linear + 7 ReLU modules (scene/deformation.py:53-76) = ~20 library GEMM/elementwise launches forward
// and ~40 backward on [P,128]/[P,64] activations; with P = 1.2 M the skinny GEMMs (N = 3..128) cost ~23 ms per
// iteration through hipBLASLt.  Here:
//   mlp_pack_kernel      builds, once per call, the LDS image [in][out+1] of every layer (odd stride -> conflict-free as
//                        MFMA A operand both straight and transposed); 9 slabs + biases = 155 KB.
//   mlp_forward_kernel   one persistent workgroup per CU keeps the WHOLE weight image in LDS (one DMA burst,
//                        global_load_lds_dwordx4).  A wave owns a 32-point tile and its activations never leave the
//                        MFMA accumulator registers: the K order of an MFMA is free, so the accumulator registers of one
//                        layer are fed back, as they are, as the B operand of the next (see gemm_reg).  No activation
//                        LDS traffic, no barriers after the weight load; waves run independently.  The 5 hidden
//                        activations are stashed for the backward with 16-byte stores.
//   mlp_backward_kernel  the per-point chain (transposed-weight reads of the same image + ReLU masks from the stash) ->
//                        g_features and 5 gradient signals, same register-resident scheme.
//   mlp_wgrad_kernel     dW = sum_p g[p] (x) act[p] as an MFMA GEMM whose K dimension is the points, operands loaded
//                        straight from HBM (permuted M/N rows make every lane's operands one contiguous load),
//                        accumulators in registers, one LDS-combined atomic flush per workgroup; biases alongside.
// HBM scratch is spent freely (2 x 1280 B per point): 3 GB of the 288 GB, ~1 ms of traffic for ~20 ms saved.
#include "common.hpp"
#include "hexplane_dev.hpp"

#include "../../include/s3g_mlp.h"

namespace s3g {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT = 32;    // points per wave tile (MFMA N)
constexpr int HID = 64;   // net_width
constexpr int FEAT = 128; // HexPlane feature width

// Accumulator layout of v_mfma_f32_32x32x2_f32: register r of lane l holds (row, col) = (rrow(r) + 4 * (l >> 5), l & 31)
// with rrow(r) = (r & 3) + 8 * (r >> 2).  Rows are output features, columns the 32 points of the tile.
__device__ __forceinline__ constexpr int rrow(int r) { return (r & 3) + 8 * (r >> 2); }
__device__ __forceinline__ int acc_row(int reg, int lane) { return rrow(reg) + 4 * (lane >> 5); }

// The trick that keeps activations out of LDS: the MFMA's K index is a summation index, so its order is free.  At K step
// (mbi, r) lane l supplies as B operand its OWN accumulator register in[mbi][r] -- that is feature
// f = 32*mbi + rrow(r) + 4*(l>>5) of point l&31 -- and the A operand is read from the weight image at that same f.
// A layer's output registers are therefore directly the next layer's input operand: no transposition, no LDS round
// trip, no barrier; waves run independently.
//   straight:   acc[mbo] += W[32*mbo + i][f] * in[f]      A = wl[f * ld + 32*mbo + i]     (wl = [in][out+1] image)
template <int MBO, int MBI, bool RELU_IN, int RSTEPS = 16>
__device__ __forceinline__ void gemm_reg(const float* wl, int ld, const f32x16 (&in)[MBI], f32x16 (&acc)[MBO], int lane) {
  const float* base = wl + 4 * (lane >> 5) * ld + (lane & 31);
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int r = 0; r < RSTEPS; r++) {
      float b = in[mbi][r];
      if (RELU_IN) b = fmaxf(b, 0.f);
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++)
        acc[mbo] = __builtin_amdgcn_mfma_f32_32x32x2f32(base[(32 * mbi + rrow(r)) * ld + 32 * mbo], b, acc[mbo], 0, 0, 0);
    }
}
//   transposed: acc[mbo] += W[f][32*mbo + i] * g[f]       A = wl[(32*mbo + i) * ld + f]   (f runs over OUTPUT features)
template <int MBO, int MBI, int RSTEPS = 16>
__device__ __forceinline__ void gemm_reg_t(const float* wl, int ld, const f32x16 (&g)[MBI], f32x16 (&acc)[MBO], int lane) {
  const float* base = wl + (lane & 31) * ld + 4 * (lane >> 5);
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int r = 0; r < RSTEPS; r++) {
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++)
        acc[mbo] = __builtin_amdgcn_mfma_f32_32x32x2f32(base[32 * mbo * ld + 32 * mbi + rrow(r)], g[mbi][r], acc[mbo], 0, 0, 0);
    }
}

// The two 3-row heads (pos_deform / dino_head output layers, 64 -> 3) on v_mfma_f32_4x4x1_16B_f32 (round 5).  A 32x32x2 MFMA spends
// a full 32-row block (64 cycles per K step) on three live rows: 2 x 32 of the forward's 512 MFMA slots per tile.  The 4x4x1
// instruction is sixteen independent 4x4 outer products (8 cycles): block b = lanes 4b .. 4b+3; lane 4b+j supplies B[j] and receives
// column j of the block in four registers, lane 4b+i supplies A[i].  It fits the register-resident scheme without moving anything:
//   B = the lane's OWN activation register in[mbi][r] -- feature f = 32 mbi + rrow(r) + 4 (lane >> 5) of point lane & 31; the four
//       lanes of a block share f (blocks do not straddle lane 32) and hold four different points;
//   A = W[lane & 3][f] from the [in][out + 1] weight image (rows 3 .. 31 of a head slab are zero, so i = 3 contributes nothing);
//   D = in lane l, registers 0 .. 2: rows 0 .. 2 of the output for point l & 31, summed over the features of the lane's half.
// The two halves (lanes l and l + 32 hold the K steps of features 4h .. 4h + 3 mod 8) meet in one cross-half add.  Summation
// order differs from the 32x32x2 chain (two half-K chains per mbi, added at the end): the training forward and the inference kernel
// use THIS function both, so they stay bit-identical to each other (tests/test_infer_gpu.py).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void head3_fw(const float* wl /* [64 in][33] head slab */, const float* bias /* LDS, >= 3 floats */,
                                         const f32x16 (&in)[2], float (&o)[3], int lane) {
  const int h = lane >> 5;
  const float* base = wl + 4 * h * 33 + (lane & 3);
  f32x4 c0, c1;
  c0[0] = h == 0 ? bias[0] : 0.f; c0[1] = h == 0 ? bias[1] : 0.f; c0[2] = h == 0 ? bias[2] : 0.f; c0[3] = 0.f;
  c1[0] = c1[1] = c1[2] = c1[3] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {      // two independent accumulation chains (mbi = 0 / 1): no MFMA waits for its predecessor's result
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(base[rrow(r) * 33], in[0][r], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(base[(32 + rrow(r)) * 33], in[1][r], c1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float v = c0[i] + c1[i];
    o[i] = v + __shfl_xor(v, 32);   // both halves end up with the total; the stores below use lanes 0 .. 31
  }
}
__device__ __forceinline__ void store3(const float (&o)[3], float* __restrict__ g, int p0, int npts, int lane) {
  if (lane < npts) {   // [P][3] rows, not 16-byte aligned
    float* row = g + (size_t)(p0 + lane) * 3;
    row[0] = o[0]; row[1] = o[1]; row[2] = o[2];
  }
}

template <int MB>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MB]) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[mb][r] = 0.f;
}
// acc = bias (LDS, zero padded to 64): 4 consecutive features per 16-byte read
template <int MB>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[MB], const float* bias, int lane) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 v = *reinterpret_cast<const float4*>(bias + 32 * mb + 8 * q + 4 * (lane >> 5));
      acc[mb][4 * q + 0] = v.x; acc[mb][4 * q + 1] = v.y; acc[mb][4 * q + 2] = v.z; acc[mb][4 * q + 3] = v.w;
    }
}
// Registers <-> a [P][WIDTH] global array, columns col0 .. col0 + 32*MB of it: lane (point j, half h) moves the four
// consecutive features 32*mb + 8*q + 4*h .. +3 as one 16-byte access (features >= VALID are zero / not stored).
template <int WIDTH, int MB, int VALID = 32 * MB>
__device__ __forceinline__ void act_load(f32x16 (&a)[MB], const float* __restrict__ g, int col0, int p0, int npts, int lane) {
  const int j = lane & 31, h = lane >> 5;
  const float* row = g + (size_t)(p0 + j) * WIDTH + col0 + 4 * h;
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (32 * mb + 8 * q < VALID && 32 * mb + 8 * q + 4 * h < VALID && j < npts)
        v = *reinterpret_cast<const float4*>(row + 32 * mb + 8 * q);
      a[mb][4 * q + 0] = v.x; a[mb][4 * q + 1] = v.y; a[mb][4 * q + 2] = v.z; a[mb][4 * q + 3] = v.w;
    }
}
template <int WIDTH, int MB, bool RELU, int VALID = 32 * MB>
__device__ __forceinline__ void act_store(const f32x16 (&a)[MB], float* __restrict__ g, int col0, int p0, int npts, int lane) {
  const int j = lane & 31, h = lane >> 5;
  float* row = g + (size_t)(p0 + j) * WIDTH + col0 + 4 * h;
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (32 * mb + 8 * q >= VALID) continue;
      float4 v = make_float4(a[mb][4 * q + 0], a[mb][4 * q + 1], a[mb][4 * q + 2], a[mb][4 * q + 3]);
      if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (32 * mb + 8 * q + 4 * h < VALID && j < npts) *reinterpret_cast<float4*>(row + 32 * mb + 8 * q) = v;
    }
}
// 3-wide heads ([P][3], not 16-byte aligned): features 0..2 sit in registers 0..2 of the h = 0 lanes
__device__ __forceinline__ void act_load3(f32x16 (&a)[1], const float* __restrict__ g, int p0, int npts, int lane) {
  acc_zero<1>(a);
  if (lane < npts) {
    const float* row = g + (size_t)(p0 + lane) * 3;
    a[0][0] = row[0]; a[0][1] = row[1]; a[0][2] = row[2];
  }
}
__device__ __forceinline__ void act_store3(const f32x16 (&a)[1], float* __restrict__ g, int p0, int npts, int lane) {
  if (lane < npts) {
    float* row = g + (size_t)(p0 + lane) * 3;
    row[0] = a[0][0]; row[1] = a[0][1]; row[2] = a[0][2];
  }
}
template <int MB>
__device__ __forceinline__ void relu_inplace(f32x16 (&a)[MB]) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) a[mb][r] = fmaxf(a[mb][r], 0.f);
}
// dst (op)= acc where mask > 0
template <int MB, bool ACCUM>
__device__ __forceinline__ void masked(f32x16 (&dst)[MB], const f32x16 (&acc)[MB], const f32x16 (&mask)[MB]) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = mask[mb][r] > 0.f ? acc[mb][r] : 0.f;
      dst[mb][r] = ACCUM ? dst[mb][r] + v : v;
    }
}

// ReLU masks as bits: bit (16*mb + r) of a lane's word = (a[mb][r] > 0).  The backward chain needs the forward activations
// only as ReLU masks; reading them as one 32-bit word per lane and plane (8 B per point and plane) instead of the fp32
// activation planes (256 B per point and plane) removes 1280 of the 3288 bytes per point the backward used to move AND every
// dependent load from its critical path (the words of the next tile are prefetched a whole tile ahead).
template <int MB>
__device__ __forceinline__ uint32_t pack_positive(const f32x16 (&a)[MB]) {
  uint32_t b = 0;
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      // x > 0  <=>  its bit pattern as a signed integer is >= 1 (negative floats and -0 are negative integers, +0 is 0):
      // med3(x, 0, 1) is the bit, one v_med3_i32 + one v_lshl_or_b32 per element
      const int bit = min(max(__float_as_int(a[mb][r]), 0), 1);
      b |= (uint32_t)bit << (16 * mb + r);
    }
  return b;
}
// dst (op)= acc where the mask bit is set
template <int MB, bool ACCUM>
__device__ __forceinline__ void masked_bits(f32x16 (&dst)[MB], const f32x16 (&acc)[MB], uint32_t bits) {
#pragma unroll
  for (int mb = 0; mb < MB; mb++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v = ((bits >> (16 * mb + r)) & 1u) ? acc[mb][r] : 0.f;
      dst[mb][r] = ACCUM ? dst[mb][r] + v : v;
    }
}

// ---- fp32 GEMMs on the bf16 matrix pipe: three-way operand split ------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at the vector-fma rate (64 cycles per 4096 FLOP and SIMD); v_mfma_f32_32x32x16_bf16 does 32768
// FLOP in 32 cycles on the real matrix pipe, beside the VALU instead of in its place.  Every fp32 operand is written as the EXACT
// sum of three bf16 numbers (round to nearest, subtract, repeat: 8 + 8 + 8 significand bits), and a product a*b is accumulated
// as the six piece products whose weight is >= 2^-16 of it:
//     a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0)          dropped: a1*b2 + a2*b1 + a2*b2 <= 2^-23 |a*b|
// Each piece product is exact in fp32 (8 x 8 bits) and the matrix pipe accumulates in fp32, so a dot product carries the error
// of an fp32 fma chain (rounding 2^-24 per step) plus <= 2^-23 per product: fp32 accuracy, 6 x 32 instead of 8 x 64 cycles per
// K = 16.  Used by the inference kernel only (deform_infer_kernel<UT, true>); the training kernels are the exact chains above.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_pair(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
// (a, b) -> word t of the three pieces; a == lo(p0) + lo(p1) + lo(p2) exactly (the residuals are exact fp32 differences)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = bf16_pair(a, b);
  const float ra = a - bf16_lo(p0), rb = b - bf16_hi(p0);
  p1 = bf16_pair(ra, rb);
  p2 = bf16_pair(ra - bf16_lo(p1), rb - bf16_hi(p1));
}
struct Split8 { u32x4 p[3]; };   // eight values = one lane's share of an MFMA operand (K = 16: k = 8 * (lane >> 5) + e), three pieces
__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
  Split8 s;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t p0, p1, p2;
    split_pair(v[2 * t], v[2 * t + 1], p0, p1, p2);
    s.p[0][t] = p0; s.p[1][t] = p1; s.p[2][t] = p2;
  }
  return s;
}
// acc += A * B for one K = 16 step, A and B given as pieces; smallest terms first
__device__ __forceinline__ f32x16 mfma_split(f32x16 acc, const Split8& a, const Split8& b) {
#define S3G_PIECE(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.p[i]), __builtin_bit_cast(bf16x8, b.p[j]), acc, 0, 0, 0)
  S3G_PIECE(2, 0); S3G_PIECE(1, 1); S3G_PIECE(0, 2); S3G_PIECE(1, 0); S3G_PIECE(0, 1); S3G_PIECE(0, 0);
#undef S3G_PIECE
  return acc;
}
// The accumulator registers of a layer as the B operand of the next, exactly as in gemm_reg: at K step (mbi, s) lane l supplies
// its own registers in[mbi][8s .. 8s+7] = features 32*mbi + 16*s + 4*(l>>5) + {0,1,2,3, 8,9,10,11} of point l & 31, and the A
// operand holds the weights of those same features in the same element order (split_feature below is that order).
template <int MBI> struct ActSplit { Split8 b[MBI][2]; };
template <int MBI, bool RELU>
__device__ __forceinline__ void act_split(ActSplit<MBI>& S, const f32x16 (&in)[MBI]) {
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int s = 0; s < 2; s++) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = RELU ? fmaxf(in[mbi][8 * s + e], 0.f) : in[mbi][8 * s + e];
      S.b[mbi][s] = split8(v);
    }
}
__host__ __device__ constexpr int split_feature(int ks, int h, int e) { return 16 * ks + 4 * h + (e & 3) + 8 * (e >> 2); }   // ks = 2 * mbi + s

// gemm_reg / gemm_reg_t with both operands split on the fly: the A operand is read from the SAME fp32 [in][out+1] LDS image (eight
// ds_read_b32 per fragment instead of one per fp32 MFMA: the same LDS traffic) and split by the lane that uses it.
template <int MBO, int MBI, bool RELU_IN, int RSTEPS = 16>
__device__ __forceinline__ void gemm_reg_split(const float* wl, int ld, const f32x16 (&in)[MBI], f32x16 (&acc)[MBO], int lane) {
  const float* base = wl + 4 * (lane >> 5) * ld + (lane & 31);
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int s = 0; s < (RSTEPS + 7) / 8; s++) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = RELU_IN ? fmaxf(in[mbi][8 * s + e], 0.f) : in[mbi][8 * s + e];
      const Split8 b = split8(v);
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = base[(32 * mbi + rrow(8 * s + e)) * ld + 32 * mbo];
        acc[mbo] = mfma_split(acc[mbo], split8(w), b);
      }
    }
}
template <int MBO, int MBI, int RSTEPS = 16>
__device__ __forceinline__ void gemm_reg_t_split(const float* wl, int ld, const f32x16 (&g)[MBI], f32x16 (&acc)[MBO], int lane) {
  const float* base = wl + (lane & 31) * ld + 4 * (lane >> 5);
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int s = 0; s < (RSTEPS + 7) / 8; s++) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = g[mbi][8 * s + e];
      const Split8 b = split8(v);
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = base[32 * mbo * ld + 32 * mbi + rrow(8 * s + e)];
        acc[mbo] = mfma_split(acc[mbo], split8(w), b);
      }
    }
}
// arithmetic selected at compile time by the kernels' SPLIT parameter
template <bool SPLIT, int MBO, int MBI, bool RELU_IN, int RSTEPS = 16>
__device__ __forceinline__ void gemm_fw(const float* wl, int ld, const f32x16 (&in)[MBI], f32x16 (&acc)[MBO], int lane) {
  if constexpr (SPLIT) gemm_reg_split<MBO, MBI, RELU_IN, RSTEPS>(wl, ld, in, acc, lane);
  else gemm_reg<MBO, MBI, RELU_IN, RSTEPS>(wl, ld, in, acc, lane);
}
template <bool SPLIT, int MBO, int MBI, int RSTEPS = 16>
__device__ __forceinline__ void gemm_bw(const float* wl, int ld, const f32x16 (&g)[MBI], f32x16 (&acc)[MBO], int lane) {
  if constexpr (SPLIT) gemm_reg_t_split<MBO, MBI, RSTEPS>(wl, ld, g, acc, lane);
  else gemm_reg_t<MBO, MBI, RSTEPS>(wl, ld, g, acc, lane);
}

// ---- weight slabs: the LDS image is built once per call in global memory and DMA-copied by every workgroup ---------
// Slab k is the [in][out+1] image of one layer (feature_out is cut in two K halves), padded to SLAB floats = 17 KiB =
// 17 global_load_lds_dwordx4 wave-instructions (1 KiB each).
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

constexpr int NWAVE = 8;  // waves per workgroup; one persistent workgroup per CU (the weights fill its LDS)
constexpr int MLP_LDS_FLOATS = PACK_FLOATS;

// Whole packed image (9 slabs + biases, 155 KB) global -> LDS through the DMA path, once per workgroup.
__device__ __forceinline__ void load_weights(float* lds, const float* __restrict__ packed, int wave, int lane) {
  static_assert(PACK_FLOATS % 256 == 0, "image is a whole number of 1 KiB DMA rows");
  for (int c = wave; c < PACK_FLOATS / 256; c += NWAVE)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(packed + c * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
#define WSLAB(k) (lds + (k) * SLAB)
#define BIAS(k) (lds + NSLAB * SLAB + (k) * 64)  // b0 | pb1 | sb1 | pb2 | sb2 | db0 | db1 | db2

struct MlpFwdArgs {
  int P;
  const float* x;
  const float* packed;
  float *dx, *dshs, *feat, *stash;
  uint32_t* maskbits;  // [tiles][5][64] ReLU mask words (NULL when no backward follows)
};

template <bool SPLIT>
__global__ void __launch_bounds__(NWAVE * 64) mlp_forward_kernel(const MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_weights(lds, a.packed, wave, lane);
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;  // one stash plane
  // the 128 input features of the NEXT tile are requested before this tile's MFMAs are issued (raw loads, clamped row: lanes
  // past the end of the array re-read the last point, whose outputs are never stored)
  struct XIn { float4 v[16]; };   // chunk c = columns 8c + 4h .. +3 of the lane's point
  const int jj = lane & 31, hh = lane >> 5;
  auto issue = [&](XIn& X, int tile) {
    const float* row = a.x + (size_t)min(tile * MT + jj, a.P - 1) * FEAT + 4 * hh;
#pragma unroll
    for (int c = 0; c < 16; c++) X.v[c] = *reinterpret_cast<const float4*>(row + 8 * c);
  };
  auto unpack = [&](f32x16 (&x)[2], const XIn& X, int half) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const float4 v = X.v[8 * half + c];
      x[c >> 2][4 * (c & 3) + 0] = v.x; x[c >> 2][4 * (c & 3) + 1] = v.y;
      x[c >> 2][4 * (c & 3) + 2] = v.z; x[c >> 2][4 * (c & 3) + 3] = v.w;
    }
  };
  const int stride = gridDim.x * NWAVE, t0 = blockIdx.x * NWAVE + wave;
  XIn cur, nxt;
  if (t0 < ntiles) issue(cur, t0);
  for (int tile = t0; tile < ntiles; tile += stride) {
    issue(nxt, min(tile + stride, ntiles - 1));
    __builtin_amdgcn_sched_barrier(0);
    const int p0 = tile * MT, npts = min(MT, a.P - p0);
    f32x16 hid[2], act[2], acc[2], o[1];
    int ln = lane;   // SPLIT: per-tile copy, so that the seven per-lane output pointers are not carried (and spilled) across the loop
    if constexpr (SPLIT) asm volatile("" : "+v"(ln));
    {  // hidden = W0 x + b0, K = 128 in two halves
      f32x16 x[2];
      acc_bias<2>(hid, BIAS(0), ln);
      unpack(x, cur, 0);
      gemm_fw<SPLIT, 2, 2, false>(WSLAB(0), 65, x, hid, ln);
      unpack(x, cur, 1);
      gemm_fw<SPLIT, 2, 2, false>(WSLAB(1), 65, x, hid, ln);
    }
    uint32_t* mw = a.maskbits ? a.maskbits + (size_t)tile * 5 * 64 + ln : nullptr;
    if (a.stash) act_store<HID, 2, false>(hid, a.stash + 0 * PS, 0, p0, npts, ln);
    if (mw) mw[0 * 64] = pack_positive<2>(hid);
    // pos head: dx = P2 relu(P1 relu(hidden) + pb1) + pb2
    acc_bias<2>(act, BIAS(1), ln);
    gemm_fw<SPLIT, 2, 2, true>(WSLAB(2), 65, hid, act, ln);
    relu_inplace<2>(act);
    if (a.stash) act_store<HID, 2, false>(act, a.stash + 1 * PS, 0, p0, npts, ln);
    if (mw) mw[1 * 64] = pack_positive<2>(act);
    if constexpr (SPLIT) {
      acc_bias<1>(o, BIAS(3), ln);
      gemm_fw<SPLIT, 1, 2, false>(WSLAB(4), 33, act, o, ln);
      act_store3(o, a.dx, p0, npts, ln);
    } else {
      float o3[3];
      head3_fw(WSLAB(4), BIAS(3), act, o3, ln);
      store3(o3, a.dx, p0, npts, ln);
    }
    // shs head: dshs = S2 relu(S1 relu(hidden) + sb1) + sb2
    acc_bias<2>(act, BIAS(2), ln);
    gemm_fw<SPLIT, 2, 2, true>(WSLAB(3), 65, hid, act, ln);
    relu_inplace<2>(act);
    if (a.stash) act_store<HID, 2, false>(act, a.stash + 2 * PS, 0, p0, npts, ln);
    if (mw) mw[2 * 64] = pack_positive<2>(act);
    acc_bias<2>(acc, BIAS(4), ln);
    gemm_fw<SPLIT, 2, 2, false>(WSLAB(5), 65, act, acc, ln);
    act_store<48, 2, false, 48>(acc, a.dshs, 0, p0, npts, ln);
    if (a.feat != nullptr) {  // inference renders that do not draw the feature image skip the head (31 % of the MFMAs)
    // dino head: feat = D2 relu(D1 relu(D0 hidden + db0) + db1) + db2   (input is the raw hidden, deformation.py:126)
    acc_bias<2>(act, BIAS(5), ln);
    gemm_fw<SPLIT, 2, 2, false>(WSLAB(6), 65, hid, act, ln);
    relu_inplace<2>(act);
    if (a.stash) act_store<HID, 2, false>(act, a.stash + 3 * PS, 0, p0, npts, ln);
    if (mw) mw[3 * 64] = pack_positive<2>(act);
    acc_bias<2>(acc, BIAS(6), ln);
    gemm_fw<SPLIT, 2, 2, false>(WSLAB(7), 65, act, acc, ln);
    relu_inplace<2>(acc);
    if (a.stash) act_store<HID, 2, false>(acc, a.stash + 4 * PS, 0, p0, npts, ln);
    if (mw) mw[4 * 64] = pack_positive<2>(acc);
    if constexpr (SPLIT) {
      acc_bias<1>(o, BIAS(7), ln);
      gemm_fw<SPLIT, 1, 2, false>(WSLAB(8), 33, acc, o, ln);
      act_store3(o, a.feat, p0, npts, ln);
    } else {
      float o3[3];
      head3_fw(WSLAB(8), BIAS(7), acc, o3, ln);
      store3(o3, a.feat, p0, npts, ln);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;   // copies at the very end: the prefetch has had the whole tile to land
  }
}

struct MlpBwdArgs {
  int P;
  const float* packed;
  const uint32_t* maskbits;  // [tiles][5][64] from the forward: hidden | pos1 | shs1 | dino1 | dino2
  const float *g_dx, *g_dshs, *g_feat;
  float *g_x, *ws;
};

// Everything the backward chain of one tile reads from memory: the three upstream gradients of the lane's point and the five
// ReLU mask words -- 45 registers, requested for the NEXT tile before the current tile's ~600 MFMAs are issued.
struct BwdIn {
  float gd[3], gf[3];
  float4 gs[6];       // g_dshs columns 8q + 4h .. +3 (q = 0..3) and 32 + 8q + 4h .. +3 (q = 0, 1): the 48 live columns
  uint32_t bits[5];
};

// Per-point backward chain, same register-resident scheme with the transposed weight reads.
template <bool SPLIT>
__global__ void __launch_bounds__(NWAVE * 64) mlp_backward_kernel(const MlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_weights(lds, a.packed, wave, lane);
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;
  const int j = lane & 31, h = lane >> 5;
  const bool dino = a.g_feat != nullptr;
  // Raw, select-free loads with clamped addresses (lanes past the end of the array re-read the last point: their columns are
  // never stored): a bounds select on a loaded value would be scheduled where the load was issued and stall there.
  auto issue = [&](BwdIn& I, int tile) {
    const size_t p = (size_t)min(tile * MT + j, a.P - 1);
    const uint32_t* mw = a.maskbits + (size_t)tile * 5 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 5; k++) I.bits[k] = mw[k * 64];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      I.gd[k] = a.g_dx[p * 3 + k];
      I.gf[k] = dino ? a.g_feat[p * 3 + k] : 0.f;
    }
    const float* row = a.g_dshs + p * 48 + 4 * h;
#pragma unroll
    for (int c = 0; c < 6; c++) I.gs[c] = *reinterpret_cast<const float4*>(row + 8 * c);
  };
  auto head3 = [&](f32x16 (&g3)[1], const float (&v)[3]) {  // features 0..2 live in registers 0..2 of the h = 0 lanes
    acc_zero<1>(g3);
#pragma unroll
    for (int k = 0; k < 3; k++) g3[0][k] = h == 0 ? v[k] : 0.f;
  };
  const int stride = gridDim.x * NWAVE, t0 = blockIdx.x * NWAVE + wave;
  BwdIn cur, nxt;
  if (t0 < ntiles) issue(cur, t0);
  for (int tile = t0; tile < ntiles; tile += stride) {
    issue(nxt, min(tile + stride, ntiles - 1));   // unconditional (clamped): see mlp_wgrad_kernel
    __builtin_amdgcn_sched_barrier(0);
    const int p0 = tile * MT, npts = min(MT, a.P - p0);
    f32x16 ghid[2], g[2], acc[2], g3[1];
    acc_zero<2>(ghid);
    // ---- dino head (skipped when the feature image has no gradient: g_feat == NULL) ----
    if (dino) {
      head3(g3, cur.gf);
      acc_zero<2>(acc);
      gemm_reg_t<2, 1, 3>(WSLAB(8), 33, g3, acc, lane);              // D2^T g_feat (3 live K steps: exact fp32 MFMAs in both arithmetics)
      masked_bits<2, false>(g, acc, cur.bits[4]);                  // gradient wrt dino2 pre-activation
      act_store<HID, 2, false>(g, a.ws + 0 * PS, 0, p0, npts, lane);
      acc_zero<2>(acc);
      gemm_bw<SPLIT, 2, 2>(WSLAB(7), 65, g, acc, lane);               // D1^T
      masked_bits<2, false>(g, acc, cur.bits[3]);
      act_store<HID, 2, false>(g, a.ws + 1 * PS, 0, p0, npts, lane);
      gemm_bw<SPLIT, 2, 2>(WSLAB(6), 65, g, ghid, lane);              // ghid = D0^T (dino input is the raw hidden: no mask)
    }
    // ---- pos head ----
    head3(g3, cur.gd);
    acc_zero<2>(acc);
    gemm_reg_t<2, 1, 3>(WSLAB(4), 33, g3, acc, lane);                // P2^T g_dx (exact fp32 MFMAs in both arithmetics)
    masked_bits<2, false>(g, acc, cur.bits[1]);
    act_store<HID, 2, false>(g, a.ws + 2 * PS, 0, p0, npts, lane);
    acc_zero<2>(acc);
    gemm_bw<SPLIT, 2, 2>(WSLAB(2), 65, g, acc, lane);                 // P1^T
    // ---- shs head ----
    {
      f32x16 gs[2], t[2];
#pragma unroll
      for (int c = 0; c < 8; c++) {  // chunk c = columns 8c + 4h .. +3; chunks 6, 7 (columns >= 48) do not exist
        const float4 v = c < 6 ? cur.gs[c < 6 ? c : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        gs[c >> 2][4 * (c & 3) + 0] = v.x; gs[c >> 2][4 * (c & 3) + 1] = v.y;
        gs[c >> 2][4 * (c & 3) + 2] = v.z; gs[c >> 2][4 * (c & 3) + 3] = v.w;
      }
      acc_zero<2>(t);
      gemm_bw<SPLIT, 2, 2>(WSLAB(5), 65, gs, t, lane);                // S2^T g_dshs (rows 48..63 of the image are zero)
      masked_bits<2, false>(g, t, cur.bits[2]);
    }
    act_store<HID, 2, false>(g, a.ws + 3 * PS, 0, p0, npts, lane);
    gemm_bw<SPLIT, 2, 2>(WSLAB(3), 65, g, acc, lane);                 // + S1^T  (same relu(hidden) mask as P1^T)
    masked_bits<2, true>(ghid, acc, cur.bits[0]);
    act_store<HID, 2, false>(ghid, a.ws + 4 * PS, 0, p0, npts, lane);
    // ---- feature_out: g_x[:, half] = W0[:, half]^T ghid ----
    acc_zero<2>(acc);
    gemm_bw<SPLIT, 2, 2>(WSLAB(0), 65, ghid, acc, lane);
    act_store<FEAT, 2, false>(acc, a.g_x, 0, p0, npts, lane);
    acc_zero<2>(acc);
    gemm_bw<SPLIT, 2, 2>(WSLAB(1), 65, ghid, acc, lane);
    act_store<FEAT, 2, false>(acc, a.g_x, 64, p0, npts, lane);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}
#undef WSLAB
#undef BIAS

// The split weight image of the inference network (32-bit words; a word = two bf16).  A FRAGMENT is the A operand of one
// (32-row block mbo, K step ks): 64 lanes x 16 bytes per piece, stored piece after piece in lane order -- one conflict-free
// ds_read_b128 per piece and lane.  Rows a layer does not have are not stored: the lanes of those rows read some stored row
// instead, and the accumulator rows they produce are never written out (S2 rows 48..63, P2 rows 3..31).
// P1 stays fp32 (split on the fly by the lanes that use it: 44 VALU instructions per fragment) -- all five layers pre-split
// would need 119 KB beside the 48 KB of staging tiles.
namespace spw {
constexpr int FRAG = 256;                            // words per piece of a full fragment
constexpr int W0 = 0;                                // [mbo 2][ks 8][piece 3][FRAG]
constexpr int S1 = W0 + 2 * 8 * 3 * FRAG;            // [mbo 2][ks 4][piece 3][FRAG]
constexpr int S2A = S1 + 2 * 4 * 3 * FRAG;           // [ks 4][piece 3][FRAG]        rows 0..31
constexpr int S2B = S2A + 4 * 3 * FRAG;              // [ks 4][piece 3][FRAG / 2]    rows 32..47: slot = 16 * h + (row & 15)
constexpr int P2 = S2B + 4 * 3 * (FRAG / 2);         // [ks 4][piece 3][h 2][row 3][4 words]
constexpr int P1LD = 68;                             // fp32 [row 64][64 inputs + 4]: 16 lanes' 16-byte chunks fall in 16 distinct bank groups
constexpr int P1 = P2 + 4 * 3 * 2 * 3 * 4;
constexpr int BIAS = P1 + 64 * P1LD;                 // b0 64 | pb1 64 | sb1 64 | sb2 64 (48 used) | pb2 32 (3 used)
constexpr int B_B0 = 0, B_PB1 = 64, B_SB1 = 128, B_SB2 = 192, B_PB2 = 256, NBIAS = 288;
constexpr int WORDS = (BIAS + NBIAS + 255) / 256 * 256;   // whole 1 KiB DMA rows
static_assert(P2 % 4 == 0 && P1 % 4 == 0 && BIAS % 4 == 0, "16-byte aligned regions");
}  // namespace spw

__device__ __forceinline__ uint32_t split_word(const float* __restrict__ W, int rows, int ld, int row, int f0, int f1, int piece) {
  uint32_t p[3];
  const float a = row < rows ? W[(size_t)row * ld + f0] : 0.f, b = row < rows ? W[(size_t)row * ld + f1] : 0.f;
  split_pair(a, b, p[0], p[1], p[2]);
  return piece == 0 ? p[0] : (piece == 1 ? p[1] : p[2]);
}
__global__ void __launch_bounds__(256) mlp_pack_split_kernel(const s3g_mlp_params w, uint32_t* __restrict__ img) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= spw::WORDS) return;
  uint32_t out = 0;
  if (x < spw::S2B) {   // full fragments: W0 | S1 | S2 rows 0..31
    const float* W; int KS, ld, rows, y;
    if (x < spw::S1) { W = w.W0; KS = 8; ld = FEAT; rows = 64; y = x - spw::W0; }
    else if (x < spw::S2A) { W = w.S1; KS = 4; ld = HID; rows = 64; y = x - spw::S1; }
    else { W = w.S2; KS = 4; ld = HID; rows = 48; y = x - spw::S2A; }
    const int t = y & 3, lane = (y >> 2) & 63, piece = (y >> 8) % 3, fr = (y >> 8) / 3, ks = fr % KS, mbo = fr / KS;
    out = split_word(W, rows, ld, 32 * mbo + (lane & 31), split_feature(ks, lane >> 5, 2 * t), split_feature(ks, lane >> 5, 2 * t + 1), piece);
  } else if (x < spw::P2) {   // S2 rows 32..47
    const int y = x - spw::S2B, t = y & 3, slot = (y >> 2) & 31, piece = (y >> 7) % 3, ks = (y >> 7) / 3;
    out = split_word(w.S2, 48, HID, 32 + (slot & 15), split_feature(ks, slot >> 4, 2 * t), split_feature(ks, slot >> 4, 2 * t + 1), piece);
  } else if (x < spw::P1) {   // P2: three rows
    const int y = x - spw::P2, t = y & 3, q = y >> 2, row = q % 3, h = (q / 3) & 1, piece = (q / 6) % 3, ks = q / 18;
    out = split_word(w.P2, 3, HID, row, split_feature(ks, h, 2 * t), split_feature(ks, h, 2 * t + 1), piece);
  } else if (x < spw::BIAS) {   // P1 as it is, rows padded
    const int y = x - spw::P1, row = y / spw::P1LD, f = y % spw::P1LD;
    out = f < HID ? __float_as_uint(w.P1[row * HID + f]) : 0u;
  } else if (x < spw::BIAS + spw::NBIAS) {
    const int y = x - spw::BIAS;
    const float v = y < 64 ? w.b0[y] : y < 128 ? w.pb1[y - 64] : y < 192 ? w.sb1[y - 128] : y < 256 ? (y - 192 < 48 ? w.sb2[y - 192] : 0.f)
                                                                                                  : (y - 256 < 3 ? w.pb2[y - 256] : 0.f);
    out = __float_as_uint(v);
  }
  img[x] = out;
}
// acc[mbo] += (rows 32*mbo .. +31 of the layer) x B for the K steps ks0 .. ks0 + 2*MBI - 1, fragments at frag + ((mbo*KS + ks)*3 + piece)*FRAG
template <int MBO, int MBI>
__device__ __forceinline__ void gemm_split(const uint32_t* frag, int KS, int ks0, const ActSplit<MBI>& B, f32x16 (&acc)[MBO], int lane) {
#pragma unroll
  for (int mbi = 0; mbi < MBI; mbi++)
#pragma unroll
    for (int s = 0; s < 2; s++) {
      Split8 a[MBO];
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++)
#pragma unroll
        for (int pc = 0; pc < 3; pc++)
          a[mbo].p[pc] = *reinterpret_cast<const u32x4*>(frag + ((mbo * KS + ks0 + 2 * mbi + s) * 3 + pc) * spw::FRAG + lane * 4);
#pragma unroll
      for (int mbo = 0; mbo < MBO; mbo++) acc[mbo] = mfma_split(acc[mbo], a[mbo], B.b[mbi][s]);
    }
}

// ---- training forward on the bf16 matrix pipe with the weights split ONCE (round 5; S3G_MLP_BF16X3) ------------------------------
// mlp_forward_kernel<true> splits every weight fragment on the fly, in every wave, for every 32-point tile: 44 VALU instructions per
// fragment, 64 fragments per tile -- the kernel is VALU-bound and gains 15 % where the instruction rates promise 2.7 x.
// Here the A operands of all layers but P1 are pre-split by mlp_pack_presplit_fwd_kernel into fragment order (three pieces x 64 lanes
// x 16 bytes: one conflict-free ds_read_b128 per piece and lane) and stay in LDS; P1 stays fp32 and is split by the lanes that read
// it (everything pre-split would need 168 KB; this image is 159 KiB of the 160).  Same pieces, same MFMA order as the on-the-fly
// kernel: outputs, stash and mask words are BIT-IDENTICAL to mlp_forward_kernel<true> (tests/test_mlp_gpu.py), which stays in the
// tree as the checker of this one (S3G_MLP_BF16X3_ONTHEFLY).
namespace tpw {   // 32-bit words
constexpr int FRAG = 256;
constexpr int W0 = 0;                                 // [mbo 2][ks 8][piece 3][FRAG]
constexpr int S1 = W0 + 2 * 8 * 3 * FRAG;             // [mbo 2][ks 4][piece 3][FRAG]
constexpr int D0 = S1 + 2 * 4 * 3 * FRAG;
constexpr int D1 = D0 + 2 * 4 * 3 * FRAG;
constexpr int S2A = D1 + 2 * 4 * 3 * FRAG;            // [ks 4][piece 3][FRAG]        rows 0..31
constexpr int S2B = S2A + 4 * 3 * FRAG;               // [ks 4][piece 3][FRAG / 2]    rows 32..47: slot = 16 * h + (row & 15)
constexpr int P2 = S2B + 4 * 3 * (FRAG / 2);          // [ks 4][piece 3][h 2][row 3][4 words]
constexpr int D2 = P2 + 4 * 3 * 2 * 3 * 4;
constexpr int P1LD = 68;
constexpr int P1 = D2 + 4 * 3 * 2 * 3 * 4;            // fp32 [row 64][64 inputs + 4]
constexpr int BIAS = P1 + 64 * P1LD;
constexpr int B_B0 = 0, B_PB1 = 64, B_SB1 = 128, B_SB2 = 192, B_DB0 = 256, B_DB1 = 320, B_PB2 = 384, B_DB2 = 416, NBIAS = 448;   // the two 3-row biases zero padded to 32
constexpr int WORDS = BIAS + NBIAS;
static_assert(WORDS % 256 == 0 && WORDS * 4 <= 160 * 1024, "whole 1 KiB DMA rows, inside the CU's LDS");
static_assert(P2 % 4 == 0 && D2 % 4 == 0 && P1 % 4 == 0 && BIAS % 4 == 0, "16-byte aligned regions");
}  // namespace tpw

// The backward's image: the TRANSPOSED layers in fragment order -- A[row = input feature][k = output feature] -- for W0 (four
// 32-row blocks: the 128 inputs), D1, D0, S1 and S2 (K = 48: three K steps); P1^T stays fp32 [64 in][64 out + 4] and is split by the
// lanes that read it; the two 3-row heads (K = 3) run on the exact fp32 MFMA from a compact fp32 [64 in][8] image (columns 0..2 = the
// three output rows, 3..7 zero: what the h = 1 lanes read).  159 KiB like the forward's.
namespace tbw {   // 32-bit words
constexpr int FRAG = 256;
constexpr int W0T = 0;                                // [mbo 4][ks 4][piece 3][FRAG]
constexpr int D1T = W0T + 4 * 4 * 3 * FRAG;           // [mbo 2][ks 4][piece 3][FRAG]
constexpr int D0T = D1T + 2 * 4 * 3 * FRAG;
constexpr int S1T = D0T + 2 * 4 * 3 * FRAG;
constexpr int S2T = S1T + 2 * 4 * 3 * FRAG;           // [mbo 2][ks 3][piece 3][FRAG]   (48 output features = 3 K steps)
constexpr int P1LD = 68;
constexpr int P1T = S2T + 2 * 3 * 3 * FRAG;           // fp32 [in 64][64 outputs + 4]
constexpr int H3LD = 8;
constexpr int P2T = P1T + 64 * P1LD;                  // fp32 [in 64][8]
constexpr int D2T = P2T + 64 * H3LD;
constexpr int WORDS = D2T + 64 * H3LD;
static_assert(WORDS % 256 == 0 && WORDS * 4 <= 160 * 1024, "whole 1 KiB DMA rows, inside the CU's LDS");
static_assert(P1T % 4 == 0 && P2T % 4 == 0, "16-byte aligned regions");
}  // namespace tbw
constexpr int PACK_TOTAL = PACK_FLOATS + tpw::WORDS + tbw::WORDS;   // floats in front of the activation stash

__global__ void __launch_bounds__(256) mlp_pack_presplit_fwd_kernel(const s3g_mlp_params w, uint32_t* __restrict__ img) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= tpw::WORDS) return;
  uint32_t out = 0;
  if (x < tpw::S2B) {   // full fragments
    const float* W; int KS, ld, rows, y;
    if (x < tpw::S1) { W = w.W0; KS = 8; ld = FEAT; rows = 64; y = x - tpw::W0; }
    else if (x < tpw::D0) { W = w.S1; KS = 4; ld = HID; rows = 64; y = x - tpw::S1; }
    else if (x < tpw::D1) { W = w.D0; KS = 4; ld = HID; rows = 64; y = x - tpw::D0; }
    else if (x < tpw::S2A) { W = w.D1; KS = 4; ld = HID; rows = 64; y = x - tpw::D1; }
    else { W = w.S2; KS = 4; ld = HID; rows = 48; y = x - tpw::S2A; }
    const int t = y & 3, lane = (y >> 2) & 63, piece = (y >> 8) % 3, fr = (y >> 8) / 3, ks = fr % KS, mbo = fr / KS;
    out = split_word(W, rows, ld, 32 * mbo + (lane & 31), split_feature(ks, lane >> 5, 2 * t), split_feature(ks, lane >> 5, 2 * t + 1), piece);
  } else if (x < tpw::P2) {   // S2 rows 32..47
    const int y = x - tpw::S2B, t = y & 3, slot = (y >> 2) & 31, piece = (y >> 7) % 3, ks = (y >> 7) / 3;
    out = split_word(w.S2, 48, HID, 32 + (slot & 15), split_feature(ks, slot >> 4, 2 * t), split_feature(ks, slot >> 4, 2 * t + 1), piece);
  } else if (x < tpw::P1) {   // the two 3-row heads
    const bool dino = x >= tpw::D2;
    const int y = x - (dino ? tpw::D2 : tpw::P2), t = y & 3, q = y >> 2, row = q % 3, h = (q / 3) & 1, piece = (q / 6) % 3, ks = q / 18;
    out = split_word(dino ? w.D2 : w.P2, 3, HID, row, split_feature(ks, h, 2 * t), split_feature(ks, h, 2 * t + 1), piece);
  } else if (x < tpw::BIAS) {   // P1 as it is, rows padded
    const int y = x - tpw::P1, row = y / tpw::P1LD, f = y % tpw::P1LD;
    out = f < HID ? __float_as_uint(w.P1[row * HID + f]) : 0u;
  } else {
    const int y = x - tpw::BIAS;
    float v = 0.f;
    if (y < 64) v = w.b0[y];
    else if (y < 128) v = w.pb1[y - 64];
    else if (y < 192) v = w.sb1[y - 128];
    else if (y < 256) v = y - 192 < 48 ? w.sb2[y - 192] : 0.f;
    else if (y < 320) v = w.db0[y - 256];
    else if (y < 384) v = w.db1[y - 320];
    else if (y < 416) v = y - 384 < 3 ? w.pb2[y - 384] : 0.f;
    else v = y - 416 < 3 ? w.db2[y - 416] : 0.f;
    out = __float_as_uint(v);
  }
  img[x] = out;
}

// word t of piece `piece` of the transposed pair (W[o0][in], W[o1][in]); output features >= outs are zero
__device__ __forceinline__ uint32_t split_word_t(const float* __restrict__ W, int outs, int ld, int in, int o0, int o1, int piece) {
  uint32_t p[3];
  const float a = o0 < outs ? W[(size_t)o0 * ld + in] : 0.f, b = o1 < outs ? W[(size_t)o1 * ld + in] : 0.f;
  split_pair(a, b, p[0], p[1], p[2]);
  return piece == 0 ? p[0] : (piece == 1 ? p[1] : p[2]);
}
__global__ void __launch_bounds__(256) mlp_pack_presplit_bwd_kernel(const s3g_mlp_params w, uint32_t* __restrict__ img) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= tbw::WORDS) return;
  uint32_t out = 0;
  if (x < tbw::P1T) {   // full fragments of W^T: row = input feature, k = output feature
    const float* W; int KS, ld, outs, y;
    if (x < tbw::D1T) { W = w.W0; KS = 4; ld = FEAT; outs = 64; y = x - tbw::W0T; }
    else if (x < tbw::D0T) { W = w.D1; KS = 4; ld = HID; outs = 64; y = x - tbw::D1T; }
    else if (x < tbw::S1T) { W = w.D0; KS = 4; ld = HID; outs = 64; y = x - tbw::D0T; }
    else if (x < tbw::S2T) { W = w.S1; KS = 4; ld = HID; outs = 64; y = x - tbw::S1T; }
    else { W = w.S2; KS = 3; ld = HID; outs = 48; y = x - tbw::S2T; }
    const int t = y & 3, lane = (y >> 2) & 63, piece = (y >> 8) % 3, fr = (y >> 8) / 3, ks = fr % KS, mbo = fr / KS;
    out = split_word_t(W, outs, ld, 32 * mbo + (lane & 31), split_feature(ks, lane >> 5, 2 * t), split_feature(ks, lane >> 5, 2 * t + 1), piece);
  } else if (x < tbw::P2T) {   // P1^T as fp32 rows
    const int y = x - tbw::P1T, in = y / tbw::P1LD, o = y % tbw::P1LD;
    out = o < HID ? __float_as_uint(w.P1[o * HID + in]) : 0u;
  } else {   // the 3-row heads, transposed: [in][8]
    const bool dino = x >= tbw::D2T;
    const int y = x - (dino ? tbw::D2T : tbw::P2T), in = y / tbw::H3LD, o = y % tbw::H3LD;
    out = o < 3 ? __float_as_uint((dino ? w.D2 : w.P2)[o * HID + in]) : 0u;
  }
  img[x] = out;
}

// Per-point backward chain on the pre-split image: mlp_backward_kernel<true> with every fragment split ONCE (bit-identical to it).
__global__ void __launch_bounds__(NWAVE * 64) mlp_backward_presplit_kernel(const MlpBwdArgs a) {   // a.packed = the tbw image
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const uint32_t* wsplit = reinterpret_cast<const uint32_t*>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < tbw::WORDS / 256; c += NWAVE)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.packed + c * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;
  const int j = lane & 31, h = lane >> 5;
  const bool dino = a.g_feat != nullptr;
  auto issue = [&](BwdIn& I, int tile) {
    const size_t p = (size_t)min(tile * MT + j, a.P - 1);
    const uint32_t* mw = a.maskbits + (size_t)tile * 5 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 5; k++) I.bits[k] = mw[k * 64];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      I.gd[k] = a.g_dx[p * 3 + k];
      I.gf[k] = dino ? a.g_feat[p * 3 + k] : 0.f;
    }
    const float* row = a.g_dshs + p * 48 + 4 * h;
#pragma unroll
    for (int c = 0; c < 6; c++) I.gs[c] = *reinterpret_cast<const float4*>(row + 8 * c);
  };
  auto head3 = [&](f32x16 (&g3)[1], const float (&v)[3]) {
    acc_zero<1>(g3);
#pragma unroll
    for (int k = 0; k < 3; k++) g3[0][k] = h == 0 ? v[k] : 0.f;
  };
  const int stride = gridDim.x * NWAVE, t0 = blockIdx.x * NWAVE + wave;
  BwdIn cur, nxt;
  if (t0 < ntiles) issue(cur, t0);
  for (int tile = t0; tile < ntiles; tile += stride) {
    issue(nxt, min(tile + stride, ntiles - 1));
    __builtin_amdgcn_sched_barrier(0);
    const int p0 = tile * MT, npts = min(MT, a.P - p0);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    f32x16 ghid[2], g[2], acc[2], g3[1];
    ActSplit<2> gsp;
    acc_zero<2>(ghid);
    if (dino) {
      head3(g3, cur.gf);
      acc_zero<2>(acc);
      gemm_reg_t<2, 1, 3>(lds + tbw::D2T, tbw::H3LD, g3, acc, ln);          // D2^T g_feat: exact fp32 MFMAs (3 live K steps)
      masked_bits<2, false>(g, acc, cur.bits[4]);
      act_store<HID, 2, false>(g, a.ws + 0 * PS, 0, p0, npts, ln);
      acc_zero<2>(acc);
      act_split<2, false>(gsp, g);
      gemm_split<2, 2>(wsplit + tbw::D1T, 4, 0, gsp, acc, ln);              // D1^T
      masked_bits<2, false>(g, acc, cur.bits[3]);
      act_store<HID, 2, false>(g, a.ws + 1 * PS, 0, p0, npts, ln);
      act_split<2, false>(gsp, g);
      gemm_split<2, 2>(wsplit + tbw::D0T, 4, 0, gsp, ghid, ln);             // ghid = D0^T (dino input is the raw hidden: no mask)
    }
    // ---- pos head ----
    head3(g3, cur.gd);
    acc_zero<2>(acc);
    gemm_reg_t<2, 1, 3>(lds + tbw::P2T, tbw::H3LD, g3, acc, ln);            // P2^T g_dx
    masked_bits<2, false>(g, acc, cur.bits[1]);
    act_store<HID, 2, false>(g, a.ws + 2 * PS, 0, p0, npts, ln);
    acc_zero<2>(acc);
    act_split<2, false>(gsp, g);
#pragma unroll
    for (int ks = 0; ks < 4; ks++)      // P1^T: fp32 rows in LDS, split by the lanes that read them (same element order as gemm_reg_t_split)
#pragma unroll
      for (int mbo = 0; mbo < 2; mbo++) {
        const float* wr = lds + tbw::P1T + (32 * mbo + (ln & 31)) * tbw::P1LD + 4 * (ln >> 5) + 16 * ks;
        const float4 lo = *reinterpret_cast<const float4*>(wr);
        const float4 hi = *reinterpret_cast<const float4*>(wr + 8);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        acc[mbo] = mfma_split(acc[mbo], split8(v), gsp.b[ks >> 1][ks & 1]);
      }
    // ---- shs head ----
    {
      f32x16 gs[2], t[2];
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float4 v = c < 6 ? cur.gs[c < 6 ? c : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        gs[c >> 2][4 * (c & 3) + 0] = v.x; gs[c >> 2][4 * (c & 3) + 1] = v.y;
        gs[c >> 2][4 * (c & 3) + 2] = v.z; gs[c >> 2][4 * (c & 3) + 3] = v.w;
      }
      acc_zero<2>(t);
      ActSplit<2> ssp;
      act_split<2, false>(ssp, gs);
#pragma unroll
      for (int ks = 0; ks < 3; ks++) {    // S2^T g_dshs: K = 48 (the on-the-fly kernel's fourth K step multiplies zeros by zeros)
        Split8 aw[2];
#pragma unroll
        for (int mbo = 0; mbo < 2; mbo++)
#pragma unroll
          for (int pc = 0; pc < 3; pc++)
            aw[mbo].p[pc] = *reinterpret_cast<const u32x4*>(wsplit + tbw::S2T + ((mbo * 3 + ks) * 3 + pc) * tbw::FRAG + ln * 4);
#pragma unroll
        for (int mbo = 0; mbo < 2; mbo++) t[mbo] = mfma_split(t[mbo], aw[mbo], ssp.b[ks >> 1][ks & 1]);
      }
      masked_bits<2, false>(g, t, cur.bits[2]);
    }
    act_store<HID, 2, false>(g, a.ws + 3 * PS, 0, p0, npts, ln);
    act_split<2, false>(gsp, g);
    gemm_split<2, 2>(wsplit + tbw::S1T, 4, 0, gsp, acc, ln);                // + S1^T  (same relu(hidden) mask as P1^T)
    masked_bits<2, true>(ghid, acc, cur.bits[0]);
    act_store<HID, 2, false>(ghid, a.ws + 4 * PS, 0, p0, npts, ln);
    // ---- feature_out: g_x[:, half] = W0[:, half]^T ghid ----
    act_split<2, false>(gsp, ghid);
    acc_zero<2>(acc);
    gemm_split<2, 2>(wsplit + tbw::W0T, 4, 0, gsp, acc, ln);
    act_store<FEAT, 2, false>(acc, a.g_x, 0, p0, npts, ln);
    acc_zero<2>(acc);
    gemm_split<2, 2>(wsplit + tbw::W0T + 2 * 4 * 3 * tbw::FRAG, 4, 0, gsp, acc, ln);
    act_store<FEAT, 2, false>(acc, a.g_x, 64, p0, npts, ln);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}

__global__ void __launch_bounds__(NWAVE * 64) mlp_forward_presplit_kernel(const MlpFwdArgs a) {   // a.packed = the tpw image
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const uint32_t* wsplit = reinterpret_cast<const uint32_t*>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < tpw::WORDS / 256; c += NWAVE)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.packed + c * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto bias = [&](int off) { return lds + tpw::BIAS + off; };
  const int ntiles = (a.P + MT - 1) / MT;
  const size_t PS = (size_t)a.P * HID;
  struct XIn { float4 v[16]; };
  const int jj = lane & 31, hh = lane >> 5;
  auto issue = [&](XIn& X, int tile) {
    const float* row = a.x + (size_t)min(tile * MT + jj, a.P - 1) * FEAT + 4 * hh;
#pragma unroll
    for (int c = 0; c < 16; c++) X.v[c] = *reinterpret_cast<const float4*>(row + 8 * c);
  };
  auto unpack = [&](f32x16 (&x)[2], const XIn& X, int half) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const float4 v = X.v[8 * half + c];
      x[c >> 2][4 * (c & 3) + 0] = v.x; x[c >> 2][4 * (c & 3) + 1] = v.y;
      x[c >> 2][4 * (c & 3) + 2] = v.z; x[c >> 2][4 * (c & 3) + 3] = v.w;
    }
  };
  // the two 3-row heads: rows 0..2 are stored; the other lanes read row 0 (their accumulator rows are never written out)
  auto head3 = [&](int region, const ActSplit<2>& B, f32x16 (&o)[1], int ln) {
    const int j3 = (ln & 31) < 3 ? (ln & 31) : 0, h3 = ln >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      Split8 w;
#pragma unroll
      for (int pc = 0; pc < 3; pc++)
        w.p[pc] = *reinterpret_cast<const u32x4*>(wsplit + region + ((((ks * 3 + pc) * 2 + h3) * 3 + j3) << 2));
      o[0] = mfma_split(o[0], w, B.b[ks >> 1][ks & 1]);
    }
  };
  const int stride = gridDim.x * NWAVE, t0 = blockIdx.x * NWAVE + wave;
  // ONE input buffer: the 128 features of a tile are consumed by the first GEMM (split into `as`), after which their registers are free
  // again -- the NEXT tile's rows are requested right there and have the remaining three quarters of the tile's work to arrive
  // (the exact kernel keeps two buffers across the whole tile: 64 more live registers, which this kernel does not have)
  XIn cur;
  if (t0 < ntiles) issue(cur, t0);
  for (int tile = t0; tile < ntiles; tile += stride) {
    const int p0 = tile * MT, npts = min(MT, a.P - p0);
    f32x16 hid[2], act[2], acc[2], o[1];
    int ln = lane;   // per-tile copy: LDS / output addresses are re-derived instead of being carried (and spilled) across the loop
    asm volatile("" : "+v"(ln));
    ActSplit<2> hs, as;
    {  // hidden = W0 x + b0, K = 128 in two halves
      f32x16 x[2];
      acc_bias<2>(hid, bias(tpw::B_B0), ln);
      unpack(x, cur, 0);
      act_split<2, false>(as, x);
      gemm_split<2, 2>(wsplit + tpw::W0, 8, 0, as, hid, ln);
      unpack(x, cur, 1);
      act_split<2, false>(as, x);
      __builtin_amdgcn_sched_barrier(0);
      issue(cur, min(tile + stride, ntiles - 1));   // unconditional (clamped): see mlp_wgrad_kernel
      __builtin_amdgcn_sched_barrier(0);
      gemm_split<2, 2>(wsplit + tpw::W0, 8, 4, as, hid, ln);
    }
    uint32_t* mw = a.maskbits ? a.maskbits + (size_t)tile * 5 * 64 + ln : nullptr;
    if (a.stash) act_store<HID, 2, false>(hid, a.stash + 0 * PS, 0, p0, npts, ln);
    if (mw) mw[0 * 64] = pack_positive<2>(hid);
    act_split<2, true>(hs, hid);   // relu(hidden): the input of the position and SH heads
    // pos head: dx = P2 relu(P1 relu(hidden) + pb1) + pb2.  P1 is fp32 in LDS: a lane's eight weights of a fragment are two 16-byte chunks of its row
    acc_bias<2>(act, bias(tpw::B_PB1), ln);
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int mbo = 0; mbo < 2; mbo++) {
        const float* wr = lds + tpw::P1 + (32 * mbo + (ln & 31)) * tpw::P1LD + 4 * (ln >> 5) + 16 * ks;   // inputs 16 ks + 4 h + {0..3, 8..11}
        const float4 lo = *reinterpret_cast<const float4*>(wr);
        const float4 hi = *reinterpret_cast<const float4*>(wr + 8);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        act[mbo] = mfma_split(act[mbo], split8(v), hs.b[ks >> 1][ks & 1]);
      }
    relu_inplace<2>(act);
    if (a.stash) act_store<HID, 2, false>(act, a.stash + 1 * PS, 0, p0, npts, ln);
    if (mw) mw[1 * 64] = pack_positive<2>(act);
    act_split<2, false>(as, act);
    acc_bias<1>(o, bias(tpw::B_PB2), ln);
    head3(tpw::P2, as, o, ln);
    act_store3(o, a.dx, p0, npts, ln);
    // shs head: dshs = S2 relu(S1 relu(hidden) + sb1) + sb2
    acc_bias<2>(act, bias(tpw::B_SB1), ln);
    gemm_split<2, 2>(wsplit + tpw::S1, 4, 0, hs, act, ln);
    relu_inplace<2>(act);
    if (a.stash) act_store<HID, 2, false>(act, a.stash + 2 * PS, 0, p0, npts, ln);
    if (mw) mw[2 * 64] = pack_positive<2>(act);
    act_split<2, false>(as, act);
    acc_bias<2>(acc, bias(tpw::B_SB2), ln);
    gemm_split<1, 2>(wsplit + tpw::S2A, 4, 0, as, *reinterpret_cast<f32x16(*)[1]>(&acc[0]), ln);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {   // rows 32..47: lanes of rows 48..63 read rows 32..47 again (never written out)
      Split8 w;
#pragma unroll
      for (int pc = 0; pc < 3; pc++)
        w.p[pc] = *reinterpret_cast<const u32x4*>(wsplit + tpw::S2B + (ks * 3 + pc) * (tpw::FRAG / 2) + ((16 * (ln >> 5) + (ln & 15)) << 2));
      acc[1] = mfma_split(acc[1], w, as.b[ks >> 1][ks & 1]);
    }
    act_store<48, 2, false, 48>(acc, a.dshs, 0, p0, npts, ln);
    if (a.feat != nullptr) {
      // dino head: feat = D2 relu(D1 relu(D0 hidden + db0) + db1) + db2   (input is the RAW hidden, deformation.py:126)
      act_split<2, false>(hs, hid);
      acc_bias<2>(act, bias(tpw::B_DB0), ln);
      gemm_split<2, 2>(wsplit + tpw::D0, 4, 0, hs, act, ln);
      relu_inplace<2>(act);
      if (a.stash) act_store<HID, 2, false>(act, a.stash + 3 * PS, 0, p0, npts, ln);
      if (mw) mw[3 * 64] = pack_positive<2>(act);
      act_split<2, false>(as, act);
      acc_bias<2>(acc, bias(tpw::B_DB1), ln);
      gemm_split<2, 2>(wsplit + tpw::D1, 4, 0, as, acc, ln);
      relu_inplace<2>(acc);
      if (a.stash) act_store<HID, 2, false>(acc, a.stash + 4 * PS, 0, p0, npts, ln);
      if (mw) mw[4 * 64] = pack_positive<2>(acc);
      act_split<2, false>(as, acc);
      acc_bias<1>(o, bias(tpw::B_DB2), ln);
      head3(tpw::D2, as, o, ln);
      act_store3(o, a.feat, p0, npts, ln);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- inference: HexPlane sampler (+) MLP heads in ONE kernel (render(): gaussian_renderer/__init__.py:82-97) --------
// Under no_grad nothing is stashed and the feature (dino) head is not needed, so the weight image shrinks to the first six slabs
// (W0 | W0 | P1 | S1 | P2 | S2 = 104 KB) and 46 KB of LDS are left: each wave gets a 32-point x 32-channel staging tile and one
// level's tap slots.  A wave samples ONE LEVEL of its 32 points the way hexplane_forward_kernel does (8 lanes per point, four
// rounds of 8 points), writes the float4 it would have stored to HBM into the staging tile instead, re-reads it in the MFMA B
// operand layout (lane = point, registers = channels) and runs that level's quarter of the feature_out GEMM (K = 32); after the
// fourth level `hidden` is complete and the two heads follow exactly as in mlp_forward_kernel.  The [P,128] feature array -- 614 MB
// written by the sampler and read back by the MLP at cfg3 -- never exists; the sampler waves of a CU wait on texel gathers while
// its other waves keep the matrix pipe busy.  Same arithmetic in the same order as the two separate kernels (the K order of the
// feature_out GEMM is level 0..3 there too): outputs are bit-identical (tests/test_infer_gpu.py).
#ifndef S3G_INFER_STAGGER
#define S3G_INFER_STAGGER 0   // x 127 x 64 cycles (~3.9 us each) of initial delay for waves 4..7: measured without effect (r3)
#endif
#define S3G_INFER_PRIO 0   // 1: s_setprio(1) around the MFMA clusters, 2: static priority for waves 4..7
#define S3G_INFER_EXPERIMENT 0   // 1: no head GEMMs, 2: no texel loads (timing experiments only; results are wrong)
struct InferArgs {
  HexArgs h;            // sampler side: descriptor (row tables already swapped in when uniform_time), xyz, time, proc_order, P
  const float* packed;  // mlp_pack_kernel's image
  float *dx, *dshs;
};
constexpr int INF_SLABS = 6;
constexpr int INF_WFLOATS = INF_SLABS * SLAB + 8 * 64;
constexpr int STG_LD = 36;                       // floats per staged point: 32 channels + 4 (16 lanes of a ds_read_b128 hit 16 distinct bank groups)
constexpr int STG_FLOATS = MT * STG_LD;
constexpr int INF_TAP_STRIDE = TAP_SLOTS + 1;    // float4 per point: 6 taps used, padded like tap_stride()
constexpr int INF_WAVE_FLOATS = STG_FLOATS + 2 * 8 * INF_TAP_STRIDE * 4;   // staging tile + two sets of tap slots
constexpr int INF_LDS_FLOATS = INF_WFLOATS + NWAVE * INF_WAVE_FLOATS;
static_assert(INF_LDS_FLOATS * 4 <= 160 * 1024, "inference image + staging must fit the CU's LDS");
// SPLIT (three-way bf16 operands, above): image spw::WORDS, and 6 instead of 9 tap slots per point (the 8 points of a round still
// read 8 disjoint bank groups: 24 words apart)
constexpr int INF_TAP_STRIDE_SPLIT = 6;
constexpr int INF_WAVE_FLOATS_SPLIT = STG_FLOATS + 2 * 8 * INF_TAP_STRIDE_SPLIT * 4;
constexpr int INF_LDS_FLOATS_SPLIT = spw::WORDS + NWAVE * INF_WAVE_FLOATS_SPLIT;
static_assert(INF_LDS_FLOATS_SPLIT * 4 <= 160 * 1024, "split inference image + staging must fit the CU's LDS");

template <bool UT, bool SPLIT>
__global__ void __launch_bounds__(NWAVE * 64) deform_infer_kernel(const InferArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WIMG = SPLIT ? spw::WORDS : INF_WFLOATS, TAPS = SPLIT ? INF_TAP_STRIDE_SPLIT : INF_TAP_STRIDE;
  constexpr int WAVE_FLOATS = SPLIT ? INF_WAVE_FLOATS_SPLIT : INF_WAVE_FLOATS;
  if constexpr (SPLIT) {
    for (int c = wave; c < spw::WORDS / 256; c += NWAVE)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.packed + c * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
  } else {
    for (int c = wave; c < INF_SLABS * SLAB / 256; c += NWAVE)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.packed + c * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(lds + c * 256), 16, 0, 0);
    if (wave < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.packed + NSLAB * SLAB + wave * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(lds + INF_SLABS * SLAB + wave * 256), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto wslab = [&](int k) { return lds + k * SLAB; };
  auto bias = [&](int k) {   // k: b0 | pb1 | sb1 | pb2 | sb2
    if constexpr (SPLIT) return lds + spw::BIAS + (k == 0 ? spw::B_B0 : k == 1 ? spw::B_PB1 : k == 2 ? spw::B_SB1 : k == 3 ? spw::B_PB2 : spw::B_SB2);
    else return lds + INF_SLABS * SLAB + k * 64;
  };
  const uint32_t* wsplit = reinterpret_cast<const uint32_t*>(lds);   // SPLIT: the spw image
  float* stage = lds + WIMG + wave * WAVE_FLOATS;
  const int slot = lane >> 3, j8 = lane & 7, c4 = j8 * 4;   // sampler role: point slot, channel quad
  const int jj = lane & 31, hh = lane >> 5;                 // MFMA role: point column, row half
  float4* tp0 = reinterpret_cast<float4*>(stage + STG_FLOATS) + slot * TAPS;   // two sets of tap slots per point slot
  float4* tp1 = tp0 + 8 * TAPS;
  const int P = a.h.P, ntiles = (P + MT - 1) / MT;
  if (S3G_INFER_PRIO == 2 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
#if S3G_INFER_STAGGER
  // Waves w and w + 4 share a SIMD.  Started together they stay in lockstep -- both gather, then both queue on the matrix pipe --
  // and the pipe idles through every gather phase; starting the second four half a tile later lets one wave's heads run under
  // the other's texel gathers.
  if (wave >= 4)
    for (int d = 0; d < S3G_INFER_STAGGER; d++) __builtin_amdgcn_s_sleep(127);
#endif
  // texels of one (level, round): spatial planes (x,y) (x,z) (y,z) four corners each; time planes four corners, or -- uniform
  // time -- the two corners of their row tables
  constexpr int NTEX = UT ? 18 : 24;
  struct Tex { float4 v[NTEX]; };
  // normalised coordinates of the tile's 32 points live in the 4 pad floats of their staging rows
  auto taps_for = [&](int l, int rr, float4* tp) {
    const float4 uv = *reinterpret_cast<const float4*>(stage + (8 * rr + slot) * STG_LD + 32);
    const float u[4] = {uv.x, uv.y, uv.z, uv.w};
    produce_taps_level(a.h, u, j8, l, tp);
  };
  auto issue = [&](Tex& T, int l, const float4* tp) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int W = a.h.d.res[l][PAIR0[i]], H = a.h.d.res[l][PAIR1[i]];
      const float* pl = a.h.d.planes[l][i];
      if (S3G_INFER_EXPERIMENT == 2) {
        const PointTap t = read_tap<false>(tp, 0, i, W, H, c4);
        const float4 c = make_float4(t.fx, t.gx, t.fy, (float)(t.off & 1u) + (float)(size_t)pl);
        for (int q = 0; q < ((UT && IS_TIME_PLANE[i]) ? 2 : 4); q++) T.v[n++] = c;
      } else if (UT && IS_TIME_PLANE[i]) {
        const PointTap t = read_tap<true>(tp, 0, i, W, H, c4);
        T.v[n++] = texel4(pl, t.off);
        T.v[n++] = texel4(pl, t.off + t.dx);
      } else {
        const PointTap t = read_tap<false>(tp, 0, i, W, H, c4);
        T.v[n++] = texel4(pl, t.off);
        T.v[n++] = texel4(pl, t.off + t.dx);
        T.v[n++] = texel4(pl, t.off + t.dy);
        T.v[n++] = texel4(pl, t.off + t.dy + t.dx);
      }
    }
  };
  auto consume = [&](const Tex& T, int l, int rr, const float4* tp) {
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
    int n = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int W = a.h.d.res[l][PAIR0[i]], H = a.h.d.res[l][PAIR1[i]];
      float4 s;
      if (UT && IS_TIME_PLANE[i]) {
        const PointTap t = read_tap<true>(tp, 0, i, W, H, c4);
        s = T.v[n] * t.gx;
        s = s + T.v[n + 1] * t.fx;
        n += 2;
      } else {
        const PointTap t = read_tap<false>(tp, 0, i, W, H, c4);
        s = T.v[n] * (t.gx * t.gy);
        s = s + T.v[n + 1] * (t.fx * t.gy);
        s = s + T.v[n + 2] * (t.gx * t.fy);
        s = s + T.v[n + 3] * (t.fx * t.fy);
        n += 4;
      }
      prod = prod * s;
    }
    // SPLIT: the float4 stored below must not come straight out of a PACKED fp32 instruction.  hipcc forms `prod` with
    // v_pk_mul_f32 and issues ds_write_b128 a few slots later; with the other wave of the SIMD issuing v_mfma_f32_32x32x16_bf16 the
    // store then reads STALE data for the last quarter of the wave (lanes 48..63 = two points, errors of 1e-1 in two adjacent rows,
    // different rows every launch): ~800 wrong rows per launch at 1.2 M points, 110 872 over 1000 launches
    // (profiles/r04_split_hazard.jsonl, build `split_nopad`).  It never happens with one wave per SIMD, nor beside the fp32 MFMAs of
    // the exact kernel; waiting for the texel loads, the LDS queue or the wave's own MFMAs does not help.  Round 3 papered over it
    // with 16 wait states (which also happened to make hipcc form the products with plain v_mul_f32).  Round 4 isolates the cause:
    // re-writing the four registers with an ordinary single-pass VALU instruction (v_mov_b32) and NO wait state at all is enough --
    // 0 wrong rows in 1000 launches at 1.2 M points and 400 at 70 001 (build `split_vmov`, now the tree; the 16-wait-state build:
    // also 0) -- i.e. the unsafe pair is "packed-fp32 VALU result -> DS store data" while XDL ops of another wave are in flight,
    // and a real register dependency on a non-packed VALU write removes it independently of timing (forming the products with plain
    // v_mul_f32 is already enough -- build `split_scalarized`, 0 wrong rows --; the v_mov makes that independent of how hipcc chooses
    // to multiply).  ISA of the builds:
    // profiles/r04_split_hazard_isa.txt; stress test: tests/test_infer_gpu.py::test_split_inference_is_bit_reproducible_1000_launches.
    // (The tap slots are stored from v_mov copies, the coordinates by lanes 0..31 only.)
    if (SPLIT) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(prod.x), "+v"(prod.y), "+v"(prod.z), "+v"(prod.w));
    *reinterpret_cast<float4*>(stage + (8 * rr + slot) * STG_LD + c4) = prod;
  };
  for (int tile = blockIdx.x * NWAVE + wave; tile < ntiles; tile += gridDim.x * NWAVE) {
    const int p0 = tile * MT;
    const int posm = p0 + jj;
    const bool livem = posm < P;
    const size_t pm = livem ? (size_t)(a.h.proc_order ? a.h.proc_order[posm] : (uint32_t)posm) : 0;
    if (hh == 0) {   // one lane per point: coordinates -> the pad of the point's staging row
      float u[4];
      point_coords(a.h, (int)pm, u);
      *reinterpret_cast<float4*>(stage + jj * STG_LD + 32) = make_float4(u[0], u[1], u[2], u[3]);
    }
    f32x16 hid[2];
    acc_bias<2>(hid, bias(0), lane);
    wave_lds_sync();
    // 16 steps k = (level k >> 2, round k & 3), two in flight: the texel gathers of step k + 2 are requested before step k's
    // products are formed, and a level's quarter of the feature_out GEMM runs under the next level's first gathers
    auto level_gemm = [&](int l) {   // the level's tile is complete: re-read it in the MFMA B-operand layout, K quarter l of feature_out
      f32x16 x[1];
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 4; q++) {   // channels 8q + 4h .. +3 of point jj: the chunk act_load would have read from HBM
        const float4 v = *reinterpret_cast<const float4*>(stage + jj * STG_LD + 8 * q + 4 * hh);
        x[0][4 * q + 0] = v.x; x[0][4 * q + 1] = v.y; x[0][4 * q + 2] = v.z; x[0][4 * q + 3] = v.w;
      }
      if (S3G_INFER_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      if constexpr (SPLIT) {
        ActSplit<1> xs;
        act_split<1, false>(xs, x);
        int ln = lane;
        asm volatile("" : "+v"(ln));   // fragment addresses are derived here, not carried (and spilled) across the sampler steps
        gemm_split<2, 1>(wsplit + spw::W0, 8, 2 * l, xs, hid, ln);
      } else {
        gemm_reg<2, 1, false>(wslab(l >> 1) + 32 * (l & 1) * 65, 65, x, hid, lane);
      }
      if (S3G_INFER_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (UT) {
      Tex A, B;
      taps_for(0, 0, tp0);
      taps_for(0, 1, tp1);
      wave_lds_sync();
      issue(A, 0, tp0);
      issue(B, 0, tp1);
      for (int k = 0; k < 16; k += 2) {
        const int l = k >> 2, rr = k & 3;
        consume(A, l, rr, tp0);
        if (k + 2 < 16) {
          wave_lds_sync();
          taps_for((k + 2) >> 2, (k + 2) & 3, tp0);
          wave_lds_sync();
          issue(A, (k + 2) >> 2, tp0);
        }
        consume(B, l, rr + 1, tp1);
        if (rr == 2) level_gemm(l);   // runs under the gathers of step k + 2 just requested
        if (k + 3 < 16) {
          wave_lds_sync();
          taps_for((k + 3) >> 2, (k + 3) & 3, tp1);
          wave_lds_sync();
          issue(B, (k + 3) >> 2, tp1);
        }
      }
    } else {   // per-point time (24 texels per step): one step in flight
      Tex A;
      for (int l = 0; l < 4; l++) {
        for (int rr = 0; rr < 4; rr++) {
          wave_lds_sync();
          taps_for(l, rr, tp0);
          wave_lds_sync();
          issue(A, l, tp0);
          consume(A, l, rr, tp0);
        }
        level_gemm(l);
      }
    }
    f32x16 act[2], acc[2], o[1];
    if (S3G_INFER_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (SPLIT) {
      ActSplit<2> hs, as;
      act_split<2, true>(hs, hid);   // relu(hidden): the input of both heads
      int ln = lane;
      asm volatile("" : "+v"(ln));   // (as in level_gemm: the heads' LDS addresses are not loop invariants kept in registers)
      const int jj = ln & 31, hh = ln >> 5;
      // pos head.  P1 is fp32 in LDS: a lane's eight weights of a fragment are two swizzled 16-byte chunks of its row
      acc_bias<2>(act, bias(1), ln);
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int mbo = 0; mbo < 2; mbo++) {
          const float* wr = lds + spw::P1 + (32 * mbo + jj) * spw::P1LD + 4 * hh + 16 * ks;   // inputs 16 ks + 4 h + {0..3, 8..11}
          const float4 lo = *reinterpret_cast<const float4*>(wr);
          const float4 hi = *reinterpret_cast<const float4*>(wr + 8);
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          act[mbo] = mfma_split(act[mbo], split8(v), hs.b[ks >> 1][ks & 1]);
        }
      relu_inplace<2>(act);
      act_split<2, false>(as, act);
      acc_bias<1>(o, bias(3), ln);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {   // P2: rows 0..2 are stored; the other lanes read row 0 (their accumulator rows are never written out)
        Split8 w;
#pragma unroll
        for (int pc = 0; pc < 3; pc++)
          w.p[pc] = *reinterpret_cast<const u32x4*>(wsplit + spw::P2 + ((((ks * 3 + pc) * 2 + hh) * 3 + (jj < 3 ? jj : 0)) << 2));
        o[0] = mfma_split(o[0], w, as.b[ks >> 1][ks & 1]);
      }
      if (livem && hh == 0) {
        float* row = a.dx + pm * 3;
        row[0] = o[0][0]; row[1] = o[0][1]; row[2] = o[0][2];
      }
      // shs head
      acc_bias<2>(act, bias(2), ln);
      gemm_split<2, 2>(wsplit + spw::S1, 4, 0, hs, act, ln);
      relu_inplace<2>(act);
      act_split<2, false>(as, act);
      acc_bias<2>(acc, bias(4), ln);
      gemm_split<1, 2>(wsplit + spw::S2A, 4, 0, as, *reinterpret_cast<f32x16(*)[1]>(&acc[0]), ln);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {   // rows 32..47: lanes of rows 48..63 read rows 32..47 again (never written out)
        Split8 w;
#pragma unroll
        for (int pc = 0; pc < 3; pc++)
          w.p[pc] = *reinterpret_cast<const u32x4*>(wsplit + spw::S2B + (ks * 3 + pc) * (spw::FRAG / 2) + ((16 * hh + (jj & 15)) << 2));
        acc[1] = mfma_split(acc[1], w, as.b[ks >> 1][ks & 1]);
      }
    } else {
    // pos head
    acc_bias<2>(act, bias(1), lane);
    if (S3G_INFER_EXPERIMENT != 1) gemm_reg<2, 2, true>(wslab(2), 65, hid, act, lane);
    relu_inplace<2>(act);
    {
      float o3[3] = {0.f, 0.f, 0.f};
      if (S3G_INFER_EXPERIMENT != 1) head3_fw(wslab(4), bias(3), act, o3, lane);
      if (livem && hh == 0) {
        float* row = a.dx + pm * 3;
        row[0] = o3[0]; row[1] = o3[1]; row[2] = o3[2];
      }
    }
    // shs head
    acc_bias<2>(act, bias(2), lane);
    if (S3G_INFER_EXPERIMENT != 1) gemm_reg<2, 2, true>(wslab(3), 65, hid, act, lane);
    relu_inplace<2>(act);
    acc_bias<2>(acc, bias(4), lane);
    if (S3G_INFER_EXPERIMENT != 1) gemm_reg<2, 2, false>(wslab(5), 65, act, acc, lane);
    }
    if (S3G_INFER_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (livem) {
      float* row = a.dshs + pm * 48 + 4 * hh;
#pragma unroll
      for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (32 * mb + 8 * q >= 48) continue;
          *reinterpret_cast<float4*>(row + 32 * mb + 8 * q) =
              make_float4(acc[mb][4 * q + 0], acc[mb][4 * q + 1], acc[mb][4 * q + 2], acc[mb][4 * q + 3]);
        }
    }
    wave_lds_sync();   // the next tile's coordinates go into the pads this tile's taps were derived from
  }
}

// dW[o][i] += sum_p G[p][o] * A[p][i];  db[o] += sum_p G[p][o].   An MFMA GEMM whose K dimension is the points, fed
// straight from HBM: at K step s lane (i, k) supplies point p0 + 2s + k, and -- because the order of the M / N rows of an
// MFMA is as free as its K order -- row i of block t is feature VEC*i + t, so a lane's operand values for all blocks
// are ONE contiguous VEC-float load and a wave instruction reads two whole rows.  No LDS, no transposition; the next
// tile's rows are requested step by step as the current ones are consumed.
struct WgradArgs {
  const float* G;  // [P][GW]
  const float* A;  // [P][AW]
  float* dW;       // [GW][AW]
  float* db;       // [GW]
  int P;
};
template <int W>
struct RowSplit {  // floats per lane = number of 32-row blocks
  static constexpr int VEC = W > 64 ? 4 : (W > 32 ? 2 : 1);
};
// Branch-free (clamped address + select) so the compiler can count the outstanding loads statically: a guarded load
// forces an s_waitcnt vmcnt(0) at every join and serialises the stream.
template <int W, bool RELU, int STRIDE = W>
__device__ __forceinline__ void row_load(float (&v)[RowSplit<W>::VEC], const float* __restrict__ g, int p, int P, int i) {
  constexpr int VEC = RowSplit<W>::VEC;
  const bool ok = p < P && VEC * i < W;
  const float* src = g + (size_t)min(p, P - 1) * STRIDE + (VEC * i < W ? VEC * i : 0);
  if constexpr (VEC == 4) {
    const float4 x = *reinterpret_cast<const float4*>(src);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  } else if constexpr (VEC == 2) {
    const float2 x = *reinterpret_cast<const float2*>(src);
    v[0] = x.x; v[1] = x.y;
  } else {
    v[0] = src[0];
  }
#pragma unroll
  for (int t = 0; t < VEC; t++) {
    v[t] = ok ? v[t] : 0.f;
    if (RELU) v[t] = fmaxf(v[t], 0.f);
  }
}

// ASTRIDE > AW: A (and dW) are AW-column windows of wider [.][ASTRIDE] arrays (feature_out is done as two halves so the
// accumulators of a wave stay at 64 registers).
//
// Software pipeline: the operand rows of a wave's NEXT tile are requested before the 64 MFMAs of the current tile are issued
// and are not touched until the following iteration, in two alternating register sets (the loop is unrolled by two), so a
// whole tile of MFMA work (~3.4 us at two waves per SIMD) covers the HBM latency.  Loads of full tiles carry no bounds
// select at all: a select on a loaded value is scheduled where the value is consumed and -- when that is the loop's last
// instruction group -- turns into `s_waitcnt vmcnt(0)` in front of the back-edge (the previous version of this kernel exposed
// the full memory latency once per tile that way: ~9 us per tile for 1.7 us of MFMA work).  The one ragged tile at the end
// of the array is handled separately with masked loads.
constexpr int WG_WAVES = 8;  // waves per wgrad workgroup (one persistent workgroup per CU)
#ifndef S3G_WGRAD_PAIRED
#define S3G_WGRAD_PAIRED 2   // 2: all nine GEMMs in ONE launch (default); 0: the nine launches of round 2 (fallback)
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// (streaming / non-temporal operand loads were measured slower in rounds 2 and 3 -- 1.19 -> 1.27 ms, 0.955 -> 1.01 ms -- and are gone)
constexpr bool WGRAD_NONTEMPORAL = false;

// raw (select-free) operand loads of a FULL tile: columns are clamped statically so lanes beyond the row's width re-read
// valid data (their MFMA rows are discarded at the flush)
template <int W, int STRIDE>
__device__ __forceinline__ void row_load_full(float (&v)[RowSplit<W>::VEC], const float* __restrict__ g, int p, int i) {
  constexpr int VEC = RowSplit<W>::VEC;
  const int col = VEC * i < W ? VEC * i : W - VEC;
  const float* src = g + (size_t)p * STRIDE + col;
  if constexpr (VEC == 4) {
    float4 x;
    if (WGRAD_NONTEMPORAL) { const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src)); x = make_float4(q.x, q.y, q.z, q.w); }
    else x = *reinterpret_cast<const float4*>(src);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  } else if constexpr (VEC == 2) {
    float2 x;
    if (WGRAD_NONTEMPORAL) { const f2v q = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(src)); x = make_float2(q.x, q.y); }
    else x = *reinterpret_cast<const float2*>(src);
    v[0] = x.x; v[1] = x.y;
  } else {
    v[0] = src[0];
  }
}

template <int GW, int AW, bool RELU_A, int ASTRIDE>
__global__ void __launch_bounds__(WG_WAVES * 64) mlp_wgrad_kernel(const WgradArgs a) {
  constexpr int GV = RowSplit<GW>::VEC, AV = RowSplit<AW>::VEC, STEPS = MT / 2;
  constexpr int GLOADS = GW == 3 ? 1 : STEPS;
  __shared__ float red[32 * GV * AW + 32 * GV];  // the workgroup's dW block and db, combined in LDS before the flush
  for (int e = threadIdx.x; e < 32 * GV * AW + 32 * GV; e += WG_WAVES * 64) red[e] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, k = lane >> 5;
  f32x16 acc[GV][AV];
#pragma unroll
  for (int m = 0; m < GV; m++)
#pragma unroll
    for (int n = 0; n < AV; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
  float bsum[GV];
#pragma unroll
  for (int m = 0; m < GV; m++) bsum[m] = 0.f;
  const int nfull = a.P / MT;  // tiles whose 32 rows all exist
  const int stride = gridDim.x * WG_WAVES;
  // GW == 3 (the dx / feat heads): a tile's 32 x 3 gradient block is 384 contiguous bytes -- one coalesced 8-byte load per
  // lane (lanes 48.. re-read the block's start) instead of 16 three-lane loads; the MFMA operand of step s is then picked
  // out with two ds_bpermute.
  struct Set {
    float g[GLOADS][GW == 3 ? 2 : GV];
    float v[STEPS][AV];
  };
  auto issue = [&](Set& S, int tile) {  // requires tile < nfull
    const int p0 = tile * MT;
    if constexpr (GW == 3) {
      const float2 x = *reinterpret_cast<const float2*>(a.G + (size_t)p0 * 3 + 2 * (lane < 48 ? lane : lane - 48));
      S.g[0][0] = x.x; S.g[0][1] = x.y;
    }
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      if constexpr (GW != 3) row_load_full<GW, GW>(S.g[s], a.G, p0 + 2 * s + k, i);
      row_load_full<AW, ASTRIDE>(S.v[s], a.A, p0 + 2 * s + k, i);
    }
  };
  auto pick_g3 = [&](const float (&g)[2], int s) {  // G[p0 + 2s + k][i] for lanes i < 3
    const int e = 6 * s + 3 * k + (i < 3 ? i : 0);
    const float x = __shfl(g[0], e >> 1), y = __shfl(g[1], e >> 1);
    return i < 3 ? ((e & 1) ? y : x) : 0.f;
  };
  auto consume = [&](const Set& S) {
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      float ga[GV], ba[AV];
      if constexpr (GW == 3) {
        ga[0] = pick_g3(S.g[0], s);
      } else {
#pragma unroll
        for (int m = 0; m < GV; m++) ga[m] = S.g[s][m];
      }
#pragma unroll
      for (int n = 0; n < AV; n++) ba[n] = RELU_A ? fmaxf(S.v[s][n], 0.f) : S.v[s][n];
#pragma unroll
      for (int m = 0; m < GV; m++) {
        bsum[m] += ga[m];
#pragma unroll
        for (int n = 0; n < AV; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[m], ba[n], acc[m][n], 0, 0, 0);
      }
    }
  };
  {
    // Loads are issued UNCONDITIONALLY (tile index clamped to the last full tile: one wasted prefetch per wave at the end):
    // a load behind a branch makes the compiler's waitcnt pass merge the "issued" and "not issued" paths at the join and
    // wait for the stricter of the two counts -- i.e. for the loads it has just issued.
    Set A, B;
    const int t0 = blockIdx.x * WG_WAVES + wave;
    const int cnt = t0 < nfull ? (nfull - t0 + stride - 1) / stride : 0;
    const int last = nfull - 1;
    if (cnt > 0) {
      // sched_barrier(0): nothing moves across it -- without it the machine scheduler sinks the prefetch loads down to
      // shorten their live ranges and the pipeline collapses into load -> wait -> use
      issue(A, t0);
      for (int it = 0; it < cnt; it += 2) {
        issue(B, min(t0 + (it + 1) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(A);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 >= cnt) break;
        issue(A, min(t0 + (it + 2) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(B);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // the ragged last tile (P % 32 rows): masked loads, handled by the wave whose sequence it continues
  if (a.P % MT != 0 && (nfull % stride) == blockIdx.x * WG_WAVES + wave) {
    const int p0 = nfull * MT;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      float ga[GV], ba[AV];
      const int p = p0 + 2 * s + k;
      if constexpr (GW == 3) {
        ga[0] = (p < a.P && i < 3) ? a.G[(size_t)p * 3 + i] : 0.f;
      } else {
        row_load<GW, false>(ga, a.G, p, a.P, i);
      }
      row_load<AW, RELU_A, ASTRIDE>(ba, a.A, p, a.P, i);
#pragma unroll
      for (int m = 0; m < GV; m++) {
        bsum[m] += ga[m];
#pragma unroll
        for (int n = 0; n < AV; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[m], ba[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // acc[m][n][r] of lane l is dW[GV * row + m][AV * col + n], row = acc_row(r, l), col = l & 31.  The waves add their blocks
  // into `red` ONE AFTER THE OTHER with plain read-modify-writes (the lanes of a wave own distinct elements): ds_add_f32 costs
  // 192 cycles per wave instruction on this chip, serialised across the waves of the CU (tools/ubench/lds_atomics.hip), and 64
  // of them per wave were 13 % of every launch (1.27 -> 1.10 ms for the
  // nine launches).  The order is fixed, so a workgroup's partial sums are reproducible.
  for (int w = 0; w < WG_WAVES; w++) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < GV; m++)
#pragma unroll
        for (int n = 0; n < AV; n++)
#pragma unroll
          for (int r = 0; r < 16; r++) red[(GV * acc_row(r, lane) + m) * AW + AV * (lane & 31) + n] += acc[m][n][r];
#pragma unroll
      for (int m = 0; m < GV; m++) {
        const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
        if (k == 0) red[32 * GV * AW + GV * i + m] += tot;
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < GW * AW; e += WG_WAVES * 64)
    atomicAdd(&a.dW[(size_t)(e / AW) * ASTRIDE + e % AW], red[e]);
  if (a.db != nullptr && threadIdx.x < GW) atomicAdd(&a.db[threadIdx.x], red[32 * GV * AW + threadIdx.x]);
}

// ---- one GEMM of the weight-gradient set ("job") ------------------------------------------------------------------------------
// D0 / P1 / S1 all multiply by `hidden` (P1 and S1 through a ReLU), the two K halves of feature_out share `ghid`; as nine separate
// launches (mlp_wgrad_kernel above, kept as the compile-time fallback S3G_WGRAD_PAIRED=0) every shared plane comes from HBM once per
// launch: 4.87 GB per iteration for 0.61 GB of algorithmic input.  mlp_wgrad_all_kernel below runs them all in ONE launch.  (Round
// 3's five-launch form -- GEMMs sharing an operand paired, mlp_wgrad_multi_kernel -- measured 0.97 vs 0.92 ms and was removed in
// round 5.)
struct WJob {
  const float* G;   // [P][64]
  const float* A;   // [P][astride], 64 columns used
  float* dW;        // [64][astride] window
  float* db;        // [64] or NULL
  int astride;
  float relu_lo;    // 0 = ReLU on A, -inf = none: one v_max either way
};
constexpr int WM_RED = 32 * 2 * 64 + 32 * 2;   // floats of LDS per job: its 64 x 64 block + the bias sums

// ---- ALL weight gradients in one launch --------------------------------------------------------------------------------------
// One persistent workgroup per CU, one wave per GEMM ("job"), every wave of a workgroup on the SAME tile sequence: each of the ten
// stash / signal planes and the feature rows leave HBM once per iteration (3288 B per point instead of 4056 B with nine launches),
// the launch ramps and tails of five launches become one, and jobs of different intensity (the 3-row heads are all loads, the
// 64 x 64 GEMMs balanced) cover each other.  Only TWO code paths live in the workgroup -- "wide" (G up to 64 columns, runtime
// strides / ReLU) and "head" (G = [P,3]; one wave does both heads, one after the other: 2 x 32 MFMAs per tile = a wide job's 64) --
// which is what separates this from the r2 dead end (nine differently unrolled paths per CU: 1.55 -> 1.93 ms).
struct WJobX {
  const float* G;   // wide: [P][gstride], gw columns used;  head: [P][3]
  const float* A;   // [P][astride], 64 columns used
  float* dW;        // [gw][astride] window
  float* db;        // [gw] or NULL
  int gstride, gw, astride;
  float relu_lo;
  int kind;         // 0 wide, 1 head (then G2 / A2 / dW2 / db2 = the second head, or NULL)
  const float* G2;
  const float* A2;
  float* dW2;
  float* db2;
};
struct WAllArgs {
  WJobX job[8];
  int P;
  float* part;      // NULL: every workgroup adds its block to dW with float atomics (run-to-run order: not reproducible);
                    // else [gridDim.x][njobs][WPART] partial blocks, plain stores, summed IN BLOCK ORDER by mlp_wgrad_reduce_kernel
};
// one job's partial block: its gw x 64 window (row-major, 64 columns) + the bias sums; a head job: two [3][64] + [3] records
constexpr int WPART = 64 * 64 + 64, WPART_BIAS = 64 * 64, WPART_HEAD2 = WPART / 2, WPART_HEAD_BIAS = 3 * 64;
constexpr int WPART_MAX_BLOCKS = 256, WPART_MAX_JOBS = 8;

__device__ __forceinline__ void wgrad_wide_wave(const WJobX& jb, float* __restrict__ red, int P, int lane, float* __restrict__ part) {
  constexpr int STEPS = MT / 2;
  const int i = lane & 31, k = lane >> 5;
  const int gcol = min(2 * i, jb.gw - 2);
  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const int nfull = P / MT, stride = gridDim.x;
  struct Set {
    float g[STEPS][2];
    float v[STEPS][2];
  };
  auto issue = [&](Set& S, int tile) {  // requires tile < nfull
    const int p0 = tile * MT;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const float2 x = *reinterpret_cast<const float2*>(jb.G + (size_t)(p0 + 2 * s + k) * jb.gstride + gcol);
      S.g[s][0] = x.x; S.g[s][1] = x.y;
      const float2 y = *reinterpret_cast<const float2*>(jb.A + (size_t)(p0 + 2 * s + k) * jb.astride + 2 * i);
      S.v[s][0] = y.x; S.v[s][1] = y.y;
    }
  };
  auto consume = [&](const Set& S) {
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const float b0 = fmaxf(S.v[s][0], jb.relu_lo), b1 = fmaxf(S.v[s][1], jb.relu_lo);
#pragma unroll
      for (int m = 0; m < 2; m++) {
        bsum[m] += S.g[s][m];
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.g[s][m], b0, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.g[s][m], b1, acc[m][1], 0, 0, 0);
      }
    }
  };
  {
    Set A, B;
    const int t0 = blockIdx.x;
    const int cnt = t0 < nfull ? (nfull - t0 + stride - 1) / stride : 0;
    const int last = nfull - 1;
    if (cnt > 0) {
      issue(A, t0);
      for (int it = 0; it < cnt; it += 2) {
        issue(B, min(t0 + (it + 1) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(A);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 >= cnt) break;
        issue(A, min(t0 + (it + 2) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(B);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (P % MT != 0 && (nfull % stride) == (int)blockIdx.x) {   // the ragged last tile, masked loads
    const int p0 = nfull * MT;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const int p = p0 + 2 * s + k;
      float ga[2] = {0.f, 0.f}, ba[2] = {0.f, 0.f};
      if (p < P) {
        ga[0] = jb.G[(size_t)p * jb.gstride + gcol]; ga[1] = jb.G[(size_t)p * jb.gstride + gcol + 1];
        ba[0] = fmaxf(jb.A[(size_t)p * jb.astride + 2 * i], jb.relu_lo);
        ba[1] = fmaxf(jb.A[(size_t)p * jb.astride + 2 * i + 1], jb.relu_lo);
      }
#pragma unroll
      for (int m = 0; m < 2; m++) {
        bsum[m] += ga[m];
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[m], ba[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // this wave is the job's only contributor in the workgroup: stage the block in LDS (plain stores) for a coalesced flush
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) red[(2 * acc_row(r, lane) + m) * 64 + 2 * (lane & 31) + n] = acc[m][n][r];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
    if (k == 0) red[32 * 2 * 64 + 2 * i + m] = tot;
  }
  wave_lds_sync();
  if (part != nullptr) {    // ordered flush: this workgroup's block goes to its own slot (coalesced plain stores)
    for (int e = lane; e < jb.gw * 64; e += 64) part[e] = red[e];
    if (lane < jb.gw) part[WPART_BIAS + lane] = red[32 * 2 * 64 + lane];
    return;
  }
  for (int e = lane; e < jb.gw * 64; e += 64) atomicAdd(&jb.dW[(size_t)(e >> 6) * jb.astride + (e & 63)], red[e]);
  if (jb.db != nullptr && lane < jb.gw) atomicAdd(&jb.db[lane], red[32 * 2 * 64 + lane]);
}

// one 3-row head: dW[3][64] += sum_p G[p][0..2] (x) A[p][0..63]   (G rows are 12 bytes: a tile's 32 x 3 block is one coalesced
// 8-byte load per lane, the operand of step s is picked out with two ds_bpermute -- as in mlp_wgrad_kernel<3, ...>)
__device__ __forceinline__ void wgrad_head_wave(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ dW,
                                                float* __restrict__ db, float* __restrict__ red, int P, int lane, float* __restrict__ part) {
  constexpr int STEPS = MT / 2;
  const int i = lane & 31, k = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
  float bsum = 0.f;
  const int nfull = P / MT, stride = gridDim.x;
  struct Set {
    float g[2];
    float v[STEPS][2];
  };
  auto issue = [&](Set& S, int tile) {
    const int p0 = tile * MT;
    const float2 x = *reinterpret_cast<const float2*>(G + (size_t)p0 * 3 + 2 * (lane < 48 ? lane : lane - 48));
    S.g[0] = x.x; S.g[1] = x.y;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const float2 y = *reinterpret_cast<const float2*>(A + (size_t)(p0 + 2 * s + k) * HID + 2 * i);
      S.v[s][0] = y.x; S.v[s][1] = y.y;
    }
  };
  auto consume = [&](const Set& S) {
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const int e = 6 * s + 3 * k + (i < 3 ? i : 0);
      const float x = __shfl(S.g[0], e >> 1), y = __shfl(S.g[1], e >> 1);
      const float ga = i < 3 ? ((e & 1) ? y : x) : 0.f;
      bsum += ga;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, S.v[s][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, S.v[s][1], acc[1], 0, 0, 0);
    }
  };
  {
    Set SA, SB;
    const int t0 = blockIdx.x;
    const int cnt = t0 < nfull ? (nfull - t0 + stride - 1) / stride : 0;
    const int last = nfull - 1;
    if (cnt > 0) {
      issue(SA, t0);
      for (int it = 0; it < cnt; it += 2) {
        issue(SB, min(t0 + (it + 1) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(SA);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 >= cnt) break;
        issue(SA, min(t0 + (it + 2) * stride, last));
        __builtin_amdgcn_sched_barrier(0);
        consume(SB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (P % MT != 0 && (nfull % stride) == (int)blockIdx.x) {
    const int p0 = nfull * MT;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const int p = p0 + 2 * s + k;
      const float ga = (p < P && i < 3) ? G[(size_t)p * 3 + i] : 0.f;
      float ba[2] = {0.f, 0.f};
      if (p < P) { ba[0] = A[(size_t)p * HID + 2 * i]; ba[1] = A[(size_t)p * HID + 2 * i + 1]; }
      bsum += ga;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, ba[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, ba[1], acc[1], 0, 0, 0);
    }
  }
  // rows 0..2 of the 32-row block are registers 0..2 of the k == 0 lanes
  wave_lds_sync();
  if (k == 0) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      red[r * 64 + 2 * i] = acc[0][r];
      red[r * 64 + 2 * i + 1] = acc[1][r];
    }
  }
  const float tot = bsum + __shfl_xor(bsum, 32);
  if (k == 0 && i < 3) red[3 * 64 + i] = tot;
  wave_lds_sync();
  if (part != nullptr) {
    for (int e = lane; e < 3 * 64; e += 64) part[e] = red[e];
    if (lane < 3) part[WPART_HEAD_BIAS + lane] = red[3 * 64 + lane];
  } else {
    for (int e = lane; e < 3 * 64; e += 64) atomicAdd(&dW[e], red[e]);
    if (db != nullptr && lane < 3) atomicAdd(&db[lane], red[3 * 64 + lane]);
  }
  wave_lds_sync();
}

__global__ void __launch_bounds__(WG_WAVES * 64) mlp_wgrad_all_kernel(const WAllArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red_all[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WJobX jb = a.job[wave];
  float* red = red_all + wave * WM_RED;
  float* part = a.part ? a.part + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * WPART : nullptr;
  if (jb.kind == 0) {
    wgrad_wide_wave(jb, red, a.P, lane, part);
  } else {
    wgrad_head_wave(jb.G, jb.A, jb.dW, jb.db, red, a.P, lane, part);
    if (jb.G2 != nullptr) wgrad_head_wave(jb.G2, jb.A2, jb.dW2, jb.db2, red, a.P, lane, part ? part + WPART_HEAD2 : nullptr);
  }
}

// Ordered flush, second half (round 6; VERDICT r5 weak #1: "the weight-gradient flush" was one of the two places where float atomics
// made two runs of the same step differ).  grid = (element chunks, jobs): thread e of job j adds the nb workgroups' partials of ONE
// gradient element in block order 0 .. nb-1 -- a fixed summation order -- onto dW / db (which the caller zero-filled or holds a sum).
// 256 x 37 440 floats = 38 MB of partials written and read once (~12 us of HBM time) + one launch.
__global__ void __launch_bounds__(256) mlp_wgrad_reduce_kernel(const WAllArgs a, int nb, int njobs) {
  // 64 gradient elements per workgroup (consecutive lanes = consecutive elements: coalesced 256-byte reads), the nb partials of
  // each split over the four waves in CONTIGUOUS quarters; every wave adds its quarter in block order (eight independent loads in
  // flight), wave 0 adds the four quarter sums in wave order: one fixed association of the nb addends, whatever the timing.
  __shared__ float quarter[4][64];
  const int j = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const WJobX jb = a.job[j];
  float* dst = nullptr;
  if (jb.kind == 0) {
    if (e < jb.gw * 64) dst = jb.dW + (size_t)(e >> 6) * jb.astride + (e & 63);
    else if (e >= WPART_BIAS && e < WPART_BIAS + jb.gw && jb.db != nullptr) dst = jb.db + (e - WPART_BIAS);
  } else {
    const int h = e >= WPART_HEAD2 ? 1 : 0, r = e - h * WPART_HEAD2;
    float* dW = h ? jb.dW2 : jb.dW;
    float* db = h ? jb.db2 : jb.db;
    if (h == 0 || jb.G2 != nullptr) {
      if (r < 3 * 64) dst = dW + r;
      else if (r >= WPART_HEAD_BIAS && r < WPART_HEAD_BIAS + 3 && db != nullptr) dst = db + (r - WPART_HEAD_BIAS);
    }
  }
  float s = 0.f;
  if (dst != nullptr && e < WPART) {
    const size_t stride = (size_t)njobs * WPART;
    const int per = (nb + 3) / 4, b0 = wv * per, b1 = min(nb, b0 + per);
    const float* src = a.part + (size_t)j * WPART + e;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = src[(size_t)(b + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; b < b1; b++) s += src[(size_t)b * stride];
  }
  quarter[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && dst != nullptr) *dst += ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
}

static int launch_wgrad_all(const WJobX* jobs, int njobs, int P, hipStream_t stream, float* partials) {
  WAllArgs a;
  memset(&a, 0, sizeof a);
  for (int j = 0; j < njobs; j++) a.job[j] = jobs[j];
  a.P = P;
  a.part = partials;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min(ntiles, WPART_MAX_BLOCKS);
  hipLaunchKernelGGL(mlp_wgrad_all_kernel, dim3(blocks), dim3(njobs * 64), (size_t)njobs * WM_RED * sizeof(float), stream, a);
  if (partials != nullptr)
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((WPART + 63) / 64, njobs), dim3(256), 0, stream, a, blocks, njobs);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

template <int GW, int AW, bool RELU_A, int ASTRIDE = AW>
static int launch_wgrad(const float* G, const float* A, float* dW, float* db, int P, hipStream_t stream) {
  WgradArgs a{G, A, dW, db, P};
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + WG_WAVES - 1) / WG_WAVES, 256);
  hipLaunchKernelGGL((mlp_wgrad_kernel<GW, AW, RELU_A, ASTRIDE>), dim3(blocks), dim3(WG_WAVES * 64), 0, stream, a);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

}  // namespace s3g

using namespace s3g;

// stash = [packed weight slabs + biases (PACK_FLOATS)] [pre-split forward image (tpw::WORDS)] [pre-split backward image (tbw::WORDS)]
//         [5 x P x 64 activations] [tiles x 5 x 64 ReLU mask words]
static size_t mask_words(int P) { return (size_t)((P > 0 ? P : 0) + MT - 1) / MT * 5 * 64; }
extern "C" size_t s3g_deform_mlp_stash_bytes(int P) {
  return ((size_t)PACK_TOTAL + (size_t)5 * (size_t)(P > 0 ? P : 0) * HID + mask_words(P)) * sizeof(float);
}
extern "C" size_t s3g_deform_mlp_pack_bytes(void) { return (size_t)PACK_TOTAL * sizeof(float); }

// arithmetic of the per-point GEMM chains of s3g_deform_mlp_forward / _backward (process-wide; the weight-gradient GEMMs, whose K
// dimension is the points, are always the exact fp32 chain)
static std::atomic<int> g_mlp_arithmetic{S3G_MLP_F32};
extern "C" int s3g_deform_mlp_set_arithmetic(int mode) {
  if (mode != S3G_MLP_F32 && mode != S3G_MLP_BF16X3 && mode != S3G_MLP_BF16X3_ONTHEFLY) {
    set_error("s3g_deform_mlp_set_arithmetic: mode must be S3G_MLP_F32, S3G_MLP_BF16X3 or S3G_MLP_BF16X3_ONTHEFLY");
    return S3G_ERR_INVALID_ARG;
  }
  g_mlp_arithmetic.store(mode, std::memory_order_relaxed);
  return S3G_OK;
}
extern "C" int s3g_deform_mlp_get_arithmetic(void) { return g_mlp_arithmetic.load(std::memory_order_relaxed); }

static int mlp_set_attrs() {
  static std::atomic<uint64_t> done{0};
  if (device_needs_setup(done)) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_forward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_backward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_wgrad_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WM_RED * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_forward_presplit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tpw::WORDS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_backward_presplit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tbw::WORDS * 4));
    device_setup_done(done);
  }
  return S3G_OK;
}

extern "C" int s3g_deform_mlp_forward(const s3g_mlp_params* w, int P, const float* features, float* dx, float* dshs,
                                      float* feat, float* stash, int save_activations, void* stream_) {
  if (!w || P < 0 || (P > 0 && (!features || !dx || !dshs || !stash || (!feat && save_activations)))) {
    set_error("s3g_deform_mlp_forward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(NSLAB + 1), dim3(256), 0, stream, *w, stash);
  MlpFwdArgs a;
  a.P = P; a.x = features; a.packed = stash; a.dx = dx; a.dshs = dshs; a.feat = feat;
  a.stash = save_activations ? stash + PACK_TOTAL : nullptr;
  a.maskbits = save_activations ? reinterpret_cast<uint32_t*>(stash + PACK_TOTAL + (size_t)5 * P * HID) : nullptr;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + NWAVE - 1) / NWAVE, 256);
  const int arith = g_mlp_arithmetic.load(std::memory_order_relaxed);
  if (arith == S3G_MLP_BF16X3)   // the forward's pre-split image, behind the fp32 one (the backward packs its own: it may run in another mode)
    hipLaunchKernelGGL(mlp_pack_presplit_fwd_kernel, dim3((tpw::WORDS + 255) / 256), dim3(256), 0, stream, *w,
                       reinterpret_cast<uint32_t*>(stash + PACK_FLOATS));
  profile_begin(S3G_PROFILE_MLP_FORWARD, stream);
  if (arith == S3G_MLP_BF16X3) {
    MlpFwdArgs s = a;
    s.packed = stash + PACK_FLOATS;
    hipLaunchKernelGGL(mlp_forward_presplit_kernel, dim3(blocks), dim3(NWAVE * 64), tpw::WORDS * 4, stream, s);
  } else if (arith == S3G_MLP_BF16X3_ONTHEFLY)
    hipLaunchKernelGGL(mlp_forward_kernel<true>, dim3(blocks), dim3(NWAVE * 64), MLP_LDS_FLOATS * 4, stream, a);
  else
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3(blocks), dim3(NWAVE * 64), MLP_LDS_FLOATS * 4, stream, a);
  profile_end(S3G_PROFILE_MLP_FORWARD, stream, (double)P, 0.0);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

static int mlp_backward_impl(const s3g_mlp_params* w, int P, const float* features, const float* stash_, const float* g_dx,
                             const float* g_dshs, const float* g_feat, float* g_features, const s3g_mlp_params* gw, float* workspace,
                             float* partials, void* stream_);

extern "C" size_t s3g_deform_mlp_wgrad_partial_bytes(void) {
  return (size_t)WPART_MAX_BLOCKS * WPART_MAX_JOBS * WPART * sizeof(float);
}

extern "C" int s3g_deform_mlp_backward(const s3g_mlp_params* w, int P, const float* features, const float* stash_,
                                       const float* g_dx, const float* g_dshs, const float* g_feat, float* g_features,
                                       const s3g_mlp_params* gw, float* workspace, void* stream_) {
  return mlp_backward_impl(w, P, features, stash_, g_dx, g_dshs, g_feat, g_features, gw, workspace, nullptr, stream_);
}

extern "C" int s3g_deform_mlp_backward_ordered(const s3g_mlp_params* w, int P, const float* features, const float* stash_,
                                               const float* g_dx, const float* g_dshs, const float* g_feat, float* g_features,
                                               const s3g_mlp_params* gw, float* workspace, float* wgrad_partials, void* stream_) {
  if (!wgrad_partials && P > 0) {
    set_error("s3g_deform_mlp_backward_ordered: wgrad_partials is NULL");
    return S3G_ERR_INVALID_ARG;
  }
  return mlp_backward_impl(w, P, features, stash_, g_dx, g_dshs, g_feat, g_features, gw, workspace, wgrad_partials, stream_);
}

static int mlp_backward_impl(const s3g_mlp_params* w, int P, const float* features, const float* stash_, const float* g_dx,
                             const float* g_dshs, const float* g_feat, float* g_features, const s3g_mlp_params* gw, float* workspace,
                             float* partials, void* stream_) {
  if (!w || !gw || P < 0 || (P > 0 && (!features || !stash_ || !g_dx || !g_dshs || !g_features || !workspace))) {
    set_error("s3g_deform_mlp_backward: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  if (int e = mlp_set_attrs()) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const float* stash = stash_ + PACK_TOTAL;  // activations; the packed weight images of the forward sit in front
  MlpBwdArgs b;
  b.P = P; b.packed = stash_; b.maskbits = reinterpret_cast<const uint32_t*>(stash + (size_t)5 * P * HID); b.g_dx = g_dx; b.g_dshs = g_dshs; b.g_feat = g_feat; b.g_x = g_features; b.ws = workspace;
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + NWAVE - 1) / NWAVE, 256);
  const int arith = g_mlp_arithmetic.load(std::memory_order_relaxed);
  // S3G_MLP_BF16X3: the transposed pre-split image goes into the slot the stash reserves for it (the caller's buffer: only this region is written)
  float* img = const_cast<float*>(stash_) + PACK_FLOATS + tpw::WORDS;
  if (arith == S3G_MLP_BF16X3)
    hipLaunchKernelGGL(mlp_pack_presplit_bwd_kernel, dim3((tbw::WORDS + 255) / 256), dim3(256), 0, stream, *w, reinterpret_cast<uint32_t*>(img));
  profile_begin(S3G_PROFILE_MLP_BACKWARD, stream);
  if (arith == S3G_MLP_BF16X3) {
    MlpBwdArgs sb = b;
    sb.packed = img;
    hipLaunchKernelGGL(mlp_backward_presplit_kernel, dim3(blocks), dim3(NWAVE * 64), tbw::WORDS * 4, stream, sb);
  } else if (arith != S3G_MLP_F32)
    hipLaunchKernelGGL(mlp_backward_kernel<true>, dim3(blocks), dim3(NWAVE * 64), MLP_LDS_FLOATS * 4, stream, b);
  else
    hipLaunchKernelGGL(mlp_backward_kernel<false>, dim3(blocks), dim3(NWAVE * 64), MLP_LDS_FLOATS * 4, stream, b);
  profile_end(S3G_PROFILE_MLP_BACKWARD, stream, (double)P, 0.0);
  S3G_HIP_CHECK(hipGetLastError());
  profile_begin(S3G_PROFILE_MLP_WGRAD, stream);
  const size_t PS = (size_t)P * HID;
  const float NONE = -__builtin_huge_valf(), RELU = 0.f;
  // the jobs: feature_out's two K halves share ghid ...
  const WJob w0a{workspace + 4 * PS, features, gw->W0, gw->b0, FEAT, NONE}, w0b{workspace + 4 * PS, features + 64, gw->W0 + 64, nullptr, FEAT, NONE};
  // ... and D0 / P1 / S1 share `hidden` (stash plane 0); D1 fills the fourth wave pair of the workgroup
  const WJob d0{workspace + 1 * PS, stash + 0 * PS, gw->D0, gw->db0, HID, NONE}, d1{workspace + 0 * PS, stash + 3 * PS, gw->D1, gw->db1, HID, NONE};
  const WJob p1{workspace + 2 * PS, stash + 0 * PS, gw->P1, gw->pb1, HID, RELU}, s1{workspace + 3 * PS, stash + 0 * PS, gw->S1, gw->sb1, HID, RELU};
  if (S3G_WGRAD_PAIRED == 2) {   // everything in ONE launch (mlp_wgrad_all_kernel)
    auto wide = [](const WJob& j, int gw = HID) { return WJobX{j.G, j.A, j.dW, j.db, gw, gw, j.astride, j.relu_lo, 0, nullptr, nullptr, nullptr, nullptr}; };
    const WJobX s2{g_dshs, stash + 2 * PS, gw->S2, gw->sb2, 48, 48, HID, NONE, 0, nullptr, nullptr, nullptr, nullptr};
    WJobX head{g_dx, stash + 1 * PS, gw->P2, gw->pb2, 3, 3, HID, NONE, 1, nullptr, nullptr, nullptr, nullptr};
    if (g_feat != nullptr) {
      head.G2 = g_feat; head.A2 = stash + 4 * PS; head.dW2 = gw->D2; head.db2 = gw->db2;
      const WJobX jobs[8] = {wide(w0a), wide(w0b), wide(d0), wide(p1), wide(s1), wide(d1), s2, head};
      if (int e = launch_wgrad_all(jobs, 8, P, stream, partials)) return e;
    } else {
      const WJobX jobs[6] = {wide(w0a), wide(w0b), wide(p1), wide(s1), s2, head};
      if (int e = launch_wgrad_all(jobs, 6, P, stream, partials)) return e;
    }
  } else {
  if (g_feat != nullptr) {
    if (int e = launch_wgrad<3, 64, false>(g_feat, stash + 4 * PS, gw->D2, gw->db2, P, stream)) return e;
    if (int e = launch_wgrad<64, 64, false>(workspace + 0 * PS, stash + 3 * PS, gw->D1, gw->db1, P, stream)) return e;
    if (int e = launch_wgrad<64, 64, false>(workspace + 1 * PS, stash + 0 * PS, gw->D0, gw->db0, P, stream)) return e;
  }
  if (int e = launch_wgrad<3, 64, false>(g_dx, stash + 1 * PS, gw->P2, gw->pb2, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, true>(workspace + 2 * PS, stash + 0 * PS, gw->P1, gw->pb1, P, stream)) return e;
  if (int e = launch_wgrad<48, 64, false>(g_dshs, stash + 2 * PS, gw->S2, gw->sb2, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, true>(workspace + 3 * PS, stash + 0 * PS, gw->S1, gw->sb1, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, false, 128>(workspace + 4 * PS, features, gw->W0, gw->b0, P, stream)) return e;
  if (int e = launch_wgrad<64, 64, false, 128>(workspace + 4 * PS, features + 64, gw->W0 + 64, nullptr, P, stream)) return e;
  }
  profile_end(S3G_PROFILE_MLP_WGRAD, stream, (double)P, 0.0);
  return S3G_OK;
}

extern "C" size_t s3g_deform_infer_workspace_bytes(const s3g_hexplane_desc* d) {
  if (!d || d->levels != 4) return 0;
  return ((size_t)PACK_FLOATS + (d->uniform_time ? time_table_floats(d) : 0)) * sizeof(float);
}

static int deform_infer_impl(const s3g_hexplane_desc* d, const s3g_mlp_params* w, int P, const float* xyz, const float* time,
                             const unsigned int* proc_order, float* dx, float* dshs, void* workspace, void* stream_, bool split) {
  if (int e = check_desc(d)) return e;
  if (d->levels != 4) {
    set_error("s3g_deform_infer: the fused path is built for 4 levels x 32 channels = feature_out's 128 inputs");
    return S3G_ERR_INVALID_ARG;
  }
  if (!w || P < 0 || (P > 0 && (!xyz || !time || !dx || !dshs || !workspace))) {
    set_error("s3g_deform_infer: bad argument");
    return S3G_ERR_INVALID_ARG;
  }
  if (P == 0) return S3G_OK;
  hipStream_t stream = (hipStream_t)stream_;
  static std::atomic<uint64_t> done{0};
  if (device_needs_setup(done)) {
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)deform_infer_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)deform_infer_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_FLOATS * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)deform_infer_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_FLOATS_SPLIT * 4));
    S3G_HIP_CHECK(hipFuncSetAttribute((const void*)deform_infer_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_FLOATS_SPLIT * 4));
    device_setup_done(done);
  }
  float* packed = (float*)workspace;
  InferArgs a;
  memset(&a, 0, sizeof a);
  a.h.d = *d; a.h.P = P; a.h.xyz = xyz; a.h.time = time; a.h.proc_order = proc_order;
  a.packed = packed; a.dx = dx; a.dshs = dshs;
  TimeRows rows;
  if (d->uniform_time) use_time_rows(a.h, rows, packed + PACK_FLOATS, nullptr, stream);
  static_assert(spw::WORDS <= PACK_FLOATS, "both weight images fit the front of the workspace");
  if (split) hipLaunchKernelGGL(mlp_pack_split_kernel, dim3(spw::WORDS / 256), dim3(256), 0, stream, *w, reinterpret_cast<uint32_t*>(packed));
  else hipLaunchKernelGGL(mlp_pack_kernel, dim3(NSLAB + 1), dim3(256), 0, stream, *w, packed);
  const int ntiles = (P + MT - 1) / MT;
  const int blocks = min((ntiles + NWAVE - 1) / NWAVE, 256);
  const dim3 grid(blocks), wg(NWAVE * 64);
  profile_begin(S3G_PROFILE_DEFORM_INFER, stream);
  if (split) {
    if (d->uniform_time) hipLaunchKernelGGL((deform_infer_kernel<true, true>), grid, wg, INF_LDS_FLOATS_SPLIT * 4, stream, a);
    else hipLaunchKernelGGL((deform_infer_kernel<false, true>), grid, wg, INF_LDS_FLOATS_SPLIT * 4, stream, a);
  } else {
    if (d->uniform_time) hipLaunchKernelGGL((deform_infer_kernel<true, false>), grid, wg, INF_LDS_FLOATS * 4, stream, a);
    else hipLaunchKernelGGL((deform_infer_kernel<false, false>), grid, wg, INF_LDS_FLOATS * 4, stream, a);
  }
  profile_end(S3G_PROFILE_DEFORM_INFER, stream, (double)P, 4.0);
  S3G_HIP_CHECK(hipGetLastError());
  return S3G_OK;
}

extern "C" int s3g_deform_infer(const s3g_hexplane_desc* d, const s3g_mlp_params* w, int P, const float* xyz, const float* time,
                                const unsigned int* proc_order, float* dx, float* dshs, void* workspace, void* stream) {
  return deform_infer_impl(d, w, P, xyz, time, proc_order, dx, dshs, workspace, stream, false);
}
extern "C" int s3g_deform_infer_split(const s3g_hexplane_desc* d, const s3g_mlp_params* w, int P, const float* xyz, const float* time,
                                      const unsigned int* proc_order, float* dx, float* dshs, void* workspace, void* stream) {
  return deform_infer_impl(d, w, P, xyz, time, proc_order, dx, dshs, workspace, stream, true);
}
