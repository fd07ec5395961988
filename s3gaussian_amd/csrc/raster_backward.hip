// Backward half of the MI355X-native differentiable Gaussian rasterizer (gfx950, wave64).
//
// Computes what CudaRasterizer::Rasterizer::backward computes (RAST/cuda_rasterizer/rasterizer_impl.cu:343-444):
//   blend_backward_kernel      <- BACKWARD::render / renderCUDA       (backward.cu:415-590)
//   geometry_backward_kernel   <- computeCov2DCUDA + preprocessCUDA   (backward.cu:144-274, 346-412) fused into one pass
//
// The reference issues 10 global float atomicAdd per contributing (pixel, Gaussian) pair (backward.cu:550-587).
// Device-scope atomics execute memory-side on MI355X (a few G/s), so this backward has NO global atomics at all and
// is bit-reproducible run to run:
//   1. the six geometry sums are re-associated so that everything depending only on the Gaussian (conic, opacity,
//      0.5*W) is factored out of the pixel sum: with v = dL/dalpha * G the kernel accumulates
//      S0=sum v, Sx=sum v*dx, Sy=sum v*dy, Sxx=sum v*dx*dx, Sxy=sum v*dx*dy, Syy=sum v*dy*dy (+3 colour, +1 depth);
//   2. each of the 10 sums is reduced across the 16 lanes of a row with DPP row-shift adds (no shuffles through memory),
//      skipped outright when no lane of the wave is touched by the Gaussian;
//   3. the 16 lane-rows of the tile (4 waves x 4 rows) park their sums in LDS slots, added in fixed order (no float atomics
//      anywhere);
//   4. once per 32-Gaussian batch each lane owns one Gaussian and stores its 10 sums as one 40-byte record at the
//      instance's position in the sorted list (coalesced: consecutive lanes -> consecutive records);
//   5. geometry_backward_kernel gathers each Gaussian's records through slot_pos[] (the instance -> position map
//      the forward's sort emitted), applies the factored-out coefficients and runs the per-Gaussian chain.
#include "geom_math.hpp"

namespace s3g {

// ---- wave64 reduction: inclusive scan with DPP, total lands in lane 63 (LLVM's gfx9 atomic-optimizer sequence) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
  return v;
}

// Sum within each row of 16 lanes only (4 DPP adds; lane 15 of every row holds its row's total).  The blend backward stops
// here and parks the four row totals in LDS: the two row_bcast steps that finish a wave total cost three instructions each
// (zero, v_mov_dpp, add) per value -- 78 of the ~240 VALU instructions of a (Gaussian, wave) visit in a VALU-bound kernel.
// The empty asm pins the last add in the straight-line block: its only use is the row leader's LDS store, so the compiler sank
// it into that branch, where it cannot be fused with the DPP move any more (v_mov_b32_dpp + v_add_f32: 13 extra VALU
// instructions per visit in a kernel that is bound by VALU issue).
__device__ __forceinline__ float row_sum_lane15(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  asm volatile("" : "+v"(v));
  return v;
}

// ---- transpose-reduce: NV values summed over the 16 lanes of a row with 29 DPP adds instead of 4 NV = 52 ----
// A plain reduction sums every value in every lane.  Here each step also splits the values between the two partners:
//   A  lane i <-> i ^ 8 (row_ror:8):        lanes 0-7 go on with values 0-7, lanes 8-15 with values 8-15;
//   B  lane i <-> 7 - i (row_half_mirror):  banks (groups of four lanes) 0 / 2 keep the low four of their eight, banks 1 / 3 the high four;
//   C, D  quad_perm butterflies inside a bank: every lane of bank b ends with the row totals of values 4b .. 4b+3.
// "keep these, take the partner's" needs no selects: a DPP instruction writes only the lanes its bank_mask enables (bank =
// four consecutive lanes of a row), so `v_add_f32_dpp r, x, x bank_mask:0x3` followed by `v_add_f32_dpp r, y, y bank_mask:0xc`
// leaves x + x' in lanes 0-7 and y + y' in lanes 8-15 of the same register.  Inline asm because the builtin (update_dpp + add)
// cannot express an add whose WRITE is masked; the s_nop covers the VALU-write -> DPP-read hazard (2 wait states) that the
// compiler does not track through inline asm.  Lanes of a bank whose values do not exist (>= NV) hold finite garbage that is
// never stored.
// first write of a register (its other lanes are overwritten by the S3G_DPP_ACC that follows, or never read)
#define S3G_DPP_SET(dst, src, ctrl, bank) \
  asm volatile("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : "=v"(dst) : "v"(src))
#define S3G_DPP_ACC(dst, src, ctrl, bank) \
  asm volatile("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : "+v"(dst) : "v"(src))
#define S3G_DPP_SELF(dst, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(dst))
template <int NV>
__device__ __forceinline__ void row_transpose_reduce(const float* v /* [NV] */, float* q /* [4] */) {
  static_assert(NV > 8 && NV <= 16, "values per visit");
  float r[8];
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < 8; i++) {
    S3G_DPP_SET(r[i], v[i], "row_ror:8", "0x3");
    if (i + 8 < NV) S3G_DPP_ACC(r[i], v[i + 8], "row_ror:8", "0xc");
  }
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < 4; i++) {
    S3G_DPP_SET(q[i], r[i], "row_half_mirror", "0x5");
    S3G_DPP_ACC(q[i], r[i + 4], "row_half_mirror", "0xa");
  }
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < 4; i++) S3G_DPP_SELF(q[i], "quad_perm:[1,0,3,2]");
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < 4; i++) S3G_DPP_SELF(q[i], "quad_perm:[2,3,0,1]");
}

// record layout (NREC floats): dcolor r,g,b | ddepth | S0 | Sx | Sy | Sxx | Sxy | Syy   [| dcolor2 r,g,b | pad]
//
// NX = 3: TWO images blended from the same geometry (the RGB+depth render and the feature render of one iteration,
// gaussian_renderer/__init__.py:127-166) are back-propagated in ONE pass.  Everything that depends only on the geometry
// -- the alpha test, exp2, the transmittance recurrence, the six S sums and their wave reductions -- is shared; the second
// image adds its three colour-gradient sums and its term of dL/dalpha.  Records grow from 10 to 14 floats.
template <int NX>
__global__ void __launch_bounds__(256)
blend_backward_kernel(int W, int H, int gx, int tiles, const uint2* __restrict__ ranges,
                      const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ point_list,
                      const float* __restrict__ bg, const float2* __restrict__ means2D,
                      const float4* __restrict__ conic_opacity, const float* __restrict__ colors,
                      const float* __restrict__ depths, const float* __restrict__ final_Ts,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                      const float* __restrict__ dL_dpixel_depths, float* __restrict__ records /*[R][NR]*/,
                      const float* __restrict__ colors2, const float* __restrict__ dL_dpixels2) {
  constexpr uint32_t BATCH = 32;  // Gaussians staged per round
  constexpr int NR = NX ? NREC + 4 : NREC;
  constexpr int SLOTS = 16;       // 4 waves x 4 rows of 16 lanes
  __shared__ StagedGaussian sg[BATCH];
  __shared__ float4 sg2[NX ? BATCH : 1];  // second image's colour
  // one slot per (wave, row): combined in fixed order -> bit-reproducible sums.  Slots are padded by 16 floats: the four row
  // leaders of a wave store the same [value][Gaussian] element of their four slots at once, and NR * BATCH floats is a
  // multiple of the 64 banks (a 4-way conflict on each of the 13 stores: SQ_LDS_BANK_CONFLICT 1.8 cycles per LDS instruction)
  constexpr int NV = NX ? NREC + 3 : NREC;   // sums per visit
  constexpr int SLOT_FLOATS = 16 * BATCH + 16;   // 16 value rows: the transpose-reduce stores four rows per bank, unconditionally
  __shared__ float acc[SLOTS][SLOT_FLOATS];

  const uint32_t tile = xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile >= (uint32_t)tiles) return;
  const uint2 rg = ranges[tile];
  const uint32_t hi = tile_hi[tile] - rg.x;  // deepest contributor of the tile (1-based); nothing behind it gets gradient
  if (hi == 0) return;
  const int tx = tile % gx, ty = tile / gx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tx * TILE_X + (tid & 15), py = ty * TILE_Y + (tid >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t pix = (size_t)py * W + px, N = (size_t)H * W;

  const float T_final = inside ? final_Ts[pix] : 0.f;
  const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
  float gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f;
  if (inside) {
    gr = dL_dpixels[pix];
    gg = dL_dpixels[N + pix];
    gb = dL_dpixels[2 * N + pix];
    gd = dL_dpixel_depths[pix];
  }
  float g2r = 0.f, g2g = 0.f, g2b = 0.f;
  if (NX && inside) {
    g2r = dL_dpixels2[pix];
    g2g = dL_dpixels2[N + pix];
    g2b = dL_dpixels2[2 * N + pix];
  }
  float bg_dot = bg[0] * gr + bg[1] * gg + bg[2] * gb;
  if (NX) bg_dot += bg[0] * g2r + bg[1] * g2g + bg[2] * g2b;

  float T = T_final;
  float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f;  // accum_rec (colour, depth)
  float last_alpha = 0.f, lr = 0.f, lg = 0.f, lb = 0.f, ld = 0.f;
  float a2r = 0.f, a2g = 0.f, a2b = 0.f, l2r = 0.f, l2g = 0.f, l2b = 0.f;  // second image

  for (uint32_t done_cnt = 0; done_cnt < hi; done_cnt += BATCH) {
    const uint32_t cnt = min(BATCH, hi - done_cnt);
    __syncthreads();  // previous batch fully consumed (sg, acc)
    for (uint32_t e = tid; e < SLOTS * SLOT_FLOATS; e += 256) (&acc[0][0])[e] = 0.f;
    if ((uint32_t)tid < cnt) {
      const uint32_t pos = hi - 1 - (done_cnt + tid);  // back to front
      const uint32_t id = point_list[rg.x + pos];
      const float2 mm = means2D[id];
      const float4 co = conic_opacity[id];
      StagedGaussian s;
      s.a = make_float4(mm.x, mm.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
      s.b = make_float4(-0.5f * LOG2E * co.z, co.w, depths[id], colors[3 * (size_t)id]);
      s.c = make_float4(colors[3 * (size_t)id + 1], colors[3 * (size_t)id + 2], 0.f, 0.f);
      sg[tid] = s;
      if (NX) sg2[tid] = make_float4(colors2[3 * (size_t)id], colors2[3 * (size_t)id + 1], colors2[3 * (size_t)id + 2], 0.f);
    }
    __syncthreads();

    for (uint32_t j = 0; j < cnt; j++) {
      const uint32_t pos = hi - 1 - (done_cnt + j);  // 0-based position in the tile list
      // the reference's `contributor` equals pos after its decrement; it skips when contributor >= last_contributor
      bool valid = pos < last_contributor;
      const float4 A = sg[j].a;
      const float4 B = sg[j].b;
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float q = gaussian_exponent2(dx, dy, A.z, A.w, B.x);
      const float G = __builtin_amdgcn_exp2f(q);
      const float alpha = fminf(0.99f, B.y * G);
      valid = valid && !(q > 0.f) && !(alpha < 1.0f / 255.0f);
      if (__ballot(valid) == 0ull) continue;  // wave-uniform: this Gaussian misses all 64 pixels of the wave

      float p_r = 0.f, p_g = 0.f, p_b = 0.f, p_d = 0.f, s0 = 0.f, sx = 0.f, sy = 0.f, sxx = 0.f, sxy = 0.f, syy = 0.f;
      float p2_r = 0.f, p2_g = 0.f, p2_b = 0.f;
      if (valid) {
        const float4 Cc = sg[j].c;
        const float cr = B.w, cg = Cc.x, cb = Cc.y, cd = B.z;
        const float rcp = __builtin_amdgcn_rcpf(1.f - alpha);
        T = T * rcp;
        const float w = alpha * T;  // dchannel_dcolor
        const float one_m_la = 1.f - last_alpha;
        ar = last_alpha * lr + one_m_la * ar;
        ag = last_alpha * lg + one_m_la * ag;
        ab = last_alpha * lb + one_m_la * ab;
        ad = last_alpha * ld + one_m_la * ad;
        lr = cr; lg = cg; lb = cb; ld = cd;
        float dL_dalpha = (cr - ar) * gr + (cg - ag) * gg + (cb - ab) * gb + (cd - ad) * gd;
        if (NX) {
          const float4 C2 = sg2[j];
          a2r = last_alpha * l2r + one_m_la * a2r;
          a2g = last_alpha * l2g + one_m_la * a2g;
          a2b = last_alpha * l2b + one_m_la * a2b;
          l2r = C2.x; l2g = C2.y; l2b = C2.z;
          dL_dalpha += (C2.x - a2r) * g2r + (C2.y - a2g) * g2g + (C2.z - a2b) * g2b;
          p2_r = w * g2r; p2_g = w * g2g; p2_b = w * g2b;
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final * rcp) * bg_dot;
        p_r = w * gr; p_g = w * gg; p_b = w * gb; p_d = w * gd;
        const float v = dL_dalpha * G;
        const float vx = v * dx, vy = v * dy;
        s0 = v; sx = vx; sy = vy; sxx = vx * dx; sxy = vx * dy; syy = vy * dy;
      }
      // wave-uniform from here: all 64 lanes take part in the DPP reductions (within rows of 16 lanes); values in record order
      const float vals[13] = {p_r, p_g, p_b, p_d, s0, sx, sy, sxx, sxy, syy, p2_r, p2_g, p2_b};
      float q4[4];
      row_transpose_reduce<NV>(vals, q4);
      if ((lane & 3) == 0) {   // the first lane of bank b stores the row totals of values 4b .. 4b+3 (rows >= NV: never read)
        float* aw = &acc[wave * 4 + (lane >> 4)][(lane & 12) * BATCH + j];
        aw[0 * BATCH] = q4[0]; aw[1 * BATCH] = q4[1]; aw[2 * BATCH] = q4[2]; aw[3 * BATCH] = q4[3];
      }
    }
    __syncthreads();
    // Epilogue of the batch: lane t owns Gaussian t of the batch and stores its record (zeros if untouched).
    if ((uint32_t)tid < cnt) {
      const uint32_t pos = hi - 1 - (done_cnt + tid);
      float2* rec = reinterpret_cast<float2*>(records + (size_t)(rg.x + pos) * NR);
#pragma unroll
      for (int k = 0; k < NR / 2; k++) {
        float lo = acc[0][2 * k * BATCH + tid], hi2 = acc[0][(2 * k + 1) * BATCH + tid];
#pragma unroll
        for (int q = 1; q < SLOTS; q++) {  // fixed order: rows of wave 0, then wave 1, ...
          lo += acc[q][2 * k * BATCH + tid];
          hi2 += acc[q][(2 * k + 1) * BATCH + tid];
        }
        rec[k] = make_float2(lo, 2 * k + 1 < NV ? hi2 : 0.f);   // the pad float of a 14-float record stays 0
      }
    }
  }
}

// =========================================================================================================
// Per-Gaussian geometry backward: conic -> cov2D -> (cov3D, mean) ; mean2D, depth -> mean ; SH ; cov3D -> scale, rot.
// One pass over the Gaussians, HBM-bound.  Formulae follow backward.cu:144-412 term by term (fp32, no contraction).
// =========================================================================================================
struct GeomBwdArgs {
  int P, D, M;
  const float* means3D;
  const int* radii;
  const float* shs;
  const uint8_t* clamped;
  const float* scales;
  const float* rotations;
  float scale_modifier;
  const float* cov3Ds;  // precomputed or the forward's own
  const float* view;
  const float* proj;
  const float* campos;
  float fx, fy, tan_fovx, tan_fovy;
  int W, H, gx;
  // gather side
  const ushort4* rect;
  const uint32_t* gauss_off;
  const uint32_t* slot_pos;
  const uint32_t* tile_hi;        // absolute end of the written records of each tile
  const float* records;           // [R][NREC]
  const float4* conic_opacity;
  const uint32_t* ctrl;           // forward's control words: ctrl[4] != 0 = asynchronous forward overflowed, nothing was binned
  // outputs (every element of every output is written, zeros for culled Gaussians)
  float* dL_dmean2D;        // [P,3]
  float* dL_dconic;         // [P,4]  optional (may be NULL)
  float* dL_dopacity;       // [P]
  float* dL_dcolor;         // [P,3]
  float* dL_dcolor2;        // [P,3]  second image's colours (two-image backward only)
  float* dL_ddepth;         // [P]    optional (may be NULL)
  float* dL_dmean3D;        // [P,3]
  float* dL_dcov3D;         // [P,6]
  float* dL_dsh;            // [P,M,3]
  float* dL_dscale;         // [P,3]
  float* dL_drot;           // [P,4]
  // optional densification bookkeeping (train.py:489-493, scene/gaussian_model.py:693-695), all three or none
  float* dens_accum;        // [P] xyz_gradient_accum += ||dL_dmean2D.xy|| where visible
  float* dens_denom;        // [P] denom += 1 where visible
  float* dens_max_radii;    // [P] max_radii2D = max(max_radii2D, radii) where visible
};

// Sum the records of the tiles k = k0, k0+stride, ... of one Gaussian's rect (row-major inside the rect), in that order.
// The walk is a chain of dependent loads (slot_pos -> record; tile_hi beside it) and the kernel is bound by that latency, not by
// bytes: GATHER_BATCH tiles are therefore in flight at a time -- all their positions and tile ends are requested first, then all
// their records, then the sums are taken in ascending k exactly as a one-at-a-time walk would take them (a tile that contributes
// nothing adds +0: the totals are bit-identical).
#ifndef S3G_GATHER_BATCH
#define S3G_GATHER_BATCH 3   // cfg3, two-image pass: 192 / 164 / 145 / 154 / 160 / 158 us for 1 / 2 / 3 / 4 / 6 / 8 (profiles/r04_gather.txt)
#endif
template <int NR>
__device__ __forceinline__ void gather_records(const GeomBwdArgs& a, const ushort4 r, uint32_t o, int k0, int stride,
                                               float* acc) {
  constexpr int B = S3G_GATHER_BATCH;
  const int w = (int)r.z - (int)r.x, n = w * ((int)r.w - (int)r.y);
  for (int k = k0; k < n; k += B * stride) {
    uint32_t pos[B], hi[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const int kk = k + j * stride;
      const int kc = kk < n ? kk : k;   // past the rect: a valid address whose value is not used
      const int ty = (int)r.y + kc / w, tx = (int)r.x + kc % w;
      pos[j] = a.slot_pos[o + kc];
      hi[j] = a.tile_hi[ty * a.gx + tx];  // tile_hi = absolute end of the positions the blend backward wrote
    }
    float2 v[B][NR / 2];
#pragma unroll
    for (int j = 0; j < B; j++) {
#pragma unroll
      for (int q = 0; q < NR / 2; q++) v[j][q] = make_float2(0.f, 0.f);
      if (k + j * stride < n && pos[j] < hi[j]) {
        const float2* rec = reinterpret_cast<const float2*>(a.records + (size_t)pos[j] * NR);
#pragma unroll
        for (int q = 0; q < NR / 2; q++) v[j][q] = rec[q];
      }
    }
#pragma unroll
    for (int j = 0; j < B; j++)
#pragma unroll
      for (int q = 0; q < NR / 2; q++) {
        acc[2 * q] += v[j][q].x;
        acc[2 * q + 1] += v[j][q].y;
      }
  }
}

template <int NX>
__global__ void __launch_bounds__(256) geometry_backward_kernel(const GeomBwdArgs a) {
  constexpr int NR = NX ? NREC + 4 : NREC;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const bool live = gid < a.P;
  const int idx = live ? gid : a.P - 1;  // keep every lane alive for the wave-cooperative gather below
  const size_t i3 = 3 * (size_t)idx;
  const bool vis = live && a.radii[idx] > 0 && a.ctrl[4] == 0u;   // overflowed asynchronous forward: zero gradients, no bookkeeping

  // ---- gather the per-instance records of this Gaussian (one per tile of its rect) ----
  float acc[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) acc[k] = 0.f;
  const ushort4 r = vis ? a.rect[idx] : make_ushort4(0, 0, 0, 0);
  const uint32_t o = vis ? a.gauss_off[idx] : 0u;
  const int n = ((int)r.z - (int)r.x) * ((int)r.w - (int)r.y);
  constexpr int BIG = 24;
  if (n > 0 && n <= BIG) gather_records<NR>(a, r, o, 0, 1, acc);
  uint64_t big = __ballot(n > BIG);
  const int lane = threadIdx.x & 63;
  while (big) {  // wave-uniform: all 64 lanes gather one large rect together, then reduce with DPP
    const int src = __ffsll((unsigned long long)big) - 1;
    big &= big - 1;
    ushort4 br;
    br.x = (unsigned short)__shfl((int)r.x, src); br.y = (unsigned short)__shfl((int)r.y, src);
    br.z = (unsigned short)__shfl((int)r.z, src); br.w = (unsigned short)__shfl((int)r.w, src);
    const uint32_t bo = (uint32_t)__shfl((int)o, src);
    float part[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) part[k] = 0.f;
    gather_records<NR>(a, br, bo, lane, 64, part);
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const float tot = __shfl(wave_sum_lane63(part[k]), 63);
      if (lane == src) acc[k] = tot;
    }
  }
  if (!live) return;
  if (!vis) {  // culled: the reference leaves its zero-initialised outputs untouched
    a.dL_dmean2D[i3] = a.dL_dmean2D[i3 + 1] = a.dL_dmean2D[i3 + 2] = 0.f;
    a.dL_dcolor[i3] = a.dL_dcolor[i3 + 1] = a.dL_dcolor[i3 + 2] = 0.f;
    if (NX) a.dL_dcolor2[i3] = a.dL_dcolor2[i3 + 1] = a.dL_dcolor2[i3 + 2] = 0.f;
    a.dL_dmean3D[i3] = a.dL_dmean3D[i3 + 1] = a.dL_dmean3D[i3 + 2] = 0.f;
    a.dL_dscale[i3] = a.dL_dscale[i3 + 1] = a.dL_dscale[i3 + 2] = 0.f;
    a.dL_dopacity[idx] = 0.f;
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = 0.f;
    if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dL_ddepth) a.dL_ddepth[idx] = 0.f;
    if (a.shs != nullptr)
      for (int k = 0; k < a.M * 3; k++) a.dL_dsh[(size_t)idx * a.M * 3 + k] = 0.f;
    return;
  }
  const float4 co = a.conic_opacity[idx];
  const float S0 = acc[4], Sx = acc[5], Sy = acc[6], Sxx = acc[7], Sxy = acc[8], Syy = acc[9];
  // dL_dG = o * dL_dalpha ; dG_ddelx = -G*dx*conic.x - G*dy*conic.y ; times 0.5*W (backward.cu:571-584)
  const float g2x = (0.5f * a.W) * co.w * (-(co.x * Sx) - co.y * Sy);
  const float g2y = (0.5f * a.H) * co.w * (-(co.z * Sy) - co.y * Sx);
  const float dcx = -0.5f * co.w * Sxx, dcy = -0.5f * co.w * Sxy, dcz = -0.5f * co.w * Syy;
  const float gdep = acc[3];
  a.dL_dmean2D[i3] = g2x; a.dL_dmean2D[i3 + 1] = g2y; a.dL_dmean2D[i3 + 2] = 0.f;
  a.dL_dcolor[i3] = acc[0]; a.dL_dcolor[i3 + 1] = acc[1]; a.dL_dcolor[i3 + 2] = acc[2];
  if (NX) { a.dL_dcolor2[i3] = acc[NREC]; a.dL_dcolor2[i3 + 1] = acc[NREC + 1]; a.dL_dcolor2[i3 + 2] = acc[NREC + 2]; }
  a.dL_dopacity[idx] = S0;
  if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(dcx, dcy, 0.f, dcz);
  if (a.dL_ddepth) a.dL_ddepth[idx] = gdep;
  if (a.dens_accum) {  // the viewspace gradient and the radius are in registers: no separate passes over [P]
    a.dens_accum[idx] += sqrtf(g2x * g2x + g2y * g2y);
    a.dens_denom[idx] += 1.f;
    a.dens_max_radii[idx] = fmaxf(a.dens_max_radii[idx], (float)a.radii[idx]);
  }

  const float3 mean = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
  const float* V = a.view;
  const float* proj = a.proj;

  // ---- computeCov2DCUDA (backward.cu:144-274) ----
  float cov3D[6];
#pragma unroll
  for (int k = 0; k < 6; k++) cov3D[k] = a.cov3Ds[6 * (size_t)idx + k];
  const Cov2DCtx c = cov2d_common(mean, a.fx, a.fy, a.tan_fovx, a.tan_fovy, cov3D, V);
  const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
  const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
  const float ca = c.cov.m[0][0] + 0.3f, cb = c.cov.m[0][1], cc = c.cov.m[1][1] + 0.3f;
  const float denom = ca * cc - cb * cb;
  float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float dcov[6];
#define TM(c_, r_) c.T.m[c_][r_]
#define VM(c_, r_) c.Vrk.m[c_][r_]
  if (denom2inv != 0.f) {
    dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
    dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
    dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
    dcov[0] = (TM(0, 0) * TM(0, 0) * dL_da + TM(0, 0) * TM(1, 0) * dL_db + TM(1, 0) * TM(1, 0) * dL_dc);
    dcov[3] = (TM(0, 1) * TM(0, 1) * dL_da + TM(0, 1) * TM(1, 1) * dL_db + TM(1, 1) * TM(1, 1) * dL_dc);
    dcov[5] = (TM(0, 2) * TM(0, 2) * dL_da + TM(0, 2) * TM(1, 2) * dL_db + TM(1, 2) * TM(1, 2) * dL_dc);
    dcov[1] = 2 * TM(0, 0) * TM(0, 1) * dL_da + (TM(0, 0) * TM(1, 1) + TM(0, 1) * TM(1, 0)) * dL_db + 2 * TM(1, 0) * TM(1, 1) * dL_dc;
    dcov[2] = 2 * TM(0, 0) * TM(0, 2) * dL_da + (TM(0, 0) * TM(1, 2) + TM(0, 2) * TM(1, 0)) * dL_db + 2 * TM(1, 0) * TM(1, 2) * dL_dc;
    dcov[4] = 2 * TM(0, 2) * TM(0, 1) * dL_da + (TM(0, 1) * TM(1, 2) + TM(0, 2) * TM(1, 1)) * dL_db + 2 * TM(1, 1) * TM(1, 2) * dL_dc;
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) dcov[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];

  const float dL_dT00 = 2 * (TM(0, 0) * VM(0, 0) + TM(0, 1) * VM(0, 1) + TM(0, 2) * VM(0, 2)) * dL_da + (TM(1, 0) * VM(0, 0) + TM(1, 1) * VM(0, 1) + TM(1, 2) * VM(0, 2)) * dL_db;
  const float dL_dT01 = 2 * (TM(0, 0) * VM(1, 0) + TM(0, 1) * VM(1, 1) + TM(0, 2) * VM(1, 2)) * dL_da + (TM(1, 0) * VM(1, 0) + TM(1, 1) * VM(1, 1) + TM(1, 2) * VM(1, 2)) * dL_db;
  const float dL_dT02 = 2 * (TM(0, 0) * VM(2, 0) + TM(0, 1) * VM(2, 1) + TM(0, 2) * VM(2, 2)) * dL_da + (TM(1, 0) * VM(2, 0) + TM(1, 1) * VM(2, 1) + TM(1, 2) * VM(2, 2)) * dL_db;
  const float dL_dT10 = 2 * (TM(1, 0) * VM(0, 0) + TM(1, 1) * VM(0, 1) + TM(1, 2) * VM(0, 2)) * dL_dc + (TM(0, 0) * VM(0, 0) + TM(0, 1) * VM(0, 1) + TM(0, 2) * VM(0, 2)) * dL_db;
  const float dL_dT11 = 2 * (TM(1, 0) * VM(1, 0) + TM(1, 1) * VM(1, 1) + TM(1, 2) * VM(1, 2)) * dL_dc + (TM(0, 0) * VM(1, 0) + TM(0, 1) * VM(1, 1) + TM(0, 2) * VM(1, 2)) * dL_db;
  const float dL_dT12 = 2 * (TM(1, 0) * VM(2, 0) + TM(1, 1) * VM(2, 1) + TM(1, 2) * VM(2, 2)) * dL_dc + (TM(0, 0) * VM(2, 0) + TM(0, 1) * VM(2, 1) + TM(0, 2) * VM(2, 2)) * dL_db;
#undef TM
#undef VM
  const float dL_dJ00 = c.W.m[0][0] * dL_dT00 + c.W.m[0][1] * dL_dT01 + c.W.m[0][2] * dL_dT02;
  const float dL_dJ02 = c.W.m[2][0] * dL_dT00 + c.W.m[2][1] * dL_dT01 + c.W.m[2][2] * dL_dT02;
  const float dL_dJ11 = c.W.m[1][0] * dL_dT10 + c.W.m[1][1] * dL_dT11 + c.W.m[1][2] * dL_dT12;
  const float dL_dJ12 = c.W.m[2][0] * dL_dT10 + c.W.m[2][1] * dL_dT11 + c.W.m[2][2] * dL_dT12;
  const float tz = 1.f / c.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
  const float dL_dtx = x_grad_mul * -a.fx * tz2 * dL_dJ02;
  const float dL_dty = y_grad_mul * -a.fy * tz2 * dL_dJ12;
  const float dL_dtz = -a.fx * tz2 * dL_dJ00 - a.fy * tz2 * dL_dJ11 + (2 * a.fx * c.t.x) * tz3 * dL_dJ02 +
                       (2 * a.fy * c.t.y) * tz3 * dL_dJ12;
  // transformVec4x3Transpose (auxiliary.h:89-97); the reference ASSIGNS here (backward.cu:273)
  float dmx = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
  float dmy = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
  float dmz = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

  // ---- preprocessCUDA backward (backward.cu:346-412) ----
  const float4 m_hom = xform_4x4(mean, proj);
  const float m_w = 1.0f / (m_hom.w + 0.0000001f);
  const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
  const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
  dmx += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
  dmy += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
  dmz += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
  const float mul3 = V[2] * mean.x + V[6] * mean.y + V[10] * mean.z + V[14];
  dmx += (V[2] - V[3] * mul3) * gdep;
  dmy += (V[6] - V[7] * mul3) * gdep;
  dmz += (V[10] - V[11] * mul3) * gdep;

  // ---- SH backward (backward.cu:20-139) ----
  if (a.shs != nullptr) {
    const int deg = a.D;
    const float3 dir_orig = make_float3(mean.x - a.campos[0], mean.y - a.campos[1], mean.z - a.campos[2]);
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = a.shs + (size_t)idx * a.M * 3;
    float* dsh = a.dL_dsh + (size_t)idx * a.M * 3;
    float dRGB[3], dRdx[3] = {0.f, 0.f, 0.f}, dRdy[3] = {0.f, 0.f, 0.f}, dRdz[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = acc[ch] * (a.clamped[3 * (size_t)idx + ch] ? 0.f : 1.f);
    for (int k = (deg + 1) * (deg + 1) * 3; k < a.M * 3; k++) dsh[k] = 0.f;  // coefficients above the active degree
#define SH(k) sh[(k)*3 + ch]
#define DSH(k, v) dsh[(k)*3 + ch] = (v)*dRGB[ch]
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      DSH(0, SH_C0);
      if (deg > 0) {
        DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
        dRdx[ch] = -SH_C1 * SH(3); dRdy[ch] = -SH_C1 * SH(1); dRdz[ch] = SH_C1 * SH(2);
        if (deg > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
          DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
          dRdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
          dRdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
          dRdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
          if (deg > 2) {
            DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy * z); DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
            DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)); DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
            DSH(14, SH_C3[5] * z * (xx - yy)); DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
            dRdx[ch] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                         SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                         SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
            dRdy[ch] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz + SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                         SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                         SH_C3[6] * SH(15) * -3.f * 2.f * xy);
            dRdz[ch] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz + SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                         SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy));
          }
        }
      }
    }
#undef SH
#undef DSH
    const float ddx = dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2];
    const float ddy = dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2];
    const float ddz = dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2];
    // dnormvdv (auxiliary.h:107-117)
    const float3 v = dir_orig;
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmx += ((+sum2 - v.x * v.x) * ddx - v.y * v.x * ddy - v.z * v.x * ddz) * invsum32;
    dmy += (-v.x * v.y * ddx + (sum2 - v.y * v.y) * ddy - v.z * v.y * ddz) * invsum32;
    dmz += (-v.x * v.z * ddx - v.y * v.z * ddy + (sum2 - v.z * v.z) * ddz) * invsum32;
  }
  a.dL_dmean3D[3 * (size_t)idx + 0] = dmx;
  a.dL_dmean3D[3 * (size_t)idx + 1] = dmy;
  a.dL_dmean3D[3 * (size_t)idx + 2] = dmz;

  // ---- computeCov3D backward (backward.cu:278-341) ----
  if (a.scales != nullptr) {
    const float4 rot = reinterpret_cast<const float4*>(a.rotations)[idx];
    const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
    const M3 R = quat_to_R(rot);
    const float3 s = make_float3(a.scale_modifier * a.scales[3 * idx], a.scale_modifier * a.scales[3 * idx + 1],
                                 a.scale_modifier * a.scales[3 * idx + 2]);
    M3 S = {{{s.x, 0.f, 0.f}, {0.f, s.y, 0.f}, {0.f, 0.f, s.z}}};
    M3 Mm = m3_mul(S, R);
    M3 dSigma = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]}, {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
    M3 M2;
#pragma unroll
    for (int cc_ = 0; cc_ < 3; cc_++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++) M2.m[cc_][rr] = 2.0f * Mm.m[cc_][rr];
    M3 dM = m3_mul(M2, dSigma);
    M3 Rt = m3_T(R), dMt = m3_T(dM);
    a.dL_dscale[3 * (size_t)idx + 0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
    a.dL_dscale[3 * (size_t)idx + 1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
    a.dL_dscale[3 * (size_t)idx + 2] = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
    const float sv[3] = {s.x, s.y, s.z};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++) dMt.m[k][rr] *= sv[k];
#define Dm(c_, r_) dMt.m[c_][r_]
    float4 dq;
    dq.x = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
    dq.y = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) - 4 * x * (Dm(2, 2) + Dm(1, 1));
    dq.z = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) - 4 * y * (Dm(2, 2) + Dm(0, 0));
    dq.w = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) - 4 * z * (Dm(1, 1) + Dm(0, 0));
#undef Dm
    reinterpret_cast<float4*>(a.dL_drot)[idx] = dq;
  } else {
    a.dL_dscale[i3] = a.dL_dscale[i3 + 1] = a.dL_dscale[i3 + 2] = 0.f;
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace s3g

using namespace s3g;

extern "C" size_t s3g_raster_backward_workspace_bytes(int P, int R) {
  (void)P;
  return ((size_t)(R > 0 ? R : 0) * NREC * sizeof(float) + 127) & ~size_t(127);
}
extern "C" size_t s3g_raster_backward2_workspace_bytes(int P, int R) {
  (void)P;
  return ((size_t)(R > 0 ? R : 0) * (NREC + 4) * sizeof(float) + 127) & ~size_t(127);
}

template <int NX>
static int raster_backward_impl(const char* who, const s3g_raster_inputs* in, const float* colors2, int R, const int* radii,
                                const void* geometry_arena, const void* binning_arena, const void* image_arena,
                                void* workspace, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix2,
                                float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                float* dL_dcolor2, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                                float* dL_dscale, float* dL_drot, const s3g_densify_accum* dens, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (dens && !(dens->xyz_gradient_accum && dens->denom && dens->max_radii2D)) {
    set_error("%s: s3g_densify_accum needs all three arrays", who);
    return S3G_ERR_INVALID_ARG;
  }
  if (!in) {
    set_error("%s: NULL inputs", who);
    return S3G_ERR_INVALID_ARG;
  }
  const int P = in->P, W = in->width, H = in->height;
  if (P == 0) return S3G_OK;
  if (!radii || !geometry_arena || !image_arena || (R > 0 && (!binning_arena || !workspace)) || !dL_dpix ||
      !dL_dpix_depth || !dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale ||
      !dL_drot || (in->shs && !dL_dsh) || (NX && (!colors2 || !dL_dpix2 || !dL_dcolor2 || !in->colors_precomp))) {
    set_error("%s: NULL array argument", who);
    return S3G_ERR_INVALID_ARG;
  }
  const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y, tiles = gx * gy;
  const bool debug = in->debug != 0;
  GeomState g = GeomState::carve(const_cast<void*>(geometry_arena), P, nullptr);
  ImageState im = ImageState::carve(const_cast<void*>(image_arena), (size_t)W * H, tiles, bin_blocks(P), nullptr);
  BinningState b = BinningState::carve(const_cast<void*>(binning_arena), (size_t)(R > 0 ? R : 0), 0, nullptr);
  float* records = reinterpret_cast<float*>(workspace);

  const float* color_ptr = in->colors_precomp ? in->colors_precomp : g.rgb;
  if (R > 0) {
    const uint32_t tile_blocks = ((uint32_t)tiles + 7u) & ~7u;
    profile_begin(S3G_PROFILE_BLEND_BACKWARD, stream);
    hipLaunchKernelGGL(blend_backward_kernel<NX>, dim3(tile_blocks), dim3(256), 0, stream, W, H, gx, tiles, im.ranges,
                       im.tile_hi, b.point_list, in->background, g.means2D, g.conic_opacity, color_ptr, g.depths,
                       im.final_T, im.n_contrib, dL_dpix, dL_dpix_depth, records, colors2, dL_dpix2);
    profile_end(S3G_PROFILE_BLEND_BACKWARD, stream, (double)R, (double)W * H);
    S3G_KERNEL_CHECK(stream, debug);
  }
  GeomBwdArgs ga;
  ga.P = P; ga.D = in->D; ga.M = in->M; ga.means3D = in->means3D; ga.radii = radii; ga.shs = in->shs;
  ga.clamped = g.clamped; ga.scales = in->scales; ga.rotations = in->rotations; ga.scale_modifier = in->scale_modifier;
  ga.cov3Ds = in->cov3D_precomp ? in->cov3D_precomp : g.cov3D;
  ga.view = in->viewmatrix; ga.proj = in->projmatrix; ga.campos = in->cam_pos;
  ga.fy = H / (2.0f * in->tan_fovy); ga.fx = W / (2.0f * in->tan_fovx);
  ga.tan_fovx = in->tan_fovx; ga.tan_fovy = in->tan_fovy;
  ga.W = W; ga.H = H; ga.gx = gx;
  ga.rect = g.rect; ga.gauss_off = g.gauss_off; ga.slot_pos = b.slot_pos; ga.tile_hi = im.tile_hi;
  ga.records = records; ga.conic_opacity = g.conic_opacity; ga.ctrl = im.ctrl;
  ga.dL_dmean2D = dL_dmean2D; ga.dL_dconic = dL_dconic; ga.dL_dopacity = dL_dopacity; ga.dL_dcolor = dL_dcolor;
  ga.dL_dcolor2 = dL_dcolor2; ga.dL_ddepth = dL_ddepth;
  ga.dL_dmean3D = dL_dmean3D; ga.dL_dcov3D = dL_dcov3D; ga.dL_dsh = dL_dsh; ga.dL_dscale = dL_dscale; ga.dL_drot = dL_drot;
  ga.dens_accum = dens ? dens->xyz_gradient_accum : nullptr;
  ga.dens_denom = dens ? dens->denom : nullptr;
  ga.dens_max_radii = dens ? dens->max_radii2D : nullptr;
  hipLaunchKernelGGL(geometry_backward_kernel<NX>, dim3((P + 255) / 256), dim3(256), 0, stream, ga);
  S3G_KERNEL_CHECK(stream, debug);
  return S3G_OK;
}

extern "C" int s3g_raster_backward(const s3g_raster_inputs* in, int R, const int* radii, const void* geometry_arena,
                                   const void* binning_arena, const void* image_arena, void* workspace,
                                   const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D, float* dL_dconic,
                                   float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                                   float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream_) {
  return raster_backward_impl<0>("s3g_raster_backward", in, nullptr, R, radii, geometry_arena, binning_arena, image_arena,
                                 workspace, dL_dpix, dL_dpix_depth, nullptr, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                 nullptr, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, nullptr, stream_);
}
extern "C" int s3g_raster_backward_accum(const s3g_raster_inputs* in, int R, const int* radii, const void* geometry_arena,
                                         const void* binning_arena, const void* image_arena, void* workspace,
                                         const float* dL_dpix, const float* dL_dpix_depth, float* dL_dmean2D,
                                         float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                                         float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                         float* dL_drot, const s3g_densify_accum* dens, void* stream_) {
  return raster_backward_impl<0>("s3g_raster_backward_accum", in, nullptr, R, radii, geometry_arena, binning_arena,
                                 image_arena, workspace, dL_dpix, dL_dpix_depth, nullptr, dL_dmean2D, dL_dconic, dL_dopacity,
                                 dL_dcolor, nullptr, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dens,
                                 stream_);
}

extern "C" int s3g_raster_backward2(const s3g_raster_inputs* in, const float* colors2, int R, const int* radii,
                                    const void* geometry_arena, const void* binning_arena, const void* image_arena,
                                    void* workspace, const float* dL_dpix, const float* dL_dpix_depth,
                                    const float* dL_dpix2, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                    float* dL_dcolor, float* dL_dcolor2, float* dL_ddepth, float* dL_dmean3D,
                                    float* dL_dcov3D, float* dL_dscale, float* dL_drot, void* stream_) {
  return raster_backward_impl<3>("s3g_raster_backward2", in, colors2, R, radii, geometry_arena, binning_arena, image_arena,
                                 workspace, dL_dpix, dL_dpix_depth, dL_dpix2, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                 dL_dcolor2, dL_ddepth, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot, nullptr, stream_);
}
extern "C" int s3g_raster_backward2_accum(const s3g_raster_inputs* in, const float* colors2, int R, const int* radii,
                                          const void* geometry_arena, const void* binning_arena, const void* image_arena,
                                          void* workspace, const float* dL_dpix, const float* dL_dpix_depth,
                                          const float* dL_dpix2, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                          float* dL_dcolor, float* dL_dcolor2, float* dL_ddepth, float* dL_dmean3D,
                                          float* dL_dcov3D, float* dL_dscale, float* dL_drot,
                                          const s3g_densify_accum* dens, void* stream_) {
  return raster_backward_impl<3>("s3g_raster_backward2_accum", in, colors2, R, radii, geometry_arena, binning_arena,
                                 image_arena, workspace, dL_dpix, dL_dpix_depth, dL_dpix2, dL_dmean2D, dL_dconic, dL_dopacity,
                                 dL_dcolor, dL_dcolor2, dL_ddepth, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot, dens,
                                 stream_);
}
