// Shared helpers for the gfx950 kernels of libs3g.so (private; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/s3g_raster.h"

namespace s3g {

// hipFuncSetAttribute is per DEVICE: a process that drives several GPUs (one Python process, two `cuda:k` tensors) must
// raise the dynamic-LDS limit on each of them.  Protocol: `if (device_needs_setup(seen)) { ...S3G_HIP_CHECK(set attrs)...;
// device_setup_done(seen); }` -- the bit is set only AFTER every attribute call succeeded (a failed attempt is retried by the
// next call instead of leaving the limit low for good), two host threads racing here both set the (idempotent) attributes
// before either launches, and device ids >= 64 simply repeat the cheap attribute calls every time instead of aliasing.
inline bool device_needs_setup(const std::atomic<uint64_t>& seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 64 || (seen.load(std::memory_order_acquire) & (1ull << dev)) == 0;
}
inline void device_setup_done(std::atomic<uint64_t>& seen) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 64) seen.fetch_or(1ull << dev, std::memory_order_release);
}

constexpr int TILE_X = 16;  // reference BLOCK_X/BLOCK_Y, RAST/cuda_rasterizer/config.h:16-17
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr int WAVE = 64;
constexpr int MAX_BIN_BLOCKS = 512;   // binning workgroups (2 per CU); each keeps a histogram of ALL tiles in LDS
constexpr int MAX_TILES_LDS = 38000;  // (160 KiB - slack) / 4 B: largest tile grid the single-pass multisplit handles
constexpr int NREC = 10;              // floats per instance gradient record written by the blend backward

// number of binning workgroups / Gaussians per workgroup for a scene of P Gaussians
inline int bin_blocks(int P) {
  int nb = (P + 255) / 256;
  return nb < 1 ? 1 : (nb > MAX_BIN_BLOCKS ? MAX_BIN_BLOCKS : nb);
}
inline int bin_chunk(int P) {
  const int nb = bin_blocks(P);
  const int per = (P + nb - 1) / nb;
  return ((per + 63) / 64) * 64;   // whole waves; the binning workgroups step through their chunk with a guarded tail
}

// ---- error plumbing ------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define S3G_HIP_CHECK(expr)                                                                     \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      s3g::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return S3G_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)
// After a kernel launch: always catch launch errors; in debug also synchronise (reference CHECK_CUDA, auxiliary.h:166-173).
#define S3G_KERNEL_CHECK(stream, debug)                          \
  do {                                                           \
    S3G_HIP_CHECK(hipGetLastError());                            \
    if (debug) S3G_HIP_CHECK(hipStreamSynchronize(stream));      \
  } while (0)

// ---- optional in-library kernel timing (bench.py's roofline leg): hipEvent pairs on the launch stream ----------
// id 0 = blend_forward_kernel, id 1 = blend_backward_kernel
void profile_begin(int id, hipStream_t stream);
void profile_end(int id, hipStream_t stream, double instances, double pixels);

// ---- arena carving (128-byte aligned sub-arrays, like the reference's obtain<>(), rasterizer_impl.h) ----
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 127) & ~size_t(127);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
  size_t bytes() const { return (off + 127) & ~size_t(127); }
};

// Per-Gaussian forward state ("GeometryState").
struct GeomState {
  float* depths;          // [P]   view-space z
  float2* means2D;        // [P]   pixel coordinates
  float4* conic_opacity;  // [P]   (conic a, b, c, opacity)
  float* cov3D;           // [P,6]
  float* rgb;             // [P,3] SH->RGB result (only when shs given)
  uint8_t* clamped;       // [P,3]
  ushort4* rect;          // [P]   tile rect (min.x, min.y, max.x, max.y), zero area when culled
  uint32_t* gauss_off;    // [P]   exclusive scan of tiles_touched: first slot of the Gaussian in slot_pos[]
  uint32_t* tile_mask;    // [P]   rects of <= 32 tiles: bit k = tile k (row-major in the rect) survives the exact cull
  static GeomState carve(void* p, size_t P, size_t* bytes) {
    Carver c(p);
    GeomState g;
    g.depths = c.take<float>(P);
    g.means2D = c.take<float2>(P);
    g.conic_opacity = c.take<float4>(P);
    g.cov3D = c.take<float>(P * 6);
    g.rgb = c.take<float>(P * 3);
    g.clamped = c.take<uint8_t>(P * 3);
    g.rect = c.take<ushort4>(P);
    g.gauss_off = c.take<uint32_t>(P);
    g.tile_mask = c.take<uint32_t>(P);
    if (bytes) *bytes = c.bytes();
    return g;
  }
};

// Per-image state ("ImageState") + the per-tile bucket bookkeeping.
struct ImageState {
  float* final_T;        // [H*W]
  uint32_t* n_contrib;   // [H*W]
  uint2* ranges;         // [tiles]  [start,end) into the sorted instance list
  uint32_t* tile_count;  // [tiles]  instances per tile
  uint32_t* tile_hi;     // [tiles]  ranges[t].x + deepest contributing list position (1-based) over the tile's pixels
  uint32_t* ctrl;        // [8]      ctrl[0]=R (total instances), ctrl[1]=max instances in one tile, ctrl[2]=error flags
  uint32_t* chunk_total; // [MAX_BIN_BLOCKS] instances emitted per binning workgroup (then its exclusive prefix)
  uint32_t* table;       // [nb][tiles] per-workgroup, per-tile instance counts (then exclusive prefix over workgroups)
  static ImageState carve(void* p, size_t N, size_t tiles, size_t nb, size_t* bytes) {
    Carver c(p);
    ImageState s;
    s.final_T = c.take<float>(N);
    s.n_contrib = c.take<uint32_t>(N);
    s.ranges = c.take<uint2>(tiles);
    s.tile_count = c.take<uint32_t>(tiles);
    s.tile_hi = c.take<uint32_t>(tiles);
    s.ctrl = c.take<uint32_t>(8);
    s.chunk_total = c.take<uint32_t>(MAX_BIN_BLOCKS);
    s.table = c.take<uint32_t>(nb * tiles);
    if (bytes) *bytes = c.bytes();
    return s;
  }
};

// Per-instance state ("BinningState").
struct BinningState {
  uint64_t* keys;        // [R]  (depth bits << 32 | gaussian index), sorted ascending inside each tile range
  uint32_t* point_list;  // [R]  gaussian index, tile-major, front-to-back
  uint32_t* slot_pos;    // [S]  slot_pos[gauss_off[g] + k] = position in point_list of g's k-th tile (row-major in its rect),
                         //      0xffffffff if that tile was culled.  S = sum of rect areas >= R; LAST in the arena so that
                         //      code which only knows R (backward, reuse) still finds it.
  static BinningState carve(void* p, size_t R, size_t S, size_t* bytes) {
    Carver c(p);
    BinningState b;
    b.keys = c.take<uint64_t>(R);
    b.point_list = c.take<uint32_t>(R);
    b.slot_pos = c.take<uint32_t>(S);
    if (bytes) *bytes = c.bytes();
    return b;
  }
};

// ---- small device helpers -------------------------------------------------------------------------------
// Row-vector 4x4 matrices are indexed column-major like the reference (auxiliary.h:58-77).
__device__ __forceinline__ float3 xform_4x3(const float3 p, const float* __restrict__ M) {
  return make_float3(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                     M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]);
}
__device__ __forceinline__ float4 xform_4x4(const float3 p, const float* __restrict__ M) {
  return make_float4(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                     M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14], M[3] * p.x + M[7] * p.y + M[11] * p.z + M[15]);
}

// XCD-aware tile order: hardware places workgroup b on XCD (b % 8); give each XCD a contiguous band of
// tiles so neighbouring tiles (which share most of their Gaussians) hit the same 4 MiB L2.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t bid, uint32_t nblocks) {
  constexpr uint32_t XCDS = 8;
  const uint32_t per = (nblocks + XCDS - 1) / XCDS;
  const uint32_t t = (bid % XCDS) * per + bid / XCDS;
  return t;  // may be >= nblocks for the tail: caller must bounds-check
}

}  // namespace s3g
