// oracle/ref_knn_shim.cpp -- TEST INFRASTRUCTURE ONLY (see build_ref.sh).
//
// Host-pointer C entry point around the REFERENCE's own SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:185-221, hipified
// into a temporary directory by build_ref.sh and linked into oracle/_ref/libref_knn.so).  Replaces the torch wrapper
// submodules/simple-knn/spatial.cu:15-26 (`distCUDA2`): same allocation of a float[P] result, same call.  Used by
// tests/ to pin oracle/knn_oracle.c and libs3g's s3g_knn_mean_dist2 against the real thing; never loaded by the product.
#include <hip/hip_runtime.h>
#include "simple_knn.h"

extern "C" int ref_knn_mean_dist2(int P, const float* points_host, float* mean_dists_host) {
  if (P <= 0) return 0;
  float3* pts = nullptr;
  float* out = nullptr;
  if (hipMalloc(&pts, sizeof(float3) * (size_t)P) != hipSuccess) return 1;
  if (hipMalloc(&out, sizeof(float) * (size_t)P) != hipSuccess) { (void)hipFree(pts); return 1; }
  (void)hipMemcpy(pts, points_host, sizeof(float3) * (size_t)P, hipMemcpyHostToDevice);
  (void)hipMemset(out, 0, sizeof(float) * (size_t)P);   // spatial.cu:20 torch::full({P}, 0.0)
  SimpleKNN::knn(P, pts, out);
  int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
  (void)hipMemcpy(mean_dists_host, out, sizeof(float) * (size_t)P, hipMemcpyDeviceToHost);
  (void)hipFree(pts);
  (void)hipFree(out);
  return rc;
}
