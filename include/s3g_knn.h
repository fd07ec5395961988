/*
 * s3g_knn.h -- C ABI of the MI355X-native simple-knn replacement (libs3g.so).
 *
 *   s3g_knn_mean_dist2  <- SimpleKNN::knn      KNN/simple_knn.h:17-19, KNN/simple_knn.cu:185-221
 *                          (called by distCUDA2, KNN/spatial.cu:15-26; bound as simple_knn._C.distCUDA2, KNN/ext.cpp:15-17)
 *
 * meanDists[i] = mean of the 3 smallest SQUARED distances from point i to the other points (self excluded by index,
 * exact duplicates count at distance 0), fp32.  With fewer than 4 points the missing neighbours are FLT_MAX and the
 * mean overflows to +inf, like the reference.
 *
 * The reference allocates its scratch with cudaMalloc/thrust per call (simple_knn.cu:188-216); here the caller
 * passes one workspace of s3g_knn_workspace_bytes(P) bytes (device, 128-byte aligned, no initialisation needed).
 * No host synchronisation; all kernels run on `stream`.
 */
#ifndef S3G_KNN_H
#define S3G_KNN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t s3g_knn_workspace_bytes(int P);
int s3g_knn_mean_dist2(int P, const float* points /* [P,3] device */, float* meanDists /* [P] device */,
                       void* workspace, void* stream /* hipStream_t */);

#ifdef __cplusplus
}
#endif
#endif
