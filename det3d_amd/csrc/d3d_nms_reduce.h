// Greedy keep pass shared by every bitmask NMS in this library (d3d_nms.hip, d3d_iou3d.hip).
// mask[i * nblk + w] bit j  <=>  box i suppresses box 64*w + j  (only words w >= i/64 are read).
// The reference does this pass on the host after a D2H copy of the whole mask
// (det3d/ops/nms/nms_gpu.py:109-126, det3d/ops/iou3d/src/iou3d.cpp:101-116); here a single
// wavefront walks the boxes in score order with the `removed` bit-vector in LDS.
#pragma once
#include "d3d_common.h"

#define NMS_MAX_BLK 256  // n <= 16384

static __global__ void __launch_bounds__(64)
    nms_reduce_kernel(const unsigned long long *__restrict__ mask, int n, int nblk, int max_keep,
                      int32_t *__restrict__ keep, int32_t *__restrict__ n_keep) {
  __shared__ unsigned long long remv[NMS_MAX_BLK];
  const int lane = threadIdx.x;
  for (int w = lane; w < nblk; w += 64) remv[w] = 0ull;
  __syncthreads();
  int nk = 0;
  for (int i = 0; i < n && nk < max_keep; ++i) {
    const int blk = i >> 6;
    const bool gone = (remv[blk] >> (i & 63)) & 1ull;
    if (!gone) {
      if (lane == 0) keep[nk] = i;
      ++nk;
      for (int w = blk + lane; w < nblk; w += 64) remv[w] |= mask[(int64_t)i * nblk + w];
    }
    __syncthreads();
  }
  if (lane == 0) *n_keep = nk;
}

