// Greedy keep pass shared by every bitmask NMS in this library (d3d_nms.hip, d3d_iou3d.hip).
// mask[i * nblk + w] bit j  <=>  box i suppresses box 64*w + j  (only words w >= i/64 are read).
// The reference does this pass on the host after a D2H copy of the whole mask
// (det3d/ops/nms/nms_gpu.py:109-126, det3d/ops/iou3d/src/iou3d.cpp:101-116); here a single
// wavefront walks the boxes in score order with the `removed` bit-vector in LDS.
#pragma once
#include "d3d_common.h"

#define NMS_MAX_BLK 256  // n <= 16384

// One wavefront resolves one score-sorted list.  Boxes are taken 64 at a time: the greedy decisions INSIDE a
// block of 64 only need the block's diagonal mask words (one per lane, exchanged with readlane -- no memory, no
// barrier), then the kept boxes' rows are OR-ed into the `removed` words of the later blocks, one word per lane.
// (Round 1 walked the boxes one by one with a barrier each: 1000 boxes = 1000 barriers ~ 0.5 ms per list.)
__device__ __forceinline__ int d3d_nms_reduce_list(const unsigned long long *__restrict__ mask, int n, int nblk,
                                                   int row_stride, int max_keep, int32_t *__restrict__ keep) {
  const int lane = threadIdx.x;
  // removed-words of the blocks still to come: lane w holds word w (nblk <= NMS_MAX_BLK = 256 -> 4 per lane)
  unsigned long long remv[NMS_MAX_BLK / 64];
#pragma unroll
  for (int q = 0; q < NMS_MAX_BLK / 64; ++q) remv[q] = 0ull;
  int nk = 0;
  for (int blk = 0; blk < nblk && nk < max_keep; ++blk) {
    const int base = blk * 64;
    const int cnt = n - base < 64 ? n - base : 64;
    // this block's removed word lives in lane (blk % 64), slot blk / 64
    unsigned long long cur = 0ull;
#pragma unroll
    for (int q = 0; q < NMS_MAX_BLK / 64; ++q)
      if (q == blk / 64) cur = __shfl(remv[q], blk & 63, 64);
    const unsigned long long diag = lane < cnt ? mask[(int64_t)(base + lane) * row_stride + blk] : 0ull;
    unsigned long long keepbits = 0ull;
    for (int j = 0; j < cnt && nk < max_keep; ++j) {
      const unsigned long long dj = __shfl(diag, j, 64);  // uniform j: a readlane
      if (!((cur >> j) & 1ull)) {
        keepbits |= 1ull << j;
        cur |= dj;
        if (lane == 0) keep[nk] = base + j;
        ++nk;
      }
    }
    // rows of the kept boxes -> removed words of the later blocks (word w = 64*q + lane)
    unsigned long long kb = keepbits;
    while (kb) {
      const int j = __ffsll((long long)kb) - 1;
      kb &= kb - 1;
      const unsigned long long *row = mask + (int64_t)(base + j) * row_stride;
#pragma unroll
      for (int q = 0; q < NMS_MAX_BLK / 64; ++q) {
        const int w = q * 64 + lane;
        if (w > blk && w < nblk) remv[q] |= row[w];
      }
    }
  }
  return nk;
}

static __global__ void __launch_bounds__(64)
    nms_reduce_kernel(const unsigned long long *__restrict__ mask, int n, int nblk, int max_keep,
                      int32_t *__restrict__ keep, int32_t *__restrict__ n_keep) {
  const int nk = d3d_nms_reduce_list(mask, n, nblk, nblk, max_keep, keep);
  if (threadIdx.x == 0) *n_keep = nk;
}
