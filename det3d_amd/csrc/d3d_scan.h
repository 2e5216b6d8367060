// Device-wide exclusive scan over a functor-defined int sequence (gfx950, wave64).
//   v[i] = f(i), i in [0,n);   g(i, v[i], exclusive_prefix(i)) is called for every i.
// Three launches: tile reduce -> spine scan (one block) -> tile scan + consumer.
// n is host-known for every user (point count, bitmap word count), so no
// device-side sizes are needed here.
#pragma once
#include "d3d_common.h"

#define D3D_SCAN_THREADS 256
#define D3D_SCAN_ITEMS 8
#define D3D_SCAN_TILE (D3D_SCAN_THREADS * D3D_SCAN_ITEMS)

__device__ __forceinline__ int d3d_wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one int per thread (blockDim.x == 256); returns
// the exclusive prefix, *block_total = sum over the block.
__device__ __forceinline__ int d3d_block_excl_scan256(int v, int *block_total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = d3d_wave_incl_scan(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < w) off += wsum[i];
  *block_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return off + inc - v;
}

template <class F>
__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    d3d_scan_reduce_kernel(F f, int64_t n, int *tile_sums) {
  const int64_t base = (int64_t)blockIdx.x * D3D_SCAN_TILE + (int64_t)threadIdx.x * D3D_SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int j = 0; j < D3D_SCAN_ITEMS; ++j) {
    int64_t i = base + j;
    if (i < n) s += f(i);
  }
  int total;
  d3d_block_excl_scan256(s, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: in-place exclusive scan of tile_sums[0..ntiles), total -> tile_sums[ntiles]
// and (optionally) *total_out.
__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    d3d_scan_spine_kernel(int *tile_sums, int ntiles, int *total_out);

template <class F, class G>
__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    d3d_scan_final_kernel(F f, G g, int64_t n, const int *tile_offsets) {
  const int64_t base = (int64_t)blockIdx.x * D3D_SCAN_TILE + (int64_t)threadIdx.x * D3D_SCAN_ITEMS;
  int v[D3D_SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int j = 0; j < D3D_SCAN_ITEMS; ++j) {
    int64_t i = base + j;
    v[j] = (i < n) ? f(i) : 0;
    s += v[j];
  }
  int total;
  int excl = d3d_block_excl_scan256(s, &total) + tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < D3D_SCAN_ITEMS; ++j) {
    int64_t i = base + j;
    if (i < n) g(i, v[j], excl);
    excl += v[j];
  }
}

static inline int d3d_scan_num_tiles(int64_t n) {
  return (int)((n + D3D_SCAN_TILE - 1) / D3D_SCAN_TILE);
}

// tile_sums must hold d3d_scan_num_tiles(n)+1 ints. total_out may be null.
template <class F, class G>
static int d3d_exclusive_scan(F f, G g, int64_t n, int *tile_sums, int *total_out,
                              hipStream_t stream) {
  const int ntiles = d3d_scan_num_tiles(n);
  if (ntiles > 0) {
    hipLaunchKernelGGL((d3d_scan_reduce_kernel<F>), dim3(ntiles), dim3(D3D_SCAN_THREADS), 0, stream,
                       f, n, tile_sums);
    D3D_LAUNCH_TRY("d3d_scan_reduce_kernel");
  }
  hipLaunchKernelGGL(d3d_scan_spine_kernel, dim3(1), dim3(D3D_SCAN_THREADS), 0, stream, tile_sums,
                     ntiles, total_out);
  D3D_LAUNCH_TRY("d3d_scan_spine_kernel");
  if (ntiles > 0) {
    hipLaunchKernelGGL((d3d_scan_final_kernel<F, G>), dim3(ntiles), dim3(D3D_SCAN_THREADS), 0,
                       stream, f, g, n, tile_sums);
    D3D_LAUNCH_TRY("d3d_scan_final_kernel");
  }
  return D3D_OK;
}
