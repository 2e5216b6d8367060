// SparseConvTensor.dense() and its backward (gfx950).  Pure data movement:
// n*c elements scattered into a zeroed [B, C, D, H, W] tensor, HBM-bound on the
// zero fill of the dense tensor (the occupied fraction is ~19 % on KITTI).
#include "d3d_common.h"
#include "d3d_fill.h"

// channels_last = 1 addresses the dense tensor as [B, H, W, C*D] (channel = c*D + z): the
// memory layout of a channels-last [B, C*D, H, W] tensor, which MIOpen's NHWC conv kernels
// consume without the NCHW<->NHWC transposes (0.63 ms/step in profiles/round1_kernel_stats_v2.txt).
template <typename T, bool GATHER>
__global__ void __launch_bounds__(256)
    dense_move_kernel(T *__restrict__ feat, const int4 *__restrict__ coords, int n, int c, int D,
                      int H, int W, int channels_last, T *__restrict__ dense) {
  // one thread per (site, channel); consecutive threads walk channels of one site so
  // the sparse side is coalesced, the dense side is a C-way strided access.
  const int64_t total = (int64_t)n * c;
  const int64_t plane = (int64_t)D * H * W;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(t / c);
    int ch = (int)(t - (int64_t)i * c);
    int4 q = coords[i];
    int64_t off = channels_last
                      ? (((int64_t)q.x * H + q.z) * W + q.w) * ((int64_t)c * D) + (int64_t)ch * D + q.y
                      : ((int64_t)q.x * c + ch) * plane + ((int64_t)q.y * H + q.z) * W + q.w;
    if (GATHER)
      feat[t] = dense[off];
    else
      dense[off] = feat[t];
  }
}

static int dense_args_ok(const void *feat, const int32_t *coords, int n, int c, int batch,
                         const int32_t *shape, int dtype, const void *dense, const char *who) {
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "%s: dtype %d", who, dtype);
  D3D_REQUIRE(n >= 0 && c >= 1 && batch >= 1, "%s: n/c/batch", who);
  D3D_REQUIRE(shape[0] > 0 && shape[1] > 0 && shape[2] > 0, "%s: spatial shape", who);
  D3D_REQUIRE(dense && (n == 0 || (feat && coords)), "%s: null pointer", who);
  return D3D_OK;
}

extern "C" int d3d_dense_scatter(const void *feat, const int32_t *coords, int n, int c, int batch,
                                 const int32_t *shape, int dtype, int channels_last, void *dense,
                                 d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = dense_args_ok(feat, coords, n, c, batch, shape, dtype, dense, "d3d_dense_scatter");
  if (rc) return rc;
  const size_t esz = dtype == D3D_BF16 ? 2 : 4;
  const size_t bytes = (size_t)batch * c * shape[0] * shape[1] * shape[2] * esz;
  D3D_HIP_TRY(d3d_fill_async(stream, dense, 0, bytes));
  if (n == 0) return D3D_OK;
  int64_t items = (int64_t)n * c;
  int nb = (int)((items + 255) / 256);
  if (nb > 8192) nb = 8192;
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL((dense_move_kernel<uint16_t, false>), dim3(nb), dim3(256), 0, stream,
                       (uint16_t *)feat, (const int4 *)coords, n, c, shape[0], shape[1], shape[2],
                       channels_last, (uint16_t *)dense);
  else
    hipLaunchKernelGGL((dense_move_kernel<float, false>), dim3(nb), dim3(256), 0, stream,
                       (float *)feat, (const int4 *)coords, n, c, shape[0], shape[1], shape[2],
                       channels_last, (float *)dense);
  D3D_LAUNCH_TRY("dense_move_kernel<scatter>");
  return D3D_OK;
}

extern "C" int d3d_dense_gather(const void *dense, const int32_t *coords, int n, int c, int batch,
                                const int32_t *shape, int dtype, int channels_last, void *feat,
                                d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = dense_args_ok(feat, coords, n, c, batch, shape, dtype, dense, "d3d_dense_gather");
  if (rc) return rc;
  if (n == 0) return D3D_OK;
  int64_t items = (int64_t)n * c;
  int nb = (int)((items + 255) / 256);
  if (nb > 8192) nb = 8192;
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL((dense_move_kernel<uint16_t, true>), dim3(nb), dim3(256), 0, stream,
                       (uint16_t *)feat, (const int4 *)coords, n, c, shape[0], shape[1], shape[2],
                       channels_last, (uint16_t *)dense);
  else
    hipLaunchKernelGGL((dense_move_kernel<float, true>), dim3(nb), dim3(256), 0, stream,
                       (float *)feat, (const int4 *)coords, n, c, shape[0], shape[1], shape[2],
                       channels_last, (float *)dense);
  D3D_LAUNCH_TRY("dense_move_kernel<gather>");
  return D3D_OK;
}
