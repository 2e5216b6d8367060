// Error reporting + the scan spine shared by the voxelizer and the rulebook builder.
#include "d3d_scan.h"

static thread_local char g_d3d_err[512] = "";

int d3d_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_d3d_err, sizeof(g_d3d_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *d3d_last_error(void) { return g_d3d_err; }
extern "C" int d3d_abi_version(void) { return D3D_ABI_VERSION; }

__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    d3d_scan_spine_kernel(int *tile_sums, int ntiles, int *total_out) {
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += D3D_SCAN_THREADS) {
    int i = base + threadIdx.x;
    int v = (i < ntiles) ? tile_sums[i] : 0;
    int total;
    int excl = d3d_block_excl_scan256(v, &total);
    int carry = carry_s;
    if (i < ntiles) tile_sums[i] = carry + excl;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_sums[ntiles] = carry_s;
    if (total_out) *total_out = carry_s;
  }
}
