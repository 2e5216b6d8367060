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

// ---- ticket words of the deterministic cross-block folds (d3d_fold.h).  One ring per device, allocated and zeroed
// on first use; a launch takes `count` consecutive words, its last-ticket blocks reset them to zero.
#include "d3d_fold.h"
#include <atomic>
#define D3D_FOLD_RING_WORDS (1 << 16)
static unsigned *g_fold_ring[16] = {};
static std::atomic<unsigned> g_fold_next[16];
static std::atomic_flag g_fold_lock = ATOMIC_FLAG_INIT;

unsigned *d3d_fold_counters(int count) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    d3d_set_error(D3D_EHIP, "d3d_fold_counters: hipGetDevice failed or device index %d >= 16", dev);
    return nullptr;
  }
  if (count < 1 || count > 1024) {
    d3d_set_error(D3D_EINVAL, "d3d_fold_counters: count=%d", count);
    return nullptr;
  }
  if (!g_fold_ring[dev]) {
    while (g_fold_lock.test_and_set(std::memory_order_acquire)) {}
    if (!g_fold_ring[dev]) {
      unsigned *p = nullptr;
      // (not legal during stream capture: run one eager step before capturing, as graph_tail.py does)
      hipError_t e = hipMalloc((void **)&p, D3D_FOLD_RING_WORDS * sizeof(unsigned));
      if (e == hipSuccess) e = hipMemset(p, 0, D3D_FOLD_RING_WORDS * sizeof(unsigned));
      if (e != hipSuccess) {
        g_fold_lock.clear(std::memory_order_release);
        d3d_set_error(D3D_EHIP, "d3d_fold_counters: allocating the ticket ring failed: %s", hipGetErrorString(e));
        return nullptr;
      }
      g_fold_ring[dev] = p;
    }
    g_fold_lock.clear(std::memory_order_release);
  }
  // consecutive words, never straddling the end of the ring
  unsigned at;
  while (true) {
    at = g_fold_next[dev].fetch_add((unsigned)count, std::memory_order_relaxed) % D3D_FOLD_RING_WORDS;
    if (at + (unsigned)count <= D3D_FOLD_RING_WORDS) break;
  }
  return g_fold_ring[dev] + at;
}
