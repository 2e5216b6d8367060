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

// ---- contexts (d3d_context.h, include/d3d.h): knobs + the ticket words of the deterministic cross-block folds
// (d3d_fold.h).  One ring of zeroed words per (context, device), allocated on first use; a launch takes `count`
// consecutive words, its last-ticket blocks reset them to zero.
#include <stdlib.h>

#include "d3d_fold.h"

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}
static void ctx_init(D3dContext *c) {
  c->k.fwd_variant = -1;                                       // -> D3D_FWD_VARIANT at the use
  c->k.wgrad_blocks = -1;
  c->k.xcd_remap = env_int("D3D_XCD_REMAP", 0);
  c->k.bn_variant = env_int("D3D_BN_VARIANT", 9);
  c->k.bn_skip_global = 0;
  c->k.bn_dx_rows = env_int("D3D_BN_DX_ROWS", 2);
  c->k.bn_stream_blocks = env_int("D3D_BN_STREAM_BLOCKS", 512);
  c->k.c2_variant = env_int("D3D_CONV2D_VARIANT", -1);
  c->k.c2_wgrad_waves = env_int("D3D_CONV2D_WGRAD_WAVES", 16);
  c->k.c2_tail = env_int("D3D_CONV2D_TAIL", 1);
  c->k.sp_exact_f32 = env_int("D3D_SPCONV_EXACT_F32", 0);
  c->k.fwd_resident = env_int("D3D_FWD_RESIDENT", -1);
  for (int d = 0; d < D3D_MAX_DEVICES; ++d) {
    c->ring[d] = nullptr;
    c->next[d].store(0u);
  }
  c->lock.clear();
}
static D3dContext *default_ctx() {
  static D3dContext *c = [] {
    D3dContext *p = new D3dContext;
    ctx_init(p);
    return p;
  }();
  return c;
}
static thread_local D3dContext *t_ctx = nullptr;
D3dContext *d3d_ctx_cur() { return t_ctx ? t_ctx : default_ctx(); }

extern "C" int d3d_ctx_create(d3d_ctx_t *ctx) {
  D3D_REQUIRE(ctx, "d3d_ctx_create: null pointer");
  D3dContext *c = new D3dContext;
  ctx_init(c);
  *ctx = (d3d_ctx_t)c;
  return D3D_OK;
}
extern "C" int d3d_ctx_bind(d3d_ctx_t ctx) {
  t_ctx = (D3dContext *)ctx;  // null: back to the process default
  return D3D_OK;
}
extern "C" int d3d_ctx_current(d3d_ctx_t *ctx) {
  D3D_REQUIRE(ctx, "d3d_ctx_current: null pointer");
  *ctx = (d3d_ctx_t)t_ctx;
  return D3D_OK;
}
extern "C" int d3d_ctx_destroy(d3d_ctx_t ctx) {
  D3dContext *c = (D3dContext *)ctx;
  D3D_REQUIRE(c && c != default_ctx(), "d3d_ctx_destroy: null or the default context");
  if (t_ctx == c) t_ctx = nullptr;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  for (int d = 0; d < D3D_MAX_DEVICES; ++d)
    if (c->ring[d]) {  // (the caller guarantees that no launch of this context is in flight)
      if (have) (void)hipSetDevice(d);
      (void)hipFree(c->ring[d]);
    }
  if (have) (void)hipSetDevice(cur);
  delete c;
  return D3D_OK;
}
extern "C" int d3d_ctx_get_tuning(int knob, int *value) {
  D3D_REQUIRE(value && knob >= 0 && knob <= 11, "d3d_ctx_get_tuning: knob %d", knob);
  const D3dKnobs &k = d3d_ctx_cur()->k;
  const int v[12] = {k.fwd_variant, k.wgrad_blocks, k.xcd_remap, k.bn_variant, k.bn_skip_global, k.bn_dx_rows,
                     k.bn_stream_blocks, k.c2_variant, k.c2_wgrad_waves, k.c2_tail, k.sp_exact_f32, k.fwd_resident};
  *value = v[knob];
  return D3D_OK;
}

unsigned *d3d_fold_counters(int count) {
  D3dContext *c = d3d_ctx_cur();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= D3D_MAX_DEVICES) {
    d3d_set_error(D3D_EHIP, "d3d_fold_counters: hipGetDevice failed or device index %d >= %d", dev, D3D_MAX_DEVICES);
    return nullptr;
  }
  if (count < 1 || count > 1024) {
    d3d_set_error(D3D_EINVAL, "d3d_fold_counters: count=%d", count);
    return nullptr;
  }
  if (!c->ring[dev]) {
    while (c->lock.test_and_set(std::memory_order_acquire)) {}
    if (!c->ring[dev]) {
      unsigned *p = nullptr;
      // (not legal during stream capture: run one eager step before capturing, as graph_tail.py does)
      hipError_t e = hipMalloc((void **)&p, D3D_FOLD_RING_WORDS * sizeof(unsigned));
      if (e == hipSuccess) e = hipMemset(p, 0, D3D_FOLD_RING_WORDS * sizeof(unsigned));
      if (e != hipSuccess) {
        c->lock.clear(std::memory_order_release);
        d3d_set_error(D3D_EHIP, "d3d_fold_counters: allocating the ticket ring failed: %s", hipGetErrorString(e));
        return nullptr;
      }
      c->ring[dev] = p;
    }
    c->lock.clear(std::memory_order_release);
  }
  // consecutive words, never straddling the end of the ring
  unsigned at;
  while (true) {
    at = c->next[dev].fetch_add((unsigned)count, std::memory_order_relaxed) % D3D_FOLD_RING_WORDS;
    if (at + (unsigned)count <= D3D_FOLD_RING_WORDS) break;
  }
  return c->ring[dev] + at;
}
