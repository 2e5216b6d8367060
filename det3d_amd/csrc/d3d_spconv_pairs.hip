// Pair-compacted sparse convolution: table build, kernel and C entry points (kernel text in d3d_spconv_pairs.h).
#include <stdlib.h>

#include <type_traits>

#include "d3d_common.h"
#include "d3d_fold.h"
#include "d3d_spconv_shared.h"

// ======================================================== pair-compacted forward / data gradient (round 6)
#include "d3d_spconv_pairs.h"

// channel pairs with a pair-compacted tiling (f32 features in split arithmetic; the 16-channel stem and the bf16 mode
// stay on the output-stationary kernels)
extern "C" int d3d_sparse_conv_pairs_supported(int cin, int cout, int dtype) {
  if (dtype != D3D_F32 || g_sp_exact_f32) return 0;
  static const int pairs[][2] = {{32, 32}, {32, 64}, {64, 32}, {64, 64}};
  for (auto &pr : pairs)
    if (pr[0] == cin && pr[1] == cout) return 1;
  return 0;
}
static inline int pair_cap64(int n_rows) { return (n_rows + D3D_PAIR_SLAB - 1) / D3D_PAIR_SLAB * D3D_PAIR_SLAB; }
static inline size_t pair_cnt_offset(int n_rows, int kvol) { return d3d_align_up((size_t)kvol * pair_cap64(n_rows) * 4); }
extern "C" size_t d3d_pair_table_bytes(int n_rows, int kvol) {
  if (n_rows < 1 || kvol < 1 || kvol > 32) return 0;
  return pair_cnt_offset(n_rows, kvol) + 2 * d3d_align_up((size_t)(pair_cap64(n_rows) / D3D_PAIR_SLAB) * 32);  // cnt, cnt_lo
}
extern "C" int d3d_pair_table_build(const int32_t *nbr, int cap, int n_rows, int kvol, void *table, size_t table_bytes,
                                    d3d_stream_t stream_) {
  D3D_REQUIRE(kvol >= 1 && kvol <= 32, "d3d_pair_table_build: kernel volume %d (1..32)", kvol);
  D3D_REQUIRE(n_rows >= 0 && n_rows <= cap, "d3d_pair_table_build: n_rows=%d cap=%d", n_rows, cap);
  if (n_rows == 0) return D3D_OK;
  D3D_REQUIRE(nbr && table, "d3d_pair_table_build: null pointer");
  D3D_REQUIRE(table_bytes >= d3d_pair_table_bytes(n_rows, kvol), "d3d_pair_table_build: table of %zu bytes < %zu (use "
              "d3d_pair_table_bytes)", table_bytes, d3d_pair_table_bytes(n_rows, kvol));
  // input rows ride in 26 bits of an entry; the table itself is read through a 2 GB buffer descriptor
  D3D_REQUIRE((int64_t)kvol * pair_cap64(n_rows) * 4 < (int64_t)D3D_RSRC_BYTES, "d3d_pair_table_build: table of %d x %d entries exceeds 2 GB",
              kvol, n_rows);
  const int nslab = pair_cap64(n_rows) / D3D_PAIR_SLAB;
  hipLaunchKernelGGL(pair_table_build_kernel, dim3((nslab + 3) / 4), dim3(256), 0, (hipStream_t)stream_, nbr, cap,
                     (const int32_t *)nullptr, n_rows, kvol, pair_cap64(n_rows), (uint32_t *)table,
                     (uint8_t *)table + pair_cnt_offset(n_rows, kvol),
                     (uint8_t *)table + pair_cnt_offset(n_rows, kvol) + d3d_align_up((size_t)nslab * 32));
  D3D_LAUNCH_TRY("pair_table_build_kernel");
  return D3D_OK;
}
extern "C" int d3d_sparse_conv_fwd_pairs(const void *feat, const void *wpack, const float *bias, const void *table,
                                         int n_out, int kvol, int cin, int cout, int dtype, void *out,
                                         const D3dConvEpilogue *epilogue, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvEpiDev ep = {};
  if (epilogue) *(D3dConvEpilogue *)&ep = *epilogue;
  ep.row_plan = nullptr;
  D3D_REQUIRE(d3d_sparse_conv_pairs_supported(cin, cout, dtype), "d3d_sparse_conv_fwd_pairs: no pair-compacted tiling for "
              "channels %d -> %d, dtype %d (ask d3d_sparse_conv_pairs_supported)", cin, cout, dtype);
  D3D_REQUIRE(!(ep.stats && ep.bn_partial), "d3d_sparse_conv_fwd_pairs: one reduction per launch (stats or bn_partial)");
  D3D_REQUIRE(!ep.bn_partial || (ep.bn_x && ep.bn_mean && ep.bn_invstd), "d3d_sparse_conv_fwd_pairs: bn_partial needs bn_x, bn_mean, bn_invstd");
  D3D_REQUIRE(n_out >= 0 && kvol >= 1 && kvol <= 32, "d3d_sparse_conv_fwd_pairs: n_out=%d kvol=%d", n_out, kvol);
  if (n_out == 0) return D3D_OK;
  D3D_REQUIRE(feat && wpack && table && out, "d3d_sparse_conv_fwd_pairs: null pointer");
  D3D_REQUIRE((int64_t)n_out * cout * 4 < (int64_t)D3D_RSRC_BYTES, "d3d_sparse_conv_fwd_pairs: %d x %d outputs exceed 2 GB", n_out, cout);
  const uint32_t *ctab = (const uint32_t *)table;
  const uint8_t *cnt = (const uint8_t *)table + pair_cnt_offset(n_out, kvol);
  const int cap64 = pair_cap64(n_out);
  const uint8_t *cnt_lo = cnt + d3d_align_up((size_t)(cap64 / D3D_PAIR_SLAB) * 32);
  // 8-wave blocks (512 rows, one block per CU) where they fill the chip, 4-wave blocks on the small levels
  const bool wide = (n_out + 511) / 512 >= 192;
  // tiling: D3D_PAIRS_TILING (tools/bench_pairs.py): 0 = 64-row slabs, 8-wave blocks where they fill the chip else 4-wave
  // blocks, weight fragments in registers (round 6, first version); 1 = 32 rows per wave, 16-wave blocks (four waves per
  // SIMD, <= 128 registers), weight fragments from LDS per k-step; 2 = 32 rows per wave, 8-wave blocks
  static const int tiling = getenv("D3D_PAIRS_TILING") ? atoi(getenv("D3D_PAIRS_TILING")) : 0;
  static const int force_w = getenv("D3D_PAIRS_WAVES") ? atoi(getenv("D3D_PAIRS_WAVES")) : 0;
  const bool w8 = force_w ? force_w == 8 : wide;
#define PAIRS_CASE(CI, CO)                                                                                                    \
  if (cin == CI && cout == CO) {                                                                                              \
    if (tiling == 1) return launch_pairs<CI, CO, 16, 2, 32, false>(feat, wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep, stream); \
    if (tiling == 2) return launch_pairs<CI, CO, 8, 2, 32, false>(feat, wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep, stream);  \
    if (tiling == 3) return launch_pairs<CI, CO, 16, 1, 32, false>(feat, wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep, stream); \
    if (w8) return launch_pairs<CI, CO, 8, 3>(feat, wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep, stream);     \
    return launch_pairs<CI, CO, 4, 3>(feat, wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep, stream);             \
  }
  PAIRS_CASE(32, 32)
  PAIRS_CASE(32, 64)
  PAIRS_CASE(64, 32)
  PAIRS_CASE(64, 64)
#undef PAIRS_CASE
  return d3d_set_error(D3D_EINVAL, "d3d_sparse_conv_fwd_pairs: channels %d -> %d", cin, cout);
}

