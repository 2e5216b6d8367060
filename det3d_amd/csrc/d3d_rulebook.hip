// Rulebook construction for submanifold and strided sparse 3D convolution (gfx950).
//
// Output-stationary neighbour tables instead of spconv's (in,out) pair lists:
//   nbr[k*cap + o] = input row feeding output row o through kernel offset k, or -1
// so the arithmetic kernel writes every output row exactly once (no atomics, no
// scatter-add, deterministic) and the gather is the only irregular access.
//
//  * submanifold: key->row hash table (fits L2), one probe per (site, offset)
//  * strided:     candidate output cells are marked in a bitmap of the output grid
//                 (atomicOr); an exclusive scan over per-word popcounts both
//                 dedupes and SORTS them -- rank(cell) = prefix[word] + popc(low bits)
//                 is the output row, ascending in ((b*D+z)*H+y)*W+x (the canonical
//                 order of SURVEY.md App. A.3) -- no sort kernel, no dense int grid
//                 (the reference's spconv v1 fills a B*D*H*W int32 grid: 1.5 GB at
//                 KITTI batch 4; the bitmap is 46 MB at full resolution, 6 MB after
//                 the first stride).
// Input counts are read from device memory so voxelize -> rulebooks -> next level
// chains on one stream with no host round trip.  All of it is integer, HBM/L2-bound.
#include "d3d_scan.h"
#include <stdlib.h>
#include "d3d_fill.h"
#include "d3d_rowplan.h"

struct RbGeom {
  int batch;
  int in_shape[3];   // z, y, x
  int out_shape[3];  // z, y, x
  int k[3];
  int s[3];
  int p[3];
  int kvol;
};

__device__ __forceinline__ uint32_t rb_lin(int b, int z, int y, int x, const int *shape) {
  return (uint32_t)((((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x);
}

// wave-aggregated pair counting: one atomic per wave when the wave shares one k.  Optional
// (pair_count == NULL skips it): kvol hot addresses serialise ~30k same-address atomics per
// launch at L2, which was >50 % of the rulebook kernels' time (profiles/round1_kernel_stats_v1.txt).
__device__ __forceinline__ void rb_count_pairs(int32_t *pair_count, int k, bool found) {
  if (pair_count == nullptr) return;
  unsigned long long m = __ballot(found);
  int k0 = __builtin_amdgcn_readfirstlane(k);
  bool uniform = __ballot(k != k0) == 0ull;
  if (uniform) {
    if ((threadIdx.x & 63) == (__ffsll((long long)__ballot(1)) - 1) && m)
      atomicAdd(&pair_count[k0], __popcll(m));
  } else if (found) {
    atomicAdd(&pair_count[k], 1);
  }
}

// ------------------------------------------------------------- submanifold
// key -> row table with packed 8-byte (key,row) entries (one load per probe) and the
// mirror symmetry of odd kernels: input j feeds output i through offset k  <=>  i feeds j
// through kvol-1-k, so only offsets k < kvol/2 are probed and both table entries are written.
__global__ void __launch_bounds__(256)
    rb_hash_build_kernel(const int4 *coords, const int32_t *n_dev, int cap, RbGeom G,
                         unsigned long long *tab, uint32_t hmask) {
  int n = *n_dev;
  n = n < cap ? n : cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int4 c = coords[i];
    d3d_kv_insert(tab, hmask, rb_lin(c.x, c.y, c.z, c.w, G.in_shape), (uint32_t)i);
  }
}

__global__ void __launch_bounds__(256)
    rb_subm_lookup_kernel(const int4 *coords, const int32_t *n_dev, int cap, RbGeom G,
                          const unsigned long long *tab, uint32_t hmask, int32_t *nbr,
                          int32_t *pair_count) {
  int n = *n_dev;
  n = n < cap ? n : cap;
  const int half = G.kvol / 2;  // centre offset
  const int64_t total = (int64_t)n * (half + 1);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(t / n);
    int i = (int)(t - (int64_t)k * n);
    if (k == half) {  // centre: every site maps to itself
      nbr[(int64_t)k * cap + i] = i;
      rb_count_pairs(pair_count, k, true);
      continue;
    }
    int4 c = coords[i];
    int kz = k / (G.k[1] * G.k[2]);
    int ky = (k / G.k[2]) % G.k[1];
    int kx = k % G.k[2];
    // cross-correlation: input = output + (k - k/2)   
    int z = c.y + kz - G.k[0] / 2;
    int y = c.z + ky - G.k[1] / 2;
    int x = c.w + kx - G.k[2] / 2;
    int r = -1;
    if (z >= 0 && z < G.in_shape[0] && y >= 0 && y < G.in_shape[1] && x >= 0 &&
        x < G.in_shape[2])
      r = d3d_kv_find(tab, hmask, rb_lin(c.x, z, y, x, G.in_shape));
    nbr[(int64_t)k * cap + i] = r;
    if (r >= 0) nbr[(int64_t)(G.kvol - 1 - k) * cap + r] = i;  // mirror pair
    rb_count_pairs(pair_count, k, r >= 0);
  }
}

__global__ void rb_mirror_counts_kernel(int32_t *pair_count, int kvol) {
  int k = threadIdx.x;
  if (k < kvol / 2) pair_count[kvol - 1 - k] = pair_count[k];
}

extern "C" size_t d3d_subm_rulebook_workspace_bytes(int cap) {
  size_t h = (size_t)d3d_next_pow2((int64_t)cap * 2 < 1024 ? 1024 : (int64_t)cap * 2);
  return d3d_align_up(h * 8) + 256;
}

extern "C" int d3d_subm_rulebook(const int32_t *coords, const int32_t *n_dev, int cap, int batch,
                                 const int32_t *spatial_shape, const int32_t *ksize, int32_t *nbr,
                                 int32_t *pair_count, void *workspace, size_t workspace_bytes,
                                 d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(cap >= 1 && batch >= 1, "d3d_subm_rulebook: cap/batch");
  D3D_REQUIRE(coords && n_dev && nbr, "d3d_subm_rulebook: null pointer");
  RbGeom G;
  G.batch = batch;
  for (int j = 0; j < 3; ++j) {
    D3D_REQUIRE(spatial_shape[j] > 0 && ksize[j] > 0 && (ksize[j] & 1),
                "d3d_subm_rulebook: shape/ksize[%d] (kernel must be odd)", j);
    G.in_shape[j] = G.out_shape[j] = spatial_shape[j];
    G.k[j] = ksize[j];
    G.s[j] = 1;
    G.p[j] = ksize[j] / 2;
  }
  G.kvol = ksize[0] * ksize[1] * ksize[2];
  D3D_REQUIRE(G.kvol <= 1024, "d3d_subm_rulebook: kernel volume %d too large", G.kvol);
  double cells = (double)batch * spatial_shape[0] * spatial_shape[1] * spatial_shape[2];
  if (cells >= 4294967295.0)
    return d3d_set_error(D3D_ECAPACITY, "d3d_subm_rulebook: %.0f cells exceed u32 keys", cells);
  if (workspace_bytes < d3d_subm_rulebook_workspace_bytes(cap))
    return d3d_set_error(D3D_ECAPACITY, "d3d_subm_rulebook: workspace too small");
  const int hsize = d3d_next_pow2((int64_t)cap * 2 < 1024 ? 1024 : (int64_t)cap * 2);
  D3dArena A(workspace, workspace_bytes);
  unsigned long long *tab = A.take<unsigned long long>(hsize);
  const int half = G.kvol / 2;
  {
    // offsets above the centre are only written through the mirror rule: default them to -1
    const D3dFillSeg segs[3] = {{tab, 0xFF, (size_t)hsize * 8},
                                {pair_count, 0, pair_count ? (size_t)G.kvol * 4 : 0},
                                {nbr + (size_t)(half + 1) * cap, 0xFF, G.kvol > 1 ? (size_t)half * cap * 4 : 0}};
    D3D_HIP_TRY(d3d_fill_async(stream, segs, 3));
  }
  int nb = (cap + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(rb_hash_build_kernel, dim3(nb), dim3(256), 0, stream, (const int4 *)coords,
                     n_dev, cap, G, tab, (uint32_t)(hsize - 1));
  D3D_LAUNCH_TRY("rb_hash_build_kernel");
  int64_t items = (int64_t)cap * (half + 1);
  int lb = (int)((items + 255) / 256);
  if (lb > 2048) lb = 2048;
  hipLaunchKernelGGL(rb_subm_lookup_kernel, dim3(lb), dim3(256), 0, stream, (const int4 *)coords,
                     n_dev, cap, G, tab, (uint32_t)(hsize - 1), nbr, pair_count);
  D3D_LAUNCH_TRY("rb_subm_lookup_kernel");
  if (G.kvol > 1 && pair_count) {
    hipLaunchKernelGGL(rb_mirror_counts_kernel, dim3(1), dim3(512), 0, stream, pair_count, G.kvol);
    D3D_LAUNCH_TRY("rb_mirror_counts_kernel");
  }
  return D3D_OK;
}

// ------------------------------------------------------------------ strided
// output cell reached from input (z,y,x) through kernel offset (kz,ky,kx):
// o = (i + p - k) / s, exact division, inside the output grid.
__device__ __forceinline__ bool rb_out_cell(const RbGeom &G, int4 c, int k, uint32_t &key) {
  int kk[3] = {k / (G.k[1] * G.k[2]), (k / G.k[2]) % G.k[1], k % G.k[2]};
  int in[3] = {c.y, c.z, c.w};
  int o[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int t = in[j] + G.p[j] - kk[j];
    if (t < 0) return false;
    int q = t / G.s[j];
    if (q * G.s[j] != t || q >= G.out_shape[j]) return false;
    o[j] = q;
  }
  key = rb_lin(c.x, o[0], o[1], o[2], G.out_shape);
  return true;
}

__global__ void __launch_bounds__(256)
    rb_conv_mark_kernel(const int4 *coords, const int32_t *n_dev, int cap, RbGeom G,
                        uint32_t *bitmap) {
  int n = *n_dev;
  n = n < cap ? n : cap;
  const int64_t total = (int64_t)n * G.kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(t / n);
    int i = (int)(t - (int64_t)k * n);
    uint32_t key;
    if (rb_out_cell(G, coords[i], k, key)) atomicOr(&bitmap[key >> 5], 1u << (key & 31));
  }
}

struct RbWordPopc {
  const uint32_t *bitmap;
  __device__ int operator()(int64_t w) const { return __popc(bitmap[w]); }
};

struct RbEmitCoords {
  const uint32_t *bitmap;
  int32_t *wordprefix;
  int4 *coords_out;
  int cap_out;
  int oshape[3];
  __device__ void operator()(int64_t w, int v, int excl) const {
    wordprefix[w] = excl;
    if (!v) return;
    uint32_t bits = bitmap[w];
    int row = excl;
    while (bits) {
      int bit = __ffs((int)bits) - 1;
      bits &= bits - 1;
      if (row < cap_out) {
        uint32_t key = ((uint32_t)w << 5) | (uint32_t)bit;
        int x = (int)(key % (uint32_t)oshape[2]);
        uint32_t r = key / (uint32_t)oshape[2];
        int y = (int)(r % (uint32_t)oshape[1]);
        r /= (uint32_t)oshape[1];
        int z = (int)(r % (uint32_t)oshape[0]);
        int b = (int)(r / (uint32_t)oshape[0]);
        coords_out[row] = make_int4(b, z, y, x);
      }
      ++row;
    }
  }
};

// The bitmap scan with 16-byte accesses: a thread owns 8 consecutive words (two uint4 loads, two int4 stores of
// their prefixes), a block the same 2048-word tile as d3d_scan.h, so the spine kernel and the tile count are
// shared.  The functor version above reads and writes one dword per lane at a 32-byte stride (every load
// instruction touches 16 cache lines for 256 useful bytes) and was 50 us per launch at KITTI scale -- more than the
// sparse convolution it prepares.  The word count is padded to a multiple of 8 (the arena aligns to 256 bytes and
// the fill zeroes the padding).
__device__ __forceinline__ void rb_load8(const uint32_t *bitmap, int64_t base, int64_t words8, uint32_t (&w)[8]) {
  if (base < words8) {
    const uint4 a = *(const uint4 *)(bitmap + base), b = *(const uint4 *)(bitmap + base + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = 0u;
  }
}

__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    rb_scan_reduce_vec_kernel(const uint32_t *__restrict__ bitmap, int64_t words8, int *__restrict__ tile_sums) {
  static_assert(D3D_SCAN_ITEMS == 8, "8 words per thread");
  const int64_t base = (int64_t)blockIdx.x * D3D_SCAN_TILE + (int64_t)threadIdx.x * 8;
  uint32_t w[8];
  rb_load8(bitmap, base, words8, w);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += __popc(w[j]);
  int total;
  d3d_block_excl_scan256(s, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(D3D_SCAN_THREADS)
    rb_scan_final_vec_kernel(const uint32_t *__restrict__ bitmap, int64_t words8, const int *__restrict__ tile_offsets,
                             RbEmitCoords emit) {
  const int64_t base = (int64_t)blockIdx.x * D3D_SCAN_TILE + (int64_t)threadIdx.x * 8;
  uint32_t w[8];
  rb_load8(bitmap, base, words8, w);
  int v[8], s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = __popc(w[j]);
    s += v[j];
  }
  int total;
  int excl = d3d_block_excl_scan256(s, &total) + tile_offsets[blockIdx.x];
  if (base >= words8) return;
  int pre[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pre[j] = excl;
    excl += v[j];
  }
  *(int4 *)(emit.wordprefix + base) = make_int4(pre[0], pre[1], pre[2], pre[3]);
  *(int4 *)(emit.wordprefix + base + 4) = make_int4(pre[4], pre[5], pre[6], pre[7]);
  if (!s) return;
  const uint32_t W = (uint32_t)emit.oshape[2], H = (uint32_t)emit.oshape[1], D = (uint32_t)emit.oshape[0];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t bits = w[j];
    if (!bits) continue;
    // decode the word's first cell once (three integer divisions), then walk its 32 cells by carries: the functor
    // version divides per emitted coordinate, which is what its 50 us per launch were spent on
    const uint32_t key0 = (uint32_t)(base + j) << 5;
    const uint32_t x0 = key0 % W;
    uint32_t r = key0 / W;
    const uint32_t y0 = r % H;
    r /= H;
    const uint32_t z0 = r % D, b0 = r / D;
    int row = pre[j];
    while (bits) {
      const int bit = __ffs((int)bits) - 1;
      bits &= bits - 1;
      if (row < emit.cap_out) {
        uint32_t x = x0 + (uint32_t)bit, y = y0, z = z0, b = b0;
        while (x >= W) { x -= W; ++y; }
        while (y >= H) { y -= H; ++z; }
        while (z >= D) { z -= D; ++b; }
        emit.coords_out[row] = make_int4((int)b, (int)z, (int)y, (int)x);
      }
      ++row;
    }
  }
}

__global__ void __launch_bounds__(256)
    rb_fill_neg1_kernel(int32_t *nbr, const int32_t *n_dev, int cap, int kvol) {
  int n = *n_dev;
  n = n < cap ? n : cap;
  const int64_t total = (int64_t)n * kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(t / n);
    int i = (int)(t - (int64_t)k * n);
    nbr[(int64_t)k * cap + i] = -1;
  }
}

__global__ void __launch_bounds__(256)
    rb_conv_fill_kernel(const int4 *coords, const int32_t *n_dev, int cap_in, RbGeom G,
                        const uint32_t *bitmap, const int32_t *wordprefix, int cap_out,
                        int32_t *nbr_out, int32_t *nbr_in, int32_t *pair_count) {
  int n = *n_dev;
  n = n < cap_in ? n : cap_in;
  const int64_t total = (int64_t)n * G.kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(t / n);
    int i = (int)(t - (int64_t)k * n);
    uint32_t key;
    int row = -1;
    if (rb_out_cell(G, coords[i], k, key)) {
      uint32_t w = key >> 5;
      row = wordprefix[w] + __popc(bitmap[w] & ((1u << (key & 31)) - 1u));
      if (row < cap_out)
        nbr_out[(int64_t)k * cap_out + row] = i;
      else
        row = -1;
    }
    nbr_in[(int64_t)k * cap_in + i] = row;
    rb_count_pairs(pair_count, k, row >= 0);
  }
}

static size_t rb_conv_words(int batch, const int32_t *out_shape) {
  double cells = (double)batch * out_shape[0] * out_shape[1] * out_shape[2];
  return ((size_t)((cells + 31) / 32) + 7) / 8 * 8;  // padded to the 8 words a scan thread owns
}

extern "C" size_t d3d_conv_rulebook_workspace_bytes(int batch, const int32_t *out_shape) {
  size_t words = rb_conv_words(batch, out_shape);
  return d3d_align_up(words * 4) * 2 + d3d_align_up(((size_t)d3d_scan_num_tiles(words) + 1) * 4) +
         256;
}

extern "C" int d3d_conv_rulebook(const int32_t *coords_in, const int32_t *n_in_dev, int cap_in,
                                 int batch, const int32_t *in_shape, const int32_t *ksize,
                                 const int32_t *stride, const int32_t *padding,
                                 const int32_t *out_shape, int32_t *coords_out,
                                 int32_t *n_out_dev, int cap_out, int32_t *nbr_out,
                                 int32_t *nbr_in, int32_t *pair_count, void *workspace,
                                 size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(cap_in >= 1 && cap_out >= 1 && batch >= 1, "d3d_conv_rulebook: capacities");
  D3D_REQUIRE(coords_in && n_in_dev && coords_out && n_out_dev && nbr_out && nbr_in,
              "d3d_conv_rulebook: null pointer");
  RbGeom G;
  G.batch = batch;
  for (int j = 0; j < 3; ++j) {
    D3D_REQUIRE(in_shape[j] > 0 && ksize[j] > 0 && stride[j] > 0 && padding[j] >= 0,
                "d3d_conv_rulebook: bad geometry on axis %d", j);
    int expect = (in_shape[j] + 2 * padding[j] - ksize[j]) / stride[j] + 1;
    D3D_REQUIRE(out_shape[j] == expect, "d3d_conv_rulebook: out_shape[%d]=%d, expected %d", j,
                out_shape[j], expect);
    G.in_shape[j] = in_shape[j];
    G.out_shape[j] = out_shape[j];
    G.k[j] = ksize[j];
    G.s[j] = stride[j];
    G.p[j] = padding[j];
  }
  G.kvol = ksize[0] * ksize[1] * ksize[2];
  double cells = (double)batch * out_shape[0] * out_shape[1] * out_shape[2];
  if (cells >= 4294967295.0)
    return d3d_set_error(D3D_ECAPACITY, "d3d_conv_rulebook: %.0f cells exceed u32 keys", cells);
  if (workspace_bytes < d3d_conv_rulebook_workspace_bytes(batch, out_shape))
    return d3d_set_error(D3D_ECAPACITY, "d3d_conv_rulebook: workspace too small");
  const size_t words = rb_conv_words(batch, out_shape);
  D3dArena A(workspace, workspace_bytes);
  uint32_t *bitmap = A.take<uint32_t>(words);
  int32_t *wordprefix = A.take<int32_t>(words);
  int *tiles = A.take<int>((size_t)d3d_scan_num_tiles(words) + 1);
  if (!tiles) return d3d_set_error(D3D_ECAPACITY, "d3d_conv_rulebook: workspace carve failed");

  {
    const D3dFillSeg segs[2] = {{bitmap, 0, (size_t)words * 4}, {pair_count, 0, pair_count ? (size_t)G.kvol * 4 : 0}};
    D3D_HIP_TRY(d3d_fill_async(stream, segs, 2));
  }
  int64_t items = (int64_t)cap_in * G.kvol;
  int nb = (int)((items + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(rb_conv_mark_kernel, dim3(nb), dim3(256), 0, stream,
                     (const int4 *)coords_in, n_in_dev, cap_in, G, bitmap);
  D3D_LAUNCH_TRY("rb_conv_mark_kernel");
  RbWordPopc F{bitmap};
  RbEmitCoords E;
  E.bitmap = bitmap;
  E.wordprefix = wordprefix;
  E.coords_out = (int4 *)coords_out;
  E.cap_out = cap_out;
  for (int j = 0; j < 3; ++j) E.oshape[j] = out_shape[j];
  static const bool scan_functor = getenv("D3D_RB_SCAN_FUNCTOR") && atoi(getenv("D3D_RB_SCAN_FUNCTOR"));
  if (scan_functor) {
    int rc = d3d_exclusive_scan(F, E, (int64_t)words, tiles, n_out_dev, stream);
    if (rc) return rc;
  } else {
    const int ntiles = d3d_scan_num_tiles((int64_t)words);
    hipLaunchKernelGGL(rb_scan_reduce_vec_kernel, dim3(ntiles), dim3(D3D_SCAN_THREADS), 0, stream, bitmap,
                       (int64_t)words, tiles);
    D3D_LAUNCH_TRY("rb_scan_reduce_vec_kernel");
    hipLaunchKernelGGL(d3d_scan_spine_kernel, dim3(1), dim3(D3D_SCAN_THREADS), 0, stream, tiles, ntiles, n_out_dev);
    D3D_LAUNCH_TRY("d3d_scan_spine_kernel");
    hipLaunchKernelGGL(rb_scan_final_vec_kernel, dim3(ntiles), dim3(D3D_SCAN_THREADS), 0, stream, bitmap,
                       (int64_t)words, tiles, E);
    D3D_LAUNCH_TRY("rb_scan_final_vec_kernel");
  }
  int64_t oitems = (int64_t)cap_out * G.kvol;
  int fb = (int)((oitems + 255) / 256);
  if (fb > 2048) fb = 2048;
  hipLaunchKernelGGL(rb_fill_neg1_kernel, dim3(fb), dim3(256), 0, stream, nbr_out, n_out_dev,
                     cap_out, G.kvol);
  D3D_LAUNCH_TRY("rb_fill_neg1_kernel");
  hipLaunchKernelGGL(rb_conv_fill_kernel, dim3(nb), dim3(256), 0, stream, (const int4 *)coords_in,
                     n_in_dev, cap_in, G, bitmap, wordprefix, cap_out, nbr_out, nbr_in,
                     pair_count);
  D3D_LAUNCH_TRY("rb_conv_fill_kernel");
  return D3D_OK;
}

// ===================================================================== row plan of a strided data gradient
// (d3d_rowplan.h)  Three launches on the geometry stream: class histogram per 1024 rows, one-block scan (which also
// writes the class boundaries and the offset lists), stable scatter.  No atomics: the plan is a pure function of the
// coordinates, so the summation order of the planned launch's BatchNorm epilogue is reproducible.
struct PlanGeom {
  int k[3], s[3], p[3];
  int ncls, kvol;
};

__device__ __forceinline__ int plan_class(const PlanGeom &G, int4 c) {
  const int in[3] = {c.y, c.z, c.w};
  int cls = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) cls = cls * G.s[j] + (in[j] + G.p[j]) % G.s[j];
  return cls;
}

// wave counts of each class in wc[wave][class]; returns this row's rank among the rows of its class in its wave
__device__ __forceinline__ int plan_wave_rank(int cls, int (*wc)[D3D_PLAN_MAX_CLASSES]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int rank = 0;
#pragma unroll
  for (int c = 0; c < D3D_PLAN_MAX_CLASSES; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wc[wave][c] = __popcll(m);
  }
  return rank;
}

__global__ void __launch_bounds__(D3D_PLAN_SORT_ROWS)
    dplan_hist_kernel(const int4 *coords, const int32_t *n_dev, int cap, PlanGeom G, int32_t *blockhist) {
  __shared__ int wc[D3D_PLAN_SORT_ROWS / 64][D3D_PLAN_MAX_CLASSES];
  int n = *n_dev;
  n = n < cap ? n : cap;
  if ((int)blockIdx.x * D3D_PLAN_SORT_ROWS >= n) return;  // capacity-sized grid: the scan only reads blocks with rows
  const int r = blockIdx.x * D3D_PLAN_SORT_ROWS + threadIdx.x;
  plan_wave_rank(r < n ? plan_class(G, coords[r]) : -1, wc);
  __syncthreads();
  if (threadIdx.x < D3D_PLAN_MAX_CLASSES) {
    int s = 0;
    for (int w = 0; w < D3D_PLAN_SORT_ROWS / 64; ++w) s += wc[w][threadIdx.x];
    blockhist[blockIdx.x * D3D_PLAN_MAX_CLASSES + threadIdx.x] = s;
  }
}

// wave c turns the block counts of class c into exclusive offsets inside the class; then the header of the plan
__global__ void __launch_bounds__(64 * D3D_PLAN_MAX_CLASSES)
    dplan_scan_kernel(int32_t *blockhist, const int32_t *n_dev, int cap, PlanGeom G, int32_t *plan) {
  __shared__ int tot[D3D_PLAN_MAX_CLASSES];
  int n = *n_dev;
  n = n < cap ? n : cap;
  const int nb = (n + D3D_PLAN_SORT_ROWS - 1) / D3D_PLAN_SORT_ROWS;
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 64) {
    const int b = b0 + lane;
    const int v = b < nb ? blockhist[b * D3D_PLAN_MAX_CLASSES + c] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (b < nb) blockhist[b * D3D_PLAN_MAX_CLASSES + c] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) tot[c] = carry;
  __syncthreads();
  if (threadIdx.x == 0) {
    plan[0] = G.ncls;
    int s = 0;
    for (int j = 0; j <= D3D_PLAN_MAX_CLASSES; ++j) {
      plan[D3D_PLAN_START + j] = s;
      if (j < G.ncls) s += tot[j];
    }
  }
  if ((int)threadIdx.x <= G.ncls) {
    int cls = (int)threadIdx.x;
    int32_t *L = plan + D3D_PLAN_LISTS + cls * D3D_PLAN_LIST_WORDS;
    const bool every = cls == G.ncls;
    int r[3];
    for (int j = 2; j >= 0; --j) {
      r[j] = cls % G.s[j];
      cls /= G.s[j];
    }
    int cnt = 0;
    unsigned long long packed[2] = {0ull, 0ull};
    for (int k = 0; k < G.kvol; ++k) {
      const int kk[3] = {k / (G.k[1] * G.k[2]), (k / G.k[2]) % G.k[1], k % G.k[2]};
      bool ok = true;
      for (int j = 0; j < 3; ++j) ok &= every || kk[j] % G.s[j] == r[j];
      if (ok) {
        if (cnt < 24) packed[cnt / 12] |= (unsigned long long)k << (5 * (cnt % 12));
        L[1 + cnt++] = k;
      }
    }
    if (cnt == 0) L[1 + cnt++] = 0;  // a class no offset reaches (kernel smaller than the stride): one empty stage
    L[0] = cnt;
    *(unsigned long long *)(L + D3D_PLAN_LIST_PACKED) = packed[0];
    *(unsigned long long *)(L + D3D_PLAN_LIST_PACKED + 2) = packed[1];
  }
}

__global__ void __launch_bounds__(D3D_PLAN_SORT_ROWS)
    dplan_scatter_kernel(const int4 *coords, const int32_t *n_dev, int cap, PlanGeom G, const int32_t *blockoff,
                         int32_t *plan) {
  __shared__ int wc[D3D_PLAN_SORT_ROWS / 64][D3D_PLAN_MAX_CLASSES];
  int n = *n_dev;
  n = n < cap ? n : cap;
  if ((int)blockIdx.x * D3D_PLAN_SORT_ROWS >= n) return;
  const int r = blockIdx.x * D3D_PLAN_SORT_ROWS + threadIdx.x;
  const int cls = r < n ? plan_class(G, coords[r]) : -1;
  const int rank = plan_wave_rank(cls, wc);
  __syncthreads();
  if (cls < 0) return;
  int pre = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += wc[w][cls];
  plan[D3D_PLAN_PERM + plan[D3D_PLAN_START + cls] + blockoff[blockIdx.x * D3D_PLAN_MAX_CLASSES + cls] + pre + rank] = r;
}

static int plan_blocks(int cap_in) { return (cap_in + D3D_PLAN_SORT_ROWS - 1) / D3D_PLAN_SORT_ROWS; }

extern "C" size_t d3d_dgrad_plan_ints(int cap_in) {
  if (cap_in < 1) return 0;
  return (size_t)D3D_PLAN_PERM + (size_t)cap_in + (size_t)plan_blocks(cap_in) * D3D_PLAN_MAX_CLASSES;
}

extern "C" int d3d_dgrad_plan_supported(const int32_t *ksize, const int32_t *stride) {
  if (!ksize || !stride) return 0;
  const int ncls = stride[0] * stride[1] * stride[2];
  const int kvol = ksize[0] * ksize[1] * ksize[2];
  for (int j = 0; j < 3; ++j)
    if (ksize[j] < 1 || stride[j] < 1) return 0;
  return ncls >= 2 && ncls <= D3D_PLAN_MAX_CLASSES && kvol <= D3D_PLAN_MAX_KVOL;
}

extern "C" int d3d_dgrad_plan_build(const int32_t *coords_in, const int32_t *n_in_dev, int cap_in,
                                    const int32_t *ksize, const int32_t *stride, const int32_t *padding,
                                    int32_t *plan, size_t plan_ints, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(coords_in && n_in_dev && ksize && stride && padding && plan, "d3d_dgrad_plan_build: null pointer");
  D3D_REQUIRE(cap_in >= 1, "d3d_dgrad_plan_build: cap_in %d", cap_in);
  D3D_REQUIRE(d3d_dgrad_plan_supported(ksize, stride),
              "d3d_dgrad_plan_build: strides %d,%d,%d / kernel %d,%d,%d outside the plan's classes (ask "
              "d3d_dgrad_plan_supported)", stride[0], stride[1], stride[2], ksize[0], ksize[1], ksize[2]);
  if (plan_ints < d3d_dgrad_plan_ints(cap_in))
    return d3d_set_error(D3D_ECAPACITY, "d3d_dgrad_plan_build: plan of %zu words < %zu", plan_ints,
                         d3d_dgrad_plan_ints(cap_in));
  PlanGeom G;
  for (int j = 0; j < 3; ++j) {
    D3D_REQUIRE(padding[j] >= 0, "d3d_dgrad_plan_build: padding");
    G.k[j] = ksize[j];
    G.s[j] = stride[j];
    G.p[j] = padding[j];
  }
  G.ncls = stride[0] * stride[1] * stride[2];
  G.kvol = ksize[0] * ksize[1] * ksize[2];
  const int nb = plan_blocks(cap_in);
  int32_t *blockhist = plan + D3D_PLAN_PERM + cap_in;
  hipLaunchKernelGGL(dplan_hist_kernel, dim3(nb), dim3(D3D_PLAN_SORT_ROWS), 0, stream, (const int4 *)coords_in,
                     n_in_dev, cap_in, G, blockhist);
  D3D_LAUNCH_TRY("dplan_hist_kernel");
  hipLaunchKernelGGL(dplan_scan_kernel, dim3(1), dim3(64 * D3D_PLAN_MAX_CLASSES), 0, stream, blockhist, n_in_dev, cap_in,
                     G, plan);
  D3D_LAUNCH_TRY("dplan_scan_kernel");
  hipLaunchKernelGGL(dplan_scatter_kernel, dim3(nb), dim3(D3D_PLAN_SORT_ROWS), 0, stream, (const int4 *)coords_in,
                     n_in_dev, cap_in, G, blockhist, plan);
  D3D_LAUNCH_TRY("dplan_scatter_kernel");
  return D3D_OK;
}
