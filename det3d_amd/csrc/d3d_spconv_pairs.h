// Pair-compacted sparse convolution (round 6) -- included by d3d_spconv.hip, which provides ConvEpiDev, epi_prepare,
// spconv_fwd_epilogue and the split helpers.
//
// The output-stationary kernels of d3d_spconv.hip walk all kvol offsets of a 16-site tile and multiply whatever the
// neighbour table holds: on the submanifold layers of a LiDAR frame 13 of 27 neighbours exist, so half of the gathered
// rows are the hardware's zero fill and half of the 3 x bf16 MFMAs multiply zeros (profiles/round5_fwd_anatomy.txt,
// section 7).  Here the table is COMPACTED first: for every slab of 64 consecutive output rows
// and every offset k, the existing pairs (output row in the slab, input row) are packed to the front of the slab's
// 64-entry segment (d3d_pair_table_build: one pass over the table per rulebook, shared by the layers, directions and
// steps that use it).  The conv kernel then runs MFMA tiles over 16 VALID pairs:
//
//   wave  = one slab (64 output rows); its f32 accumulators live in LDS ([64][COUT], 16 KB at 64 channels)
//   tile  = 16 consecutive pairs of one (slab, offset):  D^T[cout][pair] = W[k]^T . X^T  as three bf16 MFMAs per
//           k-step on hi / lo operands (the split arithmetic of d3d_spconv.hip), then every lane -- 4 consecutive
//           couts of ONE pair -- adds its 4 values to the pair's output row in the slab (16-byte read, add, write)
//   block = WAVES slabs sharing W[k] through a double-buffered LDS stage, one rendezvous per offset
//
// A row's sum is accumulated by one wave in program order (offsets ascending; inside an offset a row occurs once), LDS
// operations of a wave complete in order: results are deterministic.  They are NOT bit-identical to the
// output-stationary kernels (there the products of all offsets run through one MFMA accumulator; here an offset's six
// MFMAs are summed first and added to the row).
//
// Measured tile counts on the benched KITTI batch (tools/ in DESIGN.md section 10): 64-row slabs issue 1.5 x fewer
// tiles than the 16-row tiles of the old kernel have active offsets on the 126k-site level (fill 0.82), 1.28 x on the
// 65k-site level, 1.96 x on the 32-channel level.
//
// Pipeline: the flat list of a wave's tiles (offset-major; an offset without pairs counts as one empty tile so that every
// wave passes every rendezvous with at least one pipeline step in between) is walked with the gathers D tiles and the
// pair entries 2 D tiles ahead, all loads unconditional (invalid lanes read through out-of-range buffer offsets) so that
// hipcc's counted vmcnt waits keep them in flight across the rendezvous.
#pragma once

#define D3D_PAIR_SLAB 64
#define D3D_PAIR_ROW_BITS 26
#define D3D_PAIR_ROW_MASK 0x3FFFFFFu

// ---------------------------------------------------------------------------------------------------- table build
// ctab[k][cap64]: segment [slab * 64, +cnt) of row k = the pairs of (slab, k), entry = input row | (output row % 64) << 26
// cnt[slab][32] : pairs per offset (u8, <= 64); cnt_lo[slab][32]: those of the slab's rows 0..31 -- entries are in ascending
//                 row order, so a segment splits into the pair lists of its two 32-row halves at cnt_lo
__global__ void __launch_bounds__(256)
    pair_table_build_kernel(const int32_t *__restrict__ nbr, int cap, const int32_t *__restrict__ n_rows_dev, int n_rows_cap,
                            int kvol, int cap64, uint32_t *__restrict__ ctab, uint8_t *__restrict__ cnt,
                            uint8_t *__restrict__ cnt_lo) {
  const int lane = threadIdx.x & 63;
  const int slab = blockIdx.x * 4 + (threadIdx.x >> 6);
  int n_rows = n_rows_dev ? *n_rows_dev : n_rows_cap;
  n_rows = n_rows < n_rows_cap ? n_rows : n_rows_cap;
  const int row = slab * D3D_PAIR_SLAB + lane;
  if (slab * D3D_PAIR_SLAB >= n_rows_cap) return;
  int mine = 0, mine_lo = 0;
  for (int k = 0; k < kvol; ++k) {
    const int v = row < n_rows ? nbr[(int64_t)k * cap + row] : -1;
    const bool ok = v >= 0;
    const unsigned long long m = __ballot(ok);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (ok) ctab[(int64_t)k * cap64 + slab * D3D_PAIR_SLAB + rank] = (uint32_t)v | ((uint32_t)lane << D3D_PAIR_ROW_BITS);
    if (lane == k) {
      mine = __popcll(m);
      mine_lo = __popcll(m & 0xffffffffull);  // pairs of rows 0..31 of the slab: they come first in the segment
    }
  }
  if (lane < 32) {
    cnt[(int64_t)slab * 32 + lane] = (uint8_t)mine;
    cnt_lo[(int64_t)slab * 32 + lane] = (uint8_t)mine_lo;
  }
}

// ---------------------------------------------------------------------------------------------------- conv kernel
// R = rows per wave: 64 (a whole slab; the weight fragments of an offset stay in registers: WREG) or 32 (half a slab, twice
// the waves: 16 per CU at 64 output channels = four per SIMD, which needs <= 128 registers -- the weight fragments are
// then read from LDS per tile and k-step)
template <int CIN, int COUT, int WAVES, int DEPTH, int R = D3D_PAIR_SLAB, bool WREG = true>
__global__ void __launch_bounds__(WAVES * 64, (WAVES >= 16 ? 4 : 2))
    spconv_pairs_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_, const float *__restrict__ bias,
                        const uint32_t *__restrict__ ctab, const uint8_t *__restrict__ cnt, const uint8_t *__restrict__ cnt_lo,
                        int cap64, int n_out, int kvol, void *__restrict__ out_, ConvEpiDev ep) {
  static_assert(CIN % 32 == 0 && COUT % 16 == 0 && COUT <= 64, "f32 features, x32 MFMA steps, slab rows of <= 256 B");
  static_assert(R == 64 || R == 32, "a wave owns a slab or half of one");
  constexpr int NT = COUT / 16;
  constexpr int KS = CIN / 32;
  constexpr bool SWZ = CIN == 64;                  // weight rows: as in spconv_fwd_resident_kernel
  constexpr int ROWB = SWZ ? CIN * 2 : CIN * 2 + 16;
  constexpr int WBYTES = 2 * COUT * ROWB;          // one offset: hi plane rows, then lo plane rows
  constexpr int NTHREADS = WAVES * 64;
  constexpr int SEGS = CIN * 2 / 16;
  constexpr int WVEC = 2 * COUT * SEGS;            // 16-byte vectors per offset in global memory
  constexpr int WPT = (WVEC + NTHREADS - 1) / NTHREADS;
  constexpr int SLABB = R * COUT * 4;              // bytes of one wave's accumulator slab
  static_assert(2 * WBYTES >= 4 * SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS), "epilogue scratch aliases the W buffers");
  extern __shared__ __attribute__((aligned(16))) unsigned char plds[];  // [WAVES] slabs | 2 x W
  unsigned char *wlds = plds + WAVES * SLABB;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int unit = blockIdx.x * WAVES + wave;       // the wave's R rows
  const int slab = R == 64 ? unit : unit >> 1;      // 64-row segment of the table
  const int half = R == 64 ? 0 : unit & 1;
  const int row0 = unit * R;
  float *myslab = (float *)(plds + wave * SLABB);

  // ---- zero the slab (16-byte stores, one row segment per lane and step)
#pragma unroll
  for (int i = 0; i < SLABB / (64 * 16); ++i) *(float4 *)((unsigned char *)myslab + (i * 64 + lane) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- the wave's tile list: lane k holds the pair count of offset k, tiles(k) = max(1, ceil(cnt / 16))
  const bool slab_live = row0 < n_out;
  int cntv = 0, basev = 0;  // pairs of this wave's rows per offset; where they start in the slab's segment
  if (lane < kvol && slab_live) {
    cntv = (int)cnt[(int64_t)slab * 32 + lane];
    if constexpr (R == 32) {
      const int lo = (int)cnt_lo[(int64_t)slab * 32 + lane];
      basev = half ? lo : 0;
      cntv = half ? cntv - lo : lo;
    }
  }
  const int tl = lane < kvol ? ((cntv + 15) >> 4 > 0 ? (cntv + 15) >> 4 : 1) : 0;
  int incl = tl;  // inclusive prefix over lanes (kvol <= 32)
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const int excl = incl - tl;
  const int T = __builtin_amdgcn_readlane(incl, 31);  // lanes >= kvol repeat the total
  const int inclq = lane < kvol ? incl : 0x7fffffff;
  // tile q -> (offset, tile in the offset, pairs of the offset); q >= T: offset = kvol
  auto locate = [&](int q, int &k, int &t, int &c) {
    k = __popcll(__ballot(inclq <= q));
    const int kk = k < kvol ? k : kvol - 1;
    t = q - __builtin_amdgcn_readlane(excl, kk);
    c = k < kvol ? __builtin_amdgcn_readlane(cntv, kk) : 0;
  };
  auto seg_base = [&](int k) -> int {  // first entry of this wave's pairs of offset k inside the slab's segment
    if constexpr (R == 64) return 0;
    return __builtin_amdgcn_readlane(basev, k < kvol ? k : kvol - 1);
  };

  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_);
  const __amdgpu_buffer_rsrc_t tab_rs = d3d_make_rsrc(ctab);
  // pair entries of tile q for this lane's pair (lr); lanes past the offset's count (and tiles past the end) read
  // nothing (out of range -> 0)
  auto load_entry = [&](int q) -> uint32_t {
    int k, t, c;
    locate(q, k, t, c);
    const bool ok = t * 16 + lr < c;
    const uint32_t off = ((uint32_t)k * (uint32_t)cap64 + (uint32_t)(slab * D3D_PAIR_SLAB + seg_base(k) + t * 16 + lr)) * 4u;
    return (uint32_t)d3d_buffer_load4(tab_rs, ok ? off : D3D_OOB_OFFSET);
  };
  struct Frag32 { uint4 a, b; };
  auto gather = [&](int q, uint32_t e, Frag32 (&f)[KS]) {
    int k, t, c;
    locate(q, k, t, c);
    const bool ok = t * 16 + lr < c;
    // (a variant in which the four k-group lanes of a pair read 64 contiguous bytes per instruction -- k-steps over
    // permuted channels, weight rows permuted to match when they are published -- measured the same: the texture path
    // coalesces within quads of ADJACENT lanes, and a pair's k-groups sit 16 lanes apart in the MFMA operand layout)
    const uint32_t base = (e & D3D_PAIR_ROW_MASK) * (uint32_t)(CIN * 4) + (uint32_t)(lg * 32);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f[ks].a = d3d_buffer_load16(feat_rs, ok ? base + (uint32_t)(ks * 128) : D3D_OOB_OFFSET);
      f[ks].b = d3d_buffer_load16(feat_rs, ok ? base + (uint32_t)(ks * 128 + 16) : D3D_OOB_OFFSET);
    }
  };
  const unsigned char *wg = (const unsigned char *)wpack_;
  const int kmax = kvol - 1;
  auto load_w = [&](int k, d3d_u32x4 (&w)[WPT]) {
    k = k < kmax ? k : kmax;
    const unsigned char *wk = wg + (size_t)k * (WVEC * 16);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int e = threadIdx.x + j * NTHREADS;
      w[j] = *(const d3d_u32x4 *)(wk + (uint32_t)((e < WVEC ? e : WVEC - 1) * 16));
    }
  };
  auto publish_w = [&](int buf, const d3d_u32x4 (&w)[WPT]) {
    unsigned char *dst = wlds + buf * WBYTES;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int e = threadIdx.x + j * NTHREADS;
      const uint32_t r = (uint32_t)e / SEGS, sg = (uint32_t)e % SEGS;
      if (WVEC % NTHREADS == 0 || e < WVEC) *(d3d_u32x4 *)(dst + r * ROWB + ((sg ^ (SWZ ? (r >> 1) & 7u : 0u)) * 16)) = w[j];
    }
  };

  // ---- pipeline slots: fr[d] = gather(q + d), en[d] = entries(q + D + d); wr = W of the next offset
  Frag32 fr[DEPTH][KS];
  uint32_t en[DEPTH], ecur[DEPTH];
  d3d_u32x4 wr[WPT];
  uint4 wh[NT][KS], wl[NT][KS];  // W[k] fragments of the offset being walked (A operands), hi and lo planes
  load_w(0, wr);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ecur[d] = load_entry(d);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) en[d] = load_entry(DEPTH + d);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) gather(d, ecur[d], fr[d]);

  // accumulator row o: its 16-byte cell c (4 consecutive columns) lives at cell c ^ swz(o) of the row, swz = o mod 16
  // (64 columns: 16 cells per 256-byte row) or (o >> 1) mod 8 (32 columns: 8 cells, two rows per 256 bytes) -- the 16 lanes
  // a ds_read_b128 serves per cycle then hit 16 distinct bank groups unless two pairs' rows agree in those bits.
  // (Round 6, first version: ds_add_f32 per value -- LDS float atomics retire ~1 lane per 4 cycles per CU, 874 us for the
  // launch the tile kernel runs in 103; the read-add-write below is what a wave-private slab needs anyway.)
  constexpr int CELLS = COUT / 4;
  auto swz = [](uint32_t o) -> uint32_t { return CELLS == 16 ? (o & 15u) : ((o >> 1) & 7u); };
  const uint32_t slab_base = (uint32_t)(wave * SLABB);

#pragma unroll 1
  for (int qb = 0; qb < T; qb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int q = qb + d;
      if (q < T) {  // uniform
        int k, t, c;
        locate(q, k, t, c);
        if (t == 0) {  // first tile of offset k: the rendezvous.  Every wave gets here exactly once per offset.
          publish_w(k & 1, wr);
          load_w(k + 1, wr);
          d3d_lds_barrier();
          if constexpr (WREG) {
            if (c > 0) {
              const unsigned char *wcur = wlds + (k & 1) * WBYTES;
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                const int cb = ((ks * 32 + lg * 8) * 2) ^ (SWZ ? ((lr >> 1) & 7) << 4 : 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                  wh[n][ks] = *(const uint4 *)(wcur + (n * 16 + lr) * ROWB + cb);
                  wl[n][ks] = *(const uint4 *)(wcur + (COUT + n * 16 + lr) * ROWB + cb);
                }
              }
            }
          }
        }
        if (c > 0) {
          // this lane ends up with couts n * 16 + lg * 4 + (0..3) of pair lr, whose output row is o: fetch the row's
          // running sums now, they arrive behind the MFMAs (lanes past the count read row 0 and write nothing)
          const uint32_t o = (ecur[d] >> D3D_PAIR_ROW_BITS) & (uint32_t)(R - 1);
          const uint32_t rowb = slab_base + o * (uint32_t)(COUT * 4);
          uint32_t ca[NT];
          f32x4_t run[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            ca[n] = rowb + ((((uint32_t)(n * 4 + lg)) ^ swz(o)) << 4);
            run[n] = *(const f32x4_t *)(plds + ca[n]);
          }
          f32x4_t acc[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          const unsigned char *wtile = wlds + (k & 1) * WBYTES;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            uint4 xh, xl;
            spconv_split8(fr[d][ks].a, fr[d][ks].b, xh, xl);
            if constexpr (!WREG) {  // this k-step's weight fragments from the LDS stage (they stay there until the next rendezvous)
              const int cb = ((ks * 32 + lg * 8) * 2) ^ (SWZ ? ((lr >> 1) & 7) << 4 : 0);
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                wh[n][ks] = *(const uint4 *)(wtile + (n * 16 + lr) * ROWB + cb);
                wl[n][ks] = *(const uint4 *)(wtile + (COUT + n * 16 + lr) * ROWB + cb);
              }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wh[n][ks]),
                                                               __builtin_bit_cast(bf16x8_t, xh), acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wh[n][ks]),
                                                               __builtin_bit_cast(bf16x8_t, xl), acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wl[n][ks]),
                                                               __builtin_bit_cast(bf16x8_t, xh), acc[n], 0, 0, 0);
          }
          if (t * 16 + lr < c) {
#pragma unroll
            for (int n = 0; n < NT; ++n) *(f32x4_t *)(plds + ca[n]) = run[n] + acc[n];
          }
        }
      }
      // refill the slot: tile q + D (its entries arrived D steps ago), then the entries of tile q + 2 D
      ecur[d] = en[d];
      gather(q + DEPTH, ecur[d], fr[d]);
      en[d] = load_entry(q + 2 * DEPTH);
    }
  }

  // ---- epilogue: the slab -> the register layout of the output-stationary kernels (4 tiles of 16 rows), then the shared
  // epilogue (bias, addend, row stores, BatchNorm statistics / backward sums folded across blocks)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int MT = R / 16;
  f32x4_t acc[MT][NT];
  int site[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t o = (uint32_t)(m * 16 + lr);
    site[m] = row0 + m * 16 + lr;
    live[m] = site[m] < n_out;
#pragma unroll
    for (int n = 0; n < NT; ++n)
      acc[m][n] = *(const f32x4_t *)(plds + slab_base + o * (uint32_t)(COUT * 4) + ((((uint32_t)(n * 4 + lg)) ^ swz(o)) << 4));
  }
  spconv_fwd_epilogue<COUT, MT, false, NTHREADS>(acc, site, live, bias, out_, ep, (float *)wlds, lr, lg);
}

template <int CIN, int COUT, int WAVES, int DEPTH, int R = D3D_PAIR_SLAB, bool WREG = true>
static int launch_pairs(const void *feat, const void *wpack, const float *bias, const uint32_t *ctab, const uint8_t *cnt,
                        const uint8_t *cnt_lo, int cap64, int n_out, int kvol, void *out, ConvEpiDev ep, hipStream_t stream) {
  const int rows_per_block = WAVES * R;
  const int nb = (n_out + rows_per_block - 1) / rows_per_block;
  if (int rc = epi_prepare(ep, nb, COUT)) return rc;
  constexpr size_t lds = (size_t)WAVES * R * COUT * 4 + 2 * (size_t)(2 * COUT * (CIN == 64 ? CIN * 2 : CIN * 2 + 16));
  static_assert(lds <= 160 * 1024, "LDS of one CU");
  D3D_HIP_TRY(hipFuncSetAttribute((const void *)spconv_pairs_kernel<CIN, COUT, WAVES, DEPTH, R, WREG>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((spconv_pairs_kernel<CIN, COUT, WAVES, DEPTH, R, WREG>), dim3(nb), dim3(WAVES * 64), lds, stream, feat,
                     wpack, bias, ctab, cnt, cnt_lo, cap64, n_out, kvol, out, ep);
  D3D_LAUNCH_TRY("spconv_pairs_kernel");
  return D3D_OK;
}
