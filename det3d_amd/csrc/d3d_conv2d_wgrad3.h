// Weight gradient of the 3x3 / stride-1 RPN convolutions in split arithmetic, second generation:
//
//   dW[ky][kx][co][ci] = sum over (b, y, x) of g[b, y, x, co] * in[b, y + ky - pad, x + kx - pad, ci]
//
// Round 3's conv2d_wgrad_kernel gives every tap its own block: the g tile and the (shifted) input tile are staged nine
// times, 32 KB of LDS-DMA per 12 MFMAs of a wave -- the same L2 -> LDS traffic wall as the round-3 forward kernel.
// Here a block owns one KERNEL ROW ky of a 128 x 128 weight tile and a range of 32-pixel SEGMENTS of image rows: per K
// step it stages the segment's g rows once and ONE input run of 34 pixels (x0 - pad .. x0 - pad + 33 of input row
// y + ky - pad); the three taps kx read that run at offsets 0, 1, 2.  34 KB of LDS-DMA per 36 MFMAs of a wave.
// Segments never cross an image row, so the zero padding is the out-of-range fill of the LDS-DMA (rows of `in` left or
// right of the map, whole input rows above / below it) and the pixels of a last, shorter segment are zero rows of g.
//
// Operands are pixel-major in LDS ([pixel][128 channels], 256-byte rows, LDS-DMA of 4 rows per instruction, chunk c of
// row r at c ^ ((r & 3) << 2)) and become MFMA operands -- the pixel axis is the GEMM's K -- through the transposing
// read ds_read_b64_tr_b16, as in conv2d_wgrad_kernel.  Schedule: the forward kernel's (d3d_conv2d_halo.h): 8 waves =
// two groups one barrier apart, load part (80 transposing reads, 4-5 LDS-DMA instructions two K steps ahead, counted
// vmcnt) against MFMA part (36 MFMAs), ring of three stages; the hazard analysis there applies verbatim.
// One partial [3][128][128] slab per block; conv2d_wgrad_reduce_kernel sums the slabs in split order.
#pragma once

// SCHED: 0 one barrier per K step, 1 staggered wave groups, 2 staggered + the LDS-DMA requests inside the MFMA part
// (d3d_conv2d_halo.h)
template <int SCHED>
__global__ void __launch_bounds__(512, 2)
    conv2d_wgrad3_kernel(const void *__restrict__ g_hi, const void *__restrict__ g_lo, const void *__restrict__ x_hi,
                         const void *__restrict__ x_lo, float *__restrict__ part, Conv2dGeom G, int cin,
                         int segs_per_split, int nsplit) {
  constexpr bool STAGGER = SCHED >= 1, LATE = SCHED == 2;
  constexpr int GPLANE = 32 * 256, XPLANE = 36 * 256;
  constexpr int STAGE = 2 * GPLANE + 2 * XPLANE;  // g hi | g lo | x hi | x lo
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3][STAGE]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wco = wave >> 2, wci = wave & 3, grp = wave >> 2;
  int logical;
  {
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ky = logical % 3;
  const int split = (logical / 3) % nsplit;
  const int wtile = logical / (3 * nsplit);
  const int cin_tiles = cin / 128;
  const int co0 = (wtile / cin_tiles) * 128, ci0 = (wtile % cin_tiles) * 128;
  const int segs = (G.wo + 31) >> 5;
  const int total = G.batch * G.ho * segs;
  const int q_begin = split * segs_per_split;
  int q_end = q_begin + segs_per_split;
  if (q_end > total) q_end = total;
  const int nks = q_end > q_begin ? q_end - q_begin : 0;

  const __amdgpu_buffer_rsrc_t gr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)g_hi, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)g_lo, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)x_hi, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)x_lo, 0, G.x_bytes, 0x00020000);

  // ---- staging.  This wave's LDS-DMA instructions: rows 4 wave .. 4 wave + 3 of the g tile and of the input run (both
  // planes); rows 32 .. 35 of the input run: plane hi by wave 0, plane lo by wave 1
  const int srow = 4 * wave + (lane >> 4);             // row of the tile this lane loads 16 bytes of
  const int srow8 = 32 + (lane >> 4);
  const uint32_t schunk = (uint32_t)(((lane & 15) ^ ((srow & 3) << 2)) * 16);  // (srow8 & 3) == (srow & 3)
  // the segment the NEXT stage() call loads: (image, row, segment of the row); wave-uniform
  int s_seg, s_y, s_b;
  {
    const int ry = q_begin / segs;
    s_seg = q_begin - ry * segs;
    s_b = ry / G.ho;
    s_y = ry - s_b * G.ho;
  }
  auto stage = [&](int buf) {
    unsigned char *sg = smem + buf * STAGE, *sx = sg + 2 * GPLANE;
    const int x0 = s_seg << 5;
    const int iy = s_y - G.pad + ky;
    const bool row_ok = iy >= 0 && iy < G.h;
    {
      const int ox = x0 + srow;
      const uint32_t goff = ox < G.wo ? (uint32_t)(((s_b * G.ho + s_y) * G.wo + ox) * (G.cout * 2) + co0 * 2) + schunk
                                      : D3D_OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_hi, (c2_lds_void *)(sg + wave * 1024), 16, (int)goff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_lo, (c2_lds_void *)(sg + GPLANE + wave * 1024), 16, (int)goff, 0, 0, 0);
    }
    {
      const int ix = x0 - G.pad + srow;
      const uint32_t xoff = (row_ok && ix >= 0 && ix < G.w)
                                ? (uint32_t)(((s_b * G.h + iy) * G.w + ix) * (cin * 2) + ci0 * 2) + schunk
                                : D3D_OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sx + wave * 1024), 16, (int)xoff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sx + XPLANE + wave * 1024), 16, (int)xoff, 0, 0, 0);
    }
    if (wave < 2) {
      const int ix = x0 - G.pad + srow8;
      const uint32_t xoff = (row_ok && ix >= 0 && ix < G.w)
                                ? (uint32_t)(((s_b * G.h + iy) * G.w + ix) * (cin * 2) + ci0 * 2) + schunk
                                : D3D_OOB_OFFSET;
      if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sx + 8 * 1024), 16, (int)xoff, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sx + XPLANE + 8 * 1024), 16, (int)xoff, 0, 0, 0);
    }
    if (++s_seg == segs) {
      s_seg = 0;
      if (++s_y == G.ho) {
        s_y = 0;
        ++s_b;
      }
    }
  };

  c2_f32x16 acc[3][2];  // [kx][co tile]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

  // ---- transposing fragment reads (see conv2d_wgrad_kernel): a 16-lane group reads a 4 (pixels) x 16 (channels) block
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const int i16 = lane & 15;
  const int prow = ((lane >> 5) << 3) + (i16 >> 2);
  const int chan = (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);
  uint32_t fo_g[2], fo_x[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = (wco * 2 + i) * 32 + chan;
    fo_g[i] = (uint32_t)(prow * 256 + ((((ch * 2) >> 4) ^ ((prow & 3) << 2)) << 4) + ((ch * 2) & 15));
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int ch = wci * 32 + chan, row = prow + t;  // tap kx = t reads the run one pixel further
    fo_x[t] = (uint32_t)(row * 256 + ((((ch * 2) >> 4) ^ ((row & 3) << 2)) << 4) + ((ch * 2) & 15));
  }
  auto tr_read = [&](uint32_t addr) -> u2 {
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };

  if (nks > 0) stage(0);
  if (nks > 1) stage(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (STAGGER && grp == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

#pragma unroll 1
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks % 3;
    // ---------------- load part
    const uint32_t gbase = lds_base + (uint32_t)(buf * STAGE);
    const uint32_t xbase = gbase + (uint32_t)(2 * GPLANE);
    u2 gr[2][2][2][2], xr[2][3][2][2];  // [substep][tile | tap][plane][first / second read]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 2; ++i) gr[kk][i][p][h] = tr_read(gbase + (uint32_t)(p * GPLANE + kk * 4096 + h * 1024) + fo_g[i]);
#pragma unroll
          for (int t = 0; t < 3; ++t) xr[kk][t][p][h] = tr_read(xbase + (uint32_t)(p * XPLANE + kk * 4096 + h * 1024) + fo_x[t]);
        }
    const bool more = ks + 2 < nks;
    int nb3 = buf + 2;
    if (nb3 >= 3) nb3 -= 3;
    if (more && !LATE) {
      stage(nb3);
      if (wave < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LATE: the requests of M(ks - 1), half a K step old
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // tie every fragment register to the wait (the compiler does not see the asm reads' latency)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(gr[kk][i][p][h]));
#pragma unroll
          for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(xr[kk][t][p][h]));
        }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- MFMA part
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      c2_bf16x8 ga[2][2], xb[3][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u4 v = {gr[kk][i][p][0][0], gr[kk][i][p][0][1], gr[kk][i][p][1][0], gr[kk][i][p][1][1]};
          ga[i][p] = __builtin_bit_cast(c2_bf16x8, v);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const u4 v = {xr[kk][t][p][0][0], xr[kk][t][p][0][1], xr[kk][t][p][1][0], xr[kk][t][p][1][1]};
          xb[t][p] = __builtin_bit_cast(c2_bf16x8, v);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][0], xb[t][0], acc[t][i], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][0], xb[t][1], acc[t][i], 0, 0, 0);
      if constexpr (LATE) {
        if (kk == 0) {  // after a third of the step's MFMAs: the requests of K step ks + 2
          __builtin_amdgcn_sched_barrier(0);
          if (more) stage(nb3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][1], xb[t][0], acc[t][i], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (STAGGER) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();

  // partial slabs: part[split][tap = 3 ky + kx][co][ci]; D rows = co, cols = ci
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float *dst = part + ((int64_t)split * 9 + (ky * 3 + t)) * ((int64_t)G.cout * cin);
    const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(int64_t)co * cin + ci] = acc[t][i][r];
      }
  }
}

// ---- 64 x 64 weight tiles: the 64-channel 3x3 layers of the PointPillars necks' first block (rpn.py:80-110 with
// ds_num_filters[0] = 64).  The same construction on 128-byte rows: a block of FOUR waves owns one kernel row of the whole
// 64 x 64 weight matrix and a range of 32-pixel segments; wave (wco, wci) holds a 32 x 32 tile for the three taps.  LDS-DMA
// instructions cover 8 rows of 128 B; chunk c of row r sits at c ^ (((r >> 1) & 3) << 1): the four pixel rows of a
// transposing read are then 64 B apart within the two 256-byte bank lines they share, and the 32-byte piece a lane
// group reads stays contiguous.  18 KB per stage, ring of three, one barrier per K step, two to three blocks per CU
// (the blocks overlap each other's load and MFMA parts; no stagger inside a block).  18 MFMAs per wave and K step.
__global__ void __launch_bounds__(256, 2)
    conv2d_wgrad3n_kernel(const void *__restrict__ g_hi, const void *__restrict__ g_lo, const void *__restrict__ x_hi,
                          const void *__restrict__ x_lo, float *__restrict__ part, Conv2dGeom G, int segs_per_split,
                          int nsplit) {
  constexpr int GPLANE = 32 * 128, XPLANE = 40 * 128;
  constexpr int STAGE = 2 * GPLANE + 2 * XPLANE;  // g hi | g lo | x hi | x lo = 18432 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3][STAGE]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wco = wave >> 1, wci = wave & 1;
  int logical;
  {
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ky = logical % 3;
  const int split = logical / 3;
  const int segs = (G.wo + 31) >> 5;
  const int total = G.batch * G.ho * segs;
  const int q_begin = split * segs_per_split;
  int q_end = q_begin + segs_per_split;
  if (q_end > total) q_end = total;
  const int nks = q_end > q_begin ? q_end - q_begin : 0;

  const __amdgpu_buffer_rsrc_t gr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)g_hi, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)g_lo, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)x_hi, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)x_lo, 0, G.x_bytes, 0x00020000);

  // staging: this wave's instructions = rows 8 wave .. 8 wave + 7 of the g tile and of the input run (both planes); rows
  // 32 .. 39 of the run (36 used): plane hi by wave 0, plane lo by wave 1
  auto swz = [](int r) { return ((r >> 1) & 3) << 1; };
  const int srow = 8 * wave + (lane >> 3), srow8 = 32 + (lane >> 3);
  const uint32_t schunk = (uint32_t)(((lane & 7) ^ swz(srow)) * 16);  // swz(srow8) == swz(srow): 32 = 0 mod 8
  int s_seg, s_y, s_b;
  {
    const int ry = q_begin / segs;
    s_seg = q_begin - ry * segs;
    s_b = ry / G.ho;
    s_y = ry - s_b * G.ho;
  }
  auto stage = [&](int buf) {
    unsigned char *sg = smem + buf * STAGE, *sx = sg + 2 * GPLANE;
    const int x0 = s_seg << 5;
    const int iy = s_y - G.pad + ky;
    const bool row_ok = iy >= 0 && iy < G.h;
    {
      const int ox = x0 + srow;
      const uint32_t goff = ox < G.wo ? (uint32_t)(((s_b * G.ho + s_y) * G.wo + ox) * 128) + schunk : D3D_OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_hi, (c2_lds_void *)(sg + wave * 1024), 16, (int)goff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_lo, (c2_lds_void *)(sg + GPLANE + wave * 1024), 16, (int)goff, 0, 0, 0);
    }
    {
      const int ix = x0 - G.pad + srow;
      const uint32_t xoff = (row_ok && ix >= 0 && ix < G.w) ? (uint32_t)(((s_b * G.h + iy) * G.w + ix) * 128) + schunk
                                                            : D3D_OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sx + wave * 1024), 16, (int)xoff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sx + XPLANE + wave * 1024), 16, (int)xoff, 0, 0, 0);
    }
    if (wave < 2) {
      const int ix = x0 - G.pad + srow8;
      const uint32_t xoff = (row_ok && ix >= 0 && ix < G.w) ? (uint32_t)(((s_b * G.h + iy) * G.w + ix) * 128) + schunk
                                                            : D3D_OOB_OFFSET;
      if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sx + 4 * 1024), 16, (int)xoff, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sx + XPLANE + 4 * 1024), 16, (int)xoff, 0, 0, 0);
    }
    if (++s_seg == segs) {
      s_seg = 0;
      if (++s_y == G.ho) {
        s_y = 0;
        ++s_b;
      }
    }
  };

  c2_f32x16 acc[3];  // [kx]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const int i16 = lane & 15;
  const int prow = ((lane >> 5) << 3) + (i16 >> 2);
  const int chan = (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);
  // (row + 4 changes the swizzle, row + 16 does not: one address per (tile | tap, first / second read))
  uint32_t fo_g[2], fo_x[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    {
      const int ch = wco * 32 + chan, row = prow + 4 * h;
      fo_g[h] = (uint32_t)(row * 128 + ((((ch * 2) >> 4) ^ swz(row)) << 4) + ((ch * 2) & 15));
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ch = wci * 32 + chan, row = prow + 4 * h + t;
      fo_x[t][h] = (uint32_t)(row * 128 + ((((ch * 2) >> 4) ^ swz(row)) << 4) + ((ch * 2) & 15));
    }
  }
  auto tr_read = [&](uint32_t addr) -> u2 {
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };

  if (nks > 0) stage(0);
  if (nks > 1) stage(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#pragma unroll 1
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks % 3;
    const uint32_t gbase = lds_base + (uint32_t)(buf * STAGE);
    const uint32_t xbase = gbase + (uint32_t)(2 * GPLANE);
    u2 gr[2][2][2], xr[2][3][2][2];  // [substep][tap][plane][first / second read]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          gr[kk][p][h] = tr_read(gbase + (uint32_t)(p * GPLANE + kk * 2048) + fo_g[h]);
#pragma unroll
          for (int t = 0; t < 3; ++t) xr[kk][t][p][h] = tr_read(xbase + (uint32_t)(p * XPLANE + kk * 2048) + fo_x[t][h]);
        }
    if (ks + 2 < nks) {
      int nb3 = buf + 2;
      if (nb3 >= 3) nb3 -= 3;
      stage(nb3);
      if (wave < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          asm volatile("" : "+v"(gr[kk][p][h]));
#pragma unroll
          for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(xr[kk][t][p][h]));
        }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      c2_bf16x8 ga[2], xb[3][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u4 v = {gr[kk][p][0][0], gr[kk][p][0][1], gr[kk][p][1][0], gr[kk][p][1][1]};
        ga[p] = __builtin_bit_cast(c2_bf16x8, v);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const u4 u = {xr[kk][t][p][0][0], xr[kk][t][p][0][1], xr[kk][t][p][1][0], xr[kk][t][p][1][1]};
          xb[t][p] = __builtin_bit_cast(c2_bf16x8, u);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], xb[t][0], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], xb[t][1], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1], xb[t][0], acc[t], 0, 0, 0);
    }
  }

  // partial slabs: part[split][tap = 3 ky + kx][co][ci] with cout = cin = 64
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float *dst = part + ((int64_t)split * 9 + (ky * 3 + t)) * (64 * 64);
    const int ci = wci * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      dst[co * 64 + ci] = acc[t][r];
    }
  }
}
