// 3x3 / stride-1 convolution of the RPN (det3d/models/necks/rpn.py:70-159) in split arithmetic, second generation:
// a block owns a PH x PW PATCH of output pixels (256 of them) x 128 output channels and keeps the patch's input rows
// WITH THEIR HALO in LDS, so that the nine taps read one staged copy.
//
// Why (round 3's conv2d_nhwc_kernel sat at 0.32 of the MFMA peak): that kernel re-stages the 128-pixel activation tile
// for every tap -- 9 LDS-DMA passes over each input element and a weight tile per 128 pixels: 32 KB of LDS-DMA per 12
// MFMAs of a wave, 1.3 GB of L2 -> LDS traffic per launch on the [4,200,176,128] map (8.4 TB/s at 155 us) with a
// per-XCD working set just past the 4 MB L2.  Here, per 32-channel chunk, the (PH+2) x (PW+2) halo patch is staged ONCE
// (1.27x the patch instead of 9x) and a weight tile serves 256 pixels: 0.42 GB per launch, 2.6 LDS-DMA instructions
// per wave and K step (24 MFMAs) instead of 4 per 12.
//
// Schedule (cdna_hip_programming.md section 5, the 8-wave ping-pong): K step = (channel chunk kc, tap); the waves form
// two groups of four (one wave of each group per SIMD).  A wave alternates a LOAD part (16 ds_read_b128 of the step's
// fragments, <= 3 LDS-DMA instructions for later steps, counted vmcnt, lgkmcnt(0)) and an MFMA part (24 MFMAs at raised
// priority), separated by raw s_barriers; group 1 runs ONE barrier behind group 0, so that on every SIMD one wave
// multiplies while the other loads.  LDS-DMA requests stay in flight across the barriers (never vmcnt(0) in the loop):
// weight tiles travel two K steps ahead through a ring of three slots, the next chunk's halo patch is staged during
// taps 0-5 of the current chunk into the other of two patch buffers.
//
// Hazards (global barrier k = group 0's k-th = group 1's (k-1)-th; L(t) / M(t) = load / MFMA part of K step t;
// group 0: L(t) < barrier 2t < M(t) < barrier 2t+1; group 1: barrier 2t < L(t) < barrier 2t+1 < M(t) < barrier 2t+2):
//   RAW  a tile read in L(t+1) was requested in L(t-1) or earlier; every wave waits for ITS requests (counted vmcnt)
//        at the end of L(t), i.e. before barrier 2t (group 0) / 2t+1 (group 1); group 0 reads after barrier 2t+1,
//        group 1 after 2t+2: one barrier more than the wait, for both.
//   WAR  the slot requested in L(t) was last read in L(t-1), and every wave retires its reads (lgkmcnt(0)) at the end
//        of its load part: group 1's L(t-1) ends before barrier 2t-1, group 0 requests after it.
//
// The last, partial round of blocks.  A launch has I = patches x (cout / 128) work items for 256 CUs with one block
// each: [4,200,176,128] -> 128 has 572, i.e. two full rounds and a third with 60 of 256 CUs busy (measured: 0.53 of the
// MFMA peak on maps whose item count is a multiple of 256, 0.38 on this one).  Items past the last multiple of 256 are
// therefore cut into FOUR blocks of 32 output channels each (WCO = NI = 1: eight waves of 32 pixels x 32 channels;
// the halo patch is staged as before, the weight tiles are a quarter): 240 short blocks fill the last round.
// (included by d3d_conv2d.hip after Conv2dGeom / Conv2dEpi)
#pragma once

template <int N> __device__ __forceinline__ void c2h_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

// how a launch's items map to blocks (host: c2h_plan; device: conv2d_halo_kernel)
struct C2hGrid {
  int tiles_x, tiles_y;  // patches per image
  int ctiles;            // output-channel tiles of ct_full channels
  int n_full;            // items [0, n_full) are whole blocks; item n_full + j is blocks n_full + cut j .. + cut - 1
  int n_items;           // patches * ctiles
  int ct_full;           // output channels of a whole block: 128, or 64 for 64-channel layers (PointPillars block 0)
  int cut;               // 32-channel blocks per cut item: ct_full / 32
};

// PH x PW = 256 output pixels per block.  SCHED 0: one barrier per K step; 1: the two wave groups run one barrier
// apart; 2: as 1, and the LDS-DMA requests are issued INSIDE the MFMA part (after the 8th and the 16th MFMA, where an
// LDS-DMA costs the issuing wave ~60 cycles instead of 100-185 next to the fragment reads: MI355X_MICROARCH.md) --
// the load part is then fragment reads + waits only.  A request of M(t) is waited for (vmcnt(0)) at the end of
// L(t+1) and read from L(t+2) on: the RAW / WAR argument above holds with the requests half a step later.
// WCO waves along cout, NI 32-channel tiles per wave: (2, 2) = 128 output channels per block, (2, 1) = 64, (1, 1) = 32
template <int CIN, int PH, int PW, int SCHED, int WCO, int NI>
__device__ __forceinline__ void c2h_body(const void *__restrict__ x_hi, const void *__restrict__ x_lo,
                                         const void *__restrict__ w_hi, const void *__restrict__ w_lo,
                                         float *__restrict__ y, const Conv2dGeom &G, const Conv2dEpi &ep,
                                         const C2hGrid &Q, int tile, int ct, int n0, int fold_b, int fold_nb,
                                         unsigned char *smem) {
  constexpr int KC = CIN / 32;                       // 32-channel chunks (64-byte rows per plane)
  constexpr int HWID = PW + 2, HR = (PH + 2) * HWID;  // halo patch: rows of the LDS image
  constexpr int XI = (HR + 15) / 16;                  // LDS-DMA instructions per plane (16 rows of 64 B each)
  constexpr int XPLANE = XI * 1024, XBUF = 2 * XPLANE;
  constexpr int CT = WCO * NI * 32;                   // output channels of this block
  constexpr int WPX = 8 / WCO, NJ = 256 / (WPX * 32); // pixel waves, 32-pixel tiles per wave
  constexpr int WPLANE = CT * 64, WSLOT = 2 * WPLANE;
  constexpr int WINS = CT / 16;                       // LDS-DMA instructions per weight plane (one per wave < WINS)
  constexpr int W0 = 2 * XBUF;                        // [2][XBUF] patch buffers, then [3][WSLOT] weight ring
  static_assert(PH * PW == 256 && XI <= 24 && CIN % 32 == 0 && NJ >= 1, "patch shape");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wco = wave % WCO, wpx = wave / WCO, grp = wave >> 2;
  const int tx = tile % Q.tiles_x, t2 = tile / Q.tiles_x, ty = t2 % Q.tiles_y, img = t2 / Q.tiles_y;
  const int oy0 = ty * PH, ox0 = tx * PW;

  // ---- staging slots.  Patch: this wave owns LDS-DMA instructions wave, wave + 8, wave + 16 (< XI) of each plane;
  // instruction i covers patch rows 16 i .. 16 i + 15, four lanes (16-byte chunks) per row, chunk c of row r at
  // physical chunk c ^ ((r >> 2) & 3) (the swizzle is applied to the SOURCE address: an LDS-DMA image is lane-linear)
  uint32_t x_off[3];
  bool x_ok[3];
#pragma unroll
  for (int pr = 0; pr < 3; ++pr) {
    const int i = wave + 8 * pr;
    const int r = 16 * i + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    const int hy = r / HWID, hx = r - hy * HWID;
    const int iy = oy0 - G.pad + hy, ix = ox0 - G.pad + hx;
    x_ok[pr] = i < XI && r < HR && iy >= 0 && iy < G.h && ix >= 0 && ix < G.w;
    x_off[pr] = (uint32_t)(((img * G.h + iy) * G.w + ix) * (CIN * 2) + c * 16);
  }
  uint32_t w_off;  // weight rows 16 wave .. 16 wave + 15 of the CT-row tile (waves < WINS)
  const bool w_mine = wave < WINS;
  {
    const int row = 16 * wave + (lane >> 2);
    w_off = (uint32_t)((n0 + (w_mine ? row : 0)) * (CIN * 2) + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
  }
  const __amdgpu_buffer_rsrc_t xr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)x_hi, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)x_lo, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)w_hi, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)w_lo, 0, G.w_bytes, 0x00020000);

  // one plane of patch instruction `pr` of chunk `kc` into patch buffer kc & 1
  auto stage_x = [&](int pr, int plane, int kc) {
    const uint32_t off = x_ok[pr] ? x_off[pr] + (uint32_t)(kc * 64) : D3D_OOB_OFFSET;
    unsigned char *dst = smem + (kc & 1) * XBUF + plane * XPLANE + (wave + 8 * pr) * 1024;
    if (plane == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)dst, 16, (int)off, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)dst, 16, (int)off, 0, 0, 0);
  };
  // both planes of this wave's 16 weight rows of K step (kc, tap) into ring slot `slot`
  auto stage_w = [&](int kc, int tap, int slot) {
    if (!w_mine) return;
    const int soff = (tap * G.cout * CIN + kc * 32) * 2;  // wave-uniform: the instruction's scalar offset
    unsigned char *dst = smem + W0 + slot * WSLOT + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_hi, (c2_lds_void *)dst, 16, (int)w_off, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_lo, (c2_lds_void *)(dst + WPLANE), 16, (int)w_off, soff, 0, 0);
  };

  // ---- fragment addresses.  32x32x16 operand: lane l holds row (l & 31), k-chunk (l >> 5) of a 16-channel substep
  const int frow = lane & 31, fk = lane >> 5;
  uint32_t a_addr[NI];  // weight tile i: byte offset inside a plane for substep 0 (substep 1: ^ 32)
  int r0[NJ];            // pixel tile j: patch row of this lane's pixel at tap (0, 0)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = wco * (NI * 32) + i * 32 + frow;
    a_addr[i] = (uint32_t)(row * 64 + ((fk ^ ((row >> 2) & 3)) << 4));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = wpx * (NJ * 32) + j * 32 + frow;
    r0[j] = (q / PW) * HWID + (q % PW);
  }

  // [substep chain][cout tile][pixel tile]; a wave with ONE tile keeps two accumulator chains (substep 0 / substep 1):
  // six back-to-back MFMAs on one accumulator would each wait for the previous one's result
  constexpr int KA = NI * NJ == 1 ? 2 : 1;
  c2_f32x16 accs[KA][NI][NJ];
#pragma unroll
  for (int a = 0; a < KA; ++a)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[a][i][j][r] = 0.f;

  // ---- prologue: the whole patch of chunk 0, the weight tiles of K steps 0 and 1
#pragma unroll
  for (int pr = 0; pr < 3; ++pr)
    if (wave + 8 * pr < XI) {
      stage_x(pr, 0, 0);
      stage_x(pr, 1, 0);
    }
  stage_w(0, 0, 0);
  stage_w(0, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr bool STAGGER = SCHED >= 1, LATE = SCHED == 2;
  if (STAGGER && grp == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

#pragma unroll 1
  for (int kc = 0; kc < KC; ++kc) {
    const unsigned char *xb = smem + (kc & 1) * XBUF;
    const bool next_chunk = kc + 1 < KC;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      constexpr int dummy = 0;
      (void)dummy;
      const int ky = tap / 3, kx = tap % 3;
      // ---------------- load part
      const unsigned char *wb = smem + W0 + (tap % 3) * WSLOT;  // (9 kc + tap) % 3 == tap % 3
      c2_bf16x8 wa[2][NI][2], xf[2][NJ][2];  // [substep][tile][plane]
      uint32_t x_addr[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = r0[j] + ky * HWID + kx;
        x_addr[j] = (uint32_t)((r << 6) + ((fk ^ ((r >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int i = 0; i < NI; ++i) wa[kk][i][p] = *(const c2_bf16x8 *)(wb + p * WPLANE + (a_addr[i] ^ (uint32_t)(kk << 5)));
#pragma unroll
          for (int j = 0; j < NJ; ++j) xf[kk][j][p] = *(const c2_bf16x8 *)(xb + p * XPLANE + (x_addr[j] ^ (uint32_t)(kk << 5)));
        }
      // requests: one plane of one patch instruction of the NEXT chunk (taps 0-5), the weight tile of K step t + 2
      const bool do_x = tap < 6 && next_chunk && (wave + 8 * (tap >> 1) < XI);
      const int tap2 = (tap + 2) % 9, kc2 = kc + (tap + 2) / 9;
      const bool do_w = kc2 < KC;
      if constexpr (!LATE) {
        if (do_x) stage_x(tap >> 1, tap & 1, kc + 1);
        if (do_w) stage_w(kc2, tap2, (tap + 2) % 3);
        // everything this wave requested BEFORE this load part has landed (the tiles of K step t + 1 among them)
        if (do_w && w_mine) {
          if (do_x) c2h_wait_vm<3>(); else c2h_wait_vm<2>();
        } else {
          if (do_x) c2h_wait_vm<1>(); else c2h_wait_vm<0>();
        }
      } else {
        c2h_wait_vm<0>();  // the requests of M(t - 1) and earlier: issued at least half a K step ago
      }
      // my fragment reads are complete: the MFMA part starts without an LDS wait, and nobody overwrites what I read
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA part: hi*hi, hi*lo, lo*hi of both substeps, four independent accumulators per round
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        auto &acc = accs[kk % KA];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk][i][0], xf[kk][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk][i][0], xf[kk][j][1], acc[i][j], 0, 0, 0);
        if constexpr (LATE) {
          // after two thirds of substep 0: the patch request; after two thirds of substep 1: the weight requests
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 0) {
            if (do_x) stage_x(tap >> 1, tap & 1, kc + 1);
          } else {
            if (do_w) stage_w(kc2, tap2, (tap + 2) % 3);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk][i][1], xf[kk][j][0], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      if (STAGGER) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last barrier

  // ---- epilogue.  C/D of 32x32: col (pixel) = lane & 31, row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  auto &acc = accs[0];
  if constexpr (KA == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += accs[1][0][0][r];
  }
  const bool want_stats = ep.stats != nullptr;
  const bool want_bnb = want_stats && ep.bn_x != nullptr;
  float s1[NI][16], s2[NI][16];
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s1[i][r] = s2[i][r] = 0.f;
  }
  int64_t pix[NJ];
  bool live[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = wpx * (NJ * 32) + j * 32 + frow;
    const int oy = oy0 + q / PW, ox = ox0 + q % PW;
    live[j] = oy < G.ho && ox < G.wo;
    pix[j] = ((int64_t)(img * G.ho + oy) * G.wo + ox) * G.cout;
  }
  if (want_bnb && ep.bn_stage) {
    // ---- BatchNorm-backward sums with the previous layer's pre-BN tile staged through LDS (round 6, last session).
    // A lane owns four channels of ONE pixel, so reading x from memory next to the stores below costs 16 loads of 16 bytes at
    // pixel stride per lane (two 32-byte pieces of every 128-byte line per instruction): +35 us on a 115 us launch, more than
    // the streaming d3d_bn_bwd_reduce it replaces (29 us).  The main loop is over and LDS is free: the block's
    // [256 pixels][CT channels] f32 tile arrives by LDS-DMA in whole pixel rows (CT * 4 contiguous bytes), chunk c of pixel
    // q at physical chunk c ^ (q mod chunks) -- the swizzle on the SOURCE address, as everywhere -- so that the 32 pixels
    // of a ds_read_b128 hit 32 different 16-byte slots.
    constexpr int CH = CT / 4;          // 16-byte chunks per pixel row
    constexpr int PPI = 64 / CH;        // pixels per LDS-DMA instruction (1 KiB)
    constexpr int IPW = 256 / PPI / 8;  // instructions per wave
    static_assert(CH >= 8 && CH <= 32 && IPW >= 1 && 256 * CT * 4 <= 2 * XBUF + 3 * WSLOT, "x tile fits the block's LDS");
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        (void *)ep.bn_x, 0, (int)((uint32_t)G.m * (uint32_t)G.cout * 4u), 0x00020000);
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int t = wave * IPW + u;
      const int q = t * PPI + lane / CH, pc = lane % CH;
      const int oy = oy0 + q / PW, ox = ox0 + q % PW;
      const bool ok = oy < G.ho && ox < G.wo;
      const uint32_t el = (uint32_t)((img * G.ho + oy) * G.wo + ox) * (uint32_t)G.cout + (uint32_t)(n0 + ((pc ^ (q & (CH - 1))) << 2));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (c2_lds_void *)(smem + t * 1024), 16, (int)(ok ? el * 4u : D3D_OOB_OFFSET),
                                               0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wco * (NI * 32) + i * 32 + 4 * (lane >> 5) + 8 * g;  // channel inside the block's tile
        const int co = n0 + cl;
        float is[4], mi[4], zs[4], zb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // the arithmetic of bn_bwd_reduce_kernel (d3d_bn.hip), as in the direct path below
          const float gm = ep.bn_gamma ? ep.bn_gamma[co + e] : 1.f, bt = ep.bn_beta ? ep.bn_beta[co + e] : 0.f;
          is[e] = ep.bn_invstd[co + e];
          mi[e] = -ep.bn_mean[co + e] * is[e];
          zs[e] = gm * is[e];
          zb[e] = fmaf(mi[e], gm, bt);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (!live[j]) continue;
          const int q = wpx * (NJ * 32) + j * 32 + frow;
          const float4 xr = *(const float4 *)(smem + q * (CT * 4) + ((((cl >> 2) ^ q) & (CH - 1)) << 4));
          const float x[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][4 * g + e];
            const float dz = (ep.bn_relu && fmaf(x[e], zs[e], zb[e]) <= 0.f) ? 0.f : v;
            s1[i][4 * g + e] = fmaf(dz, fmaf(x[e], is[e], mi[e]), s1[i][4 * g + e]);
            s2[i][4 * g + e] += dz;
          }
        }
      }
    }
    // the output tile after the sums: waited for in front of them (vmcnt(0) covers loads and stores alike), its stores
    // would cost the block their whole drain
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + wco * (NI * 32) + i * 32 + 4 * (lane >> 5) + 8 * g;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (live[j])
            *(float4 *)(y + pix[j] + co) =
                make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
  } else
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int co_base = n0 + wco * (NI * 32) + i * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co_base + 8 * g;
      float is[4], mi[4], zs[4], zb[4];
      if (want_bnb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gm = ep.bn_gamma ? ep.bn_gamma[co + e] : 1.f, bt = ep.bn_beta ? ep.bn_beta[co + e] : 0.f;
          is[e] = ep.bn_invstd[co + e];
          mi[e] = -ep.bn_mean[co + e] * is[e];
          zs[e] = gm * is[e];
          zb[e] = fmaf(mi[e], gm, bt);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (live[j]) {
          *(float4 *)(y + pix[j] + co) = make_float4(v[0], v[1], v[2], v[3]);
          if (want_bnb) {
            const float4 xr = *(const float4 *)((const float *)ep.bn_x + pix[j] + co);
            x[0] = xr.x; x[1] = xr.y; x[2] = xr.z; x[3] = xr.w;
          }
        }
        if (want_stats && live[j]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (want_bnb) {
              const float dz = (ep.bn_relu && fmaf(x[e], zs[e], zb[e]) <= 0.f) ? 0.f : v[e];
              s1[i][4 * g + e] = fmaf(dz, fmaf(x[e], is[e], mi[e]), s1[i][4 * g + e]);
              s2[i][4 * g + e] += dz;
            } else {
              s1[i][4 * g + e] += v[e];
              s2[i][4 * g + e] = fmaf(v[e], v[e], s2[i][4 * g + e]);
            }
          }
        }
      }
    }
  }
  if (want_stats) {
    // as conv2d_nhwc_kernel: halving butterfly over the 32 pixel lanes (a lane ends with NV / 32 of its NV partial
    // sums, index v = (lane & 31) * (NV / 32) + k), the pixel waves through LDS in wave order, the blocks through the
    // fixed-order fold -- one fold per 128-channel tile `ct`; a 32-channel block hands in zeros for the other 96
    constexpr int NV = NI * 32, NK = NV / 32;
    __syncthreads();
    float *red = (float *)smem;  // [WPX pixel waves][2][128] then tot [2][128] then flag
    if (CT < 128) {
      for (int i = threadIdx.x; i < WPX * 256; i += 512) red[i] = 0.f;
      __syncthreads();
    }
    float vals[NV];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        vals[i * 32 + r * 2 + 0] = s1[i][r];
        vals[i * 32 + r * 2 + 1] = s2[i][r];
      }
#pragma unroll
    for (int d = 16, n = NV; d >= 1; d >>= 1, n >>= 1) {
      const bool up = (lane & d) != 0;
#pragma unroll
      for (int k = 0; k < n / 2; ++k) {
        const float send = up ? vals[k] : vals[k + n / 2];
        const float keep = up ? vals[k + n / 2] : vals[k];
        vals[k] = keep + __shfl_xor(send, d, 64);
      }
    }
    const int c_in_tile = n0 - ct * Q.ct_full;  // first channel of this block inside its channel tile
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int v = (lane & 31) * NK + k;
      const int i = v >> 5, r = (v >> 1) & 15, stat = v & 1;
      const int cl = c_in_tile + wco * (NI * 32) + i * 32 + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3);
      red[wpx * 256 + stat * 128 + cl] = vals[k];
    }
    __syncthreads();
    float *tot = red + WPX * 256;
    if (threadIdx.x < 256) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < WPX; ++wv) t += red[wv * 256 + threadIdx.x];
      tot[threadIdx.x] = t;
    }
    __syncthreads();
    float *ws = ep.stats + 2 * G.cout + (size_t)ct * ep.ws_stride;
    if (d3d_fold_block_at<512>(tot, 256, ws, ep.counters + ct * ep.ctr_stride, ep.fold_g, (int *)(tot + 256), fold_b,
                               fold_nb)) {
      const int i = threadIdx.x;
      if (i < 256 && (i & 127) < Q.ct_full) ep.stats[(i >> 7) * G.cout + ct * Q.ct_full + (i & 127)] = ws[i];
    }
  }
}

template <int CIN, int PH, int PW, int SCHED>
__global__ void __launch_bounds__(512, 2)
    conv2d_halo_kernel(const void *__restrict__ x_hi, const void *__restrict__ x_lo, const void *__restrict__ w_hi,
                       const void *__restrict__ w_lo, float *__restrict__ y, Conv2dGeom G, Conv2dEpi ep, C2hGrid Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  const int fold_full = Q.n_full / Q.ctiles;  // whole blocks per 128-channel tile
  const int rem = Q.n_items - Q.n_full;       // items cut into four blocks each
  if (bid < Q.n_full) {
    // XCD-aware order over the whole blocks: XCD i works on the i-th contiguous eighth of the items (bijective)
    const int nb = Q.n_full, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile = item / Q.ctiles, ct = item - tile * Q.ctiles;
    const int rem_ct = rem > ct ? (rem - ct + Q.ctiles - 1) / Q.ctiles : 0;
    if (Q.ct_full == 128)
      c2h_body<CIN, PH, PW, SCHED, 2, 2>(x_hi, x_lo, w_hi, w_lo, y, G, ep, Q, tile, ct, ct * 128, tile,
                                         fold_full + Q.cut * rem_ct, smem);
    else
      c2h_body<CIN, PH, PW, SCHED, 2, 1>(x_hi, x_lo, w_hi, w_lo, y, G, ep, Q, tile, ct, ct * 64, tile,
                                         fold_full + Q.cut * rem_ct, smem);
  } else {
    const int j = (bid - Q.n_full) / Q.cut, part = (bid - Q.n_full) - j * Q.cut;
    const int item = Q.n_full + j;
    const int tile = item / Q.ctiles, ct = item - tile * Q.ctiles;  // == j % ctiles: n_full is a multiple of ctiles
    const int rem_ct = (rem - ct + Q.ctiles - 1) / Q.ctiles;
    c2h_body<CIN, PH, PW, SCHED, 1, 1>(x_hi, x_lo, w_hi, w_lo, y, G, ep, Q, tile, ct, ct * Q.ct_full + part * 32,
                                       fold_full + (j / Q.ctiles) * Q.cut + part, fold_full + Q.cut * rem_ct, smem);
  }
}
