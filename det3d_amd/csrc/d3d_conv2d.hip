// Dense 2D convolution on channels-last maps for the RPN (det3d/models/necks/rpn.py:70-159: 3x3 / stride 1-2 /
// 128-256 channels; deblocks 1x1 and 2x2; the reference runs them on cuDNN), gfx950, as an implicit GEMM on MFMA:
//
//   out[p, co] = sum_{tap, ci} x[pix(p, tap), ci] * W[tap][co][ci]       p = (b, oy, ox) row-major
//
// Why not MIOpen (round 2): its assembly igemm ran the 128->128 3x3 layers at 12 % of the bf16 MFMA peak, wanted
// 30 zero-fill launches and 30 cast kernels per step around it, summed its weight gradients with atomics
// (run-to-run noise, not capturable in a hipGraph with find-mode solvers) and has no precision between bf16 and
// the 16x slower f32 -- while north_star's 1e-3 box tolerance needs more than bf16.  Here:
//   * block = 128 output pixels x 128 output channels, 4 waves of 64 x 64 (2 x 2 MFMA 32x32x16 tiles each), D^T
//     form: A = W tile (rows = cout), B = x tile (cols = pixels), so a lane ends up with runs of 4 consecutive
//     channels of ONE pixel -- 8 / 16-byte row stores, no transposes;
//   * both operands are staged by LDS-DMA (buffer_load ... lds, 16 B per lane): rows of BK channels, XOR-swizzled
//     through the SOURCE address (the LDS image of an LDS-DMA is lane-linear) so that ds_read_b128 of 32 different
//     rows at one k-chunk is conflict-free; zero padding = out-of-range buffer offsets (the hardware writes zeros);
//   * two LDS stages, one workgroup barrier per K step: the loads of step s+1 fly while step s multiplies;
//   * SPLIT mode (the contract-compliant precision): activations and weights arrive as two bf16 planes
//     (hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits), three MFMAs per fragment pair (hi*hi + hi*lo + lo*hi),
//     f32 accumulate, f32 output: |error| ~ 2^-16 of the products against 2^-8 for bf16, at 3 bf16 MFMAs = 1/5 of
//     the time of the exact-f32 MFMA.
//   * optional epilogue: per-channel sum / sum of squares of the produced rows (BatchNorm statistics), folded
//     deterministically across blocks (d3d_fold.h).
// The data gradient of a stride-1 layer is the same kernel on the output gradient with mirrored taps and
// [tap][cin][cout] weights; the weight gradient is conv2d_wgrad_kernel below.
#include <stdlib.h>

#include "d3d_common.h"
#include "d3d_fold.h"

typedef __attribute__((ext_vector_type(8))) __bf16 c2_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c2_f32x16;
typedef __attribute__((address_space(3))) void c2_lds_void;

struct Conv2dGeom {
  int batch, h, w, ho, wo;   // input / output map
  int cout;                  // all output channels (grid.y tiles of 128)
  int ksz, stride, pad;      // square kernel
  int m;                     // batch * ho * wo output pixels
  int x_bytes;               // num_records of the activation descriptor
  int w_bytes;               // ... of the weight descriptor
};

struct Conv2dEpi {
  float *stats;        // [2 * cout] result, then one fold workspace per 128-channel tile; or nullptr
  unsigned *counters;  // ticket words, `ctr_stride` per 128-channel tile
  int fold_g, ws_stride, ctr_stride;
  // what is summed per output channel: (sum y | sum y^2) of the produced rows -- BatchNorm statistics -- or, when
  // bn_x is set (a data-gradient launch: the rows produced are dL/dz of the BatchNorm + ReLU that made this conv's
  // input), (sum dz * xhat | sum dz) of that layer = what d3d_bn_bwd_reduce computes, with x = its pre-BN activations
  const void *bn_x;
  const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
  int bn_relu;
  int bn_stage;  // conv2d_halo_kernel: the bn_x tile through LDS (whole pixel rows by LDS-DMA) instead of 16-byte loads at pixel stride
};

// TR launches (the data gradient of a stride-2 layer, round 4): the rows of a block are the pixels of ONE parity class
// (iy & 1, ix & 1) of the gradient map dx -- a pixel of class (py, px) only receives the kernel taps with
// (py + pad - ky) and (px + pad - kx) even, 1 / 2 / 2 / 4 of the nine for a 3x3 layer, and its sources are the pixels
// (i + dy, j + dx) of g with i = iy >> 1, j = ix >> 1, dy = (py + pad - ky) / 2 in {-1, 0, 1}: a stride-1 problem per class
struct Conv2dTr {
  int tiles_end[4];   // cumulative pixel tiles of the classes (py * 2 + px) in launch order
  int hc[4], wc[4];   // class grids: rows / columns of dx with that parity
  int ntap[4];
  unsigned taps[4];   // 8 bits per tap: (dy + 1) | (dx + 1) << 2 | weight slot << 4
};

__device__ __forceinline__ uint16_t c2_f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// BK channels per K step; rows of RB = 2 * BK bytes = CPR 16-byte chunks.  Logical chunk c of row r lives in
// physical chunk c ^ swz(r), swz(r) = (r / rows-per-256-B) % CPR: the 16 rows of a ds_read_b128 lane group then hit
// 16 different 16-byte slots of the 256-byte bank row (lane groups: MI355X_MICROARCH.md, LDS table).
template <int BK> struct C2Tile {
  static constexpr int RB = BK * 2;
  static constexpr int CPR = RB / 16;
  static constexpr int RPB = 256 / RB;
  static constexpr int PLANE = 128 * RB;            // bytes of one 128-row operand plane
  static constexpr int INSTR = 2 * CPR / 4;         // LDS-DMA instructions per wave and plane (4 waves, 64 chunks each)
  __device__ static __forceinline__ int swz(int row) { return (row / RPB) % CPR; }
};

// WPX: waves along the pixel axis (2 -> 128-pixel tiles, two blocks per CU; 4 -> 256-pixel tiles, one block per CU:
//      the weight tile is shared by twice the pixels, 6 instead of 8 LDS-DMA instructions per wave and K step)
// ILV: the LDS-DMA instructions of step s+1 are issued in KSUB portions between the MFMA groups of step s instead of
//      all in front of them (an LDS-DMA costs the issuing wave 60-180 cycles: MI355X_MICROARCH.md; spread out they hide
//      behind the wave's own MFMAs)
//  HB: half-depth K steps (32 channels in bf16, 16 in split mode): half the LDS per block, four blocks per CU instead of
//      two -- more waves to cover the wait + barrier at the top of every K step, at twice the barriers per FLOP
//  PXT: 32-pixel MFMA tiles per wave (2 -> 64 x 64 wave tiles; 1 -> 64 x 32: with WPX = 4 the same 128-pixel block on 8
//      waves, twice the waves per CU to cover the waits of the two-stage loop, 1.5x the LDS fragment reads)
template <int CIN, bool SPLIT, bool OUT_F32, int WPX, bool ILV, bool HB = false, int PXT = 2, bool TR = false>
__global__ void __launch_bounds__(128 * WPX, (WPX * PXT == 4 ? (HB ? 4 : 2) : (HB ? 2 : 1)))
    conv2d_nhwc_kernel(const void *__restrict__ x_hi, const void *__restrict__ x_lo, const void *__restrict__ w_hi,
                       const void *__restrict__ w_lo, void *__restrict__ y, Conv2dGeom G, Conv2dEpi ep, Conv2dTr R) {
  constexpr int BK = (SPLIT ? 32 : 64) / (HB ? 2 : 1);
  using T = C2Tile<BK>;
  constexpr int WAVES = 2 * WPX, NTH = WAVES * 64, BM = 32 * PXT * WPX;
  constexpr int KC = CIN / BK;                 // K steps per tap
  constexpr int NP = SPLIT ? 2 : 1;            // planes per operand
  constexpr int APLANE = BM * T::RB, BPLANE = 128 * T::RB;
  constexpr int STAGE = NP * (APLANE + BPLANE);  // A planes then B planes
  constexpr int KSUB = BK / 16;                // MFMA k-substeps per K step
  constexpr int AI = BM * T::CPR / 64 / WAVES;   // LDS-DMA instructions per wave: activation plane
  constexpr int BI = 128 * T::CPR / 64 / WAVES;  // ... weight plane
  static_assert(CIN % BK == 0 && AI >= 1 && BI >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][STAGE]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wco = wave / WPX, wpx = wave % WPX;  // 2 x WPX waves: cout half, pixel quarter / half
  // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs; give XCD i the i-th contiguous
  // eighth of the pixel tiles so that the halo rows its tiles share stay in ITS L2 (bijective for any grid size)
  int tile;
  {
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // TR: the block's parity class (uniform), its grid and its first pixel within the class
  int cls = 0, cls_h = 1, cls_w = 1, cls_m = 0;
  if constexpr (TR) {
    cls = (tile >= R.tiles_end[0] ? 1 : 0) + (tile >= R.tiles_end[1] ? 1 : 0) + (tile >= R.tiles_end[2] ? 1 : 0);
    tile -= cls ? R.tiles_end[cls - 1] : 0;
    cls_h = R.hc[cls];
    cls_w = R.wc[cls];
    cls_m = G.batch * cls_h * cls_w;
  }
  const int m0 = tile * BM;
  const int n0 = (int)blockIdx.y * 128;

  // ---- per-thread staging slots: AI rows of the activation tile, BI rows of the weight tile
  uint32_t a_off[AI];            // byte offset of tap (0,0) of the row's pixel + this lane's (swizzled) chunk
  uint32_t a_vy[AI], a_vx[AI];   // bit k set: tap row / column k is inside the map
  uint32_t b_off[BI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int idx = (wave * AI + j) * 64 + lane;
    const int row = idx / T::CPR, cp = idx % T::CPR;
    const int c = cp ^ T::swz(row);
    const int m = m0 + row;
    if constexpr (TR) {
      // row = pixel (b, i, jj) of the class grid; slot k of the masks = source row / column i + k - 1 of g
      const int mm = m < cls_m ? m : cls_m - 1;
      const int b = mm / (cls_h * cls_w), rem = mm - b * (cls_h * cls_w);
      const int i = rem / cls_w, jj = rem - i * cls_w;
      uint32_t vy = 0, vx = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vy |= (uint32_t)((i + k - 1 >= 0 && i + k - 1 < G.h) ? 1 : 0) << k;
        vx |= (uint32_t)((jj + k - 1 >= 0 && jj + k - 1 < G.w) ? 1 : 0) << k;
      }
      if (m >= cls_m) vy = vx = 0;
      a_vy[j] = vy;
      a_vx[j] = vx;
      a_off[j] = (uint32_t)(((b * G.h + i) * G.w + jj) * (CIN * 2) + c * 16);
      continue;
    }
    const int mm = m < G.m ? m : G.m - 1;
    const int b = mm / (G.ho * G.wo), rem = mm - b * (G.ho * G.wo);
    const int oy = rem / G.wo, ox = rem - oy * G.wo;
    const int iy0 = oy * G.stride - G.pad, ix0 = ox * G.stride - G.pad;
    uint32_t vy = 0, vx = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vy |= (uint32_t)((k < G.ksz && iy0 + k >= 0 && iy0 + k < G.h) ? 1 : 0) << k;
      vx |= (uint32_t)((k < G.ksz && ix0 + k >= 0 && ix0 + k < G.w) ? 1 : 0) << k;
    }
    if (m >= G.m) vy = vx = 0;
    a_vy[j] = vy;
    a_vx[j] = vx;
    a_off[j] = (uint32_t)(((b * G.h + iy0) * G.w + ix0) * (CIN * 2) + c * 16);  // may wrap for invalid taps: masked
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int idx = (wave * BI + j) * 64 + lane;
    const int row = idx / T::CPR, cp = idx % T::CPR;
    b_off[j] = (uint32_t)((n0 + row) * (CIN * 2) + (cp ^ T::swz(row)) * 16);
  }
  const __amdgpu_buffer_rsrc_t xr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)x_hi, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)w_hi, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)(SPLIT ? x_lo : x_hi), 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)(SPLIT ? w_lo : w_hi), 0, G.w_bytes, 0x00020000);

  const int nks = (TR ? R.ntap[cls] : G.ksz * G.ksz) * KC;
  // stage the slots with (slot % PARTS) == part of K step `ks` (tap = ks / KC, channel block = ks % KC) into LDS
  // stage `buf`; slots 0 .. AI-1 are activation rows, AI .. AI+BI-1 weight rows
  auto stage = [&](int ks, int buf, int part, int parts) {
    const int tap = ks / KC, kc = ks - tap * KC;
    int ky = tap / G.ksz, kx = tap - ky * G.ksz, wslot = tap;
    if constexpr (TR) {  // (ky, kx) = mask slots of the tap's source offset (dy + 1, dx + 1)
      const unsigned code = (R.taps[cls] >> (8 * tap)) & 0xffu;
      ky = (int)(code & 3u);
      kx = (int)((code >> 2) & 3u);
      wslot = (int)(code >> 4);
    }
    const uint32_t tap_off = TR ? (uint32_t)(((ky - 1) * G.w + (kx - 1)) * (CIN * 2) + kc * (BK * 2))
                                : (uint32_t)((ky * G.w + kx) * (CIN * 2) + kc * (BK * 2));
    const uint32_t w_tap_off = (uint32_t)(wslot * G.cout * (CIN * 2) + kc * (BK * 2));
    unsigned char *sa = smem + buf * STAGE, *sb = sa + NP * APLANE;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      if (j % parts != part) continue;
      const bool ok = ((a_vy[j] >> ky) & 1u) && ((a_vx[j] >> kx) & 1u);
      const uint32_t off = ok ? a_off[j] + tap_off : D3D_OOB_OFFSET;
      const int lds_off = (wave * AI + j) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sa + lds_off), 16, (int)off, 0, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sa + APLANE + lds_off), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      if ((AI + j) % parts != part) continue;
      const uint32_t off = b_off[j] + w_tap_off;
      const int lds_off = (wave * BI + j) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_hi, (c2_lds_void *)(sb + lds_off), 16, (int)off, 0, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_lo, (c2_lds_void *)(sb + BPLANE + lds_off), 16, (int)off, 0, 0, 0);
    }
  };

  c2_f32x16 acc[2][PXT];  // [cout tile][pixel tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PXT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: lane reads row (lane & 31) of a 32-row tile, k-chunk (lane >> 5) of each 16-wide substep
  const int frow = lane & 31, fk = lane >> 5;
  int a_rd[2], b_rd[PXT];  // byte offsets of (row, chunk 0) + the row's swizzle, per tile
  int a_sw[2], b_sw[PXT];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int wr = wco * 64 + t * 32 + frow;  // weight-tile row (cout)
    a_rd[t] = wr * T::RB;
    a_sw[t] = T::swz(wr);
  }
#pragma unroll
  for (int t = 0; t < PXT; ++t) {
    const int xr = wpx * (32 * PXT) + t * 32 + frow;  // activation-tile row (pixel)
    b_rd[t] = xr * T::RB;
    b_sw[t] = T::swz(xr);
  }

  stage(0, 0, 0, 1);
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks & 1;
    // my loads of step ks have landed; after the barrier everyone's have, and every wave has finished reading the
    // other stage (step ks - 1), which the next loads overwrite
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = ks + 1 < nks;
    if (!ILV && more) stage(ks + 1, buf ^ 1, 0, 1);
    const unsigned char *sa = smem + buf * STAGE, *sb = sa + NP * APLANE;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
      const int chunk = kk * 2 + fk;
      c2_bf16x8 wa[2][NP], xb[PXT][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          wa[t][p] = *(const c2_bf16x8 *)(sb + p * BPLANE + a_rd[t] + ((chunk ^ a_sw[t]) << 4));
#pragma unroll
        for (int t = 0; t < PXT; ++t)
          xb[t][p] = *(const c2_bf16x8 *)(sa + p * APLANE + b_rd[t] + ((chunk ^ b_sw[t]) << 4));
      }
      if (ILV && more) stage(ks + 1, buf ^ 1, kk, KSUB);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][0], xb[j][0], acc[i][j], 0, 0, 0);
          if constexpr (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][0], xb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][1], xb[j][0], acc[i][j], 0, 0, 0);
          }
        }
      if (ILV) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue.  C/D of 32x32: col (pixel) = lane & 31, row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const bool want_stats = ep.stats != nullptr;
  float s1[2][16], s2[2][16];  // per cout tile: this lane's channels, summed over its pixels
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s1[i][r] = s2[i][r] = 0.f;
  }
  const bool want_bnb = want_stats && ep.bn_x != nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int co_base = n0 + wco * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co_base + 8 * g;
      // BatchNorm-backward mode: per-channel coefficients of z = x * zs + zb (the ReLU mask) and xhat = x * is + mi,
      // the arithmetic of bn_bwd_reduce_kernel / bn_bwd_dx_kernel (d3d_bn.hip)
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
      for (int j = 0; j < PXT; ++j) {
        int m = m0 + wpx * (32 * PXT) + j * 32 + (lane & 31);
        bool live = m < G.m;
        if constexpr (TR) {  // class pixel -> row of dx
          live = m < cls_m;
          const int mm = live ? m : 0;
          const int b = mm / (cls_h * cls_w), rem = mm - b * (cls_h * cls_w);
          const int ii = rem / cls_w, jj = rem - ii * cls_w;
          m = (b * G.ho + 2 * ii + (cls >> 1)) * G.wo + 2 * jj + (cls & 1);
        }
        float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (OUT_F32) {
          if (live) *(float4 *)((float *)y + (int64_t)m * G.cout + co) = make_float4(v[0], v[1], v[2], v[3]);
          if (want_bnb && live) {
            const float4 xr = *(const float4 *)((const float *)ep.bn_x + (int64_t)m * G.cout + co);
            x[0] = xr.x; x[1] = xr.y; x[2] = xr.z; x[3] = xr.w;
          }
        } else {
          const uint16_t h0 = c2_f32_to_bf16(v[0]), h1 = c2_f32_to_bf16(v[1]), h2 = c2_f32_to_bf16(v[2]),
                         h3 = c2_f32_to_bf16(v[3]);
          if (live) {
            uint2 pk;
            pk.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
            pk.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
            *(uint2 *)((uint16_t *)y + (int64_t)m * G.cout + co) = pk;
          }
          v[0] = __uint_as_float((uint32_t)h0 << 16); v[1] = __uint_as_float((uint32_t)h1 << 16);
          v[2] = __uint_as_float((uint32_t)h2 << 16); v[3] = __uint_as_float((uint32_t)h3 << 16);
          if (want_bnb && live) {
            const uint2 xr = *(const uint2 *)((const uint16_t *)ep.bn_x + (int64_t)m * G.cout + co);
            x[0] = __uint_as_float(xr.x << 16); x[1] = __uint_as_float(xr.x & 0xffff0000u);
            x[2] = __uint_as_float(xr.y << 16); x[3] = __uint_as_float(xr.y & 0xffff0000u);
          }
        }
        if (want_stats && live) {  // sums over the values as stored
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
    // The lane holds 64 partial sums (2 cout tiles x 16 channels x {sum, sumsq}) of ITS pixels; 32 lanes (pixels)
    // share them.  Halving butterfly: at distance d = 16, 8, 4, 2, 1 a lane keeps one half of its values and hands
    // the other half to its partner -- 32 + 16 + 8 + 4 + 2 = 62 cross-lane moves instead of 5 x 64, fixed order.
    // Afterwards lane l holds the two totals with index v = ((l & 31) << 1) | k, k = 0, 1.
    __syncthreads();  // every wave is done with the operand stages
    float *red = (float *)smem;  // [WPX pixel waves][2][128] then tot [2][128] then flag
    float vals[64];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        vals[i * 32 + r * 2 + 0] = s1[i][r];
        vals[i * 32 + r * 2 + 1] = s2[i][r];
      }
#pragma unroll
    for (int d = 16, n = 64; d >= 1; d >>= 1, n >>= 1) {
      const bool up = (lane & d) != 0;
#pragma unroll
      for (int k = 0; k < n / 2; ++k) {
        const float send = up ? vals[k] : vals[k + n / 2];
        const float keep = up ? vals[k + n / 2] : vals[k];
        vals[k] = keep + __shfl_xor(send, d, 64);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = ((lane & 31) << 1) | k;         // = i * 32 + r * 2 + stat
      const int i = v >> 5, r = (v >> 1) & 15, stat = v & 1;
      const int cl = wco * 64 + i * 32 + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3);  // channel within the block
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
    float *ws = ep.stats + 2 * G.cout + (size_t)blockIdx.y * ep.ws_stride;
    if (d3d_fold_block<NTH>(tot, 256, ws, ep.counters + blockIdx.y * ep.ctr_stride, ep.fold_g, (int *)(tot + 256))) {
      // this block holds the channel tile's totals (thread i wrote ws[i]): place them in the [sum | sumsq] vector
      const int i = threadIdx.x;
      if (i < 256) ep.stats[(i >> 7) * G.cout + n0 + (i & 127)] = ws[i];
    }
  }
}

#include "d3d_conv2d_halo.h"

// ---------------------------------------------------------------------------------------------------- weight gradient
//   dW[tap][co][ci] = sum_p g[p][co] * x[pix(p, tap)][ci]         (stride 1)
// A GEMM whose K dimension is the pixel axis, which is NOT contiguous in either operand (channels are): tiles of
// KP pixels x 128 channels are staged row-major by LDS-DMA and read as MFMA operands with the transposing LDS read
// (ds_read_b64_tr_b16: four pixels of one channel per lane).  One block per (pixel split, tap, 128 x 128 weight tile):
// a block owns one tap, one 128 x 128 weight tile and a contiguous range of pixels; it writes ONE partial tile, the
// partial tiles are summed in a fixed order by conv2d_wgrad_reduce_kernel: deterministic, no atomics.
// WCI: waves along the input-channel axis of the 128 x 128 weight tile (2 -> four waves of 64 x 64; 4 -> eight waves of
// 64 x 32: twice the waves per CU to cover the waits of the two-stage loop, as in the forward kernel)
template <bool SPLIT, int WCI>
__global__ void __launch_bounds__(128 * WCI, 2)
    conv2d_wgrad_kernel(const void *__restrict__ g_hi, const void *__restrict__ g_lo, const void *__restrict__ x_hi,
                        const void *__restrict__ x_lo, float *__restrict__ part, Conv2dGeom G, int cin,
                        int pixels_per_split, int nsplit) {
  constexpr int KP = SPLIT ? 32 : 64;          // pixels per K step
  constexpr int NP = SPLIT ? 2 : 1;
  constexpr int PLANE = KP * 256;              // KP rows of 128 channels (256 B)
  constexpr int STAGE = 2 * NP * PLANE;        // g planes then x planes
  constexpr int CIT = 4 / WCI;                 // 32-channel MFMA tiles per wave along cin
  constexpr int INSTR = PLANE / 1024 / (2 * WCI);  // LDS-DMA instructions per wave and plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wco = wave / WCI, wci = wave % WCI;
  // 1-D grid; XCD-aware order with the tap fastest: the 9 taps of a pixel range read the same g rows and (shifted) x
  // rows, and consecutive logical ids stay on one XCD, i.e. behind one L2
  int logical;
  {
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int taps = G.ksz * G.ksz;
  const int tap = logical % taps;
  const int split = (logical / taps) % nsplit;
  const int wtile = logical / (taps * nsplit);
  const int cin_tiles = cin / 128;
  const int co0 = (wtile / cin_tiles) * 128, ci0 = (wtile % cin_tiles) * 128;
  const int ky = tap / G.ksz, kx = tap - ky * G.ksz;
  const int p_begin = split * pixels_per_split;
  int p_end = p_begin + pixels_per_split;
  if (p_end > G.m) p_end = G.m;
  const int nks = (p_end - p_begin + KP - 1) / KP;

  // staging: one LDS-DMA instruction = 4 pixel rows x 256 B; physical chunk cp of row r holds logical chunk
  // cp ^ ((r & 3) << 2): the 4 pixel rows of a transposing read then sit in 4 different 64-byte quarters of the
  // 256-byte bank row
  const __amdgpu_buffer_rsrc_t gr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)g_hi, 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)(SPLIT ? g_lo : g_hi), 0, G.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)x_hi, 0, G.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)(SPLIT ? x_lo : x_hi), 0, G.x_bytes, 0x00020000);
  // per staging slot: the pixel this lane loads in the NEXT stage() call, as (image, row, column); stage() is called
  // for ks = 0, 1, 2, ... in order and advances the slot by KP pixels (no divisions in the loop)
  int sp[INSTR], sb_[INSTR], sy[INSTR], sx_[INSTR];
  uint32_t schunk[INSTR];
#pragma unroll
  for (int j = 0; j < INSTR; ++j) {
    const int inst = wave * INSTR + j;
    const int row = inst * 4 + (lane >> 4), cp = lane & 15;
    schunk[j] = (uint32_t)((cp ^ ((row & 3) << 2)) * 16);
    const int p = p_begin + row;
    sp[j] = p;
    const int pc = p < G.m ? p : G.m - 1;
    sb_[j] = pc / (G.ho * G.wo);
    const int rem = pc - sb_[j] * (G.ho * G.wo);
    sy[j] = rem / G.wo;
    sx_[j] = rem - sy[j] * G.wo;
  }
  auto stage = [&](int buf) {
    unsigned char *sg = smem + buf * STAGE, *sx = sg + NP * PLANE;
#pragma unroll
    for (int j = 0; j < INSTR; ++j) {
      const int inst = wave * INSTR + j;
      const bool live = sp[j] < p_end;
      const int iy = sy[j] * G.stride - G.pad + ky, ix = sx_[j] * G.stride - G.pad + kx;
      const bool inside = live && iy >= 0 && iy < G.h && ix >= 0 && ix < G.w;
      const uint32_t goff = live ? (uint32_t)(sp[j] * (G.cout * 2) + co0 * 2) + schunk[j] : D3D_OOB_OFFSET;
      const uint32_t xoff = inside ? (uint32_t)(((sb_[j] * G.h + iy) * G.w + ix) * (cin * 2) + ci0 * 2) + schunk[j]
                                   : D3D_OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_hi, (c2_lds_void *)(sg + inst * 1024), 16, (int)goff, 0, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(gr_lo, (c2_lds_void *)(sg + PLANE + inst * 1024), 16, (int)goff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_hi, (c2_lds_void *)(sx + inst * 1024), 16, (int)xoff, 0, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr_lo, (c2_lds_void *)(sx + PLANE + inst * 1024), 16, (int)xoff, 0, 0, 0);
      // advance by KP pixels
      sp[j] += KP;
      sx_[j] += KP;
      while (sx_[j] >= G.wo) {
        sx_[j] -= G.wo;
        if (++sy[j] >= G.ho) {
          sy[j] = 0;
          ++sb_[j];
        }
      }
    }
  };

  c2_f32x16 acc[2][CIT];  // [co tile][ci tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposing fragment read.  MFMA 32x32x16 operand: lane l holds row (l & 31), k = (l >> 5) * 8 .. + 8.  One
  // ds_read_b64_tr_b16 serves a 16-lane group with a 4 (pixels) x 16 (channels) block: lane i of the group passes the
  // address of pixel row i / 4, channels (i % 4) * 4 .. + 4 and receives channel i of the 4 pixels.
  //   lanes  0-15: channels  0-15, pixels k0 + 0..3 (first read), +4..7 (second read)
  //   lanes 16-31: channels 16-31           "
  //   lanes 32-63: the same channels, pixels + 8
  const int i16 = lane & 15;
  const int prow = ((lane >> 5) << 3) + (i16 >> 2);                 // pixel row within a 16-pixel substep (first read)
  const int chan = (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);     // channel within a 32-channel tile
  // The reads are inline asm: through the builtin, hipcc orders every transposing read behind the LDS-DMA
  // instructions of the NEXT stage that were issued just before (s_waitcnt vmcnt(0) in front of the first read: the
  // loads it should overlap with were drained first).  The stage being read was waited for at the top of the loop
  // (vmcnt(0) + barrier); the asm block below waits for its own reads (lgkmcnt) before the MFMAs use them.
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  // byte offset inside a plane of this lane's 8 bytes for (32-channel tile, substep 0, first read); the second read is
  // 4 pixel rows (1024 B) further, a substep 16 rows (4096 B): the swizzle term (r & 3) is the same for all of them
  uint32_t fo[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ch = t * 32 + chan;
    fo[t] = (uint32_t)(prow * 256 + ((((ch * 2) >> 4) ^ ((prow & 3) << 2)) << 4) + ((ch * 2) & 15));
  }
  auto tr_read = [&](uint32_t addr) -> u2 {
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };

  if (nks > 0) stage(0);
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks + 1 < nks) stage(buf ^ 1);
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
      u2 gr[2][NP][2], xr[CIT][NP][2];  // [tile][plane][first / second read]
      const uint32_t gbase = lds_base + (uint32_t)(buf * STAGE + kk * 4096);
      const uint32_t xbase = gbase + (uint32_t)(NP * PLANE);
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int t = 0; t < 2; ++t) gr[t][p][h] = tr_read(gbase + (uint32_t)(p * PLANE + h * 1024) + fo[wco * 2 + t]);
#pragma unroll
          for (int t = 0; t < CIT; ++t) xr[t][p][h] = tr_read(xbase + (uint32_t)(p * PLANE + h * 1024) + fo[wci * CIT + t]);
        }
      // wait for the reads; the operands are tied to the wait (first statement) / ordered behind it (second, empty
      // statement: volatile asm statements keep their order) so that no MFMA is scheduled in front of it
      if constexpr (NP == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(gr[0][0][0]), "+v"(gr[0][0][1]), "+v"(gr[1][0][0]), "+v"(gr[1][0][1]), "+v"(gr[0][1][0]),
                       "+v"(gr[0][1][1]), "+v"(gr[1][1][0]), "+v"(gr[1][1][1]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gr[0][0][0]), "+v"(gr[0][0][1]), "+v"(gr[1][0][0]), "+v"(gr[1][0][1]));
#pragma unroll
      for (int t = 0; t < CIT; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(xr[t][p][0]), "+v"(xr[t][p][1]));
      c2_bf16x8 ga[2][NP], xb[CIT][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const u4 gv = {gr[t][p][0][0], gr[t][p][0][1], gr[t][p][1][0], gr[t][p][1][1]};
          ga[t][p] = __builtin_bit_cast(c2_bf16x8, gv);
        }
#pragma unroll
        for (int t = 0; t < CIT; ++t) {
          const u4 xv = {xr[t][p][0][0], xr[t][p][0][1], xr[t][p][1][0], xr[t][p][1][1]};
          xb[t][p] = __builtin_bit_cast(c2_bf16x8, xv);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][0], xb[j][0], acc[i][j], 0, 0, 0);
          if constexpr (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][0], xb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[i][1], xb[j][0], acc[i][j], 0, 0, 0);
          }
        }
    }
  }
  // partial tile: part[split][tap][co][ci] (full cout x cin extent), D rows = co, cols = ci
  float *dst = part + ((int64_t)split * (G.ksz * G.ksz) + tap) * ((int64_t)G.cout * cin);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j) {
      const int ci = ci0 + wci * (32 * CIT) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(int64_t)co * cin + ci] = acc[i][j][r];
      }
    }
}

#include "d3d_conv2d_wgrad3.h"

// dw = sum over splits of part[s], in a fixed order (block = 8 float4 elements x 32 slab groups: a thread adds slabs g,
// g + 32, ... in that order, the groups are folded through LDS in group order), written in nn.Conv2d's parameter layout
// [cout][cin][kh][kw]; part is [split][tap][cout][cin]
__global__ void __launch_bounds__(256)
    conv2d_wgrad_reduce_kernel(const float4 *__restrict__ part, int splits, int taps, int cout, int cin,
                               float *__restrict__ dw) {
  __shared__ float4 red[32][8];
  const int64_t n4 = (int64_t)taps * cout * cin / 4;
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int64_t e = (int64_t)blockIdx.x * 8 + el;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n4) {
#pragma unroll 4
    for (int s0 = g; s0 < splits; s0 += 32) {
      const float4 u = part[(int64_t)s0 * n4 + e];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  }
  red[g][el] = a;
  __syncthreads();
  if (g == 0 && e < n4) {
    float4 r = red[0][el];
#pragma unroll
    for (int j = 1; j < 32; ++j) {
      r.x += red[j][el].x; r.y += red[j][el].y; r.z += red[j][el].z; r.w += red[j][el].w;
    }
    const int64_t e1 = e * 4;  // (tap, co, ci .. ci + 3)
    const int ci = (int)(e1 % cin), co = (int)((e1 / cin) % cout), tap = (int)(e1 / ((int64_t)cin * cout));
    float *o = dw + ((int64_t)co * cin + ci) * taps + tap;
    o[0] = r.x; o[taps] = r.y; o[2 * taps] = r.z; o[3 * taps] = r.w;
  }
}

// nn.Conv2d weight [cout][cin][kh][kw] f32 -> the kernels' operands, one launch:
//   fwd   [tap][cout][cin]            W[tap][co][ci] = w[co][ci][ky][kx]
//   dgrad [mirrored tap][cin][cout]   (the forward operand of the transposed problem: stride-1 data gradient)
// each as bf16 (hi only) or as hi / lo planes (split mode)
__global__ void __launch_bounds__(256)
    conv2d_pack_kernel(const float *__restrict__ w, int cout, int cin, int taps, uint16_t *__restrict__ f_hi,
                       uint16_t *__restrict__ f_lo, uint16_t *__restrict__ d_hi, uint16_t *__restrict__ d_lo) {
  const int total = cout * cin * taps;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int tap = t % taps, ci = (t / taps) % cin, co = t / (taps * cin);
    const float v = w[t];
    const uint16_t h = c2_f32_to_bf16(v);
    const uint16_t l = c2_f32_to_bf16(v - __uint_as_float((uint32_t)h << 16));
    const int fi = (tap * cout + co) * cin + ci;
    const int di = ((taps - 1 - tap) * cin + ci) * cout + co;
    if (f_hi) f_hi[fi] = h;
    if (f_lo) f_lo[fi] = l;
    if (d_hi) d_hi[di] = h;
    if (d_lo) d_lo[di] = l;
  }
}

// f32 [n] -> hi = bf16(v), lo = bf16(v - hi) planes (n % 4 == 0)
__global__ void __launch_bounds__(256)
    conv2d_split_kernel(const float4 *__restrict__ x, int64_t n4, uint2 *__restrict__ hi, uint2 *__restrict__ lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = x[i];
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = c2_f32_to_bf16(f[j]);
      l[j] = c2_f32_to_bf16(f[j] - __uint_as_float(h[j] << 16));
    }
    hi[i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    lo[i] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
  }
}

// ---------------------------------------------------------------------------------------------------- host side
// tiling variant of the forward kernel (d3d_set_tuning knob 7, D3D_CONV2D_VARIANT): bit 0 = interleaved LDS-DMA issue,
// bit 1 = 256-pixel tiles (8 waves, one block per CU), bit 2 = half-depth K steps (twice the blocks per CU),
// 8 = 128-pixel tiles on 8 waves (64 x 32 wave tiles; the default: 62.6 vs 67.7 us bf16, 143.8 vs 150.5 us split on the
// [4,200,176,128] map), 9 = 8 + interleaved issue
// waves per block of the weight-gradient kernel (knob 8, D3D_CONV2D_WGRAD_WAVES): 8 (64 x 32 wave tiles) or 4;
// 16 / 17: conv2d_wgrad3_kernel for the 3x3 stride-1 layers in split mode (staggered wave groups / one barrier per K
// step), other layers as with 8
// measured (profiles/round4_conv2d_variants.txt, split): [4,200,176,128] -> 128: 166 (8) / 130 (17) / 115 us (16);
// [4,128,128,256] -> 256: 281 / 206 / 182 us

static int c2_check(const char *who, int batch, int h, int w, int cin, int cout, int ksz, int stride, int pad,
                    int mode) {
  D3D_REQUIRE(mode == D3D_CONV_BF16 || mode == D3D_CONV_SPLIT, "%s: mode %d", who, mode);
  D3D_REQUIRE(batch >= 1 && h >= 1 && w >= 1, "%s: map %d x %d x %d", who, batch, h, w);
  D3D_REQUIRE(cin == 64 || cin == 128 || cin == 256 || ((cin == 384 || cin == 512) && ksz == 1),
              "%s: cin=%d (64, 128, 256; 384 / 512 for 1x1 layers)", who, cin);
  // (64 output channels: the 3x3 stride-1 layers in split mode only -- conv2d_halo_kernel's 64-channel block shape)
  D3D_REQUIRE((cout >= 128 && cout % 128 == 0) || (cout == 64 && ksz == 3 && stride == 1 && mode == D3D_CONV_SPLIT),
              "%s: cout=%d (multiples of 128; 64 for split 3x3 stride-1 layers)", who, cout);
  D3D_REQUIRE(ksz >= 1 && ksz <= 3 && stride >= 1 && stride <= 2 && pad >= 0 && pad < ksz,
              "%s: kernel %d stride %d pad %d", who, ksz, stride, pad);
  D3D_REQUIRE((int64_t)batch * h * w * cin * 2 < (int64_t)D3D_RSRC_BYTES, "%s: the input map exceeds 2 GB", who);
  return D3D_OK;
}

extern "C" int d3d_conv2d_supported(int cin, int cout, int ksz, int stride) {
  // (the same rule as c2_check: 384 / 512 input channels -- the concatenated deblock maps the stacked task heads read --
  // exist for 1x1 layers only)
  return (cin == 64 || cin == 128 || cin == 256 || ((cin == 384 || cin == 512) && ksz == 1)) && cout >= 128 &&
         cout % 128 == 0 && ksz >= 1 && ksz <= 3 && (stride == 1 || stride == 2);
}
// 3x3 stride-1 layers in split mode also come with 64 output channels (conv2d_halo_kernel; forward and data gradient)
extern "C" int d3d_conv2d_supported_split3x3(int cin, int cout) {
  return (cin == 64 || cin == 128 || cin == 256) && (cout == 64 || (cout >= 128 && cout % 128 == 0));
}

// conv2d_halo_kernel's grid (d3d_conv2d_halo.h): items = 16 x 16 patches x 128-channel tiles; the items past the last
// multiple of 256 (one block per CU and round) are cut into four 32-channel blocks each when that fills the last round
// better than whole blocks would -- at most 128 such items, i.e. two rounds of short blocks (D3D_CONV2D_TAIL=0: never)
struct C2hPlan {
  C2hGrid q;
  int blocks;    // grid size
  int fold_nb;   // blocks folding into one 128-channel tile's statistics (the largest over the tiles)
};
static C2hPlan c2h_plan(int batch, int ho, int wo, int cout) {
  C2hPlan p;
  p.q.tiles_x = (wo + 15) / 16;
  p.q.tiles_y = (ho + 15) / 16;
  p.q.ct_full = cout % 128 == 0 ? 128 : 64;
  p.q.cut = p.q.ct_full / 32;
  p.q.ctiles = cout / p.q.ct_full;
  p.q.n_items = batch * p.q.tiles_x * p.q.tiles_y * p.q.ctiles;
  const int rem = p.q.n_items % 256;
  // the statistics fold numbers a tile's blocks as (whole blocks) + (cut blocks): that needs the whole blocks to end on a
  // channel-tile boundary -- always true for 1 / 2 / 4 tiles (256 | n_full), not for cout = 384 / 640 / 768
  const bool cut = g_c2_tail && rem > 0 && rem <= 128 && (p.q.n_items - rem) % p.q.ctiles == 0;
  p.q.n_full = cut ? p.q.n_items - rem : p.q.n_items;
  p.blocks = p.q.n_full + p.q.cut * (p.q.n_items - p.q.n_full);
  const int rem_items = p.q.n_items - p.q.n_full;
  p.fold_nb = p.q.n_full / p.q.ctiles + p.q.cut * ((rem_items + p.q.ctiles - 1) / p.q.ctiles);
  return p;
}

extern "C" size_t d3d_conv2d_stats_ws_floats(int batch, int ho, int wo, int cout) {
  const int64_t m = (int64_t)batch * ho * wo;
  if (m >= 1 && cout == 64) {  // 64-channel layers: the halo-patch kernel only (one 64-channel tile, 128-wide slabs)
    const C2hPlan hp = c2h_plan(batch, ho, wo, cout);
    if (hp.fold_nb > D3D_FOLD_MAX_BLOCKS) return 0;
    return 2 * (size_t)cout + d3d_fold_ws_floats(hp.fold_nb, 256);
  }
  if (m < 1 || cout < 128 || cout % 128) return 0;
  const int64_t nb = (m + 127) / 128;
  if (nb > D3D_FOLD_MAX_BLOCKS) return 0;
  // either tiling of the forward kernels: 128-pixel runs, or 16 x 16 patches (conv2d_halo_kernel)
  const int nh = c2h_plan(batch, ho, wo, cout).fold_nb;
  size_t a = d3d_fold_ws_floats((int)nb, 256), b = d3d_fold_ws_floats(nh, 256);
  return 2 * (size_t)cout + (size_t)(cout / 128) * (a > b ? a : b);
}

static int c2_geom(const char *who, Conv2dGeom &G, int batch, int h, int w, int cin, int cout, int ksz, int stride,
                   int pad) {
  G.batch = batch; G.h = h; G.w = w;
  G.ho = (h + 2 * pad - ksz) / stride + 1;
  G.wo = (w + 2 * pad - ksz) / stride + 1;
  D3D_REQUIRE(G.ho >= 1 && G.wo >= 1, "%s: empty output", who);
  G.cout = cout; G.ksz = ksz; G.stride = stride; G.pad = pad;
  const int64_t m = (int64_t)batch * G.ho * G.wo;
  D3D_REQUIRE(m < ((int64_t)1 << 30) && m * cout * 2 < (int64_t)D3D_RSRC_BYTES, "%s: output too large", who);
  G.m = (int)m;
  G.x_bytes = (int)((int64_t)batch * h * w * cin * 2);
  G.w_bytes = ksz * ksz * cout * cin * 2;
  return D3D_OK;
}

extern "C" int d3d_conv2d_nhwc_fwd(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo, void *y,
                                   int batch, int h, int w, int cin, int cout, int ksz, int stride, int pad,
                                   int mode, float *stats_ws, size_t stats_ws_floats, const D3dDenseBnBackward *bn,
                                   d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = c2_check("d3d_conv2d_nhwc_fwd", batch, h, w, cin, cout, ksz, stride, pad, mode);
  if (rc) return rc;
  const bool split = mode == D3D_CONV_SPLIT;
  D3D_REQUIRE(x_hi && w_hi && y && (!split || (x_lo && w_lo)), "d3d_conv2d_nhwc_fwd: null pointer");
  Conv2dGeom G;
  rc = c2_geom("d3d_conv2d_nhwc_fwd", G, batch, h, w, cin, cout, ksz, stride, pad);
  if (rc) return rc;
  // measured on the [4,200,176,128] map (profiles/round3_conv2d_variants.txt): bf16 62.3 (8) / 70.7 (9) / 66.6 (0) us,
  // split 145.6 (8) / 138.7 (9) / 146.9 (0) us
  // defaults (profiles/round4_conv2d_variants.txt, split).  [4,200,176,128] -> 128 (572 items: the last round is cut
  // into 32-channel blocks): 9 = 157 us, 17 = 138, 16 = 125, 18 = 117 (with the statistics epilogue 173 / 148 / 145 /
  // 128); [4,128,128,256] -> 256 (512 items, whole blocks only): 232 / 182 / 169 / 174.  So: 18 where the launch has
  // short blocks, 16 where it has not.
  int variant = g_c2_variant >= 0 ? g_c2_variant : (split ? (ksz == 3 && stride == 1 ? 16 : 9) : 8);
  if (g_c2_variant < 0 && variant == 16) {
    const C2hPlan hp0 = c2h_plan(batch, G.ho, G.wo, cout);
    if (hp0.q.n_full != hp0.q.n_items) variant = 18;
  }
  // 16 / 17: the halo-patch kernel (d3d_conv2d_halo.h; 3x3 stride-1 layers in split mode), with / without the staggered
  // wave groups; other layers under these settings take the round-3 default
  const bool halo = (variant >= 16 && variant <= 18) && split && ksz == 3 && stride == 1;  // 18: requests inside the MFMA part
  if ((variant >= 16 && variant <= 18) && !halo) variant = split ? 9 : 8;
  D3D_REQUIRE(halo || cout % 128 == 0, "d3d_conv2d_nhwc_fwd: 64 output channels need the halo-patch kernel (variant %d)", variant);
  if (halo) {
    const C2hPlan hp = c2h_plan(batch, G.ho, G.wo, cout);
    Conv2dEpi eh = {};
    eh.fold_g = 16;
    D3D_REQUIRE(!bn || stats_ws, "d3d_conv2d_nhwc_fwd: the BatchNorm-backward sums need the statistics workspace");
    D3D_REQUIRE(!bn || (bn->x && bn->mean && bn->invstd), "d3d_conv2d_nhwc_fwd: bn needs x, mean, invstd");
    if (bn) {
      eh.bn_x = bn->x; eh.bn_mean = bn->mean; eh.bn_invstd = bn->invstd; eh.bn_gamma = bn->gamma; eh.bn_beta = bn->beta;
      eh.bn_relu = bn->relu;
      static const int stage = getenv("D3D_DENSE_BN_STAGE") ? atoi(getenv("D3D_DENSE_BN_STAGE")) : 1;
      eh.bn_stage = stage;
    }
    if (stats_ws) {
      D3D_REQUIRE(hp.fold_nb <= D3D_FOLD_MAX_BLOCKS, "d3d_conv2d_nhwc_fwd: %d blocks > %d for the statistics fold",
                  hp.fold_nb, D3D_FOLD_MAX_BLOCKS);
      const size_t need = d3d_conv2d_stats_ws_floats(batch, G.ho, G.wo, cout);
      if (need == 0 || stats_ws_floats < need)
        return d3d_set_error(D3D_ECAPACITY, "d3d_conv2d_nhwc_fwd: statistics workspace %zu floats < %zu", stats_ws_floats, need);
      const D3dFoldPlan fp = d3d_fold_plan(hp.fold_nb, 256);  // the group size of the largest fold serves every tile's
      eh.stats = stats_ws;
      eh.fold_g = fp.g;
      eh.ws_stride = (int)d3d_fold_ws_floats(hp.fold_nb, 256);
      eh.ctr_stride = d3d_fold_counter_words(fp);
      eh.counters = d3d_fold_counters(eh.ctr_stride * hp.q.ctiles);
      if (!eh.counters) return D3D_EHIP;
    }
    constexpr int HALO_XI = (18 * 18 + 15) / 16;
    const size_t hlds = 2 * (size_t)(2 * HALO_XI * 1024) + 3 * (size_t)(2 * 128 * 64);
#define C2_HALO(CIN, STG)                                                                                        \
  do {                                                                                                          \
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_halo_kernel<CIN, 16, 16, STG>,                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds));                   \
    hipLaunchKernelGGL((conv2d_halo_kernel<CIN, 16, 16, STG>), dim3(hp.blocks), dim3(512), hlds, stream,        \
                       x_hi, x_lo, w_hi, w_lo, (float *)y, G, eh, hp.q);                                        \
  } while (0)
#define C2_HALO_CASE(CIN)                        \
  if (cin == CIN) {                              \
    if (variant == 16) C2_HALO(CIN, 1);          \
    else if (variant == 18) C2_HALO(CIN, 2);     \
    else C2_HALO(CIN, 0);                        \
  }
    C2_HALO_CASE(64)
    C2_HALO_CASE(128)
    C2_HALO_CASE(256)
#undef C2_HALO_CASE
#undef C2_HALO
    D3D_LAUNCH_TRY("conv2d_halo_kernel");
    return D3D_OK;
  }
  if (cin > 256) variant = split ? 9 : 8;  // (384 / 512 input channels exist in the default tilings only)
  const bool w8 = variant == 8 || variant == 9;  // 128-pixel tiles on 8 waves
  const int wpx = ((variant & 2) || w8) ? 4 : 2;
  const int bm = w8 ? 128 : 64 * wpx;
  const int nb = (G.m + bm - 1) / bm;
  Conv2dEpi ep = {};
  ep.fold_g = 16;
  D3D_REQUIRE(!bn || stats_ws, "d3d_conv2d_nhwc_fwd: the BatchNorm-backward sums need the statistics workspace");
  D3D_REQUIRE(!bn || (bn->x && bn->mean && bn->invstd), "d3d_conv2d_nhwc_fwd: bn needs x, mean, invstd");
  if (bn) {
    ep.bn_x = bn->x; ep.bn_mean = bn->mean; ep.bn_invstd = bn->invstd; ep.bn_gamma = bn->gamma; ep.bn_beta = bn->beta;
    ep.bn_relu = bn->relu;
  }
  if (stats_ws) {
    D3D_REQUIRE(nb <= D3D_FOLD_MAX_BLOCKS, "d3d_conv2d_nhwc_fwd: %d pixel tiles > %d for the statistics fold", nb,
                D3D_FOLD_MAX_BLOCKS);
    const size_t need = d3d_conv2d_stats_ws_floats(batch, G.ho, G.wo, cout);
    if (stats_ws_floats < need)
      return d3d_set_error(D3D_ECAPACITY, "d3d_conv2d_nhwc_fwd: statistics workspace %zu floats < %zu", stats_ws_floats, need);
    const D3dFoldPlan fp = d3d_fold_plan(nb, 256);
    ep.stats = stats_ws;
    ep.fold_g = fp.g;
    ep.ws_stride = (int)d3d_fold_ws_floats(nb, 256);
    ep.ctr_stride = d3d_fold_counter_words(fp);
    ep.counters = d3d_fold_counters(ep.ctr_stride * (cout / 128));
    if (!ep.counters) return D3D_EHIP;
  }
  const bool hb = variant == 4;
  size_t lds = 2 * (size_t)((split ? 2 : 1) * (bm + 128) * (split ? 64 : 128)) / (hb ? 2 : 1);  // two stages of A + B planes
  if (stats_ws && lds < (size_t)(wpx * 256 + 256 + 4) * 4) lds = (size_t)(wpx * 256 + 256 + 4) * 4;  // epilogue scratch
#define C2_LAUNCH(CIN, SPLIT, WPX, ILV, HB, PXT)                                                                \
  do {                                                                                                          \
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_nhwc_kernel<CIN, SPLIT, SPLIT, WPX, ILV, HB, PXT>,     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
    hipLaunchKernelGGL((conv2d_nhwc_kernel<CIN, SPLIT, SPLIT, WPX, ILV, HB, PXT>), dim3(nb, cout / 128),        \
                       dim3(128 * WPX), lds, stream, x_hi, x_lo, w_hi, w_lo, y, G, ep, Conv2dTr{});             \
  } while (0)
#define C2_VARIANTS(CIN, SPLIT)                                         \
  do {                                                                  \
    if (variant == 1) C2_LAUNCH(CIN, SPLIT, 2, true, false, 2);         \
    else if (variant == 2) C2_LAUNCH(CIN, SPLIT, 4, false, false, 2);   \
    else if (variant == 3) C2_LAUNCH(CIN, SPLIT, 4, true, false, 2);    \
    else if (variant == 4) C2_LAUNCH(CIN, SPLIT, 2, false, true, 2);    \
    else if (variant == 8) C2_LAUNCH(CIN, SPLIT, 4, false, false, 1);   \
    else if (variant == 9) C2_LAUNCH(CIN, SPLIT, 4, true, false, 1);    \
    else C2_LAUNCH(CIN, SPLIT, 2, false, false, 2);                     \
  } while (0)
#define C2_CASE(CIN)                             \
  if (cin == CIN) {                              \
    if (split) C2_VARIANTS(CIN, true);           \
    else C2_VARIANTS(CIN, false);                \
  }
  C2_CASE(64)
  C2_CASE(128)
  C2_CASE(256)
  // 384 / 512 input channels: the concatenated neck outputs in front of the multi-group heads (1x1 convolutions; the
  // default tilings only)
#define C2_WIDE(CIN)                                         \
  if (cin == CIN) {                                          \
    if (split) C2_LAUNCH(CIN, true, 4, true, false, 1);      \
    else C2_LAUNCH(CIN, false, 4, false, false, 1);          \
  }
  C2_WIDE(384)
  C2_WIDE(512)
#undef C2_WIDE
#undef C2_CASE
#undef C2_VARIANTS
#undef C2_LAUNCH
  D3D_LAUNCH_TRY("conv2d_nhwc_kernel");
  return D3D_OK;
}

// data gradient of a stride-2 3x3 layer (ABI 8): g = dL/dy [batch, ho, wo, cout] -> dx [batch, h, w, cin] of the layer
// y = conv(x, w, stride 2, pad); w_hi / w_lo = the layer's data-gradient planes of d3d_conv2d_pack_weights
// ([mirrored tap][cin][cout]).  conv2d_nhwc_kernel in TR mode: one block = 128 pixels of one parity class of dx.
extern "C" int d3d_conv2d_nhwc_dgrad_s2(const void *g_hi, const void *g_lo, const void *w_hi, const void *w_lo, void *dx,
                                        int batch, int h, int w, int cin, int cout, int ksz, int pad, int mode,
                                        d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(mode == D3D_CONV_BF16 || mode == D3D_CONV_SPLIT, "d3d_conv2d_nhwc_dgrad_s2: mode %d", mode);
  const bool split = mode == D3D_CONV_SPLIT;
  D3D_REQUIRE(g_hi && w_hi && dx && (!split || (g_lo && w_lo)), "d3d_conv2d_nhwc_dgrad_s2: null pointer");
  D3D_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && ksz == 3 && pad >= 0 && pad < ksz, "d3d_conv2d_nhwc_dgrad_s2: map %d x %d x %d, kernel %d pad %d (3x3 layers)", batch, h, w, ksz, pad);
  D3D_REQUIRE((cout == 128 || cout == 256) && cin >= 128 && cin % 128 == 0,
              "d3d_conv2d_nhwc_dgrad_s2: channels %d -> %d (cout 128 | 256, cin a multiple of 128)", cin, cout);
  const int ho = (h + 2 * pad - ksz) / 2 + 1, wo = (w + 2 * pad - ksz) / 2 + 1;
  D3D_REQUIRE(ho >= 1 && wo >= 1, "d3d_conv2d_nhwc_dgrad_s2: empty output map");
  D3D_REQUIRE((int64_t)batch * ho * wo * cout * 2 < (int64_t)D3D_RSRC_BYTES && (int64_t)batch * h * w * cin * 4 < ((int64_t)1 << 33),
              "d3d_conv2d_nhwc_dgrad_s2: map too large");
  // the kernel's input map is g, its output map dx
  Conv2dGeom G = {};
  G.batch = batch; G.h = ho; G.w = wo; G.ho = h; G.wo = w;
  G.cout = cin; G.ksz = ksz; G.stride = 1; G.pad = pad;
  G.m = batch * h * w;
  G.x_bytes = (int)((int64_t)batch * ho * wo * cout * 2);
  G.w_bytes = (int)((int64_t)ksz * ksz * cin * cout * 2);
  Conv2dTr T = {};
  const int bm = 128;
  int tiles = 0;
  for (int c = 0; c < 4; ++c) {
    const int py = c >> 1, px = c & 1;
    T.hc[c] = (h - py + 1) / 2;
    T.wc[c] = (w - px + 1) / 2;
    if (T.hc[c] < 0) T.hc[c] = 0;
    if (T.wc[c] < 0) T.wc[c] = 0;
    tiles += (batch * T.hc[c] * T.wc[c] + bm - 1) / bm;
    T.tiles_end[c] = tiles;
    int n = 0;
    unsigned code = 0;
    for (int ky = 0; ky < ksz; ++ky) {
      if ((py + pad - ky) & 1) continue;
      const int dy = (py + pad - ky) / 2;  // even numerator: exact, also when negative
      for (int kx = 0; kx < ksz; ++kx) {
        if ((px + pad - kx) & 1) continue;
        const int dxx = (px + pad - kx) / 2;
        const int wslot = ksz * ksz - 1 - (ky * ksz + kx);  // the pack's mirrored order holds tap (ky, kx) there
        code |= (unsigned)((dy + 1) | ((dxx + 1) << 2) | (wslot << 4)) << (8 * n);
        ++n;
      }
    }
    D3D_REQUIRE(n >= 1 && n <= 4, "d3d_conv2d_nhwc_dgrad_s2: %d taps in class %d", n, c);
    T.ntap[c] = n;
    T.taps[c] = code;
  }
  Conv2dEpi ep = {};
  ep.fold_g = 16;
  const size_t lds = 2 * (size_t)((split ? 2 : 1) * (bm + 128) * (split ? 64 : 128));
#define C2_TR(CIN, SPLIT)                                                                                          \
  do {                                                                                                             \
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_nhwc_kernel<CIN, SPLIT, SPLIT, 4, true, false, 1, true>,  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
    hipLaunchKernelGGL((conv2d_nhwc_kernel<CIN, SPLIT, SPLIT, 4, true, false, 1, true>), dim3(tiles, cin / 128),   \
                       dim3(512), lds, stream, g_hi, g_lo, w_hi, w_lo, dx, G, ep, T);                              \
  } while (0)
  if (cout == 128) {
    if (split) C2_TR(128, true);
    else C2_TR(128, false);
  } else {
    if (split) C2_TR(256, true);
    else C2_TR(256, false);
  }
#undef C2_TR
  D3D_LAUNCH_TRY("conv2d_nhwc_kernel (strided data gradient)");
  return D3D_OK;
}

// weight gradient of the layer above (stride 1 or 2): g = dL/dy [batch, ho, wo, cout], x = the layer's input;
// dw [cout][cin][ksz][ksz] f32 (nn.Conv2d's parameter layout).  workspace: d3d_conv2d_wgrad_workspace_bytes.
static int c2_wgrad_splits(int m, int blocks_per_split) {
  // ~2 blocks per CU over (splits x taps x weight tiles); >= 8 K steps per block
  int s = (512 + blocks_per_split - 1) / blocks_per_split;
  const int max_s = (m + 511) / 512;
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}
// conv2d_wgrad3_kernel (3x3 stride 1, split mode): blocks over (splits x 3 kernel rows x weight tiles).  One block per CU
// (256) is the faster launch alone on the device, but the kernel runs on the second stream NEXT TO the data gradient of its
// layer, and neither its blocks (104 KB of LDS) nor conv2d_halo_kernel's (132 KB) share a CU with the other: with a block on
// every CU the data-gradient chain -- the critical path, 188 us per layer against 115 us here -- waits for whole
// weight-gradient blocks.  Round 6 (last session), 128 -> 128 layers of the KITTI RPN, same box, two runs each: 256 blocks
// 675.7 frames/s, 224: 673.9, 192: 674.0, 160: 679.5, 144: 674.7, 128: 683.8 / 684.5, 112: 681.2, 96: 675.9, 80: 659, 64:
// 626 -> 128 for one weight tile.  The 256 -> 256 layers of the CBGS necks measure the same either way (242.1 / 242.1,
// 113.7-114.4) and keep 256.  D3D_C2_WGRAD3_BLOCKS overrides.
static int c2_wgrad3_splits(int total_segs, int wtiles) {
  static const int env_blocks = getenv("D3D_C2_WGRAD3_BLOCKS") ? atoi(getenv("D3D_C2_WGRAD3_BLOCKS")) : 0;
  const int blocks = env_blocks > 0 ? env_blocks : (wtiles == 1 ? 128 : 256);
  int s = blocks / (3 * wtiles);
  if (s > total_segs) s = total_segs;
  return s < 1 ? 1 : s;
}
// conv2d_wgrad3n_kernel (64 x 64 weights, 3x3 stride 1, split mode): ~3 four-wave blocks per CU over (splits x 3 rows)
static int c2_wgrad3n_splits(int total_segs) {
  int s = 256;
  if (s > total_segs) s = total_segs;
  return s < 1 ? 1 : s;
}
extern "C" size_t d3d_conv2d_wgrad_workspace_bytes(int batch, int ho, int wo, int cin, int cout, int ksz) {
  const int64_t m = (int64_t)batch * ho * wo;
  if (m < 1) return 0;
  if (cin == 64 && cout == 64 && ksz == 3)
    return (size_t)c2_wgrad3n_splits(batch * ho * ((wo + 31) / 32)) * 9 * 64 * 64 * sizeof(float);
  if (cin < 128 || cout < 128) return 0;
  size_t s = (size_t)c2_wgrad_splits((int)m, ksz * ksz * (cin / 128) * (cout / 128));
  if (ksz == 3 && cin >= 128 && cout >= 128) {
    const size_t s3 = (size_t)c2_wgrad3_splits(batch * ho * ((wo + 31) / 32), (cin / 128) * (cout / 128));
    if (s3 > s) s = s3;
  }
  return s * ksz * ksz * cout * cin * sizeof(float);
}

extern "C" int d3d_conv2d_nhwc_wgrad(const void *g_hi, const void *g_lo, const void *x_hi, const void *x_lo,
                                     float *dw, int batch, int h, int w, int cin, int cout, int ksz, int stride,
                                     int pad, int mode, void *workspace, size_t workspace_bytes,
                                     d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(mode == D3D_CONV_BF16 || mode == D3D_CONV_SPLIT, "d3d_conv2d_nhwc_wgrad: mode %d", mode);
  const bool narrow = cin == 64 && cout == 64 && ksz == 3 && stride == 1 && mode == D3D_CONV_SPLIT;
  D3D_REQUIRE(narrow || (cin % 128 == 0 && cout % 128 == 0 && cin >= 128 && cout >= 128 && cin <= 512 && cout <= 512),
              "d3d_conv2d_nhwc_wgrad: channels %d -> %d (multiples of 128 up to 512; 64 -> 64 for split 3x3 stride-1 layers)",
              cin, cout);
  D3D_REQUIRE(ksz >= 1 && ksz <= 3 && stride >= 1 && stride <= 2 && pad >= 0 && pad < ksz,
              "d3d_conv2d_nhwc_wgrad: kernel %d stride %d pad %d", ksz, stride, pad);
  const bool split = mode == D3D_CONV_SPLIT;
  D3D_REQUIRE(g_hi && x_hi && dw && (!split || (g_lo && x_lo)), "d3d_conv2d_nhwc_wgrad: null pointer");
  Conv2dGeom G;
  int rc = c2_geom("d3d_conv2d_nhwc_wgrad", G, batch, h, w, cin, cout, ksz, stride, pad);
  if (rc) return rc;
  D3D_REQUIRE((int64_t)batch * h * w * cin * 2 < (int64_t)D3D_RSRC_BYTES, "d3d_conv2d_nhwc_wgrad: the input map exceeds 2 GB");
  G.w_bytes = (int)((int64_t)G.m * cout * 2);  // the descriptor of g
  if (narrow) {
    const int total_segs = batch * G.ho * ((G.wo + 31) / 32);
    int ns = c2_wgrad3n_splits(total_segs);
    const int sps = (total_segs + ns - 1) / ns;
    ns = (total_segs + sps - 1) / sps;
    const size_t needn = (size_t)c2_wgrad3n_splits(total_segs) * 9 * 64 * 64 * sizeof(float);
    if (!workspace || workspace_bytes < needn)
      return d3d_set_error(D3D_ECAPACITY, "d3d_conv2d_nhwc_wgrad: workspace %zu B < %zu B", workspace_bytes, needn);
    const size_t ldsn = 3 * (size_t)(2 * 32 * 128 + 2 * 40 * 128);
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_wgrad3n_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
    hipLaunchKernelGGL(conv2d_wgrad3n_kernel, dim3(ns * 3), dim3(256), ldsn, stream, g_hi, g_lo, x_hi, x_lo,
                       (float *)workspace, G, sps, ns);
    D3D_LAUNCH_TRY("conv2d_wgrad3n_kernel");
    const int64_t n4 = (int64_t)9 * 64 * 64 / 4;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, stream,
                       (const float4 *)workspace, ns, 9, 64, 64, dw);
    D3D_LAUNCH_TRY("conv2d_wgrad_reduce_kernel");
    return D3D_OK;
  }
  const int splits = c2_wgrad_splits(G.m, ksz * ksz * (cin / 128) * (cout / 128));
  const size_t need = d3d_conv2d_wgrad_workspace_bytes(batch, G.ho, G.wo, cin, cout, ksz);
  if (!workspace || workspace_bytes < need)
    return d3d_set_error(D3D_ECAPACITY, "d3d_conv2d_nhwc_wgrad: workspace %zu B < %zu B", workspace_bytes, need);
  if (split && ksz == 3 && stride == 1 && g_c2_wgrad_waves >= 16 && g_c2_wgrad_waves <= 18) {
    // second generation (d3d_conv2d_wgrad3.h): a block per (segment range, kernel row, weight tile), three taps each
    const int total_segs = batch * G.ho * ((G.wo + 31) / 32);
    const int wtiles = (cout / 128) * (cin / 128);
    int ns = c2_wgrad3_splits(total_segs, wtiles);
    const int sps = (total_segs + ns - 1) / ns;
    ns = (total_segs + sps - 1) / sps;
    const size_t lds3 = 3 * (size_t)(2 * 32 * 256 + 2 * 36 * 256);
    const dim3 grid3(ns * 3 * wtiles);
#define C2_WG3(STG)                                                                                              \
  do {                                                                                                           \
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_wgrad3_kernel<STG>,                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));                    \
    hipLaunchKernelGGL((conv2d_wgrad3_kernel<STG>), grid3, dim3(512), lds3, stream, g_hi, g_lo, x_hi, x_lo,      \
                       (float *)workspace, G, cin, sps, ns);                                                     \
  } while (0)
    if (g_c2_wgrad_waves == 16) C2_WG3(1); else if (g_c2_wgrad_waves == 18) C2_WG3(2); else C2_WG3(0);
#undef C2_WG3
    D3D_LAUNCH_TRY("conv2d_wgrad3_kernel");
    const int64_t n4 = (int64_t)9 * cout * cin / 4;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, stream,
                       (const float4 *)workspace, ns, 9, cout, cin, dw);
    D3D_LAUNCH_TRY("conv2d_wgrad_reduce_kernel");
    return D3D_OK;
  }
  const int kp = split ? 32 : 64;
  int pps = (G.m + splits - 1) / splits;
  pps = (pps + kp - 1) / kp * kp;
  const int nsplit = (G.m + pps - 1) / pps;
  const size_t lds = 2 * (size_t)(2 * (split ? 2 : 1) * kp * 256);
  const dim3 grid(nsplit * ksz * ksz * (cout / 128) * (cin / 128));
#define C2_WG(SPLIT, WCI)                                                                                     \
  do {                                                                                                        \
    D3D_HIP_TRY(hipFuncSetAttribute((const void *)conv2d_wgrad_kernel<SPLIT, WCI>,                            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
    hipLaunchKernelGGL((conv2d_wgrad_kernel<SPLIT, WCI>), grid, dim3(128 * WCI), lds, stream, g_hi, g_lo, x_hi, \
                       x_lo, (float *)workspace, G, cin, pps, nsplit);                                        \
  } while (0)
  if (g_c2_wgrad_waves == 4) {  // (16 / 17 on a layer conv2d_wgrad3_kernel does not serve: as 8)
    if (split) C2_WG(true, 2); else C2_WG(false, 2);
  } else {
    if (split) C2_WG(true, 4); else C2_WG(false, 4);
  }
#undef C2_WG
  D3D_LAUNCH_TRY("conv2d_wgrad_kernel");
  const int64_t n4 = (int64_t)ksz * ksz * cout * cin / 4;
  hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, stream,
                     (const float4 *)workspace, nsplit, ksz * ksz, cout, cin, dw);
  D3D_LAUNCH_TRY("conv2d_wgrad_reduce_kernel");
  return D3D_OK;
}

extern "C" int d3d_conv2d_pack_weights(const float *weight, int cout, int cin, int ksz, void *fwd_hi, void *fwd_lo,
                                       void *dgrad_hi, void *dgrad_lo, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(weight && cout >= 1 && cin >= 1 && ksz >= 1 && ksz <= 7, "d3d_conv2d_pack_weights: arguments");
  D3D_REQUIRE(fwd_hi || dgrad_hi, "d3d_conv2d_pack_weights: nothing to write");
  const int total = cout * cin * ksz * ksz;
  int nb = (total + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(conv2d_pack_kernel, dim3(nb), dim3(256), 0, stream, weight, cout, cin, ksz * ksz,
                     (uint16_t *)fwd_hi, (uint16_t *)fwd_lo, (uint16_t *)dgrad_hi, (uint16_t *)dgrad_lo);
  D3D_LAUNCH_TRY("conv2d_pack_kernel");
  return D3D_OK;
}

extern "C" int d3d_split_planes(const float *x, int64_t n, void *hi, void *lo, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(x && hi && lo && n >= 0 && n % 4 == 0, "d3d_split_planes: arguments (n %% 4 == 0)");
  if (n == 0) return D3D_OK;
  int64_t nb = (n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(conv2d_split_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float4 *)x, n / 4,
                     (uint2 *)hi, (uint2 *)lo);
  D3D_LAUNCH_TRY("conv2d_split_kernel");
  return D3D_OK;
}
