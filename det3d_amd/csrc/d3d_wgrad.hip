// Sparse-conv weight gradient for gfx950, site-tile stationary (round 2):
//
//   dW[k][ci][co] = sum_o  feat[nbr[k][o]][ci] * dout[o][co]
//
// Round 1 ran one block per (kernel offset, site split): every dout tile was fetched once per offset
// (27x) and 1024 blocks x 4096 float atomics were folded into one 216 KB dW (99 us for 64->64, 3-6x the
// compulsory HBM traffic, profiles/round1_pmc_fetch_size.txt).  Here a 16-wave block owns a contiguous
// range of output sites and a GROUP of KG kernel offsets whose [Cin x Cout] accumulators all live in
// registers (64->64: 9 offsets x 16 MFMA tiles = one tile per wave and offset, 36 VGPRs):
//   * a dout tile of STEP sites is staged in LDS ONCE and reused for the KG offsets (27 / KG reads of dout
//     per launch instead of 27; KG = 27 for Cin*Cout <= 2048, i.e. dout is read exactly once);
//   * the gathered feature rows of one (tile, offset) stage are 16 KB = one 16-byte vector per thread;
//     they are prefetched DEPTH stages ahead in registers (DEPTH x 16 KB in flight per CU) with their
//     neighbour indices prefetched a further DEPTH stages ahead, so the two dependent L2 round trips
//     (index, then row) of the gather overlap ~2*DEPTH stages of LDS + MFMA work;
//   * sites become the MFMA k dimension through ds_read_b64_tr_b16 on [16 sites][16 ch] LDS tiles (32 sites per
//     mfma_f32_16x16x32_bf16); when Cin*Cout has fewer than 16 MFMA tiles the waves sharing a tile split the sites
//     of each stage and keep separate partial sums, so all 16 waves work in every stage;
//   * f32 (parity mode) runs the same pipeline on mfma_f32_16x16x4f32 with plain [site][ch] LDS rows;
//   * no atomics: every block stores its partial dW to a workspace slice and a second, streaming kernel
//     sums the splits in a fixed order => the weight gradient is bit-reproducible run to run.
// Traffic per launch: P*Cin*e gathered (L2) + (27/KG)*N*Cout*e + 27*N*4 (table) + 2 * splits*|dW|*4.
#include <stdlib.h>

#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_spconv_internal.h"

typedef __attribute__((ext_vector_type(4))) short wg_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short wg_s16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4_t;

typedef __attribute__((ext_vector_type(2))) float wg_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 wg_bf16x2_t;
// f32 pair -> hi = bf16(v), lo = bf16(v - hi) (round to nearest even; v - hi is exact)
__device__ __forceinline__ void wg_split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  const wg_f32x2_t v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wg_bf16x2_t));
  const wg_f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, wg_bf16x2_t));
}

#define WG_WAVES 16
#define WG_THREADS (WG_WAVES * 64)

// compile-time geometry shared by the kernel and the host-side planner
template <int CIN, int COUT, bool BF16, int WAVES = WG_WAVES> struct WgGeom {
  static constexpr int EPV = BF16 ? 8 : 4;                       // elements per 16-byte vector
  static constexpr int CIT = CIN / 16, COT = COUT / 16, TILES = CIT * COT;
  static constexpr int FSEG = CIN / EPV, DSEG = COUT / EPV;      // 16-byte segments per row
  static constexpr int STEP = WAVES * 64 / FSEG;                 // sites per stage (1 vector / thread)
  static constexpr int KSTEPS = STEP / 32;                       // MFMA k-steps (32 sites each) per stage
  static constexpr int TPW = TILES >= WAVES ? TILES / WAVES : 1;     // MFMA tiles per wave
  static constexpr int CSPLIT_ = TILES >= WAVES ? 1 : WAVES / TILES;
  static constexpr int CSPLIT = CSPLIT_ < KSTEPS ? CSPLIT_ : KSTEPS;       // waves sharing one tile: they
  //   split the sites of a stage and each keeps its own partial sum -> CSPLIT partial slabs per block (f32 rows of
  //   16 channels give 8 k-steps per stage: 8 of the 16 waves multiply, all 16 load)
};
static int wg_csplit(int cin, int cout, int dtype, int waves) {
  const int tiles = (cin / 16) * (cout / 16);
  const int step = waves * 64 / (cin / (dtype == D3D_BF16 ? 8 : 4));
  int cs = tiles >= waves ? 1 : waves / tiles;
  if (cs > step / 32) cs = step / 32;
  return cs < 1 ? 1 : cs;
}

// One block = 16 waves, one group of KG kernel offsets, one contiguous range of output sites.
//   bf16: LDS tiles [16 sites][16 ch], ds_read_b64_tr_b16 turns sites into the MFMA k dimension, two of them
//         feed one mfma_f32_16x16x32_bf16 (32 sites per instruction)
//   f32 : SPLIT arithmetic (round 3; round 2 ran the exact mfma_f32_16x16x4f32 here at 216 us for 64x64 against
//         64 us in bf16): the f32 rows are split where they are written to LDS -- hi = bf16(v), lo = bf16(v - hi)
//         into two planes of the same [16 sites][16 ch] tiles -- and every product is three bf16 MFMAs
//         (hi*hi + hi*lo + lo*hi), f32 accumulation: relative error ~2^-16 of the products
// BREG (round 4): the dout fragments of a tile do not depend on the kernel offset -- a wave reads ITS fragments (one cout
// tile, its k-steps, hi / lo) once per tile into registers instead of once per stage: half of the transposing LDS reads
// of a stage gone (64 x 64 f32: 16 -> 8 per wave), for QN * NPL * 4 registers
// WAVES: 16 (one block per CU) or 8 (two per CU, half the sites per stage, twice the MFMA tiles per wave)
template <int CIN, int COUT, int KG, int DEPTH, bool BF16, bool BREG = false, int WAVES = WG_WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 16 ? 1 : 4)
    spconv_wgrad_tile_kernel(const void *__restrict__ feat_, const void *__restrict__ dout_,
                             const int32_t *__restrict__ nbr, int cap, int n_out, int rows_per_split,
                             int kvol, float *__restrict__ part, int xcd_remap) {
  using G = WgGeom<CIN, COUT, BF16, WAVES>;
  constexpr int NTH = WAVES * 64;
  constexpr int CIT = G::CIT, COT = G::COT, TILES = G::TILES, TPW = G::TPW, CSPLIT = G::CSPLIT;
  constexpr int FSEG = G::FSEG, DSEG = G::DSEG, STEP = G::STEP, EPV = G::EPV;
  constexpr int DN = STEP * DSEG;
  constexpr int DV = (DN + NTH - 1) / NTH;  // dout vectors per thread and tile
  // one plane: [chunk of 16 sites][ch tile][16 sites][16 ch] bf16; f32 inputs fill two planes (hi, lo)
  constexpr int NPL = BF16 ? 1 : 2;
  constexpr int FBUF = STEP * CIN, DBUF = STEP * COUT;  // elements per plane
  constexpr int ESZ = BF16 ? 2 : 4;                      // bytes per element in HBM
  constexpr int KSTEPS = G::KSTEPS;                      // 32 sites per MFMA k-step, split over the CSPLIT waves
  static_assert(KSTEPS % CSPLIT == 0 && KSTEPS >= 1, "sites of a stage must split evenly over the waves sharing a tile");
  __shared__ __attribute__((aligned(16))) unsigned char lds_f[2 * NPL * FBUF * 2];
  __shared__ __attribute__((aligned(16))) unsigned char lds_d[2 * NPL * DBUF * 2];

  // XCD-aware order (see spconv_fwd_mfma_kernel): workgroups are dealt round-robin to the 8 XCDs in linear
  // (x fastest) order; give XCD x the x-th contiguous eighth of the (site split, offset group) pairs so that
  // its L2 sees one slab of sites.  A bijection on the grid when the block count is a multiple of 8.
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_remap) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
    const int logical = (lin & 7) * (total >> 3) + (lin >> 3);
    bx = logical % gridDim.x;
    by = logical / gridDim.x;
  }
  const int k0 = bx * KG;
  const int s0 = by * rows_per_split;
  int s1 = s0 + rows_per_split;
  if (s1 > n_out) s1 = n_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int my_tile = wave % TILES;  // first (only, if TILES <= 16) MFMA tile of this wave
  const int my_part = wave / TILES;  // which share of a stage's sites (0 when CSPLIT == 1)

  wg_f32x4_t acc[KG][TPW];
#pragma unroll
  for (int a = 0; a < KG; ++a)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[a][t] = wg_f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (s1 - s0 + STEP - 1) / STEP;
  const int frow = threadIdx.x / FSEG, fseg = threadIdx.x - frow * FSEG;

  // Every load below is UNCONDITIONAL (clamped address, result masked afterwards) and the KG stages of a tile are
  // fully unrolled: the loop body is straight-line code with a fixed number of loads per stage, so hipcc's
  // s_waitcnt bookkeeping can emit exact counted vmcnt(N) waits and DEPTH stages really stay in flight (a
  // conditional load anywhere gives the compiler a path with fewer loads and it falls back to vmcnt(0)).  The
  // barrier is the LDS-only one: __syncthreads() would drain the whole pipeline every stage.
  // Masking happens on the ADDRESS side (buffer loads return 0 past the descriptor's range), never on the
  // loaded data, and the neighbour index is masked where it is consumed, DEPTH stages after its load.
  const int last_tile = ntiles - 1;
  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_), dout_rs = d3d_make_rsrc(dout_);
  auto index_site = [&](int tile) -> int {
    tile = tile < last_tile ? tile : last_tile;  // uniform clamp: tail stages re-load valid rows
    return s0 + tile * STEP + frow;
  };
  auto load_index = [&](int tile, int kk) -> int {  // raw value; garbage-free: the address is clamped
    const int site = index_site(tile);
    const int sc = site < s1 ? site : s1 - 1;
    const int32_t *nk = nbr + (int64_t)(k0 + kk) * cap;  // uniform base, 32-bit lane offset
    return nk[(uint32_t)sc];
  };
  auto load_feat = [&](int i, int tile) -> uint4 {  // i: raw index of a stage of `tile`
    const bool ok = i >= 0 && index_site(tile) < s1;
    const uint32_t off = ok ? (uint32_t)i * (uint32_t)(CIN * ESZ) + (uint32_t)(fseg * 16) : D3D_OOB_OFFSET;
    return d3d_buffer_load16(feat_rs, off);
  };
  uint4 dr[DV];
  auto load_dout = [&](int tile) {
    tile = tile < last_tile ? tile : last_tile;
    const int base = s0 + tile * STEP;
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int e = threadIdx.x + j * NTH;
      const int ec = (DN % NTH != 0 && e >= DN) ? DN - 1 : e;
      const int r = ec / DSEG, seg = ec - r * DSEG;
      const int site = base + r;
      const uint32_t off = site < s1 ? (uint32_t)site * (uint32_t)(COUT * ESZ) + (uint32_t)(seg * 16) : D3D_OOB_OFFSET;
      dr[j] = d3d_buffer_load16(dout_rs, off);
    }
  };
  // LDS element offset (within a plane) of 16-byte HBM segment `seg` of row r: 8 bf16 or 4 f32 channels
  auto lds_off = [&](int r, int seg, int ctiles) -> int {
    const int c = seg * EPV;
    return ((r >> 4) * ctiles + (c >> 4)) * 256 + (r & 15) * 16 + (c & 15);
  };
  // registers -> LDS; f32 rows are split into the hi / lo planes on the way
  auto put = [&](unsigned char *buf, int plane_elems, int el, const uint4 &v) {
    if constexpr (BF16) {
      *(uint4 *)(buf + el * 2) = v;
    } else {
      uint32_t h0, l0, h1, l1;
      wg_split2(__uint_as_float(v.x), __uint_as_float(v.y), h0, l0);
      wg_split2(__uint_as_float(v.z), __uint_as_float(v.w), h1, l1);
      *(uint2 *)(buf + el * 2) = make_uint2(h0, h1);
      *(uint2 *)(buf + (plane_elems + el) * 2) = make_uint2(l0, l1);
    }
  };

  // pipeline registers: slot d holds the rows of the next stage with (stage % DEPTH) == d and the neighbour
  // indices of the stage DEPTH after that one.  DEPTH divides KG, so inside the unrolled tile body the slot of
  // offset kk is the compile-time constant kk % DEPTH.
  static_assert(KG % DEPTH == 0, "prefetch depth must divide the offset group");
  constexpr int QN = KSTEPS / CSPLIT;  // k-steps of a stage per wave
  static_assert(!BREG || (16 % COT == 0 || COT % 16 == 0), "BREG: the tiles of a wave share one cout tile");
  wg_bf16x8_t breg[BREG ? QN : 1][NPL];
  uint4 fr[DEPTH];
  int ix[DEPTH];
  if (ntiles > 0) {
    load_dout(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ix[d] = load_index(d / KG, d % KG);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      fr[d] = load_feat(ix[d], d / KG);
      ix[d] = load_index((d + DEPTH) / KG, (d + DEPTH) % KG);
    }
  }

  for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
      constexpr int dummy = 0;
      (void)dummy;
      const int d = kk % DEPTH;                 // compile-time after unrolling
      const int st = tile * KG + kk;
      unsigned char *bf = lds_f + (st & 1) * (NPL * FBUF * 2);
      unsigned char *bd = lds_d + (tile & 1) * (NPL * DBUF * 2);
      // stage `st` -> LDS.  The buffer's previous readers (stage st-2) finished their LDS reads before they
      // arrived at the barrier of stage st-1, so one barrier per stage suffices
      put(bf, FBUF, lds_off(frow, fseg, CIT), fr[d]);
      if (kk == 0) {
#pragma unroll
        for (int j = 0; j < DV; ++j) {
          const int e = threadIdx.x + j * NTH;
          if (DN % NTH != 0 && e >= DN) break;
          const int r = e / DSEG, seg = e - r * DSEG;
          put(bd, DBUF, lds_off(r, seg, COT), dr[j]);
        }
      }
      d3d_lds_barrier();
      // refill: rows of stage st+DEPTH (their indices arrived DEPTH stages ago), indices of stage st+2*DEPTH,
      // and -- once per tile -- the next tile's dout rows
      fr[d] = load_feat(ix[d], tile + (kk + DEPTH) / KG);
      ix[d] = load_index(tile + (kk + 2 * DEPTH) / KG, (kk + 2 * DEPTH) % KG);
      if (kk == 0) load_dout(tile + 1);
      // 32 sites = LDS chunks 2*ks and 2*ks+1.  Each lane passes the address of ITS 8 bytes in the
      // row-major 4x16 block of sites [lg*4, lg*4+4); the hardware returns the transposed column lr.
      // A and B use the same site order, so any consistent k labelling is a valid sum.
      auto operand = [&](const unsigned char *plane, int ctiles, int ct, int ks) -> wg_bf16x8_t {
        const uint16_t *p16 = (const uint16_t *)plane;
        wg_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) wg_s16x4_t *)(p16 + ((2 * ks) * ctiles + ct) * 256 + lg * 64 + lr * 4));
        wg_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) wg_s16x4_t *)(p16 + ((2 * ks + 1) * ctiles + ct) * 256 + lg * 64 + lr * 4));
        return __builtin_bit_cast(wg_bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      if constexpr (BREG) {
        if (kk == 0 && my_part < CSPLIT) {  // this tile's dout fragments: once for the KG stages
          const int cot0 = my_tile % COT;
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            breg[q][0] = operand(bd, COT, cot0, q * CSPLIT + my_part);
            if constexpr (!BF16) breg[q][1] = operand(bd + DBUF * 2, COT, cot0, q * CSPLIT + my_part);
          }
        }
        if (my_part < CSPLIT) {
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int ks = q * CSPLIT + my_part;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
              const int tl = my_tile + t * WAVES;
              const int cit = tl / COT;
              const wg_bf16x8_t ah = operand(bf, CIT, cit, ks);
              acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, breg[q][0], acc[kk][t], 0, 0, 0);
              if constexpr (!BF16) {
                const wg_bf16x8_t al = operand(bf + FBUF * 2, CIT, cit, ks);
                acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, breg[q][1], acc[kk][t], 0, 0, 0);
                acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, breg[q][0], acc[kk][t], 0, 0, 0);
              }
            }
          }
        }
      } else {
#pragma unroll 4
        for (int q = 0; q < (my_part < CSPLIT ? KSTEPS / CSPLIT : 0); ++q) {
          const int ks = q * CSPLIT + my_part;  // this wave's k-steps of the stage
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int tl = my_tile + t * WAVES;
            const int cit = tl / COT, cot = tl - cit * COT;
            const wg_bf16x8_t ah = operand(bf, CIT, cit, ks), bh = operand(bd, COT, cot, ks);
            acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[kk][t], 0, 0, 0);
            if constexpr (!BF16) {
              const wg_bf16x8_t al = operand(bf + FBUF * 2, CIT, cit, ks), bl = operand(bd + DBUF * 2, COT, cot, ks);
              acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[kk][t], 0, 0, 0);
              acc[kk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[kk][t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // partial dW of this (offset group, site split, wave share): plain stores, summed by wgrad_reduce_kernel
  if (my_part >= CSPLIT) return;  // (f32, 16 x 16: more waves than k-steps -- these only loaded)
  float *pbase = part + ((int64_t)by * CSPLIT + my_part) * kvol * (CIN * COUT);
#pragma unroll
  for (int a = 0; a < KG; ++a) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tl = my_tile + t * WAVES;
      const int cit = tl / COT, cot = tl - cit * COT;
      float *o = pbase + ((int64_t)(k0 + a) * CIN + cit * 16 + lg * 4) * COUT + cot * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * COUT] = acc[a][t][r];
    }
  }
}

// dw[e] = sum over slabs of part[s][e] in a fixed order.  Block = 8 float4 elements x 32 slab groups: thread (element, g)
// adds slabs g, g + 32, ... in that order (their loads are independent: the unrolled loop keeps four in flight), the 32
// groups are folded through LDS in group order -- bit-reproducible.  (Rounds 2-6 ran 32 elements x 8 groups with up to 17
// slabs per thread: 864 blocks for the 64 x 64 layers, 25-28 us for 60 MB; this shape has 3456 blocks of short chains.)
__global__ void __launch_bounds__(256)
    wgrad_reduce_kernel(const float4 *__restrict__ part, int slabs, int64_t total4, float4 *__restrict__ dw) {
  __shared__ float4 red[32][8];
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int64_t e = (int64_t)blockIdx.x * 8 + el;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total4) {
#pragma unroll 4
    for (int s = g; s < slabs; s += 32) {
      const float4 u = part[(int64_t)s * total4 + e];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  }
  red[g][el] = a;
  __syncthreads();
  if (g == 0 && e < total4) {
    float4 r = red[0][el];
#pragma unroll
    for (int j = 1; j < 32; ++j) {
      r.x += red[j][el].x; r.y += red[j][el].y; r.z += red[j][el].z; r.w += red[j][el].w;
    }
    dw[e] = r;
  }
}

struct WgPlan {
  int kg, step, csplit, splits, rows_per_split, waves;
};

// kernel variant: 0 = the shape's default (BREG except bf16 64 x 128; f32 64 x 64 at prefetch depth 3), 1 = dout fragments
// re-read from LDS every stage (round 3), 2 = dout fragments in registers (BREG) on 16 waves at the round-3 prefetch depth, 3 = as 0.
// 0 also picks 8-wave blocks (two independent blocks per CU: one writes its stage while the other multiplies) for the f32
// shapes that have them
static int wg_variant() {
  static const int v = getenv("D3D_WGRAD_VARIANT") ? atoi(getenv("D3D_WGRAD_VARIANT")) : 0;
  return g_d3d_wgrad_blocks >= 100000 ? g_d3d_wgrad_blocks / 100000 : v;
}

// waves per block
static int wg_waves(int kvol, int cin, int cout, int dtype) {
  const bool has8 = dtype == D3D_F32 && kvol % 9 == 0 &&
                    ((cin == 64 && (cout == 64 || cout == 128)) || (cin == 32 && (cout == 32 || cout == 64)));
  // default where instantiated (f32, tools/tune_fwd_wide.py: 64 x 64 262 -> 251 us, 32 x 64 127 -> 117, 32 x 32 95 -> 89,
  // 64 -> 128 448 -> 416); variants 1 / 2 keep the 16-wave kernels
  const int var = wg_variant();
  return (has8 && (var == 0 || var == 3)) ? 8 : WG_WAVES;
}

// Blocks to aim for.  One per CU by default; the f32 layers that take three offsets per block (nine offset groups: 64->128
// and 128x128) get two rounds -- 24 splits x 9 = 216 blocks left 40 CUs idle, 56 x 9 = 504 fills two rounds (measured on the
// CBGS rulebooks: 1049 -> 850 us and 514 -> 472 us per launch; the nine-offset layers are flat or slower with more splits)
// (tuning knob 1 carries two numbers: value % 100000 = blocks, value / 100000 = kernel variant, see wg_variant)
static int wg_target_blocks(int kg, int dtype, int n_out) {
  static const int v = getenv("D3D_WGRAD_BLOCKS") ? atoi(getenv("D3D_WGRAD_BLOCKS")) : 0;
  if (g_d3d_wgrad_blocks > 0 && g_d3d_wgrad_blocks % 100000 > 0) return g_d3d_wgrad_blocks % 100000;
  if (v >= 1) return v;
  // 208, not one block per CU (round 6): the weight gradient runs on a second stream NEXT TO the data-gradient chain, and
  // with ~50 CUs left to that chain the step is 1.3-1.9 % faster than with 256 (same box, 540 timed steps each: 192 blocks
  // 646.4 / 645.5 / 648.2 frames/s, 224: 645.1, 256: 634.0 / 638.5 / 642.3, 160: 632.9, 128: 641.0, 384: 630.1); alone on
  // the device 256 is the faster launch
  // (on the 800 k-row layers of nuScenes CBGS the two counts measure the same: 237.5 vs 238.5 frames/s -- 256 stays there)
  return (dtype == D3D_F32 && kg == 3) ? 512 : (n_out < 300000 ? 208 : 256);
}

// offsets per block: the accumulators of a wave are 4 * KG * TPW registers
static int wg_kgroup(int kvol, int cin, int cout, int dtype) {
  const int tiles = (cin / 16) * (cout / 16);
  // (f32 = split arithmetic: two more operand sets live per MFMA group; 9 offsets x 2 tiles per wave spill)
  const int kg = (tiles <= 16 || (tiles <= 32 && dtype == D3D_BF16)) ? 9 : 3;
  if (kvol % kg == 0 && kvol >= kg) return kg;
  return kvol % 3 == 0 ? 3 : 1;
}

static bool wg_supported(int cin, int cout) {
  return (cin == 16 && (cout == 16 || cout == 32)) || (cin == 32 && (cout == 32 || cout == 64)) ||
         (cin == 64 && (cout == 64 || cout == 128)) || (cin == 128 && cout == 128);
}

static WgPlan wg_plan(int n_out, int kvol, int cin, int cout, int dtype) {
  WgPlan p;
  const int epv = dtype == D3D_BF16 ? 8 : 4;
  const int tiles_mfma = (cin / 16) * (cout / 16);
  p.kg = wg_kgroup(kvol, cin, cout, dtype);
  p.waves = wg_waves(kvol, cin, cout, dtype);
  p.step = p.waves * 64 / (cin / epv);
  p.csplit = wg_csplit(cin, cout, dtype, p.waves);
  const int kgroups = kvol / p.kg;
  const int tiles = (n_out + p.step - 1) / p.step;
  int splits = (wg_target_blocks(p.kg, dtype, n_out) * (WG_WAVES / p.waves) + kgroups - 1) / kgroups;
  int max_splits = (tiles + 1) / 2;  // >= 2 tiles per block
  if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  if (splits >= 16) splits = splits / 8 * 8;  // XCD-aware order needs (offset groups x splits) % 8 == 0
  int tiles_per_split = (tiles + splits - 1) / splits;
  p.rows_per_split = tiles_per_split * p.step;
  p.splits = (n_out + p.rows_per_split - 1) / p.rows_per_split;
  if (p.splits < 1) p.splits = 1;
  if (splits >= 16 && p.splits < splits) p.splits = splits;  // keep the multiple of 8: trailing splits are empty
  return p;
}

extern "C" size_t d3d_sparse_conv_wgrad_workspace_bytes(int n_out, int kvol, int cin, int cout, int dtype) {
  if ((dtype != D3D_BF16 && dtype != D3D_F32) || !wg_supported(cin, cout) || n_out <= 0) return 256;
  WgPlan p = wg_plan(n_out, kvol, cin, cout, dtype);
  return (size_t)p.splits * p.csplit * kvol * cin * cout * sizeof(float) + 256;
}

template <int CI, int CO, int KG, int DEPTH, bool BF16, bool BREG = false, int WAVES = WG_WAVES>
static int wg_launch(const void *feat, const void *dout, const int32_t *nbr, int cap, int n_out, int kvol,
                     const WgPlan &p, float *part, hipStream_t stream) {
  const int remap = g_xcd_remap && ((kvol / KG) * p.splits) % 8 == 0 && p.splits >= 8;
  hipLaunchKernelGGL((spconv_wgrad_tile_kernel<CI, CO, KG, DEPTH, BF16, BREG, WAVES>), dim3(kvol / KG, p.splits),
                     dim3(WAVES * 64), 0, stream, feat, dout, nbr, cap, n_out, p.rows_per_split, kvol, part, remap);
  D3D_LAUNCH_TRY("spconv_wgrad_tile_kernel");
  return D3D_OK;
}

#define WG_ARGS feat, dout, nbr, cap, n_out, kvol, p, part, stream
// (channels) -> the offset group wg_kgroup() picks for a 27-offset kernel with its prefetch depth, plus the
// 3-offset ((3,1,1) kernels) and 1-offset fallbacks
#define WG_CASE(CI, CO, KGMAIN, DMAIN, DF32, DBREG, B16)                                                     \
  if (cin == CI && cout == CO) {                                                                       \
    const int var = wg_variant();                                                                      \
    if (bf16) {                                                                                        \
      if (p.kg == KGMAIN && (var == 2 || (var == 0 && B16)))                                           \
        return wg_launch<CI, CO, KGMAIN, DMAIN, true, true>(WG_ARGS);                                  \
      if (p.kg == KGMAIN) return wg_launch<CI, CO, KGMAIN, DMAIN, true>(WG_ARGS);                      \
      if (p.kg == 3) return wg_launch<CI, CO, 3, 3, true>(WG_ARGS);                                    \
      if (p.kg == 1) return wg_launch<CI, CO, 1, 1, true>(WG_ARGS);                                    \
    } else {                                                                                           \
      if (p.kg == KGMAIN && var == 1) return wg_launch<CI, CO, KGMAIN, DF32, false>(WG_ARGS);          \
      if (p.kg == KGMAIN && var == 2) return wg_launch<CI, CO, KGMAIN, DF32, false, true>(WG_ARGS);    \
      if (p.kg == KGMAIN) return wg_launch<CI, CO, KGMAIN, DBREG, false, true>(WG_ARGS);               \
      if (p.kg == 3 && var == 1) return wg_launch<CI, CO, 3, 3, false>(WG_ARGS);                       \
      if (p.kg == 3) return wg_launch<CI, CO, 3, 3, false, true>(WG_ARGS);                             \
      if (p.kg == 1) return wg_launch<CI, CO, 1, 1, false>(WG_ARGS);                                   \
    }                                                                                                  \
  }

static int wg_dispatch(const void *feat, const void *dout, const int32_t *nbr, int cap, int n_out, int kvol,
                       int cin, int cout, bool bf16, const WgPlan &p, float *part, hipStream_t stream) {
  // prefetch depth: DEPTH stages x 16 KB of gathered rows in flight per CU, bounded by the 128 VGPRs a
  // 16-wave block leaves each lane (accumulators: 4 * KG * TPW)
  // f32: the dout fragments of a tile live in registers (BREG; tools/tune_fwd_wide.py on a 264k-site rulebook, bit-identical
  // sums: 64 x 64 332 -> 263 us at depth 3, 32 x 64 156 -> 124, 32 x 32 105 -> 95, 128 x 128 810 -> 770)
  // bf16: 64 x 64 158 -> 135, 128 x 128 452 -> 427, 32 x 64 83 -> 71; not 64 x 128 (nine offsets x two tiles: the extra
  // registers spill, 229 -> 340 us)
#define WG_CASE8(CI, CO, KG8)                                                                           \
  if (cin == CI && cout == CO && !bf16 && p.waves == 8 && p.kg == KG8)                                  \
    return wg_launch<CI, CO, KG8, 3, false, true, 8>(WG_ARGS);
  WG_CASE8(32, 32, 9)
  WG_CASE8(32, 64, 9)
  WG_CASE8(64, 64, 9)
  WG_CASE8(64, 128, 3)
  if (p.waves != WG_WAVES) return d3d_set_error(D3D_EINVAL, "wgrad tile kernel: no 8-wave kernel for %d -> %d (kg %d)", cin, cout, p.kg);
  WG_CASE(16, 16, 9, 3, 3, 3, true)
  WG_CASE(16, 32, 9, 3, 3, 3, true)
  WG_CASE(32, 32, 9, 3, 3, 3, true)
  WG_CASE(32, 64, 9, 3, 3, 3, true)
  WG_CASE(64, 64, 9, 9, 9, 3, true)
  WG_CASE(64, 128, 9, 3, 3, 3, false)
  WG_CASE(128, 128, 3, 3, 3, 3, true)
  return d3d_set_error(D3D_EINVAL, "wgrad tile kernel: unsupported channels %d -> %d (kg %d)", cin, cout, p.kg);
}

extern "C" int d3d_sparse_conv_wgrad(const void *feat, const void *dout, const int32_t *nbr, int cap,
                                     int n_out, int kvol, int cin, int cout, int dtype, float *dw,
                                     void *workspace, size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_sparse_conv_wgrad: dtype %d", dtype);
  D3D_REQUIRE(n_out >= 0 && n_out <= cap && kvol >= 1 && cin >= 1 && cout >= 1,
              "d3d_sparse_conv_wgrad: bad sizes");
  D3D_REQUIRE(cin * cout <= 16384, "d3d_sparse_conv_wgrad: cin*cout=%d > 16384", cin * cout);
  D3D_REQUIRE(dw, "d3d_sparse_conv_wgrad: null dw");
  // D3D_WGRAD_LEGACY=1: the round-1 kernels (one block per offset, float atomics) for A/B timing
  static const int legacy = getenv("D3D_WGRAD_LEGACY") ? atoi(getenv("D3D_WGRAD_LEGACY")) : 0;
  const bool exact = g_sp_exact_f32 && dtype == D3D_F32;  // d3d_set_tuning knob 10 (test knob): VALU f32 kernel
  const bool tile_path = !legacy && !exact && wg_supported(cin, cout) && n_out > 0 &&
                         (int64_t)n_out * cout * (dtype == D3D_BF16 ? 2 : 4) < (int64_t)D3D_RSRC_BYTES;  // 2 GB descriptors
  if (!tile_path) return d3d_wgrad_legacy(feat, dout, nbr, cap, n_out, kvol, cin, cout, dtype, dw, stream);
  D3D_REQUIRE(feat && dout && nbr, "d3d_sparse_conv_wgrad: null pointer");
  const size_t need = d3d_sparse_conv_wgrad_workspace_bytes(n_out, kvol, cin, cout, dtype);
  if (!workspace || workspace_bytes < need)
    return d3d_set_error(D3D_ECAPACITY, "d3d_sparse_conv_wgrad: workspace %zu B < %zu B", workspace_bytes, need);
  WgPlan p = wg_plan(n_out, kvol, cin, cout, dtype);
  float *part = (float *)workspace;
  int rc = wg_dispatch(feat, dout, nbr, cap, n_out, kvol, cin, cout, dtype == D3D_BF16, p, part, stream);
  if (rc) return rc;
  const int64_t total4 = (int64_t)kvol * cin * cout / 4;
  const int nb = (int)((total4 + 7) / 8);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, stream, (const float4 *)part,
                     p.splits * p.csplit, total4, (float4 *)dw);
  D3D_LAUNCH_TRY("wgrad_reduce_kernel");
  return D3D_OK;
}
