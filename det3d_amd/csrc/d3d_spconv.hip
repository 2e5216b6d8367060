// Sparse 3D convolution arithmetic for gfx950 (wave64, MFMA), output-stationary:
//
//   out[o, :] = bias + sum_k  feat[nbr[k][o], :] @ W[k]          (fwd and dgrad)
//   dW[k]     = sum_o  feat[nbr[k][o], :]^T (x) dout[o, :]       (wgrad)
//
// spconv (the reference's un-vendored dependency) runs 27 x {gather kernel, cuBLAS
// GEMM, scatter-add kernel} per layer through HBM.  Here one launch per layer
// gathers rows straight into MFMA operand registers and keeps the [sites x Cout]
// tile in accumulators across all kernel offsets, so the only HBM/L2 traffic is
// the algorithmic one: P gathered rows + the index table + each output row once.
// The path is memory-bound (<= 60 FLOP/B, ridge ~300): MFMA is used only on the
// dense [16 sites x Cin] . [Cin x Cout] tile so that arithmetic never shows up.
//
// MFMA operand mapping (16x16 tiles; C/D: col = lane&15, row = (lane>>4)*4 + reg):
//   D^T form:  A = W[k]^T tile (rows = cout), B = gathered features (cols = sites)
//   => each lane ends up with 4 CONSECUTIVE couts of ONE site: 8/16-byte row stores,
//   and the B operand for lane (site = lane&15, kgroup = lane>>4) is one contiguous
//   16-byte piece of the gathered row -- no LDS, no transposes on the hot path.
#include <stdlib.h>

#include <type_traits>

#include "d3d_common.h"
#include "d3d_fill.h"
#include "d3d_fold.h"
#include "d3d_rowplan.h"
#include "d3d_spconv_internal.h"
#include "d3d_spconv_shared.h"

// tuning hook (tools/tune_fwd.py): alternative tilings of the dominant 64->64 bf16 layer, selectable per process
// (D3D_FWD_VARIANT) or at run time (d3d_set_tuning)
// (the knobs live in the calling thread's D3dContext: d3d_context.h, d3d_core.hip)
static int fwd_variant_env() {
  static const int v = getenv("D3D_FWD_VARIANT") ? atoi(getenv("D3D_FWD_VARIANT")) : 0;
  return v;
}
// XCD-aware tile order: measured neutral-to-slightly-negative on MI355X for these gathers (profiles/round2_tune_fwd.txt:
// 50.6 vs 50.6-54 us on the 126k-site 64->64 layer, 578 vs 566 frames/s end to end), so it is off by default
// knob 10 (test knob): f32 features on the EXACT-f32 kernels -- every product one fmaf on the VALU
// (spconv_fwd_generic_kernel for forward and data gradient, spconv_wgrad_generic_kernel with one split per offset, i.e. a
// deterministic sum) instead of three bf16 MFMAs on hi / lo pairs.  d3d_sparse_conv_has_epilogue and
// d3d_sparse_conv_dgrad_plan_supported then answer 0 for f32, the packers write plain f32 copies, the host side keeps
// the per-layer path.  10-50x slower: for "split vs exact f32" comparisons, not for use.
// knob 11: chunk-resident-weights forward kernel for f32 features (spconv_fwd_resident_kernel).  -1 = the measured
// default per channel pair (FWD_RESIDENT_DEFAULT below), 0 = off (spconv_fwd_mfma_kernel everywhere), 1..3 = tiling set
extern "C" int d3d_set_tuning(int knob, int value) {
  D3D_REQUIRE(knob >= 0 && knob <= 11, "d3d_set_tuning: unknown knob %d (11 = chunk-resident-weights f32 forward kernel: -1 default, 0 off, 1..3 tiling set, 10 = exact-f32 sparse kernels for f32 features (test knob), 9 = cut the last partial round of the dense 3x3 kernel into 32-channel blocks, 8 = waves of the dense weight-gradient kernel, 7 = dense conv tiling variant, 0 = forward 64->64 tiling variant, "
              "1 = target block count of the weight-gradient kernel, 2 = XCD-aware tile order on/off, 3 = BatchNorm "
              "reduction variant for large inputs, 4 = skip the BatchNorm global atomics (timing diagnosis), 5 = rows "
              "in flight in bn_bwd_dx, 6 = grid cap of the BatchNorm streaming sweeps)", knob);
  if (knob == 0) g_fwd_variant = value;
  else if (knob == 1) g_d3d_wgrad_blocks = value;
  else if (knob == 2) g_xcd_remap = value;
  else if (knob == 3) g_bn_variant = value;
  else if (knob == 4) g_bn_skip_global = value;
  else if (knob == 5) g_bn_dx_rows = value;
  else if (knob == 6) g_bn_stream_blocks = value;
  else if (knob == 7) g_c2_variant = value;
  else if (knob == 8) g_c2_wgrad_waves = value;
  else if (knob == 9) g_c2_tail = value;
  else if (knob == 10) g_sp_exact_f32 = value;
  else g_fwd_resident = value;
  return D3D_OK;
}


// ===================================================================== forward
// One wave = MT tiles of 16 output sites x all COUT columns; a block is WAVES waves that share W[k]: the
// block stages W[k] (COUT x CIN) in LDS once per kernel offset (double buffered, +16 B row padding so
// ds_read_b128 of 16 rows is conflict-free).  The gathered rows go straight from global memory into MFMA
// B-operand registers.
//
// Round 2: the offset loop is a DEPTH-stage software pipeline that really stays in flight.  Round 1 had the
// structure (prefetch one offset ahead) but not the effect: (1) __syncthreads() makes hipcc drain every
// outstanding global load before each barrier, (2) rotating the prefetch registers with copies (fcur = fnxt)
// waits for the load that was just issued, (3) any conditional load gives the compiler a path with fewer
// loads and its s_waitcnt falls back to vmcnt(0).  So the kernel was purely latency bound (41 us for 154 MB,
// measured HBM traffic a third of that).  Now:
//   * the loop is unrolled by DEPTH with statically indexed register slots (no copies);
//   * every load is unconditional: invalid neighbours (index -1, rows past n_out) get an out-of-range buffer
//     offset and the hardware returns zeros; indices / weights past the last offset re-read the last one;
//   * the barrier orders LDS only (d3d_lds_barrier);
//   * W[k] travels through the same vmcnt stream as the gathers (loads retire in order), so it is prefetched
//     exactly as deep as they are: slot d holds gather(k+DEPTH), idx(k+2*DEPTH), W(k+1+DEPTH).
// A 16-wave block with ~100 VGPRs runs once per CU with DEPTH x WAVES x MT x 16 rows in flight (64->64,
// DEPTH 5: 160 KB per CU, against 64 KB for the 8-waves-per-SIMD / depth-1 version).
//
// PLAN (round 3, d3d_rowplan.h): the data gradient of a strided layer.  Tile rows go through the plan's permutation
// (rows of one parity class are consecutive), and a block whose rows share a class walks only that class's offsets:
// stage j of the loop is offset list[j], `kvol` becomes the length of the list.  Stages past the end of a short list
// gather nothing (out-of-range offsets) instead of re-reading the last offset.
template <int CIN, int COUT, int MT, int WAVES, int DEPTH, bool BF16, bool PLAN = false>
__global__ void __launch_bounds__(WAVES * 64, (MT * (COUT / 16) >= 16 ? 2 : 4))
    spconv_fwd_mfma_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_,
                           const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                           int cap, int n_out, int kvol_, void *__restrict__ out_, ConvEpiDev ep,
                           int xcd_remap) {
  // f32 features (BF16 = false) are multiplied in SPLIT arithmetic (round 3): a gathered f32 fragment is split in
  // registers into two bf16 vectors hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits), the weights arrive
  // pre-split as two bf16 planes, and each product is three bf16 MFMAs hi*hi + hi*lo + lo*hi with f32 accumulation:
  // relative error ~2^-16 of the products instead of 2^-8 (bf16), at a fifth of the exact-f32 MFMA's time (round 2
  // ran mfma_f32_16x16x4f32 here: 160 us for the 64->64 layer against 45 us in bf16).
  constexpr int NT = COUT / 16;
  constexpr int ESZ = BF16 ? 2 : 4;       // bytes per feature element in HBM
  constexpr int NPL = BF16 ? 1 : 2;       // bf16 planes per weight matrix
  constexpr int KW = (CIN % 32 == 0) ? 32 : 16;  // channels per k-step
  constexpr int KS = CIN / KW;
  constexpr int EPL = KW / 4;             // elements per lane per step (8 | 4)
  constexpr int ROWB = CIN * 2 + 16;      // padded LDS row of one bf16 plane (bytes)
  constexpr int WBYTES = NPL * COUT * ROWB;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int SEGS = CIN * 2 / 16;      // 16-byte segments per weight row (one plane)
  constexpr int WVEC = NPL * COUT * SEGS; // 16-byte vectors in W[k] (all planes)
  constexpr int WPT = (WVEC + NTHREADS - 1) / NTHREADS;
  static_assert(2 * WBYTES >= 4 * SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS), "epilogue scratch aliases the W buffers");
  __shared__ __attribute__((aligned(16))) unsigned char wlds[2 * WBYTES];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with its
  // own L2.  Output rows are sorted by cell index, so neighbouring tiles gather neighbouring rows; handing XCD x
  // the x-th CONTIGUOUS eighth of the tiles keeps each L2's working set to ~1/8 of the feature matrix (+ halo)
  // instead of all of it.  (Placement only affects speed: the mapping is a bijection on [0, gridDim.x).)
  int tile_block = blockIdx.x;
  if (xcd_remap) tile_block = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // gridDim.x % 8 == 0
  const int row0 = (tile_block * WAVES + wave) * (MT * 16);
  if (tile_block * WAVES * MT * 16 >= n_out) return;  // padding blocks of the rounded-up grid (whole block)

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int site[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    site[m] = row0 + m * 16 + lr;
    live[m] = site[m] < n_out;
  }
  int kvol = kvol_;                 // stages of the offset loop
  // PLAN: stage -> kernel offset, 5 bits per stage in two scalar registers (a load inside the loop would be a vector
  // load -- hipcc keeps loads behind a barrier off the scalar cache -- and its wait would drain the gather pipeline);
  // a block across a class boundary walks every offset: identity
  unsigned long long kp0 = 0, kp1 = 0;
  bool kident = true;
  if constexpr (PLAN) {
    const int32_t *pl = ep.row_plan;  // uniform addresses: scalar loads
    const int ncls = pl[0];
    const int rfirst = tile_block * (WAVES * MT * 16);
    int rlast = rfirst + WAVES * MT * 16;
    rlast = (rlast < n_out ? rlast : n_out) - 1;
    int cf = 0, cl = 0;
#pragma unroll
    for (int c = 1; c < D3D_PLAN_MAX_CLASSES; ++c) {
      const int st = pl[D3D_PLAN_START + c];
      cf += (c < ncls && rfirst >= st) ? 1 : 0;
      cl += (c < ncls && rlast >= st) ? 1 : 0;
    }
    const int32_t *klist = pl + D3D_PLAN_LISTS + (cf == cl ? cf : ncls) * D3D_PLAN_LIST_WORDS;
    kvol = klist[0];
    kident = kvol == kvol_;
    kp0 = *(const unsigned long long *)(klist + D3D_PLAN_LIST_PACKED);
    kp1 = *(const unsigned long long *)(klist + D3D_PLAN_LIST_PACKED + 2);
#pragma unroll
    for (int m = 0; m < MT; ++m) site[m] = live[m] ? pl[D3D_PLAN_PERM + site[m]] : 0;
  }
  auto stage_offset = [&](int j) -> int {
    if (!PLAN || kident) return j;
    return (int)((j < 12 ? kp0 >> (5 * j) : kp1 >> (5 * (j - 12))) & 31ull);
  };
  // gathered-row fragment of one (tile, k-step): EPL elements = 16 B (8 bf16 | 4 f32), 8 B (4 bf16, Cin = 16) or
  // 32 B (8 f32)
  struct Frag32 { uint4 a, b; };
  using FragT = typename std::conditional<BF16, typename std::conditional<KW == 16, uint2, uint4>::type,
                                          typename std::conditional<KW == 16, uint4, Frag32>::type>::type;
  // MFMA operand of one k-step: 8 bf16 (KW = 32) or 4 bf16 (KW = 16)
  using OpT = typename std::conditional<KW == 16, uint2, uint4>::type;
  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_);
  const uint32_t lane_off = (uint32_t)(lg * EPL * ESZ);
  const int kmax = kvol - 1;
  // neighbour index of this lane's sites at offset k (clamped: past the last offset / the last row it re-reads
  // a valid entry; the value is then ignored through `live` or never used)
  auto load_idx = [&](int k, int (&ix)[MT]) {
    k = k < kmax ? k : kmax;
    k = stage_offset(k);
    const int32_t *nk = nbr + (int64_t)k * cap;  // uniform base, 32-bit lane offset
#pragma unroll
    for (int m = 0; m < MT; ++m) ix[m] = nk[(uint32_t)(live[m] ? site[m] : (PLAN ? 0 : n_out - 1))];
  };
  // returns the wave's ballot of lanes with a real neighbour: a (tile group, offset) stage whose 16*MT sites all
  // miss the neighbour skips its LDS reads and MFMAs (corner offsets of submanifold layers are mostly empty)
  auto gather = [&](int k, const int (&ix)[MT], FragT (&f)[MT][KS]) -> unsigned long long {
    bool any = false;
    const bool stage = !PLAN || k <= kmax;  // PLAN: lists are short, a stage past the end must not re-gather
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool ok = live[m] && ix[m] >= 0 && stage;
      any |= ok;
      const uint32_t base = ok ? (uint32_t)ix[m] * (uint32_t)(CIN * ESZ) + lane_off : D3D_OOB_OFFSET;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint32_t off = ok ? base + (uint32_t)(ks * KW * ESZ) : D3D_OOB_OFFSET;
        if constexpr (sizeof(FragT) == 32) {
          f[m][ks].a = d3d_buffer_load16(feat_rs, off);
          f[m][ks].b = d3d_buffer_load16(feat_rs, ok ? off + 16u : D3D_OOB_OFFSET);
        } else if constexpr (sizeof(FragT) == 16) {
          f[m][ks] = d3d_buffer_load16(feat_rs, off);
        } else {
          f[m][ks] = d3d_buffer_load8(feat_rs, off);
        }
      }
    }
    return __ballot(any);
  };
  const unsigned char *wg = (const unsigned char *)wpack_;
  auto load_w = [&](int k, d3d_u32x4 (&w)[WPT]) {  // first-class vectors: arrays of HIP's uint4 struct stay in scratch
    k = k < kmax ? k : kmax;
    k = stage_offset(k);
    const unsigned char *wk = wg + (size_t)k * (WVEC * 16);  // uniform
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
      if (WVEC % NTHREADS == 0 || e < WVEC) *(d3d_u32x4 *)(dst + (e / SEGS) * ROWB + (e % SEGS) * 16) = w[j];  // (planes follow each other row by row)
    }
  };

  // ---- prologue: W(0) -> LDS[0]; slots: fr[d] = gather(d), ix[d] = idx(d + DEPTH), wr[d] = W(d + 1)
  FragT fr[DEPTH][MT][KS];
  int ix[DEPTH][MT];
  d3d_u32x4 wr[DEPTH][WPT];
  unsigned long long okm[DEPTH];  // wave-uniform
  {
    d3d_u32x4 w0[WPT];
    load_w(0, w0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_idx(d, ix[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(d + 1, wr[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      okm[d] = gather(d, ix[d], fr[d]);
      load_idx(d + DEPTH, ix[d]);
    }
    publish_w(0, w0);
  }
  d3d_lds_barrier();

  for (int kb = 0; kb < kvol; kb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = kb + d;
      const unsigned char *wcur = wlds + (k & 1) * WBYTES;
      if (k < kvol && okm[d] != 0ull) {  // uniform; tail stages of the last group only keep the load counts fixed
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ch = ks * KW + lg * EPL;
          // B operands of this k-step: the gathered fragments as they are (bf16) or split into hi / lo (f32)
          OpT xh[MT], xl[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if constexpr (BF16) {
              xh[m] = fr[d][m][ks];
            } else if constexpr (KW == 32) {
              spconv_split8(fr[d][m][ks].a, fr[d][m][ks].b, xh[m], xl[m]);
            } else {
              spconv_split4(fr[d][m][ks], xh[m], xl[m]);
            }
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const OpT wa = *(const OpT *)(wcur + (n * 16 + lr) * ROWB + ch * 2);
            OpT wl;
            if constexpr (!BF16) wl = *(const OpT *)(wcur + (COUT + n * 16 + lr) * ROWB + ch * 2);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              if constexpr (KW == 32) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xh[m]), acc[m][n], 0, 0, 0);
                if constexpr (!BF16) {
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xl[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, wl), __builtin_bit_cast(bf16x8_t, xh[m]), acc[m][n], 0, 0, 0);
                }
              } else {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                    __builtin_bit_cast(s16x4_t, wa), __builtin_bit_cast(s16x4_t, xh[m]), acc[m][n], 0, 0, 0);
                if constexpr (!BF16) {
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                      __builtin_bit_cast(s16x4_t, wa), __builtin_bit_cast(s16x4_t, xl[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                      __builtin_bit_cast(s16x4_t, wl), __builtin_bit_cast(s16x4_t, xh[m]), acc[m][n], 0, 0, 0);
                }
              }
            }
          }
        }
      }
      // W(k+1) -> the other LDS buffer (its last readers finished before the previous barrier), then refill the
      // three slots this stage freed; all loads unconditional, in consumption order
      publish_w((k + 1) & 1, wr[d]);
      load_w(k + 1 + DEPTH, wr[d]);
      okm[d] = gather(k + DEPTH, ix[d], fr[d]);   // offset k + DEPTH (its indices arrived DEPTH stages ago)
      load_idx(k + 2 * DEPTH, ix[d]);
      d3d_lds_barrier();
    }
  }

  // every wave is past the last barrier of the offset loop: the W buffers are free for the epilogue's sums
  spconv_fwd_epilogue<COUT, MT, BF16, NTHREADS>(acc, site, live, bias, out_, ep, (float *)wlds, lr, lg);
}

// ---- chunk-resident weights (bf16, CIN % 32 == 0) ------------------------------------------------------------
// The kernel above exchanges W[k] through LDS once per kernel offset, i.e. one workgroup barrier per offset: the
// waves of a block run in lock step, so they all issue their gathers in the same cycles (SQ_VMEM_TA_CMD_FIFO_FULL
// 20 % of the time), then all read LDS (29 % of the LDS cycles are bank conflicts on the padded rows), then all
// queue for the MFMA pipe.  Counter passes on the 64->64 layer (profiles/round2_pmc_fwd.txt) show no unit above
// 50 % (TA 44-50 %, LDS 28 %, MFMA 17-20 %, VALU 16 %): the time is the SUM of the phases, not their maximum.
// Here W stays resident in LDS for KC offsets at a time (64->64: 9 offsets = 72 KB, unpadded rows with an XOR
// swizzle of the 16-byte cells instead of padding, two blocks per CU) and is fenced by ONE barrier pair per chunk:
// 3 instead of 27 rendezvous, and between them every wave walks its offsets at its own pace, so gather issue, LDS
// reads and MFMAs of different waves overlap.
//   weight cell (16 B) c of a chunk lives in LDS cell c ^ ((c >> 4) & 15): the 16 lanes of an MFMA A-fragment read
//   (rows r..r+15, same segment) then hit 16 distinct 16-byte bank groups.
template <int CIN, int COUT, int MT, int WAVES, int DEPTH, int KC, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)  // second value: waves per SIMD
    spconv_fwd_chunk_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_,
                            const float *__restrict__ bias, const int32_t *__restrict__ nbr, int cap, int n_out,
                            int kvol, void *__restrict__ out_, ConvEpiDev ep) {
  constexpr int NT = COUT / 16;
  constexpr int ESZ = 2, KW = 32, EPL = 8;
  constexpr int KS = CIN / KW;
  constexpr int SEGS = CIN * ESZ / 16;   // 16-byte cells per weight row
  constexpr int WVEC = COUT * SEGS;      // cells per kernel offset
  constexpr int CHUNK = KC * WVEC;       // cells per chunk
  constexpr int NTHREADS = WAVES * 64;
  constexpr int GPW = (CHUNK + NTHREADS - 1) / NTHREADS;  // global_load_lds instructions per wave and chunk
  static_assert(CIN % 32 == 0 && COUT % 16 == 0, "bf16 x32 MFMA tiles");
  static_assert(WVEC % 256 == 0, "the swizzle must not see the offset index");
  static_assert(CHUNK % 64 == 0 && KC % DEPTH == 0, "chunk / pipeline shape");
  static_assert(CHUNK * 16 >= 4 * SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS), "epilogue scratch aliases the W buffers");
  __shared__ __attribute__((aligned(16))) unsigned char wlds[CHUNK * 16];
  float *red = (float *)wlds;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = (blockIdx.x * WAVES + wave) * (MT * 16);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int site[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    site[m] = row0 + m * 16 + lr;
    live[m] = site[m] < n_out;
  }
  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_);
  const uint32_t lane_off = (uint32_t)(lg * EPL * ESZ);
  const int kmax = kvol - 1;
  auto load_idx = [&](int k, int (&ix)[MT]) {
    k = k < kmax ? k : kmax;
    const int32_t *nk = nbr + (int64_t)k * cap;
#pragma unroll
    for (int m = 0; m < MT; ++m) ix[m] = nk[(uint32_t)(live[m] ? site[m] : n_out - 1)];
  };
  auto gather = [&](const int (&ix)[MT], uint4 (&f)[MT][KS]) -> unsigned long long {
    bool any = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool ok = live[m] && ix[m] >= 0;
      any |= ok;
      const uint32_t base = ok ? (uint32_t)ix[m] * (uint32_t)(CIN * ESZ) + lane_off : D3D_OOB_OFFSET;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        f[m][ks] = d3d_buffer_load16(feat_rs, ok ? base + (uint32_t)(ks * KW * ESZ) : D3D_OOB_OFFSET);
    }
    return __ballot(any);
  };
  // this lane's A-fragment cell inside one offset (swizzled), bytes.  The swizzle term of row n*16 + lr only depends
  // on n through the bits of (n*16*SEGS) >> 4 below 16, so a few registers cover all n (the rest is an immediate)
  constexpr int NPER = (16 * SEGS / 16) >= 16 ? 1 : 16 / (16 * SEGS / 16);  // row tiles until the term repeats
  constexpr int NW = NT < NPER ? NT : NPER;
  uint32_t woff[NW][KS];
#pragma unroll
  for (int n = 0; n < NW; ++n)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t c = (uint32_t)((n * 16 + lr) * SEGS + ks * (KW * ESZ / 16) + lg);
      c ^= (c >> 4) & 15u;
      woff[n][ks] = c * 16u;
    }
  const unsigned char *wg = (const unsigned char *)wpack_;
  const uint32_t wcells = (uint32_t)kvol * WVEC;
  // W[k0 .. k0+KC) -> LDS through registers, SUB loads in flight per thread.  (global_load_lds would save the
  // registers and the ds_write pass, but with an LDS-DMA possibly in flight hipcc waits vmcnt(0) before EVERY use
  // of a gathered row -- measured: the offset pipeline collapses to depth 0.)
  auto load_chunk = [&](int k0) {
    constexpr int SUB = 3;
#pragma unroll
    for (int j0 = 0; j0 < GPW; j0 += SUB) {
      d3d_u32x4 w[SUB];
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        if (j0 + u >= GPW) continue;
        uint32_t g = (uint32_t)k0 * WVEC + (uint32_t)((j0 + u) * NTHREADS) + threadIdx.x;
        g = g < wcells ? g : wcells - 1;  // offsets past the last one (kvol < KC): never read
        w[u] = *(const d3d_u32x4 *)(wg + (size_t)g * 16);
      }
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        if (j0 + u >= GPW) continue;
        const uint32_t e = (uint32_t)((j0 + u) * NTHREADS) + threadIdx.x;
        if (CHUNK % NTHREADS == 0 || e < (uint32_t)CHUNK) *(d3d_u32x4 *)(wlds + (e ^ ((e >> 4) & 15u)) * 16u) = w[u];
      }
    }
  };

  uint4 fr[DEPTH][MT][KS];
  int ix[DEPTH][MT];
  unsigned long long okm[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_idx(d, ix[d]);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    okm[d] = gather(ix[d], fr[d]);
    load_idx(d + DEPTH, ix[d]);
  }

  for (int k0 = 0; k0 < kvol; k0 += KC) {
    if (k0) d3d_lds_barrier();  // every wave has finished reading the previous chunk
    load_chunk(k0);
    d3d_lds_barrier();
#pragma unroll 1
    for (int kb = k0; kb < k0 + KC && kb < kvol; kb += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = kb + d;
        if (k < kvol && okm[d] != 0ull) {
          const unsigned char *wk = wlds + (uint32_t)(k - k0) * (WVEC * 16);
          // all A fragments of this offset are requested before the first MFMA: one exposed LDS round trip per
          // offset (the MFMAs then drain them with partial lgkmcnt waits) instead of one per fragment
          uint4 wa[KS][NT];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              wa[ks][n] = *(const uint4 *)(wk + woff[n % NW][ks] + (uint32_t)((n / NW) * NW * 16 * SEGS * 16));
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, wa[ks][n]), __builtin_bit_cast(bf16x8_t, fr[d][m][ks]), acc[m][n], 0, 0, 0);
          // keep that order: left alone, the scheduler minimises registers and emits read / wait / 2 MFMAs eight times
          __builtin_amdgcn_sched_group_barrier(0x100, KS * NT, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, KS * NT * MT, 0);
        }
        okm[d] = gather(ix[d], fr[d]);  // offset k + DEPTH
        load_idx(k + 2 * DEPTH, ix[d]);
      }
    }
  }
  spconv_fwd_epilogue<COUT, MT, true, NTHREADS>(acc, site, live, bias, out_, ep, red, lr, lg);
}

// ---- chunk-resident weights, f32 features in split arithmetic (round 5) --------------------------------------
// spconv_fwd_mfma_kernel exchanges W[k] through LDS once per kernel offset: 27 workgroup rendezvous per block.  On the
// 16 / 32-channel layers a stage is 6-12 MFMAs per wave, so the launch is the latency of 27 barrier-separated stages
// (`<16,16>` 0.07, `<32,32>` 0.16-0.22 of the algorithmic roofline: VERDICT round 4 item 4).  Here KC weight matrices
// (both bf16 planes, the padded rows of the kernel above) sit in LDS at a time -- all 27 where they fit -- fenced by
// one barrier pair per chunk; between the fences every wave walks its offsets at its own pace with its gathers
// DEPTH offsets ahead.  Tile shapes, the order of a row's products (offset, k-step, hi*hi / hi*lo / lo*hi) and the
// epilogue are those of spconv_fwd_mfma_kernel: the produced rows are bit-identical (the statistics too where the
// block shape is the same).  No row plan (the strided layers' data gradients stay on the kernel above).
template <int CIN, int COUT, int MT, int WAVES, int DEPTH, int KC, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)  // second value: waves per SIMD
    spconv_fwd_resident_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_,
                               const float *__restrict__ bias, const int32_t *__restrict__ nbr, int cap, int n_out,
                               int kvol, void *__restrict__ out_, ConvEpiDev ep) {
  constexpr int NT = COUT / 16;
  constexpr int ESZ = 4;                          // f32 features
  constexpr int KW = (CIN % 32 == 0) ? 32 : 16;   // channels per k-step
  constexpr int KS = CIN / KW;
  constexpr int EPL = KW / 4;                     // elements per lane per step (8 | 4)
  // LDS row of one bf16 plane: padded by 16 B as in the kernel above, or -- 64 input channels, where nine padded
  // matrices would not fit -- 128 B unpadded with the 16-byte cell c of row r at cell c ^ ((r >> 1) & 7): the 16 lanes
  // ds_read_b128 serves per cycle ({lr 0-3, 12-15} of one k-group with {lr 4-11} of the next) then hit 16 distinct cells
  constexpr bool SWZ = CIN == 64;
  constexpr int ROWB = SWZ ? CIN * 2 : CIN * 2 + 16;
  constexpr int WBYTES = 2 * COUT * ROWB;         // one offset: hi plane rows, then lo plane rows
  constexpr int NTHREADS = WAVES * 64;
  constexpr int SEGS = CIN * 2 / 16;              // 16-byte vectors per weight row
  constexpr int WVEC = 2 * COUT * SEGS;           // vectors per offset in global memory
  constexpr int CHUNKV = KC * WVEC;
  constexpr int GPW = (CHUNKV + NTHREADS - 1) / NTHREADS;
  static_assert(KC % DEPTH == 0, "chunk / pipeline shape");
  static_assert((SEGS & (SEGS - 1)) == 0 && (WVEC & (WVEC - 1)) == 0, "vector -> (offset, row, segment) by shifts");
  static_assert(KC * WBYTES >= 4 * SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS), "epilogue scratch aliases the W chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char rwlds[];  // KC * WBYTES

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = (blockIdx.x * WAVES + wave) * (MT * 16);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int site[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    site[m] = row0 + m * 16 + lr;
    live[m] = site[m] < n_out;
  }
  struct Frag32 { uint4 a, b; };
  using FragT = typename std::conditional<KW == 16, uint4, Frag32>::type;  // 4 | 8 f32 of a gathered row
  using OpT = typename std::conditional<KW == 16, uint2, uint4>::type;     // 4 | 8 bf16: one MFMA operand
  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_);
  const uint32_t lane_off = (uint32_t)(lg * EPL * ESZ);
  const int kmax = kvol - 1;
  auto load_idx = [&](int k, int (&ix)[MT]) {
    k = k < kmax ? k : kmax;
    const int32_t *nk = nbr + (int64_t)k * cap;
#pragma unroll
    for (int m = 0; m < MT; ++m) ix[m] = nk[(uint32_t)(live[m] ? site[m] : n_out - 1)];
  };
  auto gather = [&](const int (&ix)[MT], FragT (&f)[MT][KS]) -> unsigned long long {
    bool any = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool ok = live[m] && ix[m] >= 0;
      any |= ok;
      const uint32_t base = ok ? (uint32_t)ix[m] * (uint32_t)(CIN * ESZ) + lane_off : D3D_OOB_OFFSET;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint32_t off = ok ? base + (uint32_t)(ks * KW * ESZ) : D3D_OOB_OFFSET;
        if constexpr (KW == 32) {
          f[m][ks].a = d3d_buffer_load16(feat_rs, off);
          f[m][ks].b = d3d_buffer_load16(feat_rs, ok ? off + 16u : D3D_OOB_OFFSET);
        } else {
          f[m][ks] = d3d_buffer_load16(feat_rs, off);
        }
      }
    }
    return __ballot(any);
  };
  const unsigned char *wg = (const unsigned char *)wpack_;
  const uint32_t wcells = (uint32_t)kvol * WVEC;
  // W[k0 .. k0 + KC) -> LDS through registers, SUB loads in flight per thread (an LDS-DMA in flight would make hipcc
  // wait vmcnt(0) before every use of a gathered row: see spconv_fwd_chunk_kernel)
  auto load_chunk = [&](int k0) {
    constexpr int SUB = GPW < 4 ? GPW : 4;
#pragma unroll
    for (int j0 = 0; j0 < GPW; j0 += SUB) {
      d3d_u32x4 w[SUB];
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        if (j0 + u >= GPW) continue;
        uint32_t g = (uint32_t)k0 * WVEC + (uint32_t)((j0 + u) * NTHREADS) + threadIdx.x;
        g = g < wcells ? g : wcells - 1;  // offsets past the last one (kvol < KC): never read
        w[u] = *(const d3d_u32x4 *)(wg + (size_t)g * 16);
      }
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        if (j0 + u >= GPW) continue;
        const uint32_t e = (uint32_t)((j0 + u) * NTHREADS) + threadIdx.x;
        const uint32_t kk = e / WVEC, r = e % WVEC;
        if (CHUNKV % NTHREADS == 0 || e < (uint32_t)CHUNKV)
          *(d3d_u32x4 *)(rwlds + kk * WBYTES + (r / SEGS) * ROWB + (((r % SEGS) ^ (SWZ ? ((r / SEGS) >> 1) & 7u : 0u)) * 16)) = w[u];
      }
    }
  };

  FragT fr[DEPTH][MT][KS];
  int ix[DEPTH][MT];
  unsigned long long okm[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_idx(d, ix[d]);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    okm[d] = gather(ix[d], fr[d]);
    load_idx(d + DEPTH, ix[d]);
  }

  for (int k0 = 0; k0 < kvol; k0 += KC) {
    if (k0) d3d_lds_barrier();  // every wave has finished reading the previous chunk
    load_chunk(k0);
    d3d_lds_barrier();
#pragma unroll 1
    for (int kb = k0; kb < k0 + KC && kb < kvol; kb += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = kb + d;
        if (k < kvol && okm[d] != 0ull) {  // uniform
          const unsigned char *wcur = rwlds + (uint32_t)(k - k0) * WBYTES;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int cb = ((ks * KW + lg * EPL) * 2) ^ (SWZ ? ((lr >> 1) & 7) << 4 : 0);  // byte offset in the row
            OpT xh[MT], xl[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              if constexpr (KW == 32) spconv_split8(fr[d][m][ks].a, fr[d][m][ks].b, xh[m], xl[m]);
              else spconv_split4(fr[d][m][ks], xh[m], xl[m]);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const OpT wa = *(const OpT *)(wcur + (n * 16 + lr) * ROWB + cb);
              const OpT wl = *(const OpT *)(wcur + (COUT + n * 16 + lr) * ROWB + cb);
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                if constexpr (KW == 32) {
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xh[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xl[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      __builtin_bit_cast(bf16x8_t, wl), __builtin_bit_cast(bf16x8_t, xh[m]), acc[m][n], 0, 0, 0);
                } else {
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                      __builtin_bit_cast(s16x4_t, wa), __builtin_bit_cast(s16x4_t, xh[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                      __builtin_bit_cast(s16x4_t, wa), __builtin_bit_cast(s16x4_t, xl[m]), acc[m][n], 0, 0, 0);
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                      __builtin_bit_cast(s16x4_t, wl), __builtin_bit_cast(s16x4_t, xh[m]), acc[m][n], 0, 0, 0);
                }
              }
            }
          }
        }
        okm[d] = gather(ix[d], fr[d]);  // offset k + DEPTH (past the last offset: the last one again, never used)
        load_idx(k + 2 * DEPTH, ix[d]);
      }
    }
  }
  // (the epilogue's reductions start behind a workgroup barrier: no wave still reads the chunk its scratch aliases)
  spconv_fwd_epilogue<COUT, MT, false, NTHREADS>(acc, site, live, bias, out_, ep, (float *)rwlds, lr, lg);
}

template <int CIN, int COUT, int MT, int WAVES, int DEPTH, int KC, int MINW = (MT * (COUT / 16) >= 16 ? 2 : 4)>
static int launch_fwd_resident(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                               int n_out, int kvol, void *out, ConvEpiDev ep, hipStream_t stream) {
  const int rows_per_block = WAVES * MT * 16;
  const int nb = (n_out + rows_per_block - 1) / rows_per_block;
  if (int rc = epi_prepare(ep, nb, COUT)) return rc;
  constexpr size_t lds = (size_t)KC * 2 * COUT * (CIN == 64 ? CIN * 2 : CIN * 2 + 16);
  static_assert(lds <= 160 * 1024, "LDS of one CU");
  D3D_HIP_TRY(hipFuncSetAttribute((const void *)spconv_fwd_resident_kernel<CIN, COUT, MT, WAVES, DEPTH, KC, MINW>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((spconv_fwd_resident_kernel<CIN, COUT, MT, WAVES, DEPTH, KC, MINW>), dim3(nb), dim3(WAVES * 64), lds,
                     stream, feat, wpack, bias, nbr, cap, n_out, kvol, out, ep);
  D3D_LAUNCH_TRY("spconv_fwd_resident_kernel");
  return D3D_OK;
}

// ---- grouped barriers: the pipelined kernel above with ONE rendezvous per DEPTH offsets -----------------------
// LDS holds two groups of DEPTH weight matrices (unpadded rows, same cell swizzle as the chunk kernel).  During
// group g every wave reads buffer g & 1 and writes W of group g+1 -- whose loads it issued a group earlier, riding the
// same in-order vmcnt stream as the gathers -- into the other buffer; the barrier at the end of the group both
// publishes those writes and retires the reads.  27 offsets = 9 barriers at DEPTH 3.
template <int CIN, int COUT, int MT, int WAVES, int DEPTH>
__global__ void __launch_bounds__(WAVES * 64, (MT * (COUT / 16) >= 16 ? 2 : 4))
    spconv_fwd_group_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_,
                            const float *__restrict__ bias, const int32_t *__restrict__ nbr, int cap, int n_out,
                            int kvol, void *__restrict__ out_, ConvEpiDev ep) {
  constexpr int NT = COUT / 16;
  constexpr int ESZ = 2, KW = 32, EPL = 8;
  constexpr int KS = CIN / KW;
  constexpr int SEGS = CIN * ESZ / 16;
  constexpr int WVEC = COUT * SEGS;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int WPT = (WVEC + NTHREADS - 1) / NTHREADS;
  static_assert(CIN % 32 == 0 && COUT % 16 == 0 && WVEC % 256 == 0, "bf16 x32 MFMA tiles, swizzle period");
  static_assert(2 * DEPTH * WVEC * 16 >= 4 * SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS), "epilogue scratch aliases the W buffers");
  __shared__ __attribute__((aligned(16))) unsigned char wlds[2 * DEPTH * WVEC * 16];
  float *red = (float *)wlds;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = (blockIdx.x * WAVES + wave) * (MT * 16);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int site[MT];
  bool live[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    site[m] = row0 + m * 16 + lr;
    live[m] = site[m] < n_out;
  }
  const __amdgpu_buffer_rsrc_t feat_rs = d3d_make_rsrc(feat_);
  const uint32_t lane_off = (uint32_t)(lg * EPL * ESZ);
  const int kmax = kvol - 1;
  auto load_idx = [&](int k, int (&ix)[MT]) {
    k = k < kmax ? k : kmax;
    const int32_t *nk = nbr + (int64_t)k * cap;
#pragma unroll
    for (int m = 0; m < MT; ++m) ix[m] = nk[(uint32_t)(live[m] ? site[m] : n_out - 1)];
  };
  auto gather = [&](const int (&ix)[MT], uint4 (&f)[MT][KS]) -> unsigned long long {
    bool any = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool ok = live[m] && ix[m] >= 0;
      any |= ok;
      const uint32_t base = ok ? (uint32_t)ix[m] * (uint32_t)(CIN * ESZ) + lane_off : D3D_OOB_OFFSET;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        f[m][ks] = d3d_buffer_load16(feat_rs, ok ? base + (uint32_t)(ks * KW * ESZ) : D3D_OOB_OFFSET);
    }
    return __ballot(any);
  };
  constexpr int NPER = SEGS >= 16 ? 1 : 16 / SEGS;
  constexpr int NW = NT < NPER ? NT : NPER;
  uint32_t woff[NW][KS];
#pragma unroll
  for (int n = 0; n < NW; ++n)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t c = (uint32_t)((n * 16 + lr) * SEGS + ks * (KW * ESZ / 16) + lg);
      c ^= (c >> 4) & 15u;
      woff[n][ks] = c * 16u;
    }
  const unsigned char *wg = (const unsigned char *)wpack_;
  auto load_w = [&](int k, d3d_u32x4 (&w)[WPT]) {
    k = k < kmax ? k : kmax;
    const unsigned char *wk = wg + (size_t)k * (WVEC * 16);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int e = threadIdx.x + j * NTHREADS;
      w[j] = *(const d3d_u32x4 *)(wk + (uint32_t)((e < WVEC ? e : WVEC - 1) * 16));
    }
  };
  auto publish_w = [&](int buf, int slot, const d3d_u32x4 (&w)[WPT]) {
    unsigned char *dst = wlds + (uint32_t)((buf * DEPTH + slot) * WVEC * 16);
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const uint32_t e = threadIdx.x + j * NTHREADS;
      if (WVEC % NTHREADS == 0 || e < (uint32_t)WVEC) *(d3d_u32x4 *)(dst + (e ^ ((e >> 4) & 15u)) * 16u) = w[j];
    }
  };

  // prologue: W(0..DEPTH) -> buffer 0; slots: fr[d] = gather(d), ix[d] = idx(d + DEPTH), wr[d] = W(d + DEPTH)
  uint4 fr[DEPTH][MT][KS];
  int ix[DEPTH][MT];
  d3d_u32x4 wr[DEPTH][WPT];
  unsigned long long okm[DEPTH];
  {
    d3d_u32x4 w0[DEPTH][WPT];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(d, w0[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_idx(d, ix[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_w(d + DEPTH, wr[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      okm[d] = gather(ix[d], fr[d]);
      load_idx(d + DEPTH, ix[d]);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) publish_w(0, d, w0[d]);
  }
  d3d_lds_barrier();

  int buf = 0;
  for (int kb = 0; kb < kvol; kb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = kb + d;
      if (k < kvol && okm[d] != 0ull) {
        const unsigned char *wk = wlds + (uint32_t)((buf * DEPTH + d) * WVEC * 16);
        uint4 wa[KS][NT];  // requested together: one exposed LDS round trip per offset, not one per fragment
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            wa[ks][n] = *(const uint4 *)(wk + woff[n % NW][ks] + (uint32_t)((n / NW) * NW * 16 * SEGS * 16));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8_t, wa[ks][n]), __builtin_bit_cast(bf16x8_t, fr[d][m][ks]), acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, KS * NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, KS * NT * MT, 0);
      }
      publish_w(buf ^ 1, d, wr[d]);      // W(k + DEPTH): slot d of the next group
      load_w(k + 2 * DEPTH, wr[d]);
      okm[d] = gather(ix[d], fr[d]);     // offset k + DEPTH
      load_idx(k + 2 * DEPTH, ix[d]);
    }
    d3d_lds_barrier();
    buf ^= 1;
  }
  spconv_fwd_epilogue<COUT, MT, true, NTHREADS>(acc, site, live, bias, out_, ep, red, lr, lg);
}

template <int CIN, int COUT, int MT, int WAVES, int DEPTH>
static int launch_fwd_group(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                            int n_out, int kvol, void *out, ConvEpiDev ep, hipStream_t stream) {
  const int rows_per_block = WAVES * MT * 16;
  const int nb = (n_out + rows_per_block - 1) / rows_per_block;
  if (int rc = epi_prepare(ep, nb, COUT)) return rc;
  hipLaunchKernelGGL((spconv_fwd_group_kernel<CIN, COUT, MT, WAVES, DEPTH>), dim3(nb), dim3(WAVES * 64), 0, stream,
                     feat, wpack, bias, nbr, cap, n_out, kvol, out, ep);
  D3D_LAUNCH_TRY("spconv_fwd_group_kernel");
  return D3D_OK;
}

template <int CIN, int COUT, int MT, int WAVES, int DEPTH, int KC, int MINW = (MT * (COUT / 16) >= 16 ? 2 : 4)>
static int launch_fwd_chunk(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                            int n_out, int kvol, void *out, ConvEpiDev ep, hipStream_t stream) {
  const int rows_per_block = WAVES * MT * 16;
  const int nb = (n_out + rows_per_block - 1) / rows_per_block;
  if (int rc = epi_prepare(ep, nb, COUT)) return rc;
  hipLaunchKernelGGL((spconv_fwd_chunk_kernel<CIN, COUT, MT, WAVES, DEPTH, KC, MINW>), dim3(nb), dim3(WAVES * 64), 0,
                     stream, feat, wpack, bias, nbr, cap, n_out, kvol, out, ep);
  D3D_LAUNCH_TRY("spconv_fwd_chunk_kernel");
  return D3D_OK;
}

// generic fallback (any cin/cout, e.g. the 4->16 / 5->16 stem): one thread per
// (site, cout), f32 accumulate.  Also the on-device second opinion in tests.
template <bool BF16>
__global__ void __launch_bounds__(256)
    spconv_fwd_generic_kernel(const void *__restrict__ feat_, const void *__restrict__ wpack_,
                              const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                              int cap, int n_out, int kvol, int cin, int cout,
                              void *__restrict__ out_) {
  const int64_t total = (int64_t)n_out * cout;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int o = (int)(t / cout);
    int co = (int)(t - (int64_t)o * cout);
    float acc = bias ? bias[co] : 0.f;
    for (int k = 0; k < kvol; ++k) {
      int i = nbr[(int64_t)k * cap + o];
      if (i < 0) continue;
      if constexpr (BF16) {
        const uint16_t *f = (const uint16_t *)feat_ + (int64_t)i * cin;
        const uint16_t *w = (const uint16_t *)wpack_ + ((int64_t)k * cout + co) * cin;
        for (int c = 0; c < cin; ++c) acc = fmaf(bf16_to_f32(f[c]), bf16_to_f32(w[c]), acc);
      } else {
        const float *f = (const float *)feat_ + (int64_t)i * cin;
        const float *w = (const float *)wpack_ + ((int64_t)k * cout + co) * cin;
        for (int c = 0; c < cin; ++c) acc = fmaf(f[c], w[c], acc);
      }
    }
    if constexpr (BF16)
      ((uint16_t *)out_)[t] = f32_to_bf16(acc);
    else
      ((float *)out_)[t] = acc;
  }
}

static int launch_fwd_generic(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                              int n_out, int kvol, int cin, int cout, int dtype, void *out, hipStream_t stream) {
  int64_t items = (int64_t)n_out * cout;
  int nb = (int)((items + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(spconv_fwd_generic_kernel<true>, dim3(nb), dim3(256), 0, stream, feat, wpack,
                       bias, nbr, cap, n_out, kvol, cin, cout, out);
  else
    hipLaunchKernelGGL(spconv_fwd_generic_kernel<false>, dim3(nb), dim3(256), 0, stream, feat,
                       wpack, bias, nbr, cap, n_out, kvol, cin, cout, out);
  D3D_LAUNCH_TRY("spconv_fwd_generic_kernel");
  return D3D_OK;
}

template <int CIN, int COUT, int MT, int WAVES, int DEPTH, bool BF16, bool PLAN = false>
static int launch_fwd(const void *feat, const void *wpack, const float *bias, const int32_t *nbr,
                      int cap, int n_out, int kvol, void *out, ConvEpiDev ep, hipStream_t stream) {
  const int rows_per_block = WAVES * MT * 16;
  int nb = (n_out + rows_per_block - 1) / rows_per_block;
  const bool reduces = ep.stats || ep.bn_partial;
  const int remap = g_xcd_remap && nb >= 16 && !reduces;  // every block of a reducing launch takes a fold ticket
  if (remap) nb = (nb + 7) / 8 * 8;
  if (int rc = epi_prepare(ep, nb, COUT)) return rc;
  hipLaunchKernelGGL((spconv_fwd_mfma_kernel<CIN, COUT, MT, WAVES, DEPTH, BF16, PLAN>), dim3(nb),
                     dim3(WAVES * 64), 0, stream, feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, remap);
  D3D_LAUNCH_TRY("spconv_fwd_mfma_kernel");
  return D3D_OK;
}

// D3D_F32_WIDE=1 runs f32 features one tile per wave on 16 waves.  Alone on the 126k-site 64->64 layer that is faster
// (97.6 vs 112.6 us, profiles/round3_tune_fwd_f32.txt), inside the step it is not: 64->64 90 vs 79 us, 32->32 79 vs
// 64 us, 32->64 73 vs 56 us, 498 vs 503 frames/s (gpurun_out/r3j20) -- the step's layers are smaller and share the GPU
// with the geometry stream.  Off.
static const int g_f32_wide = getenv("D3D_F32_WIDE") ? atoi(getenv("D3D_F32_WIDE")) : 0;
#define FWD_CASE(CI, CO, MT, WV, DB, DF)                                                            \
  if (cin == CI && cout == CO) {                                                                    \
    if (dtype == D3D_BF16)                                                                          \
      return launch_fwd<CI, CO, MT, WV, DB, true>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream); \
    if (g_f32_wide && CI >= 32)                                                                     \
      return launch_fwd<CI, CO, 1, 16, (DF > 2 ? 2 : DF), false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream); \
    return launch_fwd<CI, CO, MT, WV, DF, false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream); \
  }

// channel pairs with an MFMA tiling below (FWD_CASE list): only these can carry the epilogue reductions
extern "C" int d3d_sparse_conv_has_epilogue(int cin, int cout, int dtype) {
  if (dtype != D3D_F32 && dtype != D3D_BF16) return 0;
  if (g_sp_exact_f32 && dtype == D3D_F32) return 0;  // knob 10: the VALU kernels carry no epilogue
  static const int pairs[][2] = {{16, 16}, {16, 32}, {32, 16}, {32, 32}, {32, 64}, {64, 32},
                                 {64, 64}, {64, 128}, {128, 64}, {128, 128}};
  for (auto &pr : pairs)
    if (pr[0] == cin && pr[1] == cout) return 1;
  return 0;
}
// floats of the buffer D3dConvEpilogue.stats / .bn_partial point to for a launch producing n_out rows:
// [2*cout] result | block slabs | group slabs, sized for the smallest tile any variant uses (64 rows per block)
extern "C" size_t d3d_conv_epilogue_ws_floats(int n_out, int cout) {
  if (n_out < 1 || cout < 1) return 0;
  int nb = (n_out + 63) / 64 + 8;
  if (nb > D3D_FOLD_MAX_BLOCKS) nb = D3D_FOLD_MAX_BLOCKS;
  return d3d_fold_ws_floats(nb, 2 * cout);
}

extern "C" int d3d_sparse_conv_fwd(const void *feat, const void *wpack, const float *bias,
                                   const int32_t *nbr, int cap, int n_out, int kvol, int cin,
                                   int cout, int dtype, void *out, d3d_stream_t stream_) {
  return d3d_sparse_conv_fwd_ex(feat, wpack, bias, nbr, cap, n_out, kvol, cin, cout, dtype, out, nullptr, stream_);
}

static int sparse_conv_fwd_impl(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                                int n_out, int kvol, int cin, int cout, int dtype, void *out,
                                const D3dConvEpilogue *epilogue, const int32_t *row_plan, d3d_stream_t stream_);

extern "C" int d3d_sparse_conv_fwd_ex(const void *feat, const void *wpack, const float *bias,
                                      const int32_t *nbr, int cap, int n_out, int kvol, int cin,
                                      int cout, int dtype, void *out, const D3dConvEpilogue *epilogue,
                                      d3d_stream_t stream_) {
  return sparse_conv_fwd_impl(feat, wpack, bias, nbr, cap, n_out, kvol, cin, cout, dtype, out, epilogue, nullptr,
                              stream_);
}

// channel pairs (gradient channels -> input channels) of the strided layers in the shipped backbones
// (scn.py:113-150: 16->32, 32->64, 64->64; :347-407: 64->128, 128->128), either precision
extern "C" int d3d_sparse_conv_dgrad_plan_supported(int cin, int cout, int dtype) {
  if (dtype != D3D_F32 && dtype != D3D_BF16) return 0;
  if (g_sp_exact_f32 && dtype == D3D_F32) return 0;
  static const int pairs[][2] = {{32, 16}, {64, 32}, {64, 64}, {128, 64}, {128, 128}};
  for (auto &pr : pairs)
    if (pr[0] == cin && pr[1] == cout) return 1;
  return 0;
}

extern "C" int d3d_sparse_conv_dgrad_planned(const void *dout, const void *wpack, const int32_t *nbr_in, int cap_in,
                                             int n_in, int kvol, int cin, int cout, int dtype, void *din,
                                             const D3dConvEpilogue *epilogue, const int32_t *plan,
                                             d3d_stream_t stream_) {
  D3D_REQUIRE(plan, "d3d_sparse_conv_dgrad_planned: null plan");
  D3D_REQUIRE(d3d_sparse_conv_dgrad_plan_supported(cin, cout, dtype),
              "d3d_sparse_conv_dgrad_planned: no planned tiling for channels %d -> %d (ask "
              "d3d_sparse_conv_dgrad_plan_supported)", cin, cout);
  D3D_REQUIRE(kvol <= D3D_PLAN_MAX_KVOL, "d3d_sparse_conv_dgrad_planned: kernel volume %d", kvol);
  return sparse_conv_fwd_impl(dout, wpack, nullptr, nbr_in, cap_in, n_in, kvol, cin, cout, dtype, din, epilogue, plan,
                              stream_);
}

static int sparse_conv_fwd_impl(const void *feat, const void *wpack, const float *bias, const int32_t *nbr, int cap,
                                int n_out, int kvol, int cin, int cout, int dtype, void *out,
                                const D3dConvEpilogue *epilogue, const int32_t *row_plan, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvEpiDev ep = {};
  if (epilogue) *(D3dConvEpilogue *)&ep = *epilogue;
  ep.row_plan = row_plan;
  D3D_REQUIRE(!(ep.stats && ep.bn_partial), "d3d_sparse_conv_fwd_ex: one reduction per launch (stats or bn_partial)");
  D3D_REQUIRE(!ep.bn_partial || (ep.bn_x && ep.bn_mean && ep.bn_invstd),
              "d3d_sparse_conv_fwd_ex: bn_partial needs bn_x, bn_mean, bn_invstd");
  D3D_REQUIRE(!(ep.stats || ep.bn_partial || ep.addend) || d3d_sparse_conv_has_epilogue(cin, cout, dtype),
              "d3d_sparse_conv_fwd_ex: the fused reductions / the addend exist on the MFMA tiles only, channels %d -> %d "
              "have none (ask d3d_sparse_conv_has_epilogue)", cin, cout);
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_sparse_conv_fwd: dtype %d", dtype);
  D3D_REQUIRE(n_out >= 0 && n_out <= cap && kvol >= 1 && cin >= 1 && cout >= 1,
              "d3d_sparse_conv_fwd: n_out=%d cap=%d kvol=%d cin=%d cout=%d", n_out, cap, kvol, cin,
              cout);
  if (n_out == 0) return D3D_OK;
  D3D_REQUIRE(feat && wpack && nbr && out, "d3d_sparse_conv_fwd: null pointer");
  // (the gathered matrix must stay below the 2 GB a buffer descriptor addresses; its row count is the caller's
  // knowledge -- det3d_amd/spconv.py checks it -- the produced matrix is checked here)
  D3D_REQUIRE((int64_t)n_out * cout * (dtype == D3D_BF16 ? 2 : 4) < (int64_t)D3D_RSRC_BYTES,
              "d3d_sparse_conv_fwd: %d x %d outputs exceed 2 GB", n_out, cout);
  // (Cin, Cout) -> tiles per wave, waves per block, pipeline depth for bf16 / f32.  Depth x gathered bytes per
  // stage is what hides the L2 gather latency; it is bounded by the 128 VGPRs of a 4-waves-per-SIMD block.
  // D3D_FWD_VARIANT (tools/tune_fwd.py) selects alternative 64->64 tilings.
  if (g_sp_exact_f32 && dtype == D3D_F32) {  // knob 10: exact f32 products (weights packed as plain f32 copies)
    D3D_REQUIRE(!row_plan, "d3d_sparse_conv_dgrad_planned: not with the exact-f32 test knob");
    return launch_fwd_generic(feat, wpack, bias, nbr, cap, n_out, kvol, cin, cout, dtype, out, stream);
  }
  if (row_plan) {
    // three to four stages per block on average: depth 2 (1 where the f32 fragments of two tiles fill the registers).
    // Block shapes from profiles/round3_dgrad_plan.txt: `<64,32>` f32 67.6 us on 16 waves, 55.2 on 8, 59.5 on 4.
#define PLAN_CASE(CI, CO, MT, WV, DB, DF)                                                                         \
    if (cin == CI && cout == CO)                                                                                  \
      return dtype == D3D_BF16                                                                                    \
                 ? launch_fwd<CI, CO, MT, WV, DB, true, true>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream) \
                 : launch_fwd<CI, CO, MT, WV, DF, false, true>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    PLAN_CASE(32, 16, 1, 4, 2, 2)
    PLAN_CASE(64, 32, 1, 8, 2, 2)
    PLAN_CASE(64, 64, 2, 8, 2, 1)
    PLAN_CASE(128, 64, 1, 16, 2, 1)
    PLAN_CASE(128, 128, 1, 16, 2, 1)
#undef PLAN_CASE
    return d3d_set_error(D3D_EINVAL, "planned data gradient: channels %d -> %d", cin, cout);
  }
  const int fwd_variant = g_fwd_variant >= 0 ? g_fwd_variant : fwd_variant_env();
  if (fwd_variant && fwd_variant < 50 && cin == 64 && cout == 64 && dtype == D3D_BF16) {
#define FWD_VAR(V, MT, WV, DP) \
    if (fwd_variant == V) return launch_fwd<64, 64, MT, WV, DP, true>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD_VAR(40, 2, 8, 3)  // the per-offset-barrier kernel that was the default before the grouped one
    FWD_VAR(1, 1, 16, 3)
    FWD_VAR(2, 1, 16, 5)
    FWD_VAR(3, 2, 8, 3)
    FWD_VAR(5, 2, 16, 3)
    FWD_VAR(7, 4, 4, 2)
    FWD_VAR(8, 4, 8, 2)
    FWD_VAR(9, 4, 4, 3)
    FWD_VAR(10, 1, 8, 4)
#undef FWD_VAR
#define FWD_CHUNK(V, MT, WV, DP, KC) \
    if (fwd_variant == V) return launch_fwd_chunk<64, 64, MT, WV, DP, KC>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD_CHUNK(20, 2, 8, 3, 9)
    FWD_CHUNK(21, 1, 16, 3, 9)
    FWD_CHUNK(22, 2, 4, 3, 9)
    FWD_CHUNK(23, 2, 8, 1, 9)
    FWD_CHUNK(24, 4, 4, 3, 9)
    FWD_CHUNK(25, 1, 8, 3, 9)
    FWD_CHUNK(26, 2, 8, 2, 8)
#undef FWD_CHUNK
    // three blocks per CU: 6 offsets (48 KB) resident, <= 85 registers
    if (fwd_variant == 27) return launch_fwd_chunk<64, 64, 1, 8, 1, 6, 6>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    if (fwd_variant == 28) return launch_fwd_chunk<64, 64, 1, 4, 1, 3, 6>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    if (fwd_variant == 29) return launch_fwd_chunk<64, 64, 1, 8, 1, 9, 4>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
#define FWD_GROUP(V, MT, WV, DP) \
    if (fwd_variant == V) return launch_fwd_group<64, 64, MT, WV, DP>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD_GROUP(30, 2, 8, 3)
    FWD_GROUP(31, 1, 16, 3)
    FWD_GROUP(32, 2, 8, 2)
    FWD_GROUP(33, 1, 8, 3)
    FWD_GROUP(34, 2, 4, 3)
    FWD_GROUP(35, 1, 8, 5)
#undef FWD_GROUP
  }
  if (fwd_variant >= 50 && cin == 64 && cout == 64 && dtype == D3D_F32) {  // f32 (split arithmetic) tilings
#define FWD32_VAR(V, MT, WV, DP) \
    if (fwd_variant == V) return launch_fwd<64, 64, MT, WV, DP, false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD32_VAR(50, 1, 16, 2)
    FWD32_VAR(51, 1, 16, 3)
    FWD32_VAR(52, 2, 8, 2)
    FWD32_VAR(53, 1, 8, 3)
    FWD32_VAR(54, 2, 16, 1)
    FWD32_VAR(55, 1, 16, 1)
#undef FWD32_VAR
  }
  if (fwd_variant >= 60 && fwd_variant < 80 && dtype == D3D_F32) {  // f32 tilings of the 128-channel layers (CBGS)
#define FWD32_WIDE(V, CI, CO, MT, WV, DP)       \
    if (fwd_variant == V && cin == CI && cout == CO) \
      return launch_fwd<CI, CO, MT, WV, DP, false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD32_WIDE(60, 128, 128, 2, 8, 1)
    FWD32_WIDE(61, 128, 128, 1, 8, 2)
    FWD32_WIDE(62, 128, 128, 1, 16, 2)
    FWD32_WIDE(60, 64, 128, 2, 8, 1)
    FWD32_WIDE(61, 64, 128, 1, 8, 2)
    FWD32_WIDE(62, 64, 128, 1, 16, 2)
    FWD32_WIDE(63, 64, 128, 2, 8, 2)
    FWD32_WIDE(60, 128, 64, 2, 8, 1)
    FWD32_WIDE(61, 128, 64, 1, 8, 2)
    FWD32_WIDE(62, 128, 64, 1, 16, 2)
    FWD32_WIDE(63, 128, 64, 2, 8, 2)
#undef FWD32_WIDE
  }
  // f32 features on the chunk-resident-weights kernel (round 5).  Tiling sets (knob 11 / D3D_FWD_RESIDENT):
  //   1: the tile shapes of the per-offset kernel below, all 27 matrices resident where they fit 48 KB, else 9
  //   2: one tile per wave on the 16 / 32-channel layers (twice the blocks); 64 -> 64 four tiles per wave on 8 waves
  //   3: 16-wave blocks with all 27 matrices resident (one block per CU)
  if (dtype == D3D_F32 && g_fwd_resident != 0) {
    int rv = g_fwd_resident;
    // the measured default (profiles/round5_tune_resident.txt): set 1 on the 16 / 32-channel layers (-7 ... -22 % alone on
    // the device; same block shape as the per-offset kernel, so the statistics stay bit-identical too); everything else
    // stays on the per-offset kernel (64 -> 64: -5 % where 512-site blocks fill the chip, +80 % where they do not)
    if (rv < 0) rv = (cin <= 32 && cout <= 32) ? 1 : 0;
#define FWD_RES(V, CI, CO, MT, WV, DP, KC)                                                                        \
    if (rv == V && cin == CI && cout == CO)                                                                        \
      return launch_fwd_resident<CI, CO, MT, WV, DP, KC>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    FWD_RES(1, 16, 16, 2, 8, 3, 27)
    FWD_RES(1, 16, 32, 2, 8, 3, 9)
    FWD_RES(1, 32, 16, 2, 8, 3, 9)
    FWD_RES(1, 32, 32, 2, 8, 3, 9)
    FWD_RES(1, 32, 64, 2, 16, 3, 9)
    FWD_RES(1, 64, 32, 1, 16, 3, 9)
    FWD_RES(1, 64, 64, 2, 16, 1, 9)
    FWD_RES(2, 16, 16, 1, 8, 9, 27)
    FWD_RES(2, 16, 32, 1, 8, 3, 9)
    FWD_RES(2, 32, 16, 1, 8, 3, 9)
    FWD_RES(2, 32, 32, 1, 8, 3, 9)
    FWD_RES(2, 32, 64, 1, 16, 3, 9)
    FWD_RES(2, 64, 64, 4, 8, 1, 9)
    FWD_RES(3, 16, 16, 1, 16, 9, 27)
    FWD_RES(3, 16, 32, 2, 16, 3, 27)
    FWD_RES(3, 32, 16, 2, 16, 3, 27)
    FWD_RES(3, 32, 32, 2, 16, 3, 27)
#undef FWD_RES
  }
  FWD_CASE(16, 16, 2, 8, 5, 4)
  FWD_CASE(16, 32, 2, 8, 5, 4)
  FWD_CASE(32, 16, 2, 8, 5, 3)
  FWD_CASE(32, 32, 2, 8, 5, 3)
  FWD_CASE(32, 64, 2, 8, 4, 2)
  FWD_CASE(64, 32, 1, 16, 5, 2)
  // 64 -> 64 bf16 (16 of the 39 launches of a KITTI step): grouped barriers, profiles/round2_tune_fwd.txt
  if (cin == 64 && cout == 64 && dtype == D3D_BF16 && (fwd_variant == 0 || fwd_variant >= 50))
    return launch_fwd_group<64, 64, 2, 8, 2>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
  FWD_CASE(64, 64, 2, 8, 3, 1)
  // the 128-channel layers of the CBGS backbones in f32 (profiles/round4_tune_fwd_wide.txt, 264k-site rulebook):
  // 128->128 two tiles per wave on 8 waves 610 us against 748 (each weight fragment read from LDS feeds six MFMAs
  // instead of three: the 16-wave tiling kept the LDS pipe 2/3 busy); 64->128 one tile on 8 waves, depth 2: 294 us
  // submanifold / 302 us strided against 310 / 337; 128->64 stays (440 us, 0.69 of the roofline)
  if (dtype == D3D_F32 && !g_f32_wide) {
    if (cin == 128 && cout == 128)
      return launch_fwd<128, 128, 2, 8, 1, false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
    if (cin == 64 && cout == 128)
      return launch_fwd<64, 128, 1, 8, 2, false>(feat, wpack, bias, nbr, cap, n_out, kvol, out, ep, stream);
  }
  FWD_CASE(64, 128, 1, 16, 3, 1)
  FWD_CASE(128, 64, 1, 16, 3, 1)
  FWD_CASE(128, 128, 1, 16, 2, 1)
  return launch_fwd_generic(feat, wpack, bias, nbr, cap, n_out, kvol, cin, cout, dtype, out, stream);
}

// ======================================================================= wgrad
// bf16 MFMA wgrad.  grid = (kvol / KG, splits): a block owns KG consecutive kernel offsets and
// a contiguous range of output sites.  Per tile of STEP sites the dout rows are staged in LDS
// ONCE and reused for the KG offsets (v1 re-read dout 27x: 100 MB FETCH_SIZE per launch,
// profiles/round1_pmc_fetch_size.txt); per (tile, offset) stage the gathered feature rows are
// staged as [16 sites][16 ch] bf16 tiles and every wave pulls transposed fragments with
// ds_read_b64_tr_b16 (sites become the MFMA k dimension).  Stages are software pipelined:
// the next stage's global loads are issued before the current stage's MFMAs, LDS is double
// buffered, one barrier per stage.  Partial sums go to dW with f32 atomics.
template <int CIN, int COUT, int KG, int STEP, bool PIPE>
__global__ void __launch_bounds__(256)
    spconv_wgrad_mfma_kernel(const uint16_t *__restrict__ feat, const uint16_t *__restrict__ dout,
                             const int32_t *__restrict__ nbr, int cap, int n_out,
                             int rows_per_split, float *__restrict__ dw) {
  constexpr int CIT = CIN / 16, COT = COUT / 16;
  constexpr int TILES = CIT * COT;
  constexpr int TPW = (TILES + 3) / 4;  // MFMA tiles per wave
  constexpr int CHUNKS = STEP / 16;
  constexpr int FSEG = CIN / 8, DSEG = COUT / 8;  // 16-byte segments per row
  constexpr int FN = STEP * FSEG, DN = STEP * DSEG;             // 16-byte vectors per tile
  constexpr int FV = (FN + 255) / 256, DV = (DN + 255) / 256;   // vectors per thread
  // [buffer][chunk][ch tile][16 sites][16 ch] bf16
  constexpr int NBUF = PIPE ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint16_t lds_f[NBUF * CHUNKS * CIT * 256];
  __shared__ __attribute__((aligned(16))) uint16_t lds_d[NBUF * CHUNKS * COT * 256];
  const int k0 = blockIdx.x * KG;
  const int s0 = blockIdx.y * rows_per_split;
  int s1 = s0 + rows_per_split;
  if (s1 > n_out) s1 = n_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  f32x4_t acc[KG][TPW];
#pragma unroll
  for (int kk = 0; kk < KG; ++kk)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[kk][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (s1 - s0 + STEP - 1) / STEP;
  const int nstages = ntiles * KG;
  uint4 fr[FV], dr[DV];

  auto load_stage = [&](int st) {
    const int tile = st / KG, kk = st - tile * KG;
    const int base = s0 + tile * STEP;
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      int e = threadIdx.x + j * 256;
      if (FN % 256 != 0 && e >= FN) break;
      int r = e / FSEG, seg = e - r * FSEG;
      int site = base + r;
      int i = site < s1 ? nbr[(int64_t)(k0 + kk) * cap + site] : -1;
      fr[j] = i >= 0 ? *(const uint4 *)(feat + (int64_t)i * CIN + seg * 8) : make_uint4(0, 0, 0, 0);
    }
    if (kk == 0) {
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        int e = threadIdx.x + j * 256;
        if (DN % 256 != 0 && e >= DN) break;
        int r = e / DSEG, seg = e - r * DSEG;
        int site = base + r;
        dr[j] = site < s1 ? *(const uint4 *)(dout + (int64_t)site * COUT + seg * 8)
                          : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_stage = [&](int st) {
    const int tile = st / KG, kk = st - tile * KG;
    uint16_t *bf = lds_f + (PIPE ? (st & 1) : 0) * (CHUNKS * CIT * 256);
#pragma unroll
    for (int j = 0; j < FV; ++j) {
      int e = threadIdx.x + j * 256;
      if (FN % 256 != 0 && e >= FN) break;
      int r = e / FSEG, seg = e - r * FSEG;
      *(uint4 *)(bf + ((r >> 4) * CIT + (seg >> 1)) * 256 + (r & 15) * 16 + (seg & 1) * 8) = fr[j];
    }
    if (kk == 0) {
      uint16_t *bd = lds_d + (PIPE ? (tile & 1) : 0) * (CHUNKS * COT * 256);
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        int e = threadIdx.x + j * 256;
        if (DN % 256 != 0 && e >= DN) break;
        int r = e / DSEG, seg = e - r * DSEG;
        *(uint4 *)(bd + ((r >> 4) * COT + (seg >> 1)) * 256 + (r & 15) * 16 + (seg & 1) * 8) = dr[j];
      }
    }
  };

  if (PIPE && nstages > 0) load_stage(0);
  for (int st = 0; st < nstages; ++st) {
    const int tile = st / KG, kk = st - tile * KG;
    if (!PIPE) {
      if (st > 0) __syncthreads();  // previous stage's fragment reads are done
      load_stage(st);
    }
    store_stage(st);
    __syncthreads();
    if (PIPE && st + 1 < nstages) load_stage(st + 1);  // in flight during the MFMAs below
    const uint16_t *bf = lds_f + (PIPE ? (st & 1) : 0) * (CHUNKS * CIT * 256);
    const uint16_t *bd = lds_d + (PIPE ? (tile & 1) : 0) * (CHUNKS * COT * 256);
#pragma unroll
    for (int kq = 0; kq < KG; ++kq) {
      if (kq != kk) continue;  // keeps acc[] statically indexed
#pragma unroll
      for (int chunk = 0; chunk < CHUNKS; ++chunk) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          int tl = wave + t * 4;
          if (tl < TILES) {
            int cit = tl / COT, cot = tl - cit * COT;
            // each lane passes the address of ITS 8 bytes in the row-major 4x16 block of
            // sites [lg*4, lg*4+4); the hardware returns the transposed column lr
            s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t *)(bf + (chunk * CIT + cit) * 256 + lg * 64 + lr * 4));
            s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t *)(bd + (chunk * COT + cot) * 256 + lg * 64 + lr * 4));
            acc[kq][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[kq][t], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < KG; ++kk)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      int tl = wave + t * 4;
      if (tl < TILES) {
        int cit = tl / COT, cot = tl - cit * COT;
        float *o = dw + ((int64_t)(k0 + kk) * CIN + cit * 16 + lg * 4) * COUT + cot * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(o + r * COUT, acc[kk][t][r]);
      }
    }
}

// generic wgrad (any dtype / channel count): block = (offset k, site split); each
// thread owns dW elements e = tid + j*256 and walks the split's sites.
template <bool BF16>
__global__ void __launch_bounds__(256)
    spconv_wgrad_generic_kernel(const void *__restrict__ feat_, const void *__restrict__ dout_,
                                const int32_t *__restrict__ nbr, int cap, int n_out,
                                int rows_per_split, int cin, int cout, float *__restrict__ dw) {
  extern __shared__ float smem[];  // [R][cin] + [R][cout]
  constexpr int R = 32;
  float *sf = smem, *sd = smem + R * cin;
  const int k = blockIdx.x;
  const int s0 = blockIdx.y * rows_per_split;
  int s1 = s0 + rows_per_split;
  if (s1 > n_out) s1 = n_out;
  const int nel = cin * cout;
  constexpr int MAXE = 64;  // supports cin*cout <= 16384
  float acc[MAXE];
#pragma unroll
  for (int j = 0; j < MAXE; ++j) acc[j] = 0.f;
  for (int base = s0; base < s1; base += R) {
    for (int e = threadIdx.x; e < R * cin; e += 256) {
      int r = e / cin, c = e - r * cin;
      int site = base + r;
      int i = site < s1 ? nbr[(int64_t)k * cap + site] : -1;
      float v = 0.f;
      if (i >= 0)
        v = BF16 ? bf16_to_f32(((const uint16_t *)feat_)[(int64_t)i * cin + c])
                 : ((const float *)feat_)[(int64_t)i * cin + c];
      sf[e] = v;
    }
    for (int e = threadIdx.x; e < R * cout; e += 256) {
      int r = e / cout, c = e - r * cout;
      int site = base + r;
      float v = 0.f;
      if (site < s1)
        v = BF16 ? bf16_to_f32(((const uint16_t *)dout_)[(int64_t)site * cout + c])
                 : ((const float *)dout_)[(int64_t)site * cout + c];
      sd[e] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
      int e = threadIdx.x + j * 256;
      if (e < nel) {
        int ci = e / cout, co = e - ci * cout;
        float a = acc[j];
        for (int r = 0; r < R; ++r) a = fmaf(sf[r * cin + ci], sd[r * cout + co], a);
        acc[j] = a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MAXE; ++j) {
    int e = threadIdx.x + j * 256;
    if (e < nel) atomicAdd(&dw[(int64_t)k * nel + e], acc[j]);
  }
}

// stem layers (Cin = 3..5 -> 16): every thread walks its share of the split's sites with
// the whole [CIN x COUT] tile in registers, then a wave shuffle reduction and one atomic
// per wave and element.  (the generic kernel below needs ~0.9 ms for this layer)
template <int CIN, int COUT, bool BF16>
__global__ void __launch_bounds__(256)
    spconv_wgrad_small_kernel(const void *__restrict__ feat_, const void *__restrict__ dout_,
                              const int32_t *__restrict__ nbr, int cap, int n_out,
                              int rows_per_split, float *__restrict__ dw) {
  const int k = blockIdx.x;
  const int s0 = blockIdx.y * rows_per_split;
  int s1 = s0 + rows_per_split;
  if (s1 > n_out) s1 = n_out;
  float acc[CIN][COUT];
#pragma unroll
  for (int a = 0; a < CIN; ++a)
#pragma unroll
    for (int b = 0; b < COUT; ++b) acc[a][b] = 0.f;
  for (int site = s0 + threadIdx.x; site < s1; site += 256) {
    int i = nbr[(int64_t)k * cap + site];
    if (i < 0) continue;
    float f[CIN], d[COUT];
#pragma unroll
    for (int a = 0; a < CIN; ++a)
      f[a] = BF16 ? bf16_to_f32(((const uint16_t *)feat_)[(int64_t)i * CIN + a])
                  : ((const float *)feat_)[(int64_t)i * CIN + a];
#pragma unroll
    for (int b = 0; b < COUT; ++b)
      d[b] = BF16 ? bf16_to_f32(((const uint16_t *)dout_)[(int64_t)site * COUT + b])
                  : ((const float *)dout_)[(int64_t)site * COUT + b];
#pragma unroll
    for (int a = 0; a < CIN; ++a)
#pragma unroll
      for (int b = 0; b < COUT; ++b) acc[a][b] = fmaf(f[a], d[b], acc[a][b]);
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < CIN; ++a)
#pragma unroll
    for (int b = 0; b < COUT; ++b) {
      float v = acc[a][b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0 && v != 0.f) atomicAdd(&dw[((int64_t)k * CIN + a) * COUT + b], v);
    }
}

#define WGRAD_SMALL_CASE(CI, CO)                                                                 \
  if (cin == CI && cout == CO) {                                                                 \
    if (dtype == D3D_BF16)                                                                       \
      hipLaunchKernelGGL((spconv_wgrad_small_kernel<CI, CO, true>), dim3(kvol, splits),          \
                         dim3(256), 0, stream, feat, dout, nbr, cap, n_out, rows_per_split, dw); \
    else                                                                                         \
      hipLaunchKernelGGL((spconv_wgrad_small_kernel<CI, CO, false>), dim3(kvol, splits),         \
                         dim3(256), 0, stream, feat, dout, nbr, cap, n_out, rows_per_split, dw); \
    D3D_LAUNCH_TRY("spconv_wgrad_small_kernel");                                                 \
    return D3D_OK;                                                                               \
  }

template <int CI, int CO, int KG, int STEP, bool PIPE>
static int launch_wgrad(const void *feat, const void *dout, const int32_t *nbr, int cap, int n_out,
                        int kvol, int splits, int rows_per_split, float *dw, hipStream_t stream) {
  hipLaunchKernelGGL((spconv_wgrad_mfma_kernel<CI, CO, KG, STEP, PIPE>), dim3(kvol / KG, splits),
                     dim3(256), 0, stream, (const uint16_t *)feat, (const uint16_t *)dout, nbr, cap,
                     n_out, rows_per_split, dw);
  D3D_LAUNCH_TRY("spconv_wgrad_mfma_kernel");
  return D3D_OK;
}

#define WGRAD_ARGS feat, dout, nbr, cap, n_out, kvol, splits, rows_per_split, dw, stream
// offsets are grouped by 3 (27 = 9 x 3, (3,1,1) = 1 x 3) when the [Cin x Cout] tile is small
// enough to keep 3 accumulator sets in registers; `variant` (tuning knob) picks tile / pipelining
#define WGRAD_CASE(CI, CO)                                                                \
  if (dtype == D3D_BF16 && cin == CI && cout == CO) {                                     \
    if (kgroups != kvol) {                                                                \
      if (variant == 0) return launch_wgrad<CI, CO, 3, 64, false>(WGRAD_ARGS);            \
      if (variant == 1) return launch_wgrad<CI, CO, 3, 64, true>(WGRAD_ARGS);             \
      return launch_wgrad<CI, CO, 3, 128, true>(WGRAD_ARGS);                              \
    }                                                                                     \
    if (variant == 0) return launch_wgrad<CI, CO, 1, 64, false>(WGRAD_ARGS);              \
    return launch_wgrad<CI, CO, 1, 64, true>(WGRAD_ARGS);                                 \
  }

int d3d_wgrad_legacy(const void *feat, const void *dout, const int32_t *nbr, int cap, int n_out, int kvol,
                     int cin, int cout, int dtype, float *dw, hipStream_t stream) {
  D3D_HIP_TRY(d3d_fill_async(stream, dw, 0, (size_t)kvol * cin * cout * sizeof(float)));
  if (n_out == 0) return D3D_OK;
  D3D_REQUIRE(feat && dout && nbr, "d3d_sparse_conv_wgrad: null pointer");
  // enough blocks to fill 256 CUs a few times over, but >= 1024 sites per split
  // tuning knobs (tools/tune_wgrad.py): D3D_WGRAD_KG=1 disables offset grouping,
  // D3D_WGRAD_BLOCKS = target number of blocks per launch
  // defaults from the sweep on a KITTI batch (126k sites, 1.67M pairs, 64x64): KG=1 pipelined
  // 64-site tiles 138 us, KG=3 pipelined 141 us, v1 (unpipelined) 149 us, 128-site tiles 199 us
  if (g_sp_exact_f32 && dtype == D3D_F32) {
    // knob 10: one block per kernel offset walks ALL sites with fmaf -- one atomicAdd per element onto the zeroed dw,
    // i.e. a deterministic, exact-f32 sum in site order
    const int rows = (n_out + 127) / 128 * 128;
    const size_t smem = (size_t)32 * (cin + cout) * sizeof(float);
    hipLaunchKernelGGL(spconv_wgrad_generic_kernel<false>, dim3(kvol, 1), dim3(256), smem, stream, feat, dout, nbr, cap,
                       n_out, rows, cin, cout, dw);
    D3D_LAUNCH_TRY("spconv_wgrad_generic_kernel");
    return D3D_OK;
  }
  static const int knob_kg = getenv("D3D_WGRAD_KG") ? atoi(getenv("D3D_WGRAD_KG")) : 1;
  static const int knob_blocks = getenv("D3D_WGRAD_LEGACY_BLOCKS") ? atoi(getenv("D3D_WGRAD_LEGACY_BLOCKS")) : 1024;
  static const int variant = getenv("D3D_WGRAD_VARIANT") ? atoi(getenv("D3D_WGRAD_VARIANT")) : 1;
  const int kgroups =
      (knob_kg == 3 && kvol % 3 == 0 && cin * cout <= 4096 && cin >= 16) ? kvol / 3 : kvol;
  int splits = (knob_blocks + kgroups - 1) / kgroups;
  int max_splits = (n_out + 1023) / 1024;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows_per_split = (n_out + splits - 1) / splits;
  rows_per_split = (rows_per_split + 127) / 128 * 128;
  splits = (n_out + rows_per_split - 1) / rows_per_split;
  WGRAD_SMALL_CASE(3, 16)
  WGRAD_SMALL_CASE(4, 16)
  WGRAD_SMALL_CASE(5, 16)
  WGRAD_CASE(16, 16)
  WGRAD_CASE(16, 32)
  WGRAD_CASE(32, 32)
  WGRAD_CASE(32, 64)
  WGRAD_CASE(64, 64)
  WGRAD_CASE(64, 128)
  WGRAD_CASE(128, 128)
  size_t smem = (size_t)32 * (cin + cout) * sizeof(float);
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(spconv_wgrad_generic_kernel<true>, dim3(kvol, splits), dim3(256), smem,
                       stream, feat, dout, nbr, cap, n_out, rows_per_split, cin, cout, dw);
  else
    hipLaunchKernelGGL(spconv_wgrad_generic_kernel<false>, dim3(kvol, splits), dim3(256), smem,
                       stream, feat, dout, nbr, cap, n_out, rows_per_split, cin, cout, dw);
  D3D_LAUNCH_TRY("spconv_wgrad_generic_kernel");
  return D3D_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight repack: the parameter keeps spconv's layout [kvol][cin][cout] f32 (checkpoint compatible);
// the conv kernels want W[k] with the contraction dimension contiguous, in the feature dtype:
//   forward  wpack_fwd  [kvol][cout][cin] = W[k]^T
//   dgrad    wpack_dgrad[kvol][cin][cout] = W[k] (k mirrored for submanifold layers)
// One launch writes both (the PyTorch-op version was transpose-copy + cast + flip-copy + cast = 4
// launches per layer and step).
// mode 0: f32 copies (channel pairs without MFMA tiles: the generic kernels read f32), 1: bf16, 2: split = two bf16
// planes per kernel offset, [k][plane][rows][cols] with plane 0 = bf16(w), plane 1 = bf16(w - plane 0)
__global__ void __launch_bounds__(256)
    spconv_pack_weights_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int flip, int mode_fwd,
                               int mode_dgrad, void *__restrict__ fwd_, void *__restrict__ dgrad_) {
  const int per_k = cin * cout;
  const int total = kvol * per_k;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int k = t / per_k, rem = t - k * per_k;
    const int ci = rem / cout, co = rem - ci * cout;
    const float v = w[t];
    const uint16_t h = f32_to_bf16(v);
    const uint16_t l = f32_to_bf16(v - bf16_to_f32(h));
    const int kd = flip ? kvol - 1 - k : k;
    if (fwd_) {
      if (mode_fwd == 0) ((float *)fwd_)[(k * cout + co) * cin + ci] = v;
      else if (mode_fwd == 1) ((uint16_t *)fwd_)[(k * cout + co) * cin + ci] = h;
      else {
        ((uint16_t *)fwd_)[((k * 2 + 0) * cout + co) * cin + ci] = h;
        ((uint16_t *)fwd_)[((k * 2 + 1) * cout + co) * cin + ci] = l;
      }
    }
    if (dgrad_) {
      if (mode_dgrad == 0) ((float *)dgrad_)[(kd * cin + ci) * cout + co] = v;
      else if (mode_dgrad == 1) ((uint16_t *)dgrad_)[(kd * cin + ci) * cout + co] = h;
      else {
        ((uint16_t *)dgrad_)[((kd * 2 + 0) * cin + ci) * cout + co] = h;
        ((uint16_t *)dgrad_)[((kd * 2 + 1) * cin + ci) * cout + co] = l;
      }
    }
  }
}

extern "C" int d3d_pack_conv_weights(const float *weight, int kvol, int cin, int cout, int dtype,
                                     int flip_dgrad, void *wpack_fwd, void *wpack_dgrad,
                                     d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(kvol >= 1 && cin >= 1 && cout >= 1, "d3d_pack_conv_weights: sizes");
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_pack_conv_weights: dtype %d", dtype);
  D3D_REQUIRE(weight && (wpack_fwd || wpack_dgrad), "d3d_pack_conv_weights: null pointer");
  const int total = kvol * cin * cout;
  int nb = (total + 255) / 256;
  if (nb > 1024) nb = 1024;
  // f32 features: the MFMA kernels multiply in split arithmetic and want bf16 hi / lo planes (same bytes as f32)
  const int mode_fwd = dtype == D3D_BF16 ? 1 : (d3d_sparse_conv_has_epilogue(cin, cout, dtype) ? 2 : 0);
  const int mode_dgrad = dtype == D3D_BF16 ? 1 : (d3d_sparse_conv_has_epilogue(cout, cin, dtype) ? 2 : 0);
  hipLaunchKernelGGL(spconv_pack_weights_kernel, dim3(nb), dim3(256), 0, stream, weight, kvol, cin, cout, flip_dgrad,
                     mode_fwd, mode_dgrad, wpack_fwd, wpack_dgrad);
  D3D_LAUNCH_TRY("spconv_pack_weights_kernel");
  return D3D_OK;
}
