// The prediction convolutions of a MultiGroupHead task group -- conv_box / conv_cls / conv_dir, three 1x1 Conv2d
// with bias on the same [B, 128, H, W] map (det3d/models/bbox_heads/mg_head.py:193-232, 553-560; cuDNN in the
// reference, MIOpen here until round 3) -- as ONE launch per direction.
//
// They are not GEMM-shaped work worth the matrix pipe: 128 input channels against 2 + 14 + 4 = 20 outputs, i.e.
// 5 KFLOP per pixel against 512 bytes of input -- a streaming pass over the map.  MIOpen ran them as 9 igemm
// launches per step plus 9 zero-fill and ~12 cast launches, summed their weight gradients with atomics (the last
// nondeterministic kernel of the step) and had no precision between bf16 and f32.  Here: exact f32 FMA arithmetic on
// the VALU for either input format, the map read once for all three convolutions, and a fixed-order two-level sum
// for the weight / bias gradients.
//   fwd    y_k[p, :] = x[p, :] @ W_k^T + b_k          x [M, 128] f32 | bf16 (channels-last rows), y_k same type
//   dgrad  dx[p, :]  = sum_k g_k[p, :] @ W_k           one pass writes the SUM over the three convolutions
//   wgrad  dW_k = g_k^T @ x, db_k = sum_p g_k          per-block partial slabs, summed in a fixed order by a second launch
// The kernels read the convolutions' own parameter tensors (weight [n_k, 128, 1, 1], bias [n_k]) and write / read three
// separate prediction tensors (the loss kernel and the decoder want them contiguous); ctot = sum n_k <= 32.
#include <stdlib.h>

#include "d3d_common.h"

// D3D_HEAD_VALU=1: f32 maps on the round-3 VALU kernels as well (A/B timing)
static const int g_hc_valu = getenv("D3D_HEAD_VALU") ? atoi(getenv("D3D_HEAD_VALU")) : 0;

#define HC_CIN 128
#define HC_MAXC 32       // total output channels of a task group (padded)
#define HC_TP 64         // pixels per tile
#define HC_XROW (HC_CIN + 4)   // padded f32 row of the staged map tile: 16-byte aligned, conflict-free column walks

struct HcSeg {           // one convolution's slice of the concatenated channel axis
  void *ptr[3];          // output (fwd) / gradient (dgrad, wgrad) tensors [M, n_k]
  int n[3], first[3];    // channels, first concatenated channel
  int count;
};

struct HcW {             // the convolutions' own parameter tensors: weight [n_k, 128] f32, bias [n_k] f32 or NULL
  const float *w[3], *b[3];
};

// W_k rows -> LDS [HC_MAXC][128] in concatenated channel order, zero rows behind ctot
__device__ __forceinline__ void hc_stage_w(const HcW &P, const HcSeg &S, float *ws) {
  for (int e = threadIdx.x; e < HC_MAXC * HC_CIN / 4; e += 256) {
    const int row = e / (HC_CIN / 4), c4 = e % (HC_CIN / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < S.count && row >= S.first[k] && row < S.first[k] + S.n[k])
        v = ((const float4 *)P.w[k])[(row - S.first[k]) * (HC_CIN / 4) + c4];
    ((float4 *)ws)[e] = v;
  }
}

__device__ __forceinline__ float hc_bf16(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t hc_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// map tile [HC_TP][128] -> LDS as f32 rows of HC_XROW floats (rows past m are zero)
template <bool BF16>
__device__ __forceinline__ void hc_stage_x(const void *__restrict__ x, int64_t p0, int64_t m, float *xs) {
  // 64 rows x 32 float4 slots = 2048 slots, 8 per thread
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int slot = threadIdx.x + j * 256;
    const int row = slot >> 5, c4 = slot & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + row < m) {
      if (BF16) {
        const uint2 r = *(const uint2 *)((const uint16_t *)x + (p0 + row) * HC_CIN + c4 * 4);
        v = make_float4(hc_bf16((uint16_t)(r.x & 0xffffu)), hc_bf16((uint16_t)(r.x >> 16)),
                        hc_bf16((uint16_t)(r.y & 0xffffu)), hc_bf16((uint16_t)(r.y >> 16)));
      } else {
        v = *(const float4 *)((const float *)x + (p0 + row) * HC_CIN + c4 * 4);
      }
    }
    *(float4 *)(xs + row * HC_XROW + c4 * 4) = v;
  }
}

// gradient tile [HC_TP][ctot] (from the up-to-three tensors) -> LDS f32 [HC_TP][HC_MAXC], zero padded
template <bool BF16>
__device__ __forceinline__ void hc_stage_g(const HcSeg &S, int64_t p0, int64_t m, float *gs) {
  for (int e = threadIdx.x; e < HC_TP * HC_MAXC; e += 256) {
    const int row = e / HC_MAXC, c = e % HC_MAXC;
    float v = 0.f;
    if (p0 + row < m) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < S.count && c >= S.first[k] && c < S.first[k] + S.n[k]) {
          const int64_t i = (p0 + row) * S.n[k] + (c - S.first[k]);
          v = BF16 ? hc_bf16(((const uint16_t *)S.ptr[k])[i]) : ((const float *)S.ptr[k])[i];
        }
    }
    gs[e] = v;
  }
}

template <bool BF16>
__global__ void __launch_bounds__(256)
    head_conv_fwd_kernel(const void *__restrict__ x, HcW P, HcSeg S, int ctot, int64_t m) {
  __shared__ __attribute__((aligned(16))) float xs[HC_TP * HC_XROW];
  __shared__ __attribute__((aligned(16))) float ws[HC_MAXC * HC_CIN];
  const int64_t p0 = (int64_t)blockIdx.x * HC_TP;
  hc_stage_w(P, S, ws);
  hc_stage_x<BF16>(x, p0, m, xs);
  __syncthreads();
  // thread = (pixel, group of 8 output channels); the group index is wave-uniform: weight reads are LDS broadcasts
  const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (cg * 8 < ctot) {
    const float *xr = xs + px * HC_XROW;
#pragma unroll 4
    for (int c4 = 0; c4 < HC_CIN / 4; ++c4) {
      const float4 xv = *(const float4 *)(xr + c4 * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 wv = *(const float4 *)(ws + (cg * 8 + j) * HC_CIN + c4 * 4);
        acc[j] = fmaf(xv.x, wv.x, acc[j]);
        acc[j] = fmaf(xv.y, wv.y, acc[j]);
        acc[j] = fmaf(xv.z, wv.z, acc[j]);
        acc[j] = fmaf(xv.w, wv.w, acc[j]);
      }
    }
  }
  const int64_t p = p0 + px;
  if (p >= m) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    if (c >= ctot) break;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < S.count && c >= S.first[k] && c < S.first[k] + S.n[k]) {
        const float v = acc[j] + (P.b[k] ? P.b[k][c - S.first[k]] : 0.f);
        const int64_t i = p * S.n[k] + (c - S.first[k]);
        if (BF16) ((uint16_t *)S.ptr[k])[i] = hc_to_bf16(v);
        else ((float *)S.ptr[k])[i] = v;
      }
  }
}

template <bool BF16>
__global__ void __launch_bounds__(256)
    head_conv_dgrad_kernel(HcW P, HcSeg S, int ctot, int64_t m, void *__restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float gs[HC_TP * HC_MAXC];
  __shared__ __attribute__((aligned(16))) float ws[HC_MAXC * HC_CIN];
  const int64_t p0 = (int64_t)blockIdx.x * HC_TP;
  hc_stage_w(P, S, ws);
  hc_stage_g<BF16>(S, p0, m, gs);
  __syncthreads();
  // thread = (4 input channels, group of 8 pixels)
  const int c4 = threadIdx.x & 31, pg = threadIdx.x >> 5;
  float4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int co = 0; co < ctot; ++co) {
    const float4 wv = *(const float4 *)(ws + co * HC_CIN + c4 * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = gs[(pg * 8 + j) * HC_MAXC + co];
      acc[j].x = fmaf(g, wv.x, acc[j].x);
      acc[j].y = fmaf(g, wv.y, acc[j].y);
      acc[j].z = fmaf(g, wv.z, acc[j].z);
      acc[j].w = fmaf(g, wv.w, acc[j].w);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t p = p0 + pg * 8 + j;
    if (p >= m) break;
    if (BF16) {
      uint2 pk;
      pk.x = (uint32_t)hc_to_bf16(acc[j].x) | ((uint32_t)hc_to_bf16(acc[j].y) << 16);
      pk.y = (uint32_t)hc_to_bf16(acc[j].z) | ((uint32_t)hc_to_bf16(acc[j].w) << 16);
      *(uint2 *)((uint16_t *)dx + p * HC_CIN + c4 * 4) = pk;
    } else {
      *(float4 *)((float *)dx + p * HC_CIN + c4 * 4) = acc[j];
    }
  }
}

// per block: a contiguous range of pixels; thread = (input channel, half of the output channels); the block writes its
// [HC_MAXC x 128] weight partial + [HC_MAXC] bias partial as one slab, head_conv_wgrad_reduce_kernel sums the slabs in
// a fixed order.  (The in-kernel ticket fold of d3d_fold.h is built for [2C]-sized vectors: with 4128 values per block
// its narrow write-through stores and the last block's serial reads cost 60 us here, measured.)
#define HC_WG_NVAL (HC_MAXC * HC_CIN + HC_MAXC)
template <bool BF16>
__global__ void __launch_bounds__(256)
    head_conv_wgrad_kernel(const void *__restrict__ x, HcSeg S, int ctot, int64_t m, int64_t pixels_per_block,
                           float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *xs = smem;                                  // [HC_TP][HC_XROW]
  float *gs = xs + HC_TP * HC_XROW;                  // [HC_TP][HC_MAXC]
  const int ci = threadIdx.x & 127, ch = threadIdx.x >> 7;  // output channels ch * 16 .. + 16
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  float bsum = 0.f;  // threads 0 .. 31 of the block: bias gradient of output channel threadIdx.x
  const int64_t pb = (int64_t)blockIdx.x * pixels_per_block;
  int64_t pe = pb + pixels_per_block;
  if (pe > m) pe = m;
  for (int64_t p0 = pb; p0 < pe; p0 += HC_TP) {
    __syncthreads();
    hc_stage_x<BF16>(x, p0, pe, xs);
    hc_stage_g<BF16>(S, p0, pe, gs);
    __syncthreads();
    for (int r = 0; r < HC_TP; ++r) {
      const float xv = xs[r * HC_XROW + ci];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g4 = *(const float4 *)(gs + r * HC_MAXC + ch * 16 + q * 4);
        acc[q * 4 + 0] = fmaf(g4.x, xv, acc[q * 4 + 0]);
        acc[q * 4 + 1] = fmaf(g4.y, xv, acc[q * 4 + 1]);
        acc[q * 4 + 2] = fmaf(g4.z, xv, acc[q * 4 + 2]);
        acc[q * 4 + 3] = fmaf(g4.w, xv, acc[q * 4 + 3]);
      }
    }
    if (threadIdx.x < HC_MAXC)
      for (int r = 0; r < HC_TP; ++r) bsum += gs[r * HC_MAXC + threadIdx.x];
  }
  float *slab = part + (size_t)blockIdx.x * HC_WG_NVAL;
#pragma unroll
  for (int j = 0; j < 16; ++j) slab[(ch * 16 + j) * HC_CIN + ci] = acc[j];
  if (threadIdx.x < HC_MAXC) slab[HC_MAXC * HC_CIN + threadIdx.x] = bsum;
}

// out[e] = sum over slabs of part[s][e]: 8 slab groups walked in parallel, folded through LDS in group order
__global__ void __launch_bounds__(256)
    head_conv_wgrad_reduce_kernel(const float4 *__restrict__ part, int slabs, float4 *__restrict__ out) {
  __shared__ float4 red[8][32];
  constexpr int N4 = HC_WG_NVAL / 4;
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < N4)
    for (int s0 = g; s0 < slabs; s0 += 8) {
      const float4 u = part[(size_t)s0 * N4 + e];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  red[g][el] = a;
  __syncthreads();
  if (g == 0 && e < N4) {
    float4 r = red[0][el];
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      r.x += red[j][el].x; r.y += red[j][el].y; r.z += red[j][el].z; r.w += red[j][el].w;
    }
    out[e] = r;
  }
}

// ---------------------------------------------------------------------------------------------------- f32 maps: MFMA
// Round 6 (second session).  The VALU kernels above re-stage the 16 KB of weights per 64-pixel block, run load -> barrier ->
// compute without overlap and spend 9 LDS reads per 32 FMAs: 50 us (forward), 28 us (data gradient), 55 + 20 us (weight
// gradient + its fold) on the [4, 200, 176, 128] map, i.e. 1.4 TB/s on passes that move 72 MB once.  For f32 maps the
// forward and the weight gradient now run on v_mfma_f32_32x32x2_f32 (the data gradient built the same way measured 27 us
// against the VALU kernel's 23 and was dropped) -- exact f32 products and accumulation, bit-for-bit an fmaf chain
// (cdna_hip_programming.md, "FP32-input MFMA"), so the arithmetic contract of this file does not change -- with the operands
// loaded straight from global memory into the MFMA lane layout (A: lane l holds A[l & 31][l >> 5], B: B[l >> 5][l & 31]):
//   fwd    D[pixel][channel]: A = 32 map rows x 2 input channels, B = the concatenated weights, 64 k-steps per 32-pixel tile;
//          a lane's float4 loads are 32-byte pairs per map row (four consecutive instructions finish a 128-byte line), the
//          64 weight operands of a lane live in registers for the whole kernel, the next tile's rows are loaded while the
//          current tile multiplies; the C layout (lane = channel, registers = pixels) stores each prediction row contiguously
//   wgrad  D[channel][input channel 4n + j]: the PIXELS are the k dimension (2 per instruction): one float4 of the map row
//          and one gradient value per lane and pixel pair feed four MFMAs; a block's waves fold their accumulators through
//          LDS in wave order, blocks write slabs, head_conv_wgrad_reduce2_kernel sums the slabs in a fixed order
// bf16 maps stay on the kernels above.
typedef __attribute__((ext_vector_type(16))) float hc_f32x16;

struct HcLane {  // where lane-channel c = lane & 31 of the concatenated axis lives: element pointer of row 0, row stride (floats)
  float *ptr;
  int stride;
};
__device__ __forceinline__ HcLane hc_lane_channel(const HcSeg &S, int c) {
  HcLane r = {nullptr, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < S.count && c >= S.first[k] && c < S.first[k] + S.n[k]) {
      r.ptr = (float *)S.ptr[k] + (c - S.first[k]);
      r.stride = S.n[k];
    }
  return r;
}
// weight element W[c][ci] (0 for padded channels) and bias of concatenated channel c
__device__ __forceinline__ const float *hc_weight_row(const HcW &P, const HcSeg &S, int c) {
  const float *r = nullptr;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < S.count && c >= S.first[k] && c < S.first[k] + S.n[k]) r = P.w[k] + (size_t)(c - S.first[k]) * HC_CIN;
  return r;
}

// k-step (t, j), t = 0..15, j = 0..3: lane half h multiplies input channel 8 t + 4 h + j.  A 32-pixel tile of the map is
// 16 KB of consecutive memory: the wave loads it with 16 fully coalesced float4 instructions, writes it to its PRIVATE LDS
// region (rows padded to 132 floats) and reads it back in the operand layout, lane (pixel, h) -> 16 x ds_read_b128 of its row.
// (Loading the operand layout straight from global memory touches 32 lines per instruction for 32 bytes each, four
// instructions per line: 31 us for this map; through LDS: see profiles/round6_head_mfma.txt.)  Wave-private region: LDS
// operations of one wave execute in order, no barrier anywhere in the loop.
#define HC_F_ROW 132
__global__ void __launch_bounds__(256, 2)
    head_conv_fwd_mfma_kernel(const float *__restrict__ x, HcW P, HcSeg S, int64_t m, int tiles) {
  __shared__ __attribute__((aligned(16))) float stage[4][32 * HC_F_ROW];
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = (int)(threadIdx.x >> 6);
  const int wave_global = (int)(blockIdx.x * 4) + wave, nwaves = (int)gridDim.x * 4;
  float *mine = stage[wave];
  // B operands: W[c][8 t + 4 h + j] for every k-step, resident
  float4 wq[16];
  {
    const float *wr = hc_weight_row(P, S, c);
#pragma unroll
    for (int t = 0; t < 16; ++t)
      wq[t] = wr ? *(const float4 *)(wr + 8 * t + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const HcLane out = hc_lane_channel(S, c);
  float bias = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < S.count && P.b[k] && c >= S.first[k] && c < S.first[k] + S.n[k]) bias = P.b[k][c - S.first[k]];

  // float4 e = 64 t + lane of the tile: map row e / 32, float4 column e % 32
  auto load_tile = [&](int tile, float4 *raw) {
    const int64_t p0 = (int64_t)tile * 32;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = 2 * t + (lane >> 5);
      const bool ok = tile < tiles && p0 + row < m;
      raw[t] = ok ? *(const float4 *)(x + (p0 + row) * HC_CIN + 4 * (lane & 31)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float4 raw[16];
  int tile = wave_global;
  load_tile(tile, raw);
  for (; tile < tiles; tile += nwaves) {
#pragma unroll
    for (int t = 0; t < 16; ++t) *(float4 *)(mine + (2 * t + (lane >> 5)) * HC_F_ROW + 4 * (lane & 31)) = raw[t];
    load_tile(tile + nwaves, raw);  // in flight during this tile's 64 MFMAs
    float4 cur[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[t] = *(const float4 *)(mine + c * HC_F_ROW + 8 * t + 4 * h);
    hc_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t].x, wq[t].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t].y, wq[t].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t].z, wq[t].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t].w, wq[t].w, acc, 0, 0, 0);
    }
    // D[i = pixel][j = channel]: lane holds channel c, register r pixel (r & 3) + 8 (r >> 2) + 4 h of the tile
    if (out.ptr) {
      const int64_t p0 = (int64_t)tile * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t p = p0 + (r & 3) + 8 * (r >> 2);
        if (p < m) out.ptr[p * out.stride] = acc[r] + bias;
      }
    }
  }
}

// pixels are the k dimension: the pair (p, p + 1) per instruction.  Block = 8 waves, wave w takes the pairs w, w + 8, ... of the
// block's range; DEPTH pairs in flight per wave
#define HC_WG2_WAVES 8
#define HC_WG2_DEPTH 4
__global__ void __launch_bounds__(HC_WG2_WAVES * 64)
    head_conv_wgrad_mfma_kernel(const float *__restrict__ x, HcSeg S, int64_t m, int64_t pixels_per_block,
                                float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float slab_s[HC_WG_NVAL];
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const HcLane g = hc_lane_channel(S, c);  // A operand: gradient of channel c at pixel p + h
  const int64_t pb = (int64_t)blockIdx.x * pixels_per_block;
  int64_t pe = pb + pixels_per_block;
  if (pe > m) pe = m;
  hc_f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;
  float4 xq[HC_WG2_DEPTH];
  float gq[HC_WG2_DEPTH];
  auto load_pair = [&](int64_t p, float4 &xv, float &gv) {  // pair starting at pixel p (even offset inside the block)
    const int64_t q = p + h;
    const bool ok = q < pe;
    xv = ok ? *(const float4 *)(x + q * HC_CIN + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    gv = (ok && g.ptr) ? g.ptr[q * g.stride] : 0.f;
  };
  const int64_t step = 2 * HC_WG2_WAVES;
  int64_t p = pb + 2 * wave;
#pragma unroll
  for (int d = 0; d < HC_WG2_DEPTH; ++d) load_pair(p + d * step, xq[d], gq[d]);
  for (; p < pe; p += HC_WG2_DEPTH * step) {
#pragma unroll
    for (int d = 0; d < HC_WG2_DEPTH; ++d) {
      const float4 xv = xq[d];
      const float gv = gq[d];
      load_pair(p + (d + HC_WG2_DEPTH) * step, xq[d], gq[d]);
      bsum += gv;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xv.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xv.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xv.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, xv.w, acc[3], 0, 0, 0);
    }
  }
  // fold the waves in wave order (fixed): D_j[i = channel][n] -> slab[i][4 n + j]; bias: lane halves in order
  for (int w = 0; w < HC_WG2_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float4 *dst = (float4 *)(slab_s + i * HC_CIN + 4 * c);
        float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        if (w > 0) {
          const float4 o = *dst;
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *dst = v;
      }
      const float other = __shfl(bsum, c + 32, 64);  // channel c: pixel parity 0 + parity 1
      if (h == 0) slab_s[HC_MAXC * HC_CIN + c] = (w > 0 ? slab_s[HC_MAXC * HC_CIN + c] : 0.f) + (bsum + other);
    }
    __syncthreads();
  }
  float4 *dst = (float4 *)(part + (size_t)blockIdx.x * HC_WG_NVAL);
  for (int e = threadIdx.x; e < HC_WG_NVAL / 4; e += HC_WG2_WAVES * 64) dst[e] = ((const float4 *)slab_s)[e];
}

// out[e] = sum over slabs of part[s][e]: block = 8 float4 elements x 32 slab groups (every thread walks slabs g, g + 32, ...
// with independent loads), the groups folded through LDS in group order
__global__ void __launch_bounds__(256)
    head_conv_wgrad_reduce2_kernel(const float4 *__restrict__ part, int slabs, float4 *__restrict__ out) {
  __shared__ float4 red[32][8];
  constexpr int N4 = HC_WG_NVAL / 4;
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + el;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < N4)
    for (int s0 = g; s0 < slabs; s0 += 32) {
      const float4 u = part[(size_t)s0 * N4 + e];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  red[g][el] = a;
  __syncthreads();
  if (g == 0 && e < N4) {
    float4 r = red[0][el];
#pragma unroll
    for (int j = 1; j < 32; ++j) {
      r.x += red[j][el].x; r.y += red[j][el].y; r.z += red[j][el].z; r.w += red[j][el].w;
    }
    out[e] = r;
  }
}

static int hc_mfma_grid(int tiles) {  // 4-wave blocks, two per CU resident: at most 2048 waves, at least one tile per wave
  int nb = (tiles + 3) / 4;
  if (nb > 512) nb = 512;
  return nb < 1 ? 1 : nb;
}

// ---------------------------------------------------------------------------------------------------- host side
static int hc_segments(const char *who, HcSeg &S, void *const *ptrs, const int *n, int count, int &ctot) {
  D3D_REQUIRE(count >= 1 && count <= 3, "%s: %d convolutions (1..3)", who, count);
  ctot = 0;
  S.count = count;
  for (int k = 0; k < 3; ++k) {
    S.ptr[k] = k < count ? ptrs[k] : nullptr;
    S.n[k] = k < count ? n[k] : 0;
    S.first[k] = ctot;
    if (k < count) {
      D3D_REQUIRE(ptrs[k] && n[k] >= 1, "%s: convolution %d: null tensor or %d channels", who, k, n[k]);
      ctot += n[k];
    }
  }
  D3D_REQUIRE(ctot <= HC_MAXC, "%s: %d output channels in total (<= %d)", who, ctot, HC_MAXC);
  return D3D_OK;
}

extern "C" int d3d_head_conv_supported(int cin, int ctot) { return cin == HC_CIN && ctot >= 1 && ctot <= HC_MAXC; }

static int hc_params(const char *who, HcW &P, const float *const *weights, const float *const *biases, int count) {
  D3D_REQUIRE(weights, "%s: null weights", who);
  for (int k = 0; k < 3; ++k) {
    P.w[k] = k < count ? weights[k] : nullptr;
    P.b[k] = (k < count && biases) ? biases[k] : nullptr;
    D3D_REQUIRE(k >= count || P.w[k], "%s: null weight %d", who, k);
  }
  return D3D_OK;
}

extern "C" int d3d_head_conv_fwd(const void *x, int64_t m, int cin, int dtype, const float *const *weights,
                                 const float *const *biases, void *const *y, const int *n, int count,
                                 d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_head_conv_fwd: dtype %d", dtype);
  D3D_REQUIRE(cin == HC_CIN && m >= 1 && x && y && n, "d3d_head_conv_fwd: cin %d (must be %d), m %lld", cin,
              HC_CIN, (long long)m);
  HcSeg S;
  HcW P;
  int ctot;
  if (int rc = hc_segments("d3d_head_conv_fwd", S, y, n, count, ctot)) return rc;
  if (int rc = hc_params("d3d_head_conv_fwd", P, weights, biases, count)) return rc;
  const unsigned nb = (unsigned)((m + HC_TP - 1) / HC_TP);
  if (dtype == D3D_F32 && !g_hc_valu && m < ((int64_t)1 << 31) * 16) {
    const int tiles = (int)((m + 31) / 32);
    hipLaunchKernelGGL(head_conv_fwd_mfma_kernel, dim3(hc_mfma_grid(tiles)), dim3(256), 0, stream, (const float *)x, P, S, m,
                       tiles);
    D3D_LAUNCH_TRY("head_conv_fwd_mfma_kernel");
    return D3D_OK;
  }
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(head_conv_fwd_kernel<true>, dim3(nb), dim3(256), 0, stream, x, P, S, ctot, m);
  else
    hipLaunchKernelGGL(head_conv_fwd_kernel<false>, dim3(nb), dim3(256), 0, stream, x, P, S, ctot, m);
  D3D_LAUNCH_TRY("head_conv_fwd_kernel");
  return D3D_OK;
}

extern "C" int d3d_head_conv_dgrad(void *const *g, const int *n, int count, int64_t m, int cin, int dtype,
                                   const float *const *weights, void *dx, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_head_conv_dgrad: dtype %d", dtype);
  D3D_REQUIRE(cin == HC_CIN && m >= 1 && g && dx && n, "d3d_head_conv_dgrad: arguments");
  HcSeg S;
  HcW P;
  int ctot;
  if (int rc = hc_segments("d3d_head_conv_dgrad", S, g, n, count, ctot)) return rc;
  if (int rc = hc_params("d3d_head_conv_dgrad", P, weights, nullptr, count)) return rc;
  const unsigned nb = (unsigned)((m + HC_TP - 1) / HC_TP);
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(head_conv_dgrad_kernel<true>, dim3(nb), dim3(256), 0, stream, P, S, ctot, m, dx);
  else
    hipLaunchKernelGGL(head_conv_dgrad_kernel<false>, dim3(nb), dim3(256), 0, stream, P, S, ctot, m, dx);
  D3D_LAUNCH_TRY("head_conv_dgrad_kernel");
  return D3D_OK;
}

static int hc_wgrad_blocks(int64_t m) {
  int64_t nb = (m + 4 * HC_TP - 1) / (4 * HC_TP);  // >= 4 tiles per block
  if (nb > 512) nb = 512;
  return nb < 1 ? 1 : (int)nb;
}
extern "C" size_t d3d_head_conv_wgrad_ws_floats(int64_t m) {
  if (m < 1) return 0;
  return (size_t)HC_WG_NVAL * (1 + (size_t)hc_wgrad_blocks(m));  // result | one slab per block
}
// ws[0 .. 32*128) = dW [32][128] (rows >= ctot are zero), ws[32*128 .. +32) = db
extern "C" int d3d_head_conv_wgrad(const void *x, void *const *g, const int *n, int count, int64_t m, int cin,
                                   int dtype, float *ws, size_t ws_floats, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_head_conv_wgrad: dtype %d", dtype);
  D3D_REQUIRE(cin == HC_CIN && m >= 1 && x && g && n && ws, "d3d_head_conv_wgrad: arguments");
  HcSeg S;
  int ctot;
  if (int rc = hc_segments("d3d_head_conv_wgrad", S, g, n, count, ctot)) return rc;
  const int nb = hc_wgrad_blocks(m);
  if (ws_floats < d3d_head_conv_wgrad_ws_floats(m))
    return d3d_set_error(D3D_ECAPACITY, "d3d_head_conv_wgrad: workspace %zu floats < %zu", ws_floats,
                         d3d_head_conv_wgrad_ws_floats(m));
  int64_t ppb = (m + nb - 1) / nb;
  ppb = (ppb + HC_TP - 1) / HC_TP * HC_TP;
  const size_t lds = (size_t)(HC_TP * HC_XROW + HC_TP * HC_MAXC) * sizeof(float);
  float *part = ws + HC_WG_NVAL;
  if (dtype == D3D_F32 && !g_hc_valu) {
    // one 8-wave block per CU at most; nb2 <= nb, so the slabs fit the workspace d3d_head_conv_wgrad_ws_floats() sized
    int nb2 = nb < 256 ? nb : 256;
    int64_t ppb2 = (m + nb2 - 1) / nb2;
    ppb2 = (ppb2 + 2 * HC_WG2_WAVES - 1) / (2 * HC_WG2_WAVES) * (2 * HC_WG2_WAVES);  // whole rounds of pixel pairs
    nb2 = (int)((m + ppb2 - 1) / ppb2);
    hipLaunchKernelGGL(head_conv_wgrad_mfma_kernel, dim3(nb2), dim3(HC_WG2_WAVES * 64), 0, stream, (const float *)x, S, m,
                       ppb2, part);
    D3D_LAUNCH_TRY("head_conv_wgrad_mfma_kernel");
    hipLaunchKernelGGL(head_conv_wgrad_reduce2_kernel, dim3((HC_WG_NVAL / 4 + 7) / 8), dim3(256), 0, stream,
                       (const float4 *)part, nb2, (float4 *)ws);
    D3D_LAUNCH_TRY("head_conv_wgrad_reduce2_kernel");
    return D3D_OK;
  }
  if (dtype == D3D_BF16)
    hipLaunchKernelGGL(head_conv_wgrad_kernel<true>, dim3(nb), dim3(256), lds, stream, x, S, ctot, m, ppb, part);
  else
    hipLaunchKernelGGL(head_conv_wgrad_kernel<false>, dim3(nb), dim3(256), lds, stream, x, S, ctot, m, ppb, part);
  D3D_LAUNCH_TRY("head_conv_wgrad_kernel");
  hipLaunchKernelGGL(head_conv_wgrad_reduce_kernel, dim3((HC_WG_NVAL / 4 + 31) / 32), dim3(256), 0, stream,
                     (const float4 *)part, nb, (float4 *)ws);
  D3D_LAUNCH_TRY("head_conv_wgrad_reduce_kernel");
  return D3D_OK;
}
