// Pieces shared by the sparse-convolution translation units (d3d_spconv.hip, d3d_spconv_pairs.hip): operand types, the
// f32 -> hi / lo bf16 split, the epilogue request as the kernels see it and the epilogue all forward kernels end with.
#pragma once
#include "d3d_common.h"
#include "d3d_fold.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round to nearest even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// f32 -> (hi, lo) bf16 pairs, hi = bf16(v) and lo = bf16(v - hi), both round-to-nearest-even (v_cvt_pk_bf16_f32);
// v - hi is exact in f32, so hi + lo carries 16 mantissa bits of v
typedef __attribute__((ext_vector_type(2))) float sp_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 sp_bf16x2_t;
__device__ __forceinline__ void spconv_split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  const sp_f32x2_t v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sp_bf16x2_t));
  const sp_f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, sp_bf16x2_t));
}
__device__ __forceinline__ void spconv_split4(const uint4 &f, uint2 &hi, uint2 &lo) {
  spconv_split2(__uint_as_float(f.x), __uint_as_float(f.y), hi.x, lo.x);
  spconv_split2(__uint_as_float(f.z), __uint_as_float(f.w), hi.y, lo.y);
}
__device__ __forceinline__ void spconv_split8(const uint4 &f0, const uint4 &f1, uint4 &hi, uint4 &lo) {
  spconv_split2(__uint_as_float(f0.x), __uint_as_float(f0.y), hi.x, lo.x);
  spconv_split2(__uint_as_float(f0.z), __uint_as_float(f0.w), hi.y, lo.y);
  spconv_split2(__uint_as_float(f1.x), __uint_as_float(f1.y), hi.z, lo.z);
  spconv_split2(__uint_as_float(f1.z), __uint_as_float(f1.w), hi.w, lo.w);
}

// the ABI's epilogue request + what the launcher adds for the deterministic cross-block fold (d3d_fold.h)
struct ConvEpiDev : D3dConvEpilogue {
  unsigned *counters;       // ticket words of this launch
  int fold_g;               // blocks per fold group
  const int32_t *row_plan;  // d3d_rowplan.h (planned launches only)
};

// fold plan of a reducing launch: capacity check of the caller's buffer, ticket words from the library's ring
static int epi_prepare(ConvEpiDev &ep, int nb, int cout) {
  ep.counters = nullptr;
  ep.fold_g = 16;
  if (!ep.stats && !ep.bn_partial) return D3D_OK;
  D3D_REQUIRE(nb <= D3D_FOLD_MAX_BLOCKS, "sparse conv epilogue reduction: %d blocks > %d", nb, D3D_FOLD_MAX_BLOCKS);
  const D3dFoldPlan fp = d3d_fold_plan(nb, 2 * cout);
  const size_t need = d3d_fold_ws_floats(nb, 2 * cout);
  if ((size_t)ep.ws_floats < need)
    return d3d_set_error(D3D_ECAPACITY, "sparse conv epilogue reduction: buffer of %d floats < %zu (use "
                         "d3d_conv_epilogue_ws_floats)", ep.ws_floats, need);
  ep.counters = d3d_fold_counters(d3d_fold_counter_words(fp));
  if (!ep.counters) return D3D_EHIP;
  ep.fold_g = fp.g;
  return D3D_OK;
}

// epilogue shared by the forward kernels: lane holds couts [n*16 + lg*4, +4) of site[m].  Optional per-channel
// reductions ride along (each saves one full sweep over the activations, see D3dConvEpilogue in include/d3d.h):
//   ep.stats      forward : sum / sum of squares of the rows just produced (BatchNorm batch statistics)
//   ep.bn_partial dgrad   : sum dz * xhat | sum dz of the BatchNorm(+ReLU) that produced this conv's input
// The sums are deterministic: the MT tiles of a lane in order, the 16 sites of a tile by shuffles, the waves of a
// block through one LDS row per wave summed in wave order, the blocks through d3d_fold_block.
// `red`: LDS scratch of SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS) floats; it may alias the weight buffers (the
// reductions start behind a workgroup barrier, so no wave still reads W).
#define SPCONV_EPI_LDS_FLOATS(COUT, NTHREADS) (((NTHREADS) / 64 + 1) * 2 * (COUT) + 4)
template <int COUT, int MT, bool BF16, int NTHREADS>
__device__ __forceinline__ void spconv_fwd_epilogue(f32x4_t (&acc)[MT][COUT / 16], const int (&site)[MT],
                                                    const bool (&live)[MT], const float *__restrict__ bias,
                                                    void *__restrict__ out_, const ConvEpiDev &ep, float *red,
                                                    int lr, int lg) {
  constexpr int NT = COUT / 16;
  constexpr int NW = NTHREADS / 64;
  const bool want_stats = ep.stats != nullptr, want_bnb = ep.bn_partial != nullptr;
  const int wave = threadIdx.x >> 6;
  if (want_stats || want_bnb) __syncthreads();  // `red` may alias LDS other waves were still reading
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = n * 16 + lg * 4;
    float r0[4] = {0.f, 0.f, 0.f, 0.f}, r1[4] = {0.f, 0.f, 0.f, 0.f};
    float bmu[4], bis[4], bga[4], bbe[4];
    if (want_bnb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bmu[j] = ep.bn_mean[co + j];
        bis[j] = ep.bn_invstd[co + j];
        bga[j] = ep.bn_gamma ? ep.bn_gamma[co + j] : 1.f;
        bbe[j] = ep.bn_beta ? ep.bn_beta[co + j] : 0.f;
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (!live[m]) continue;
      f32x4_t v = acc[m][n];
      if (bias) {
        v[0] += bias[co + 0];
        v[1] += bias[co + 1];
        v[2] += bias[co + 2];
        v[3] += bias[co + 3];
      }
      if (ep.addend) {  // accumulating launch (residual blocks): out = conv + addend
        if constexpr (BF16) {
          const uint2 ar = *(const uint2 *)((const uint16_t *)ep.addend + (int64_t)site[m] * COUT + co);
          v[0] += bf16_to_f32((uint16_t)(ar.x & 0xffffu)); v[1] += bf16_to_f32((uint16_t)(ar.x >> 16));
          v[2] += bf16_to_f32((uint16_t)(ar.y & 0xffffu)); v[3] += bf16_to_f32((uint16_t)(ar.y >> 16));
        } else {
          const float4 ar = *(const float4 *)((const float *)ep.addend + (int64_t)site[m] * COUT + co);
          v[0] += ar.x; v[1] += ar.y; v[2] += ar.z; v[3] += ar.w;
        }
      }
      float w[4];  // the values as stored (rounded to the feature dtype): what the next pass will read
      if constexpr (BF16) {
        uint16_t *o = (uint16_t *)out_ + (int64_t)site[m] * COUT + co;
        const uint16_t h0 = f32_to_bf16(v[0]), h1 = f32_to_bf16(v[1]), h2 = f32_to_bf16(v[2]), h3 = f32_to_bf16(v[3]);
        uint2 pk;
        pk.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
        pk.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
        *(uint2 *)o = pk;
        w[0] = bf16_to_f32(h0); w[1] = bf16_to_f32(h1); w[2] = bf16_to_f32(h2); w[3] = bf16_to_f32(h3);
      } else {
        float *o = (float *)out_ + (int64_t)site[m] * COUT + co;
        *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
      }
      if (want_stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r0[j] += w[j];
          r1[j] = fmaf(w[j], w[j], r1[j]);
        }
      }
      if (want_bnb) {
        float x[4];
        if constexpr (BF16) {
          const uint2 xr = *(const uint2 *)((const uint16_t *)ep.bn_x + (int64_t)site[m] * COUT + co);
          x[0] = bf16_to_f32((uint16_t)(xr.x & 0xffffu)); x[1] = bf16_to_f32((uint16_t)(xr.x >> 16));
          x[2] = bf16_to_f32((uint16_t)(xr.y & 0xffffu)); x[3] = bf16_to_f32((uint16_t)(xr.y >> 16));
        } else {
          const float4 xr = *(const float4 *)((const float *)ep.bn_x + (int64_t)site[m] * COUT + co);
          x[0] = xr.x; x[1] = xr.y; x[2] = xr.z; x[3] = xr.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // same arithmetic as bn_bwd_reduce_kernel (d3d_bn.hip)
          const float xh = (x[j] - bmu[j]) * bis[j];
          const float dz = (ep.bn_relu && fmaf(xh, bga[j], bbe[j]) <= 0.f) ? 0.f : w[j];
          r0[j] = fmaf(dz, xh, r0[j]);
          r1[j] += dz;
        }
      }
    }
    if (want_stats || want_bnb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a0 = r0[j], a1 = r1[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {  // the 16 sites of a tile sit in lanes lr = 0..15
          a0 += __shfl_xor(a0, o, 64);
          a1 += __shfl_xor(a1, o, 64);
        }
        if (lr == 0) {  // one lane per (wave, channel): plain stores
          red[wave * 2 * COUT + co + j] = a0;
          red[wave * 2 * COUT + COUT + co + j] = a1;
        }
      }
    }
  }
  if (want_stats || want_bnb) {
    __syncthreads();
    float *tot = red + NW * 2 * COUT;
    for (int i = threadIdx.x; i < 2 * COUT; i += NTHREADS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 2 * COUT + i];
      tot[i] = t;
    }
    __syncthreads();
    // (sum | sum sq) or (dgamma | dbeta): this block's slab, folded across blocks in a fixed order
    d3d_fold_block<NTHREADS>(tot, 2 * COUT, want_stats ? ep.stats : ep.bn_partial, ep.counters, ep.fold_g,
                             (int *)(tot + 2 * COUT));
  }
}

